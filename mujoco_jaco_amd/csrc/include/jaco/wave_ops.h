// Wavefront-level primitives for gfx950 (64 lanes). One env == one wavefront == one workgroup,
// so every "sync" below is wave-scoped.  All cross-lane calls must be reached by all 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

#define JACO_WAVE 64
#define JDEV __device__ __forceinline__

JDEV int lane_id() { return (int)threadIdx.x; }
JDEV int env_id() { return (int)blockIdx.x; }

// LDS hand-off between lanes of the (single-wave) workgroup.  One wavefront issues its LDS instructions in program order and
// the LDS unit serves them in that order, so a read issued after a write of another lane of the SAME wave sees it: nothing
// has to be waited for in hardware.  What must be stopped is the compiler, which reasons per thread ("lane k's store to
// x[k] cannot alias its load of x[k + 1]") and could move the load up: a wavefront-scope fence pair + the scheduling barrier
// pins the order without emitting an instruction.  (__syncthreads() here meant s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier at every
// one of the ~100 stage boundaries per substep: each drained the write acknowledgements and every model load in flight.)
#ifdef JACO_SYNCTHREADS
JDEV void wave_sync() { __syncthreads(); }
#else
JDEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif

// Broadcast from a wave-uniform source lane (v_readlane_b32).
JDEV float wave_bcast(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
JDEV int wave_bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
// A value every lane already holds (e.g. read from the same LDS word): moved to an SGPR so that loop bounds, table
// indices and model loads derived from it become scalar (s_load / s_cbranch) instead of per-lane work.
JDEV int wave_uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Arbitrary per-lane gather (ds_bpermute_b32).
JDEV float wave_shfl(float v, int src) { return __shfl(v, src, 64); }
JDEV int wave_shfl_i(int v, int src) { return __shfl(v, src, 64); }

JDEV unsigned long long wave_ballot(bool p) { return __ballot(p); }
// number of set bits of `mask` below this lane
JDEV int wave_prefix_count(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
JDEV int popc64(unsigned long long m) { return __popcll(m); }
JDEV int ffs64(unsigned long long m) { return __ffsll((long long)m) - 1; }  // index of lowest set bit, -1 if none

template <int CTRL>
JDEV float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
JDEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

// DPP butterflies inside each 16-lane row (quad swap, quad pair swap, half mirror, mirror), then 4 readlanes.
JDEV float wave_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return (wave_bcast(v, 0) + wave_bcast(v, 16)) + (wave_bcast(v, 32) + wave_bcast(v, 48));
}
// two independent sums, interleaved step by step: a DPP operand written by the instruction right in front of it costs two wait states
JDEV void wave_sum2(float a, float b, float* ra, float* rb) {
  a += dpp_f<0xB1>(a); b += dpp_f<0xB1>(b);
  a += dpp_f<0x4E>(a); b += dpp_f<0x4E>(b);
  a += dpp_f<0x141>(a); b += dpp_f<0x141>(b);
  a += dpp_f<0x140>(a); b += dpp_f<0x140>(b);
  const float a0 = wave_bcast(a, 0), a1 = wave_bcast(a, 16), a2 = wave_bcast(a, 32), a3 = wave_bcast(a, 48);
  const float b0 = wave_bcast(b, 0), b1 = wave_bcast(b, 16), b2 = wave_bcast(b, 32), b3 = wave_bcast(b, 48);
  *ra = (a0 + a1) + (a2 + a3);
  *rb = (b0 + b1) + (b2 + b3);
}
JDEV float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return fmaxf(fmaxf(wave_bcast(v, 0), wave_bcast(v, 16)), fmaxf(wave_bcast(v, 32), wave_bcast(v, 48)));
}
JDEV float wave_min(float v) { return -wave_max(-v); }
// maximum over the lane's own 32-lane half, in every lane (row butterflies, then one exchange with the neighbouring row)
JDEV float half_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return fmaxf(v, wave_shfl(v, lane_id() ^ 16));
}

// argmax with lowest-index tie-break; returns the winning index in all lanes, *best gets the value.
#define JACO_ARGMAX_STEP(CTRL)                              \
  {                                                         \
    float v2 = dpp_f<CTRL>(v);                              \
    int i2 = dpp_i<CTRL>(idx);                              \
    bool take = (v2 > v) || (v2 == v && i2 < idx);          \
    v = take ? v2 : v;                                      \
    idx = take ? i2 : idx;                                  \
  }
JDEV int wave_argmax(float v, int idx, float* best) {
  JACO_ARGMAX_STEP(0xB1)
  JACO_ARGMAX_STEP(0x4E)
  JACO_ARGMAX_STEP(0x141)
  JACO_ARGMAX_STEP(0x140)
  float bv = wave_bcast(v, 0);
  int bi = wave_bcast_i(idx, 0);
#pragma unroll
  for (int r = 16; r < 64; r += 16) {
    float v2 = wave_bcast(v, r);
    int i2 = wave_bcast_i(idx, r);
    bool take = (v2 > bv) || (v2 == bv && i2 < bi);
    bv = take ? v2 : bv;
    bi = take ? i2 : bi;
  }
  *best = bv;
  return bi;
}
// the same inside each 16-lane row: every lane gets its own row's winner
JDEV int row_argmax(float v, int idx, float* best) {
  JACO_ARGMAX_STEP(0xB1)
  JACO_ARGMAX_STEP(0x4E)
  JACO_ARGMAX_STEP(0x141)
  JACO_ARGMAX_STEP(0x140)
  *best = v;
  return idx;
}
#undef JACO_ARGMAX_STEP

JDEV unsigned long long wave_clock() { return __builtin_amdgcn_s_memtime(); }   // free-running shader clock
JDEV int grid_size() { return (int)gridDim.x; }
JDEV int jaco_atomic_inc(int* p) { return atomicAdd(p, 1); }
// (release = this workgroup's earlier publications are visible to whoever reads the new count; costs an L2 write-back on
// multi-XCD parts, so workgroups with nothing to publish use the relaxed form)
JDEV int jaco_atomic_dec(int* p, bool release) {
  return release ? __hip_atomic_fetch_sub(p, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_fetch_sub(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// device-scope publish / observe for the light -> heavy work list (the two tiers run concurrently in different workgroups)
// Write-through stores (visible device-wide once acknowledged, no L2 write-back needed afterwards): used for the state of
// an env that is handed to another workgroup.  INVARIANT: every store whose value a hand-off carries to another workgroup
// (state rows, task row, flags, marker / pin poses, cost, remaining) must be st_wt / or_wt, followed by dev_stores_done()
// before the work-list entry is published; the reader does one agent-scope acquire (dev_acquire) after seeing the entry.
JDEV void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
JDEV void st_wt_i(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
JDEV void st_wt_u(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
JDEV void or_wt(unsigned* p, unsigned v) { atomicOr(p, v); }
// dev_stores_done(): every earlier global store of this lane (the write-through state rows of a hand-off) has been
// acknowledged by memory before anything after it issues.  The wait is explicit: a workgroup-scope release fence alone emits
// no s_waitcnt on gfx950 (single-wave workgroup, non-tgsplit), so nothing would order the sc1 stores before the list publish.
JDEV void dev_stores_done() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
JDEV void dev_fence() { __threadfence(); }
JDEV void dev_store_release(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
// polling: coherent relaxed load (no cache invalidation -- an acquire per poll would keep flushing the XCD's L2 under the
// light tier); one acquire fence once the awaited value has been seen
JDEV int dev_load_relaxed(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
JDEV void dev_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
JDEV void wave_sleep() { __builtin_amdgcn_s_sleep(127); }   // ~8k cycles

// Individually rounded product (never contracted into an fma with a neighbouring add): the error-free transformations of the
// compensated state update (physics_kernel.h, comp_add) need the rounded product itself as an operand.
JDEV float fmul_rn(float a, float b) { return __fmul_rn(a, b); }

// (the host emulator fills a workgroup's LDS with garbage here -- real LDS is not zeroed between workgroups -- so that a read of a
// never-written word shows up in the CPU tests; no code on the device)
#define JEMU_POISON(x)

// Pin three already-loaded values in VGPRs here: keeps the optimiser from sinking their loads into a (divergent) branch.
JDEV void keep_loaded(float& a, float& b, float& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }

// Hide the origin of a per-lane value (the lane id) from the optimiser: whatever is derived from it afterwards cannot be hoisted above this point.
JDEV int wave_opaque_i(int v) { asm volatile("" : "+v"(v)); return v; }

// Hide a (wave-uniform) pointer's provenance from the optimiser: stops it from hoisting per-lane model
// loads out of the substep loop and keeping them live (in VGPRs) across the whole loop body.
template <class T>
JDEV const T* opaque_ptr(const T* p) {
  unsigned long long v = (unsigned long long)p;
  asm volatile("" : "+s"(v));
  // the asm hides the pointer's provenance; restate that it is constant global memory so loads through it become global_load / 
  // s_load (a generic pointer would compile to flat_load, which also ties up the LDS counter)
  return (const T*)(const __attribute__((address_space(4))) T*)v;
}

// 32x32 f32 accumulator tile of v_mfma_f32_32x32x2_f32: element (row, col) lives in lane (col + 32 * ((row >> 2) & 1)),
// register (row & 3) + 4 * (row >> 3).  One call adds the rank-2 product A[32x2] * B[2x32]:
// lane l supplies a = A[l & 31][l >> 5] and b = B[l >> 5][l & 31].  Exact f32 fmaf chain (no reduced precision).
typedef float jaco_f32x16 __attribute__((ext_vector_type(16)));
struct acc32x32 { jaco_f32x16 v; };
JDEV void acc_zero(acc32x32& c) {
#pragma unroll
  for (int i = 0; i < 16; i++) c.v[i] = 0.f;
}
JDEV void wave_mfma_32x32x2(float a, float b, acc32x32& c) { c.v = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c.v, 0, 0, 0); }
// 16x16 f32 tile of v_mfma_f32_16x16x4_f32: element (row, col) lives in lane (col + 16 * (row >> 2)), register row & 3.
// One call adds the rank-4 product A[16x4] * B[4x16]: lane l supplies a = A[l & 15][l >> 4] and b = B[l >> 4][l & 15].
typedef float jaco_f32x4 __attribute__((ext_vector_type(4)));
struct acc16x16 { jaco_f32x4 v; };
JDEV void acc_zero(acc16x16& c) { c.v[0] = c.v[1] = c.v[2] = c.v[3] = 0.f; }
JDEV void wave_mfma_16x16x4(float a, float b, acc16x16& c) { c.v = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c.v, 0, 0, 0); }
