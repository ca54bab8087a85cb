// TEST INFRASTRUCTURE: emulation of mujoco_jaco_amd/csrc/include/jaco/wave_ops.h on the host.
// 64 fibers run the kernel body in lockstep; every cross-lane primitive is a rendezvous.
// Stricter than hardware: a missing wave_sync() around an LDS hand-off shows up as stale data.
#pragma once
#include <hip/hip_runtime.h>

#define JACO_WAVE 64
#define JACO_EMULATED 1
#define JDEV static inline

union EmuWord { float f; int i; unsigned long long u; };
extern EmuWord emu_x[2][64];
extern unsigned emu_cnt[64];
extern long emu_counter[16];   // event counters the kernel source bumps under JACO_EMULATED (0 / 1: all-pairs / list passes of the broadphase, 2: entries tested by list passes,
                              // 3: MPR calls, 4: MPR hits, 5 / 6: support queries of hit / miss calls)

JDEV int lane_id() { return emu_cur_lane; }
JDEV int env_id() { return emu_block; }
JDEV void wave_sync() { emu_collective(); }

JDEV int emu_post_f(float v) { int p = emu_cnt[emu_cur_lane]++ & 1; emu_x[p][emu_cur_lane].f = v; return p; }
JDEV int emu_post_i(int v) { int p = emu_cnt[emu_cur_lane]++ & 1; emu_x[p][emu_cur_lane].i = v; return p; }

JDEV float wave_bcast(float v, int src) { int p = emu_post_f(v); emu_collective(); return emu_x[p][src].f; }
JDEV int wave_bcast_i(int v, int src) { int p = emu_post_i(v); emu_collective(); return emu_x[p][src].i; }
JDEV void st_wt(float* p, float v) { *p = v; }
JDEV void st_wt_i(int* p, int v) { *p = v; }
JDEV void st_wt_u(unsigned* p, unsigned v) { *p = v; }
JDEV void or_wt(unsigned* p, unsigned v) { *p |= v; }
JDEV void dev_stores_done() {}
JDEV void dev_fence() {}
JDEV void dev_store_release(int* p, int v) { *p = v; }
JDEV int dev_load_relaxed(const int* p) { return *p; }
JDEV void dev_acquire() {}
JDEV void wave_sleep() {}
JDEV unsigned long long wave_clock() { return 0ull; }
JDEV int wave_uniform_i(int v) { return wave_bcast_i(v, 0); }
JDEV void keep_loaded(float&, float&, float&) {}
// real LDS is not zeroed between workgroups: fill it with garbage (0x7f7f7f7f: 3.4e38 as a float, 2 139 062 143 as an int) before the kernel body runs
#define JEMU_POISON(x) do { if (emu_cur_lane == 0) memset((void*)&(x), 0x7f, sizeof(x)); emu_collective(); } while (0)
JDEV float fmul_rn(float a, float b) { volatile float p = a * b; return p; }
JDEV int wave_opaque_i(int v) { return v; }
JDEV float wave_shfl(float v, int src) { int p = emu_post_f(v); emu_collective(); return emu_x[p][src & 63].f; }
JDEV int wave_shfl_i(int v, int src) { int p = emu_post_i(v); emu_collective(); return emu_x[p][src & 63].i; }
JDEV unsigned long long wave_ballot(bool pr) {
  int p = emu_post_i(pr ? 1 : 0);
  emu_collective();
  unsigned long long m = 0;
  for (int l = 0; l < 64; l++) if (emu_x[p][l].i) m |= 1ull << l;
  return m;
}
JDEV int wave_prefix_count(unsigned long long mask) { return __builtin_popcountll(mask & ((1ull << emu_cur_lane) - 1)); }
JDEV int popc64(unsigned long long m) { return __builtin_popcountll(m); }
JDEV int ffs64(unsigned long long m) { return m ? __builtin_ctzll(m) : -1; }

// same association order as the DPP butterfly of the device version (bitwise-identical sums)
JDEV float wave_sum(float v) {
  int p = emu_post_f(v);
  emu_collective();
  float row[4];
  for (int r = 0; r < 4; r++) {
    float t[16];
    for (int l = 0; l < 16; l++) t[l] = emu_x[p][16 * r + l].f;
    float a[16];
    for (int l = 0; l < 16; l++) a[l] = t[l] + t[l ^ 1];
    for (int l = 0; l < 16; l++) t[l] = a[l] + a[l ^ 2];
    for (int l = 0; l < 16; l++) a[l] = t[l] + t[(l & 8) | (7 - (l & 7))];
    for (int l = 0; l < 16; l++) t[l] = a[l] + a[15 - l];
    row[r] = t[0];
  }
  return (row[0] + row[1]) + (row[2] + row[3]);
}
JDEV void wave_sum2(float a, float b, float* ra, float* rb) { *ra = wave_sum(a); *rb = wave_sum(b); }
JDEV float wave_max(float v) {
  int p = emu_post_f(v);
  emu_collective();
  float m = emu_x[p][0].f;
  for (int l = 1; l < 64; l++) m = fmaxf(m, emu_x[p][l].f);
  return m;
}
JDEV float wave_min(float v) { return -wave_max(-v); }
JDEV float half_max(float v) {
  int p = emu_post_f(v);
  emu_collective();
  const int h0 = emu_cur_lane & 32;
  float m = emu_x[p][h0].f;
  for (int l = h0 + 1; l < h0 + 32; l++) m = fmaxf(m, emu_x[p][l].f);
  return m;
}
JDEV int wave_argmax(float v, int idx, float* best) {
  int p = emu_post_f(v);
  emu_collective();
  float vs[64];  // copy before the second post: a lane running ahead may reuse buffer p right after it
  for (int l = 0; l < 64; l++) vs[l] = emu_x[p][l].f;
  int q = emu_post_i(idx);
  emu_collective();
  float bv = vs[0];
  int bi = emu_x[q][0].i;
  for (int l = 1; l < 64; l++) {
    float v2 = vs[l];
    int i2 = emu_x[q][l].i;
    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
  }
  *best = bv;
  return bi;
}
JDEV int row_argmax(float v, int idx, float* best) {
  int p = emu_post_f(v);
  emu_collective();
  float vs[64];
  for (int l = 0; l < 64; l++) vs[l] = emu_x[p][l].f;
  int q = emu_post_i(idx);
  emu_collective();
  const int r0 = emu_cur_lane & 48;
  float bv = vs[r0];
  int bi = emu_x[q][r0].i;
  for (int l = r0 + 1; l < r0 + 16; l++) {
    float v2 = vs[l];
    int i2 = emu_x[q][l].i;
    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
  }
  *best = bv;
  return bi;
}

extern int emu_grid;
JDEV int grid_size() { return emu_grid; }
JDEV int jaco_atomic_inc(int* p) { return (*p)++; }
JDEV int jaco_atomic_dec(int* p, bool) { return (*p)--; }

template <class T>
JDEV const T* opaque_ptr(const T* p) {
  asm volatile("" : "+r"(p));
  return p;
}

struct acc16x16 { float v[4]; };
JDEV void acc_zero(acc16x16& c) { for (int i = 0; i < 4; i++) c.v[i] = 0.f; }
JDEV void wave_mfma_16x16x4(float a, float b, acc16x16& c) {
  int p = emu_post_f(a);
  emu_collective();
  float as[64];
  for (int l = 0; l < 64; l++) as[l] = emu_x[p][l].f;
  int q = emu_post_f(b);
  emu_collective();
  int lane = emu_cur_lane, col = lane & 15;
  for (int reg = 0; reg < 4; reg++) {
    int row = 4 * (lane >> 4) + reg;
    float acc = c.v[reg];
    for (int k = 0; k < 4; k++) acc = fmaf(as[row + 16 * k], emu_x[q][col + 16 * k].f, acc);
    c.v[reg] = acc;
  }
}
struct acc32x32 { float v[16]; };
JDEV void acc_zero(acc32x32& c) { for (int i = 0; i < 16; i++) c.v[i] = 0.f; }
JDEV void wave_mfma_32x32x2(float a, float b, acc32x32& c) {
  int p = emu_post_f(a);
  emu_collective();
  float as[64];
  for (int l = 0; l < 64; l++) as[l] = emu_x[p][l].f;
  int q = emu_post_f(b);
  emu_collective();
  int lane = emu_cur_lane, col = lane & 31;
  for (int reg = 0; reg < 16; reg++) {
    int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    float acc = c.v[reg];
    for (int k = 0; k < 2; k++) acc = fmaf(as[row + 32 * k], emu_x[q][col + 32 * k].f, acc);
    c.v[reg] = acc;
  }
}
