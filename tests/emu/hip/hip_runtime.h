// TEST INFRASTRUCTURE: host stand-in for <hip/hip_runtime.h> so that the *unmodified* kernel
// sources under mujoco_jaco_amd/csrc/ compile with g++ for the lockstep wavefront emulator.
// Never part of the product build (the product is compiled by hipcc for gfx950 only).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __restrict__ __restrict
#define __launch_bounds__(...)
#define __constant__ static

extern int emu_cur_lane;
extern int emu_block;
struct EmuThreadIdxX { struct { operator unsigned() const { return (unsigned)emu_cur_lane; } } x; };
struct EmuBlockIdxX { struct { operator unsigned() const { return (unsigned)emu_block; } } x; };
static EmuThreadIdxX threadIdx;
static EmuBlockIdxX blockIdx;

void emu_collective();
static inline void __syncthreads() { emu_collective(); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }

static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline int atomicCAS(int* p, int cmp, int v) { int o = *p; if (o == cmp) *p = v; return o; }
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
