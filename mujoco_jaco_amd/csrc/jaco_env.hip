// libjaco_env.so: C ABI (include/jaco_env.h) over the gfx950 physics kernel.
// Host side owns model constants, per-env state rows and launch plumbing; all arithmetic is in
// physics_kernel.h.  There is no CPU fallback: without a HIP device jaco_create fails.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstddef>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/jaco_env.h"
#include "model_blob.h"
#include "physics_kernel.h"

// the physics kernels live in their own translation units (kernels.hip, one per kernel)
#define JACO_DECLARE_LAUNCHER(n) void jaco_launch_kernel_##n(unsigned grid, hipStream_t st, const JacoStepArgs& A);
JACO_DECLARE_LAUNCHER(0) JACO_DECLARE_LAUNCHER(1) JACO_DECLARE_LAUNCHER(2) JACO_DECLARE_LAUNCHER(3)
JACO_DECLARE_LAUNCHER(4) JACO_DECLARE_LAUNCHER(5) JACO_DECLARE_LAUNCHER(6) JACO_DECLARE_LAUNCHER(7) JACO_DECLARE_LAUNCHER(8)
enum { JK_STEP = 0, JK_LISTED = 1, JK_MEDIUM = 2, JK_MEDIUM_DRAIN = 3, JK_HEAVY = 4, JK_HEAVY_DRAIN = 5, JK_HUGE = 6, JK_HUGE_DRAIN = 7, JK_ARM = 8 };
void jaco_launch_kernel(int k, unsigned grid, hipStream_t st, const JacoStepArgs& A) {
  switch (k) {
    case 0: jaco_launch_kernel_0(grid, st, A); break;
    case 1: jaco_launch_kernel_1(grid, st, A); break;
    case 2: jaco_launch_kernel_2(grid, st, A); break;
    case 3: jaco_launch_kernel_3(grid, st, A); break;
    case 4: jaco_launch_kernel_4(grid, st, A); break;
    case 5: jaco_launch_kernel_5(grid, st, A); break;
    case 6: jaco_launch_kernel_6(grid, st, A); break;
    case 7: jaco_launch_kernel_7(grid, st, A); break;
    default: jaco_launch_kernel_8(grid, st, A); break;
  }
}
#define JLAUNCHK(h, k, grid, st, A) do { jaco_launch_kernel((k), (grid), (st), (A)); (h)->nlaunch++; } while (0)

static_assert(JFLAG_CON_OVERFLOW == JACO_FLAG_CON_OVERFLOW && JFLAG_EFC_OVERFLOW == JACO_FLAG_EFC_OVERFLOW &&
                  JFLAG_CAND_OVERFLOW == JACO_FLAG_CAND_OVERFLOW && JFLAG_NAN == JACO_FLAG_NAN &&
                  JFLAG_SOLVER_MAXITER == JACO_FLAG_SOLVER_MAXITER && JFLAG_HEAVY_TIER == JACO_FLAG_HEAVY_TIER,
              "flag bits of the kernel and the public header must agree");

// Queue control words (ints): per tier t (0 medium, 1 heavy, 2 huge) JQ_COUNT + t appended, JQ_TAKEN + t claimed, JQ_LIMIT + t workers
// that start, JQ_RESERVE + t workers that stay when the queue runs dry; JQ_LIGHT light workgroups still running; JQ_ROUTED envs
// queued at once by the light grid (hint > 0), JQ_HINTED + t how many of them per tier (counted by the ordering pass).
#ifndef JACO_HEAVY_GRID
#define JACO_HEAVY_GRID 1024u   // 4 heavy-tier workgroups per CU
#endif
enum { JQ_COUNT = 0, JQ_TAKEN = 3, JQ_LIMIT = 6, JQ_LIGHT = 9, JQ_RESERVE = 10, JQ_ROUTED = 13, JQ_HINTED = 14, JQ_PREV_COUNT = 17, JQ_PREV_HINTED = 20, JQ_LASTMODE = 23, JQ_ROUND1 = 24, JQ_TICKET = 26, JQ_WORDS = 27 };   // (JQ_ROUND1 + t, t = 0, 1: the queue's length before the second drain round, -1 = no second round)

struct JacoHandle {
  JacoModelDev model_host;
  JacoModelDev* model_dev = nullptr;
  float* hull_dev = nullptr;
  float* qpos0_dev = nullptr;   // [nq] reset pose, uploaded once (resets never touch host memory)
  float *qpos = nullptr, *qvel = nullptr, *qacc_ws = nullptr, *sensordata = nullptr, *dbg = nullptr;
  float *qpos_lo = nullptr, *qvel_lo = nullptr;   // low-order parts of the compensated state (physics_kernel.h, comp_add): zero after any state write from outside
  unsigned* flags = nullptr;
  int* stats = nullptr;
  int* remaining = nullptr;
  // tier queues (medium, heavy, huge): lists [3][num_envs] and the control words JQ_* below
  int *qlist = nullptr, *qctl = nullptr;   // both double-buffered: launch N uses lists [qsel][6 B] and control words [qsel][JQ_WORDS]
  int qsel = 0;                               // buffer of the next launch
  bool q_ready = false;                       // ... already prepared by the previous launch's routing kernel
  int merge_prepare = 1;                      // option "merge_prepare": 0 = every launch prepares its own buffer with jaco_prepare_kernel (the launch set of rounds 1-4; comparison / regression tests)
  int* hint = nullptr;                        // [num_envs] tier the env's last step needed
  bool reset_listed = false;                  // jaco_reset in progress: h->order holds the list of the masked envs
  int* routed_mark = nullptr;                 // [num_envs] id of the launch that queued the env for a bigger tier before it started
  int launch_id = 0;
  hipStream_t side[3] = {nullptr, nullptr, nullptr};   // the tiers' resident workers run here, concurrently with the light grid
  hipEvent_t ev_pre = nullptr, ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  int pair_list = 1;   // option "pair_list"
  float* sepdir = nullptr;   // [num_envs][JMAXPAIR][4] separating-direction cache of the hull narrowphase (collision.h)
  int sep_cache = 1;         // option "sep_cache"
  int mpr_pairs = JACO_MPR_PAIRS;   // option "mpr_pairs" (libraries built with -DJACO_MPR_PAIRS=1 only): hull candidates go through MPR two at a time, one per half wave (collision.h mpr_pair2); bit-identical results
  int arm_kernel = 1;        // option "arm_kernel": contact-free steps use the contact-free instantiation (0: the general kernel with its runtime flag, comparison)
  int concurrent = 1, workers = 1024, workers_heavy = 256, workers_huge = 32, tier_return = 1, use_hints = 2, handdown = 1;   // options "concurrent_heavy", "heavy_workers", "hints"
  float *task_rows = nullptr, *cache = nullptr;
  float* terminal = nullptr;   // [num_envs][2] (success flag, wb) latched by every terminal step
  float* terminal_obs = nullptr;   // [num_envs][26] observation of the terminal step (auto_reset)
  float* goal_buf = nullptr;       // recorded reaching goals (jaco_set_init_buffer; library-owned copy), or nullptr
  int goal_n = 0, goal_stride = 0;
  float* marker = nullptr;    // [num_envs][2][12] poses of the "hand" / "subgoal_reach" markers (mocap bodies the task layer moves)
  unsigned* cost = nullptr;   // per env: shader-clock ticks its last step took (>> 4)
  int* order = nullptr;       // launch order of the env-level light kernel: expensive envs first
  unsigned* order_ctl = nullptr;   // histogram / cursors / cost sum / bucket reference of the ordering passes
  int schedule = 1;           // option "schedule": 0 = launch envs in index order
  int auto_reset = 0;         // option "auto_reset"
  int min_nsub_sched = 2;     // option "min_nsub_sched": shortest step (substeps) that gets the resident tier workers
  int min_nsub_order = 8;     // option "min_nsub_order": shortest step that gets the cost-ordered launch (two small launches in front of the light grid); at frame_skip 4 the
                              // ordering does not pay: 19.4-19.7 M env-steps/s with it, 20.1 M without (round 5, tools/gpu_cfg4_options.sh)
  const float* noise = nullptr;
  const float* subgoal = nullptr;   // obs_mode 1: the policy's sub-goal offsets for the "subgoal_reach" marker
  int obs_mode = 0;
  unsigned long long* prof = nullptr;
  std::vector<float> qpos0;
  int num_envs = 0, device = 0, frame_skip = 50, task = 0, disable_contact = 0;
  uint64_t seed = 0;
  std::string err;
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;        // around a whole step launch set (routing .. drains)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> kevents;       // around the light-tier kernel alone (what rocprofv3 reports for it)
  size_t events_used = 0;
  long long nlaunch = 0;   // kernel launches since the last jaco_launch_count
};

static std::string g_create_error;

// every kernel launch of the library goes through here: counted per handle (jaco_launch_count: bench.py's launches_per_step)
#define JLAUNCH(h, ...) do { hipLaunchKernelGGL(__VA_ARGS__); (h)->nlaunch++; } while (0)

#define HIPCHK(h, call)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                  \
      return JACO_EHIP;                                                                              \
    }                                                                                                \
  } while (0)

// Every entry point that touches HIP runs on the handle's own GPU, whatever the caller's current device.
#define ENTER(h) HIPCHK(h, hipSetDevice((h)->device))

extern "C" const char* jaco_last_error(const JacoHandle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static int upload_model(JacoHandle* h) {
  HIPCHK(h, hipMemcpy(h->model_dev, &h->model_host, sizeof(JacoModelDev), hipMemcpyHostToDevice));
  return JACO_OK;
}

extern "C" int jaco_create(const JacoConfig* cfg, JacoHandle** out) {
  if (!cfg || !out || !cfg->model_blob || cfg->num_envs <= 0) { g_create_error = "jaco_create: bad arguments"; return JACO_EINVAL; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
    g_create_error = "jaco_create: no usable HIP device (this library has no CPU path)";
    return JACO_ENODEV;
  }
  JacoHandle* h = new JacoHandle();
  std::vector<float> hull;
  std::string err;
  if (jaco_model_from_blob(cfg->model_blob, cfg->model_blob_size, &h->model_host, &hull, &err)) {
    g_create_error = "jaco_create: " + err;
    delete h;
    return JACO_EINVAL;
  }
  h->num_envs = cfg->num_envs; h->device = cfg->device; h->frame_skip = cfg->frame_skip > 0 ? cfg->frame_skip : 50;
  h->task = cfg->task; h->seed = cfg->seed;
#define CREATECHK(call)                                                                    \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      g_create_error = std::string("jaco_create: " #call ": ") + hipGetErrorString(e_);    \
      jaco_destroy(h);                                                                     \
      return JACO_EHIP;                                                                    \
    }                                                                                      \
  } while (0)
  CREATECHK(hipSetDevice(cfg->device));
  const JacoModelDev& m = h->model_host;
  size_t B = (size_t)cfg->num_envs;
  CREATECHK(hipMalloc(&h->model_dev, sizeof(JacoModelDev)));
  CREATECHK(hipMalloc(&h->hull_dev, hull.size() * sizeof(float) + 16));
  CREATECHK(hipMalloc(&h->qpos, B * m.nq * sizeof(float)));
  CREATECHK(hipMalloc(&h->qvel, B * m.nv * sizeof(float)));
  CREATECHK(hipMalloc(&h->qacc_ws, B * m.nv * sizeof(float)));
  CREATECHK(hipMalloc(&h->qpos_lo, B * m.nq * sizeof(float)));
  CREATECHK(hipMalloc(&h->qvel_lo, B * m.nv * sizeof(float)));
  CREATECHK(hipMemset(h->qpos_lo, 0, B * m.nq * sizeof(float)));
  CREATECHK(hipMemset(h->qvel_lo, 0, B * m.nv * sizeof(float)));
  CREATECHK(hipMalloc(&h->sensordata, B * (m.nsensor > 0 ? m.nsensor : 1) * sizeof(float)));
  CREATECHK(hipMalloc(&h->flags, B * sizeof(unsigned)));
  CREATECHK(hipMalloc(&h->stats, B * 4 * sizeof(int)));
  CREATECHK(hipMalloc(&h->dbg, JDBG_SIZE * sizeof(float)));
  CREATECHK(hipMalloc(&h->remaining, B * sizeof(int)));
  CREATECHK(hipMalloc(&h->qlist, 2 * 6 * B * sizeof(int)));   // (per tier 2 B slots: an env can come by twice, see the second drain round)
  CREATECHK(hipMalloc(&h->qctl, 2 * JQ_WORDS * sizeof(int)));
  CREATECHK(hipMemset(h->qctl, 0, 2 * JQ_WORDS * sizeof(int)));
  // (the separating-direction cache, 16 B x JMAXPAIR per env = 0.8 GB at 65 536 envs, is allocated by the first launch that uses it: launch_step)
  CREATECHK(hipMalloc(&h->hint, B * sizeof(int)));
  CREATECHK(hipMemset(h->hint, 0, B * sizeof(int)));
  CREATECHK(hipMalloc(&h->routed_mark, B * sizeof(int)));
  CREATECHK(hipMemset(h->routed_mark, 0, B * sizeof(int)));
  {
    int lo = 0, hi = 0;
    CREATECHK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // (hi = numerically lowest = highest priority)
    CREATECHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    CREATECHK(hipEventCreateWithFlags(&h->ev_pre, hipEventDisableTiming));
    for (int t = 0; t < 3; t++) {
      CREATECHK(hipStreamCreateWithPriority(&h->side[t], hipStreamNonBlocking, hi));
      CREATECHK(hipEventCreateWithFlags(&h->ev_join[t], hipEventDisableTiming));
    }
  }
  CREATECHK(hipMemset(h->remaining, 0, B * sizeof(int)));
  CREATECHK(hipMalloc(&h->marker, B * 24 * sizeof(float)));
  CREATECHK(hipMalloc(&h->cost, B * sizeof(unsigned)));
  CREATECHK(hipMalloc(&h->order, B * sizeof(int)));
  CREATECHK(hipMalloc(&h->order_ctl, 72 * sizeof(unsigned)));
  CREATECHK(hipMemset(h->order_ctl, 0, 72 * sizeof(unsigned)));
  CREATECHK(hipMemset(h->cost, 0, B * sizeof(unsigned)));
  CREATECHK(hipMalloc(&h->task_rows, B * JTASK_N * sizeof(float)));
  CREATECHK(hipMalloc(&h->cache, B * JCACHE_N * sizeof(float)));
  CREATECHK(hipMalloc(&h->terminal, B * 2 * sizeof(float)));
  CREATECHK(hipMemset(h->terminal, 0, B * 2 * sizeof(float)));
  CREATECHK(hipMalloc(&h->terminal_obs, B * 26 * sizeof(float)));
  CREATECHK(hipMemset(h->terminal_obs, 0, B * 26 * sizeof(float)));
  CREATECHK(hipMemset(h->task_rows, 0, B * JTASK_N * sizeof(float)));
  CREATECHK(hipMemset(h->cache, 0, B * JCACHE_N * sizeof(float)));
  CREATECHK(hipMemcpy(h->model_dev, &h->model_host, sizeof(JacoModelDev), hipMemcpyHostToDevice));
  CREATECHK(hipMemcpy(h->hull_dev, hull.data(), hull.size() * sizeof(float), hipMemcpyHostToDevice));
  CREATECHK(hipMemset(h->flags, 0, B * sizeof(unsigned)));
  CREATECHK(hipMemset(h->stats, 0, B * 4 * sizeof(int)));
  CREATECHK(hipMemset(h->sensordata, 0, B * (m.nsensor > 0 ? m.nsensor : 1) * sizeof(float)));
  // qpos0 from the raw view of the blob (same joint order as the fused view)
  {
    const char* p = (const char*)cfg->model_blob;
    int n = *(const int32_t*)(p + 8);
    size_t off = 16;
    for (int i = 0; i < n; i++) {
      std::string name(p + off, strnlen(p + off, 32));
      int code = *(const int32_t*)(p + off + 32), count = *(const int32_t*)(p + off + 36);
      off += 40;
      size_t nb = (size_t)count * (code == 0 ? 8 : 4);
      if (name == "qpos0" && code == 0) {
        h->qpos0.resize(count);
        for (int k = 0; k < count; k++) h->qpos0[k] = (float)((const double*)(p + off))[k];
      }
      off += nb + ((8 - nb % 8) % 8);
    }
    if ((int)h->qpos0.size() != m.nq) { g_create_error = "jaco_create: qpos0 missing"; jaco_destroy(h); return JACO_EINVAL; }
    CREATECHK(hipMalloc(&h->qpos0_dev, m.nq * sizeof(float)));
    CREATECHK(hipMemcpy(h->qpos0_dev, h->qpos0.data(), m.nq * sizeof(float), hipMemcpyHostToDevice));
  }
  *out = h;
  int rc = jaco_reset_state(h, nullptr);
  if (rc) { g_create_error = h->err; jaco_destroy(h); *out = nullptr; return rc; }
  (void)hipDeviceSynchronize();
  return JACO_OK;
}

extern "C" int jaco_destroy(JacoHandle* h) {
  if (!h) return JACO_EINVAL;
  (void)hipSetDevice(h->device);
  for (auto& e : h->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto& e : h->kevents) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (int t = 0; t < 3; t++) {
    if (h->side[t]) { (void)hipStreamSynchronize(h->side[t]); (void)hipStreamDestroy(h->side[t]); }
    if (h->ev_join[t]) (void)hipEventDestroy(h->ev_join[t]);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_pre) (void)hipEventDestroy(h->ev_pre);
  void* ptrs[] = {h->model_dev, h->hull_dev, h->qpos, h->qvel, h->qacc_ws, h->qpos_lo, h->qvel_lo, h->sensordata, h->flags, h->stats, h->dbg, h->prof, h->remaining, h->qlist, h->qctl, h->hint, h->routed_mark, h->task_rows, h->cache, h->cost, h->order, h->marker, h->order_ctl, h->qpos0_dev, h->sepdir, h->terminal, h->terminal_obs, h->goal_buf};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete h;
  return JACO_OK;
}

extern "C" int jaco_dims(const JacoHandle* h, int* nq, int* nv, int* nu, int* nsensor, int* nobs, int* nact) {
  if (!h) return JACO_EINVAL;
  if (nq) *nq = h->model_host.nq;
  if (nv) *nv = h->model_host.nv;
  if (nu) *nu = h->model_host.nu;
  if (nsensor) *nsensor = h->model_host.nsensor;
  if (nobs) *nobs = 26;
  if (nact) *nact = (h->task == JACO_TASK_REACHING || h->task == JACO_TASK_PUSHING) ? 6 : 7;
  return JACO_OK;
}
extern "C" int jaco_num_envs(const JacoHandle* h) { return h ? h->num_envs : JACO_EINVAL; }

extern "C" int jaco_set_state(JacoHandle* h, const float* qpos, const float* qvel, const float* qacc_ws, void* stream) {
  if (!h) return JACO_EINVAL;
  ENTER(h);
  hipStream_t st = (hipStream_t)stream;
  size_t B = h->num_envs;
  if (qpos) HIPCHK(h, hipMemcpyAsync(h->qpos, qpos, B * h->model_host.nq * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (qpos) HIPCHK(h, hipMemsetAsync(h->hint, 0, B * sizeof(int), st));
  if (qpos) HIPCHK(h, hipMemsetAsync(h->qpos_lo, 0, B * h->model_host.nq * sizeof(float), st));   // the state handed in is exactly the floats
  if (qvel) HIPCHK(h, hipMemsetAsync(h->qvel_lo, 0, B * h->model_host.nv * sizeof(float), st));   // a state from outside starts in the light tier: results depend on the state alone
  if (qvel) HIPCHK(h, hipMemcpyAsync(h->qvel, qvel, B * h->model_host.nv * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (qacc_ws) HIPCHK(h, hipMemcpyAsync(h->qacc_ws, qacc_ws, B * h->model_host.nv * sizeof(float), hipMemcpyDeviceToDevice, st));
  return JACO_OK;
}
extern "C" int jaco_get_state(JacoHandle* h, float* qpos, float* qvel, float* qacc_ws, void* stream) {
  if (!h) return JACO_EINVAL;
  ENTER(h);
  hipStream_t st = (hipStream_t)stream;
  size_t B = h->num_envs;
  if (qpos) HIPCHK(h, hipMemcpyAsync(qpos, h->qpos, B * h->model_host.nq * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (qvel) HIPCHK(h, hipMemcpyAsync(qvel, h->qvel, B * h->model_host.nv * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (qacc_ws) HIPCHK(h, hipMemcpyAsync(qacc_ws, h->qacc_ws, B * h->model_host.nv * sizeof(float), hipMemcpyDeviceToDevice, st));
  return JACO_OK;
}

__global__ void jaco_fill_rows_kernel(float* dst, const float* row, int n, int nenv) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)n * nenv) dst[i] = row[i % n];
}
extern "C" int jaco_reset_state(JacoHandle* h, void* stream) {
  if (!h) return JACO_EINVAL;
  ENTER(h);
  hipStream_t st = (hipStream_t)stream;
  const JacoModelDev& m = h->model_host;
  size_t B = h->num_envs;
  size_t total = B * m.nq;
  JLAUNCH(h, jaco_fill_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, h->qpos, h->qpos0_dev, m.nq, (int)B);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemsetAsync(h->qvel, 0, B * m.nv * sizeof(float), st));
  HIPCHK(h, hipMemsetAsync(h->qacc_ws, 0, B * m.nv * sizeof(float), st));
  HIPCHK(h, hipMemsetAsync(h->qpos_lo, 0, B * m.nq * sizeof(float), st));
  HIPCHK(h, hipMemsetAsync(h->qvel_lo, 0, B * m.nv * sizeof(float), st));
  HIPCHK(h, hipMemsetAsync(h->hint, 0, B * sizeof(int), st));
  // markers back to their XML rest pose (sim.reset() restores mocap_pos / mocap_quat)
  const float* rest = (const float*)((const char*)h->model_dev + offsetof(JacoModelDev, marker_rest));
  JLAUNCH(h, jaco_fill_rows_kernel, dim3((unsigned)((B * 24 + 255) / 256)), dim3(256), 0, st, h->marker, rest, 24, (int)B);
  HIPCHK(h, hipGetLastError());
  return JACO_OK;
}

#define JACO_PLACING_HOLD_SUBSTEPS 150   // reset_frame_skip (env_mujoco_util.py:114)
struct EnvIO { int mode = 0; const float* action = nullptr; float* obs = nullptr; float* reward = nullptr; unsigned char* done = nullptr; const unsigned char* mask = nullptr;
               bool listed = false; };   // listed: h->order[0 .. order_ctl[67]) holds the envs of `mask` (written by jaco_reset_kernel): launch a small grid over that list

// Launch order for the next env step: envs sorted by the cost of their previous step, most expensive first (32 buckets of
// 1/8 of the mean cost).  An env step is ~1.5 ms of one wavefront and the expensive ones (hull-hull narrowphase, the
// controller's pseudo-inverse branch) take 2-8x the mean; started last they would leave the chip idle behind them.
// Costs persist from step to step (contact state, arm configuration), so last step's cost predicts this step's.
// Two small multi-block passes over the cost array (histogram, then scatter), wave-aggregated global atomics; the bucket
// width comes from the mean cost of the launch before (one launch stale).  The order within a bucket is arbitrary: envs
// are independent.  Scratch `oc` (unsigned[72]): [0..31] histogram, [32..63] scatter cursors, [64..65] cost sum of this
// launch (u64), [66] mean cost used as the bucket reference; [0..65] are zeroed by jaco_prepare_kernel.
static __device__ __forceinline__ unsigned jaco_cost_bucket(unsigned c, unsigned ref) {
  unsigned long long q = (unsigned long long)c * 8ull / ref;
  return q > 30ull ? 30u : (unsigned)q;   // (bucket 31 belongs to the envs that start in a bigger tier: they lead the order)
}
__global__ __launch_bounds__(1024) void jaco_order_hist_kernel(const unsigned* cost, unsigned* oc, int n, const int* mark, int launch_id) {
  __shared__ unsigned lh[32];
  __shared__ unsigned long long lsum;
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x), lane = (int)threadIdx.x & 63;
  if (threadIdx.x < 32) lh[threadIdx.x] = 0u;
  if (threadIdx.x == 0) lsum = 0ull;
  __syncthreads();
  const bool valid = i < n;
  const unsigned ref = oc[66] + 1u;
  const unsigned c = valid ? cost[i] : 0u;
  const bool routed = valid && mark && mark[i] == launch_id;
  const unsigned b = routed ? 31u : jaco_cost_bucket(c, ref);   // (envs already queued for a bigger tier: their light workgroups only have to leave)
  unsigned long long sum = routed ? 0u : c;
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) atomicAdd(&lsum, sum);
  unsigned long long todo = __ballot(valid);
  while (todo) {   // one LDS atomic per distinct bucket in the wave
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned lb = (unsigned)__shfl((int)b, leader, 64);
    const unsigned long long same = __ballot(valid && b == lb);
    if (lane == leader) atomicAdd(&lh[lb], (unsigned)__popcll(same));
    todo &= ~same;
  }
  __syncthreads();
  if (threadIdx.x < 32 && lh[threadIdx.x]) atomicAdd(&oc[threadIdx.x], lh[threadIdx.x]);   // one global atomic per bucket per block
  if (threadIdx.x == 32) atomicAdd(reinterpret_cast<unsigned long long*>(oc + 64), lsum);
}
__global__ __launch_bounds__(1024) void jaco_order_scatter_kernel(const unsigned* cost, unsigned* oc, int* order, int n, const int* mark, int launch_id) {
  __shared__ unsigned base[32], lh[32], gb[32];
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x), lane = (int)threadIdx.x & 63;
  if (threadIdx.x < 32) lh[threadIdx.x] = 0u;
  if (threadIdx.x == 0) { unsigned p = 0; for (int b = 31; b >= 0; b--) { base[b] = p; p += oc[b]; } }   // most expensive bucket first
  __syncthreads();
  const bool valid = i < n;
  const unsigned ref = oc[66] + 1u;
  const unsigned b = (valid && mark && mark[i] == launch_id) ? 31u : jaco_cost_bucket(valid ? cost[i] : 0u, ref);
  unsigned off = 0;   // position of this env among the block's members of its bucket
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned lb = (unsigned)__shfl((int)b, leader, 64);
    const unsigned long long same = __ballot(valid && b == lb);
    unsigned pos = 0;
    if (lane == leader) pos = atomicAdd(&lh[lb], (unsigned)__popcll(same));
    pos = (unsigned)__shfl((int)pos, leader, 64);
    if (valid && b == lb) off = pos + __popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  __syncthreads();
  if (threadIdx.x < 32) gb[threadIdx.x] = lh[threadIdx.x] ? base[threadIdx.x] + atomicAdd(&oc[32 + threadIdx.x], lh[threadIdx.x]) : 0u;   // reserve the block's range
  __syncthreads();
  if (valid) order[gb[b] + off] = i;
  // the last block to finish turns this launch's cost sum into the bucket reference of the NEXT launch (was a one-thread launch of its own):
  // every block has read oc[66] by now (hist and scatter must bucket with the same reference)
  __shared__ int last_block;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last_block = atomicAdd(&oc[68], 1u) == gridDim.x - 1u;
  __syncthreads();
  if (last_block && threadIdx.x == 0) {
    const unsigned long long total = atomicAdd(reinterpret_cast<unsigned long long*>(oc + 64), 0ull);
    oc[66] = (unsigned)(total / (unsigned long long)(n > 0 ? n : 1));
    oc[68] = 0u;
  }
}
// ... and sizes this launch's workers from what the ordering pass has just counted: per tier, the envs that start there
// (a medium worker serves ~8 of them in a third of a step, a heavy one ~4, a huge one ~2) plus a reserve for overflows that
// only show up during the step (a tenth of the previous launch's total demand)
// Envs whose previous step ended in a bigger tier go there at once: queued here, before the launch, so that the tier's workers
// find them when they start; the hint is consumed (it is re-earned during the step by whichever tier is really needed).
// Between the two drain rounds: every workgroup of a drain ends on one claim beyond the queue's end, so `taken` has overshot the
// end of round one by exactly the grid size; the second round starts where the first one really stopped.
__global__ void jaco_drain_round2_kernel(int* ctl, int medium_grid, int heavy_grid) {
  ctl[JQ_TAKEN + 0] -= medium_grid;
  ctl[JQ_TAKEN + 1] -= heavy_grid;
  ctl[JQ_ROUND1 + 0] = ctl[JQ_TAKEN + 0];   // = the medium queue's length when its first drain ended
  ctl[JQ_ROUND1 + 1] = ctl[JQ_COUNT + 1];   // (the heavy queue only grows again in the second medium drain)
}
// Queue state is double-buffered: launch N works on buffer N & 1 (control words + lists).  What used to be a launch of its own in front of
// every step -- lists back to -1, counters zeroed, the workers sized from the previous launch's demand -- is now done for the NEXT launch's
// buffer by this launch's routing kernel (nobody touches that buffer during this launch set: the launch that used it is complete in
// stream order), so an env step in a row of env steps starts with ONE small kernel instead of two.  Launches that do not route (reset-time
// forward passes, ctrl level without hints) and the first launch after one of those run jaco_prepare_kernel on their own buffer.
//
// Sizing the workers (one thread): how many of the launched workers of each tier start, and how many stay resident when the queue runs
// dry, follows the previous launch's demand for that tier (an idle worker still holds LDS the light grid could use): a medium worker
// serves an env in ~1/20 of a step, a heavy / huge one in ~1/8 - 1/4.  `prev` = control words of the launch before (final), `ctl` = this launch's.
static __device__ void jaco_queue_limits(const int* prev, int* ctl, int mode, int wm, int wh, int wg, int light_wgs) {
  // (the demand that counts is that of the last real step, not of a reset-time forward pass in between)
  for (int t = 0; t < 3; t++) {
    if (prev[JQ_LASTMODE] <= 1) {
      // (what came by a second time -- handed down by the heavy drain, overflowed again after that -- is not new demand)
      ctl[JQ_PREV_COUNT + t] = (t < 2 && prev[JQ_ROUND1 + t] >= 0) ? prev[JQ_ROUND1 + t] : prev[JQ_COUNT + t];
      ctl[JQ_PREV_HINTED + t] = prev[JQ_HINTED + t];
    } else { ctl[JQ_PREV_COUNT + t] = prev[JQ_PREV_COUNT + t]; ctl[JQ_PREV_HINTED + t] = prev[JQ_PREV_HINTED + t]; }
  }
  ctl[JQ_LASTMODE] = mode; ctl[JQ_ROUND1] = -1; ctl[JQ_ROUND1 + 1] = -1;
  const int* pc = ctl + JQ_PREV_COUNT;
  const int want[3] = {16 + pc[0] / 10, 8 + pc[1] / 4, 2 + pc[2] / 2}, cap[3] = {wm, wh, wg};
  for (int t = 0; t < 3; t++) {
    const int w = want[t] < cap[t] ? want[t] : cap[t];
    // reserve: overflows that only show up during a step = last step's demand minus what it had queued at once
    int late = pc[t] - ctl[JQ_PREV_HINTED + t];
    late = late < 0 ? 0 : late;
    const int base = t == 0 ? 8 : (t == 1 ? 4 : 1);   // (a resident heavy / huge worker holds the LDS of 3 / 5 light envs)
    ctl[JQ_LIMIT + t] = w; ctl[JQ_RESERVE + t] = base + late / 10;
  }
  ctl[JQ_LIGHT] = light_wgs;
}
// a buffer ready for use: nothing queued, nothing claimed (JQ_PREV_*, JQ_LASTMODE, JQ_LIMIT, JQ_RESERVE, JQ_LIGHT are written by jaco_queue_limits)
static __device__ void jaco_queue_clear(int* ctl) {
  for (int t = 0; t < 3; t++) { ctl[JQ_COUNT + t] = 0; ctl[JQ_TAKEN + t] = 0; ctl[JQ_HINTED + t] = 0; }
  ctl[JQ_ROUTED] = 0; ctl[JQ_TICKET] = 0; ctl[JQ_ROUND1] = -1; ctl[JQ_ROUND1 + 1] = -1;
}
// Envs whose previous step ended in a bigger tier go there at once: queued here, before the launch, so that the tier's workers find them
// when they start; the hint is consumed (it is re-earned during the step by whichever tier is really needed).  The last block to finish
// sizes this launch's workers: per tier the previous launch's demand (jaco_queue_limits), then the envs that start there (a medium
// worker serves ~8 of them in a third of a step, a heavy one ~4, a huge one ~2) plus the reserve for overflows that only show up during
// the step -- and clears the other buffer's control words for the next launch.  Every thread resets its share of the other buffer's lists.
__global__ void jaco_route_kernel(int* hint, int* mark, int launch_id, int* lists, int* ctl, int* remaining, unsigned* cost, int n, int nsub, int wm, int wh, int wg,
                                  int* next_lists, int* other_ctl, unsigned* oc, int mode) {
  const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (e < 66) oc[e] = 0u;
  const int t = e < n ? hint[e] : 0;
  if (t > 0) {
    hint[e] = 0;
    mark[e] = launch_id;
    remaining[e] = nsub;
    cost[e] = 0u;   // (the tiers add what they spend)
    lists[(size_t)(t - 1) * 2 * n + atomicAdd(&ctl[JQ_COUNT + t - 1], 1)] = e;
  }
  __shared__ int last_block;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last_block = atomicAdd(&ctl[JQ_TICKET], 1) == (int)gridDim.x - 1;
  // the other buffer's lists back to "nothing published" (behind the ticket: these stores need no fence, the end of the kernel makes them visible)
  if (e < n) for (int j = 0; j < 6; j++) next_lists[(size_t)j * n + e] = -1;
  __syncthreads();
  if (!last_block) return;
  // (one thread doing this on global memory is ~50 dependent round trips = 15 us in front of every step: the two buffers' control words come in
  //  and go out in parallel, the arithmetic in between runs on their copies in LDS)
  __shared__ int sp[JQ_WORDS], sc[JQ_WORDS];
  if (threadIdx.x < JQ_WORDS) {
    sp[threadIdx.x] = other_ctl[threadIdx.x];
    sc[threadIdx.x] = atomicAdd(&ctl[threadIdx.x], 0);   // (device-scope read: the other blocks' appends)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    jaco_queue_limits(sp, sc, mode, wm, wh, wg, n);
    const int cap[3] = {wm, wh, wg}, per[3] = {8, 4, 2};
    int routed = 0;
    for (int q = 0; q < 3; q++) {
      const int hinted = sc[JQ_COUNT + q];
      sc[JQ_HINTED + q] = hinted;
      routed += hinted;
      const int reserve = sc[JQ_RESERVE + q] < cap[q] ? sc[JQ_RESERVE + q] : cap[q];
      const int want = reserve + (hinted + per[q] - 1) / per[q];
      sc[JQ_LIMIT + q] = want < cap[q] ? want : cap[q];
      sc[JQ_RESERVE + q] = reserve;
    }
    sc[JQ_ROUTED] = routed;
    sc[JQ_LIGHT] = n - routed;   // the light workgroups of queued envs leave without being counted
    sc[JQ_TICKET] = 0;
    jaco_queue_clear(sp);   // (its demand figures have been carried over: the next launch's buffer)
  }
  __syncthreads();
  if (threadIdx.x < JQ_WORDS) { ctl[threadIdx.x] = sc[threadIdx.x]; other_ctl[threadIdx.x] = sp[threadIdx.x]; }
}
// queue preparation as a launch of its own (launches that do not route, and the first launch after one of those): this launch's lists = -1,
// its counters zeroed, its workers sized from the launch before
__global__ void jaco_prepare_kernel(int* ctl, const int* prev_ctl, int* lists, int n, int wm, int wh, int wg, unsigned* oc, int mode, int light_wgs) {
  int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < 6 * n) lists[i] = -1;
  if (i < 66) oc[i] = 0u;
  if (i == 0) {
    jaco_queue_clear(ctl);
    jaco_queue_limits(prev_ctl, ctl, mode, wm, wh, wg, light_wgs);
  }
}

static int launch_step(JacoHandle* h, const float* ctrl, int nsub, hipStream_t st, float* dbg, int dbg_env, const EnvIO& io = EnvIO()) {
  if ((!ctrl && io.mode == 0) || nsub <= 0) { h->err = "jaco_physics_step: bad arguments"; return JACO_EINVAL; }
#if JNV > 21
  // the two-arm layout serves the sim-interface tier (jaco_physics_step and the state accessors).  The env tier is one arm's task layer; the
  // reference's own env loop is single-robot too (_step_simulation stacks ONE gripper command onto the controller output,
  // env_mujoco_util.py:73-83: 15 values for the dual model's 18 controls)
  if (io.mode != 0) { h->err = "this build of the library (two-arm layout) steps models at the ctrl level only: jaco_physics_step / jaco_get_state / jaco_set_state"; return JACO_EINVAL; }
#endif
  ENTER(h);
  JacoStepArgs A{};
  A.model = h->model_dev; A.hull = h->hull_dev; A.qpos = h->qpos; A.qvel = h->qvel; A.qacc_ws = h->qacc_ws; A.qpos_lo = h->qpos_lo; A.qvel_lo = h->qvel_lo;
  A.ctrl = ctrl ? ctrl : h->qvel;   // env modes compute ctrl in-kernel; the pointer only has to be readable
  A.sensordata = h->sensordata; A.flags = h->flags; A.stats = h->stats; A.nenv = h->num_envs; A.nsub = nsub;
  if (h->sep_cache && !h->disable_contact && h->model_host.npair > 0 && !h->sepdir) {   // first use: a handle that never runs contacts with the cache on never pays for it
    const size_t bytes = (size_t)h->num_envs * JMAXPAIR * 4 * sizeof(float);
    HIPCHK(h, hipMalloc(&h->sepdir, bytes));
    HIPCHK(h, hipMemsetAsync(h->sepdir, 0, bytes, st));   // (entries are re-validated by a support query before use: zeros are "no direction known")
  }
  A.disable_contact = h->disable_contact; A.no_pairlist = !h->pair_list; A.mpr_pairs = h->mpr_pairs; A.sepdir = h->sep_cache ? h->sepdir : nullptr; A.no_tier_return = !h->tier_return; A.dbg = dbg; A.dbg_env = dbg_env; A.prof = h->prof;
  // this launch's queue buffer (lists + control words) and the other one: the previous launch's, and the next launch's
  int* const qctl = h->qctl + h->qsel * JQ_WORDS;
  int* const qctl_other = h->qctl + (h->qsel ^ 1) * JQ_WORDS;
  int* const qlist = h->qlist + (size_t)h->qsel * 6 * h->num_envs;
  int* const qlist_other = h->qlist + (size_t)(h->qsel ^ 1) * 6 * h->num_envs;
  A.remaining = h->remaining; A.light_left = qctl + JQ_LIGHT; A.hint = h->use_hints ? h->hint : nullptr; A.hint_mode = h->use_hints;
  for (int t = 0; t < 3; t++) { A.q[t].list = qlist + (size_t)t * 2 * h->num_envs; A.q[t].count = qctl + JQ_COUNT + t; A.q[t].taken = qctl + JQ_TAKEN + t; A.q[t].limit = qctl + JQ_LIMIT + t; A.q[t].reserve = qctl + JQ_RESERVE + t; }
  A.routed_mark = nullptr; A.launch_id = ++h->launch_id;
  A.env_mode = io.mode; A.task_id = h->task; A.nact = (h->task == JACO_TASK_REACHING || h->task == JACO_TASK_PUSHING) ? 6 : 7; A.seed = h->seed;
  A.task = h->task_rows; A.cache = h->cache; A.action = io.action; A.noise = h->noise; A.obs_mode = h->obs_mode; A.subgoal = h->subgoal; A.obs = io.obs; A.reward = io.reward; A.done = io.done; A.terminal = h->terminal; A.terminal_obs = h->terminal_obs; A.goal_buf = h->goal_buf; A.goal_n = h->goal_n; A.goal_stride = h->goal_stride; A.mask = io.mask; A.marker = h->marker;
  A.cost = h->cost;
  // auto-reset folds draws + sim.forward() + observation into the step wave: the tasks whose reset is nothing more (placing holds the
  // object for 150 substeps, grasping pre-reaches: those keep the explicit jaco_reset)
  A.auto_reset = h->auto_reset && io.mode == 1 && (h->task == JACO_TASK_PICKING || h->task == JACO_TASK_REACHING || h->task == JACO_TASK_PICKANDPLACE || h->task == JACO_TASK_PUSHING);
  A.qpos0 = h->qpos0_dev;
  const bool reorder = io.mode == 1 && h->schedule && nsub >= h->min_nsub_order && h->num_envs >= 4096;
  std::pair<hipEvent_t, hipEvent_t>*ev = nullptr, *kev = nullptr;
  if (h->timing && io.mode <= 1) {   // (the masked forward passes of resets are not the kernel being measured)
    if (h->events_used == h->events.size()) {
      hipEvent_t a, b;
      HIPCHK(h, hipEventCreate(&a));
      HIPCHK(h, hipEventCreate(&b));
      h->events.emplace_back(a, b);
      HIPCHK(h, hipEventCreate(&a));
      HIPCHK(h, hipEventCreate(&b));
      h->kevents.emplace_back(a, b);
    }
    kev = &h->kevents[h->events_used];
    ev = &h->events[h->events_used++];
    HIPCHK(h, hipEventRecord(ev->first, st));
  }
  // Contact-free step (option disable_contact, or a model without a collidable pair): its own lean kernel, alone -- no capacity can
  // overflow (at most one limit row per joint), so there is nothing to route, order, serve or drain.
  if (h->arm_kernel && (h->disable_contact || h->model_host.npair == 0) && io.mode <= 1) {
    A.disable_contact = 1; A.hint = nullptr;
    if (kev) HIPCHK(h, hipEventRecord(kev->first, st));
    JLAUNCHK(h, JK_ARM, (unsigned)h->num_envs, st, A);
    if (kev) HIPCHK(h, hipEventRecord(kev->second, st));
    HIPCHK(h, hipGetLastError());
    if (ev) HIPCHK(h, hipEventRecord(ev->second, st));
    return JACO_OK;
  }
  // masked reset: the reset kernel has listed its envs; a small grid walks that list instead of 65 536 workgroups finding out one by
  // one that they have nothing to do (0.9 ms per launch)
  unsigned light_grid = (unsigned)h->num_envs;
  if (io.listed && io.mask) { A.order = h->order; A.nslots = reinterpret_cast<const int*>(h->order_ctl + 67); light_grid = light_grid < 1024u ? light_grid : 1024u; }
  // Light grid for every env.  An env that overflows the light capacities is handed over (queue) to the medium tier, and on
  // to the heavy / huge tiers if need be; an env whose previous step ended in a bigger tier is queued there straight away.
  // Each tier's persistent worker workgroups run concurrently on their own higher-priority stream: started just before the
  // light grid they are resident from the beginning (a bigger workgroup needs more LDS than a finishing light workgroup
  // frees and would otherwise starve behind the light grid, leaving serial tails of several ms per env step).  The drains that
  // follow in stream order serve whatever the workers did not (all of it when concurrency is off: full grids, which is also
  // what carries the load when most envs overflow).
  // (the contact-free path above, take_action and terminal_inspection use no queue: they leave the buffers as they are)
  const bool routes = A.hint && io.mode <= 1 && !(io.listed && io.mask);
  if (!h->q_ready || !routes) {   // this launch's buffer has not been prepared by the launch before
    JLAUNCH(h, jaco_prepare_kernel, dim3((unsigned)((6 * h->num_envs + 255) / 256)), dim3(256), 0, st, qctl, qctl_other, qlist, h->num_envs, h->workers, h->workers_heavy, h->workers_huge, h->order_ctl, io.mode, (int)light_grid);
    HIPCHK(h, hipGetLastError());
  }
  if (routes) {   // queue the envs whose last step ended in a bigger tier, size the tiers' workers, prepare the next launch's buffer
    JLAUNCH(h, jaco_route_kernel, dim3((unsigned)((h->num_envs + 255) / 256)), dim3(256), 0, st, h->hint, h->routed_mark, A.launch_id, qlist, qctl, h->remaining, h->cost, h->num_envs, nsub,
            h->workers, h->workers_heavy, h->workers_huge, qlist_other, qctl_other, h->order_ctl, io.mode);
    A.routed_mark = h->routed_mark;
  }
  h->q_ready = routes && h->merge_prepare;   // (the routing kernel has prepared the other buffer either way; with the option off the next launch prepares it again)
  h->qsel ^= 1;
  // The resident workers go first: their workgroups need 20 - 68 KB of LDS on one CU, and once the light grid (13 KB per workgroup,
  // 65 536 of them) has filled the chip such a hole only opens when the grid runs out -- a huge-tier env queued at t = 0 would then
  // start its 10 ms of work when everything else is done.  Launched before the ordering pass, they have three small kernels of
  // head start on the light grid (and the event record below adds a barrier packet in front of it).
  // (short steps too: at frame_skip 4 the serial drains behind the light grid were 1.2 ms of a 5.1 ms step, profiles/r03_trace_fs4.txt)
  const bool conc = h->concurrent && io.mode == 1 && nsub >= h->min_nsub_sched && h->num_envs >= 4096;
  if (conc) {
    HIPCHK(h, hipEventRecord(h->ev_fork, st));
    for (int t = 2; t >= 0; t--) HIPCHK(h, hipStreamWaitEvent(h->side[t], h->ev_fork, 0));
    JLAUNCHK(h, JK_HUGE, (unsigned)h->workers_huge, h->side[2], A);
    JLAUNCHK(h, JK_HEAVY, (unsigned)h->workers_heavy, h->side[1], A);
    JLAUNCHK(h, JK_MEDIUM, (unsigned)h->workers, h->side[0], A);
    HIPCHK(h, hipGetLastError());
    for (int t = 0; t < 3; t++) HIPCHK(h, hipEventRecord(h->ev_join[t], h->side[t]));
  }
  if (reorder) {
    const unsigned ob = (unsigned)((h->num_envs + 1023) / 1024);
    JLAUNCH(h, jaco_order_hist_kernel, dim3(ob), dim3(1024), 0, st, h->cost, h->order_ctl, h->num_envs, A.routed_mark, A.launch_id);
    JLAUNCH(h, jaco_order_scatter_kernel, dim3(ob), dim3(1024), 0, st, h->cost, h->order_ctl, h->order, h->num_envs, A.routed_mark, A.launch_id);
    HIPCHK(h, hipGetLastError());
    A.order = h->order;
  }
  if (conc) HIPCHK(h, hipEventRecord(h->ev_pre, st));
  if (kev) HIPCHK(h, hipEventRecord(kev->first, st));
  if (A.nslots || io.mode >= 2) JLAUNCHK(h, JK_LISTED, light_grid, st, A);   // (resets: forward passes, placing hold)
  else JLAUNCHK(h, JK_STEP, light_grid, st, A);
  if (kev) HIPCHK(h, hipEventRecord(kev->second, st));
  HIPCHK(h, hipGetLastError());
  if (conc) for (int t = 0; t < 3; t++) HIPCHK(h, hipStreamWaitEvent(st, h->ev_join[t], 0));   // (joining each tier's workers only in front of the drain that needs them was tried: no gain, every cross-stream wait costs its ~15 us wherever it sits)
  const unsigned ne = (unsigned)h->num_envs;
  // drain grids = the tiers' full occupancy on 256 CUs (8 / 4 / 2 workgroups per CU by LDS and registers); slots are claimed one at a time
  unsigned mg = ne < 2048 ? ne : 2048, hg = ne < JACO_HEAVY_GRID ? ne : JACO_HEAVY_GRID, gg = ne < 512 ? ne : 512;
  if (io.mode == 2) { mg = mg < 64 ? mg : 64; hg = hg < 64 ? hg : 64; gg = gg < 256 ? gg : 256; }   // (reset-time forward passes: overflows of the light tier go straight to the last one)
  if (io.mode == 4 || io.mode == 5) { if (ev) HIPCHK(h, hipEventRecord(ev->second, st)); return JACO_OK; }   // take_action / terminal_inspection run no substep: nothing can overflow
  if (io.mode != 2) JLAUNCHK(h, JK_MEDIUM_DRAIN, mg, st, A);   // (mode 2 queues for the last tier only)
  if (h->handdown && io.mode == 1 && nsub >= JACO_HANDDOWN_MIN) {   // (a hand-down needs JACO_HANDDOWN_MIN substeps left to pay: shorter steps never hand down, and skip the round)
    // the heavy tier holds 4 envs per CU: an env that needed it for a few substeps is passed back down to a second medium drain
    // (8 per CU) rather than kept there for the rest of its step; what overflows again is served by a second, final heavy drain
    A.handdown = 1;
    JLAUNCHK(h, JK_HEAVY_DRAIN, hg, st, A);
    A.handdown = 0;
    JLAUNCH(h, jaco_drain_round2_kernel, dim3(1), dim3(1), 0, st, qctl, (int)mg, (int)hg);
    JLAUNCHK(h, JK_MEDIUM_DRAIN, mg, st, A);
  }
  if (io.mode != 2) JLAUNCHK(h, JK_HEAVY_DRAIN, hg, st, A);
  JLAUNCHK(h, JK_HUGE_DRAIN, gg, st, A);
  HIPCHK(h, hipGetLastError());
  if (ev) HIPCHK(h, hipEventRecord(ev->second, st));
  return JACO_OK;
}

extern "C" int jaco_physics_step(JacoHandle* h, const float* ctrl_dev, int nsub, void* stream) {
  if (!h) return JACO_EINVAL;
  return launch_step(h, ctrl_dev, nsub, (hipStream_t)stream, nullptr, -1);
}
// ---- env level: reset / step with the reference's Gym-style semantics, batched -------------------------
struct JacoResetArgs {
  const float* qpos0; float* qpos; float* qvel; float* qacc_ws; float* qpos_lo; float* qvel_lo; float* task; const unsigned char* mask; float* marker; const float* marker_rest;
  int nenv, nq, nv, task_id, has_free; unsigned long long seed;
  float base[3];   // link1 position: the reaching goal's orientation looks along base -> goal (env_mujoco_util.py:201-205)
  int* list; unsigned* list_count;   // the reset envs, for the launches that follow (forward pass, placing hold)
  GoalBuffer goals;
};
__global__ void jaco_reset_kernel(JacoResetArgs R) {
  int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (e >= R.nenv || (R.mask && !R.mask[e])) return;
  if (R.mask) R.list[atomicAdd(R.list_count, 1u)] = e;
  float* t = R.task + (size_t)e * JTASK_N;
  float* q = R.qpos + (size_t)e * R.nq;
  for (int k = 0; k < R.nq; k++) { q[k] = R.qpos0[k]; R.qpos_lo[(size_t)e * R.nq + k] = 0.f; }
  for (int k = 0; k < R.nv; k++) { R.qvel[(size_t)e * R.nv + k] = 0.f; R.qacc_ws[(size_t)e * R.nv + k] = 0.f; R.qvel_lo[(size_t)e * R.nv + k] = 0.f; }
  for (int k = 0; k < 24; k++) R.marker[(size_t)e * 24 + k] = R.marker_rest[k];   // sim.reset(): markers back to their XML pose
  reset_draws(R.task_id, R.seed, (unsigned)e, R.has_free, R.base, q, t, R.goals);          // (env_logic.h: shared with the in-kernel auto-reset)
}

extern "C" int jaco_forward(JacoHandle* h, float* obs_dev, void* stream) {
  if (!h || !obs_dev) return JACO_EINVAL;
  EnvIO io; io.mode = 2; io.obs = obs_dev;
  return launch_step(h, nullptr, 1, (hipStream_t)stream, nullptr, -1, io);
}
extern "C" int jaco_placing_hold(JacoHandle* h, const uint8_t* mask_dev, int nsub, void* stream) {
  if (!h || nsub <= 0) return JACO_EINVAL;
  if (h->model_host.eeobj_body < 0 || h->model_host.nq < 23) { h->err = "jaco_placing_hold: the model has no EE_obj frame / object body"; return JACO_EINVAL; }
  EnvIO hold; hold.mode = 3; hold.mask = mask_dev; hold.listed = h->reset_listed;
  return launch_step(h, nullptr, nsub, (hipStream_t)stream, nullptr, -1, hold);
}
#define JACO_PREREACH_MAX_SUBSTEPS 4000   // cap of the grasping reset's two `while True` loops (typically ~600-1500 substeps at 0.4 m/s)
extern "C" int jaco_grasping_prereach(JacoHandle* h, const uint8_t* mask_dev, int max_substeps, float* obs_dev, void* stream) {
  if (!h || !obs_dev || max_substeps <= 0) return JACO_EINVAL;
  if (h->model_host.obj_body < 0 || h->model_host.nq < 23) { h->err = "jaco_grasping_prereach: the model has no object body"; return JACO_EINVAL; }
  EnvIO pre; pre.mode = 6; pre.mask = mask_dev; pre.obs = obs_dev; pre.listed = h->reset_listed;
  return launch_step(h, nullptr, max_substeps, (hipStream_t)stream, nullptr, -1, pre);
}
extern "C" int jaco_reset(JacoHandle* h, const uint8_t* mask_dev, float* obs_dev, void* stream) {
  if (!h || !obs_dev) return JACO_EINVAL;
  ENTER(h);
  hipStream_t st = (hipStream_t)stream;
  const JacoModelDev& m = h->model_host;
  const float* rest = (const float*)((const char*)h->model_dev + offsetof(JacoModelDev, marker_rest));
  JacoResetArgs R{h->qpos0_dev, h->qpos, h->qvel, h->qacc_ws, h->qpos_lo, h->qvel_lo, h->task_rows, mask_dev, h->marker, rest, h->num_envs, m.nq, m.nv, h->task, m.nq >= 23, h->seed, {m.base_pos[0], m.base_pos[1], m.base_pos[2]}, h->order, h->order_ctl + 67, GoalBuffer{h->goal_buf, h->goal_n, h->goal_stride}};
  if (mask_dev) HIPCHK(h, hipMemsetAsync(h->order_ctl + 67, 0, sizeof(unsigned), st));
  h->reset_listed = mask_dev != nullptr;
  JLAUNCH(h, jaco_reset_kernel, dim3((unsigned)((h->num_envs + 255) / 256)), dim3(256), 0, st, R);
  HIPCHK(h, hipGetLastError());
  if (h->task == JACO_TASK_PLACING || h->task == JACO_TASK_CARRYING || h->task == JACO_TASK_RELEASING) {   // object into the hand, 150 held substeps while the fingers close (env_mujoco_util.py:106-117)
    int rc = jaco_placing_hold(h, mask_dev, JACO_PLACING_HOLD_SUBSTEPS, stream);
    if (rc) { h->reset_listed = false; return rc; }
  }
  if (h->task == JACO_TASK_GRASPING || h->task == JACO_TASK_CARRYING) {   // the pre-reach loops (env_mujoco_util.py:123-170); the observation comes from their last substep's mjData
    int rc = jaco_grasping_prereach(h, mask_dev, JACO_PREREACH_MAX_SUBSTEPS, obs_dev, stream);
    h->reset_listed = false;
    return rc;
  }
  // sim.forward() + _get_observation for the reset envs (the others keep their observation row and controller cache)
  EnvIO io; io.mode = 2; io.obs = obs_dev; io.mask = mask_dev; io.listed = h->reset_listed;
  h->reset_listed = false;
  return launch_step(h, nullptr, 1, st, nullptr, -1, io);
}
extern "C" int jaco_step(JacoHandle* h, const float* action_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream) {
  if (!h || !action_dev || !obs_dev || !reward_dev || !done_dev) return JACO_EINVAL;
  EnvIO io; io.mode = 1; io.action = action_dev; io.obs = obs_dev; io.reward = reward_dev; io.done = done_dev;
  return launch_step(h, nullptr, h->frame_skip, (hipStream_t)stream, nullptr, -1, io);
}
extern "C" int jaco_take_action(JacoHandle* h, const float* action_dev, void* stream) {
  if (!h || !action_dev) return JACO_EINVAL;
  EnvIO io; io.mode = 4; io.action = action_dev;
  return launch_step(h, nullptr, 1, (hipStream_t)stream, nullptr, -1, io);
}
extern "C" int jaco_terminal_inspection(JacoHandle* h, uint8_t* done_dev, float* bonus_dev, void* stream) {
  if (!h || !done_dev || !bonus_dev) return JACO_EINVAL;
  EnvIO io; io.mode = 5; io.reward = bonus_dev; io.done = done_dev;
  return launch_step(h, nullptr, 1, (hipStream_t)stream, nullptr, -1, io);
}
extern "C" int jaco_set_noise(JacoHandle* h, const float* noise_dev) {
  if (!h) return JACO_EINVAL;
  h->noise = noise_dev;
  return JACO_OK;
}
extern "C" int jaco_set_subgoal(JacoHandle* h, const float* subgoal_dev) {
  if (!h) return JACO_EINVAL;
  h->subgoal = subgoal_dev;
  return JACO_OK;
}
extern "C" int jaco_set_init_buffer(JacoHandle* h, const float* rows_dev, int nrows, int row_floats, void* stream) {
  if (!h) return JACO_EINVAL;
  ENTER(h);
  if (h->goal_buf) { HIPCHK(h, hipStreamSynchronize((hipStream_t)stream)); HIPCHK(h, hipFree(h->goal_buf)); h->goal_buf = nullptr; h->goal_n = h->goal_stride = 0; }
  if (!rows_dev) return JACO_OK;
  if (nrows < 2 || row_floats < 7) { h->err = "jaco_set_init_buffer: needs at least 2 rows (np.random.randint(0, len - 1)) of at least 7 floats"; return JACO_EINVAL; }
  HIPCHK(h, hipMalloc(&h->goal_buf, (size_t)nrows * row_floats * sizeof(float)));
  HIPCHK(h, hipMemcpyAsync(h->goal_buf, rows_dev, (size_t)nrows * row_floats * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  h->goal_n = nrows; h->goal_stride = row_floats;
  return JACO_OK;
}
extern "C" int jaco_get_task_state(JacoHandle* h, float* out_dev, void* stream) {
  if (!h || !out_dev) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemcpyAsync(out_dev, h->task_rows, (size_t)h->num_envs * JTASK_N * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return JACO_OK;
}
extern "C" int jaco_set_task_state(JacoHandle* h, const float* in_dev, void* stream) {
  if (!h || !in_dev) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemcpyAsync(h->task_rows, in_dev, (size_t)h->num_envs * JTASK_N * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return JACO_OK;
}
extern "C" int jaco_task_row_floats(void) { return JTASK_N; }
extern "C" int jaco_get_terminal_obs(JacoHandle* h, float* out_dev, void* stream) {
  if (!h || !out_dev) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemcpyAsync(out_dev, h->terminal_obs, (size_t)h->num_envs * 26 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return JACO_OK;
}
extern "C" int jaco_get_last_terminal(JacoHandle* h, float* out_dev, void* stream) {
  if (!h || !out_dev) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemcpyAsync(out_dev, h->terminal, (size_t)h->num_envs * 2 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return JACO_OK;
}
extern "C" int jaco_get_markers(JacoHandle* h, float* out_dev, void* stream) {
  if (!h || !out_dev) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemcpyAsync(out_dev, h->marker, (size_t)h->num_envs * 24 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return JACO_OK;
}
extern "C" int jaco_set_markers(JacoHandle* h, const float* in_dev, void* stream) {
  if (!h || !in_dev) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemcpyAsync(h->marker, in_dev, (size_t)h->num_envs * 24 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return JACO_OK;
}
extern "C" int jaco_set_frame_skip(JacoHandle* h, int frame_skip) {
  if (!h || frame_skip <= 0) return JACO_EINVAL;
  h->frame_skip = frame_skip;
  return JACO_OK;
}

extern "C" long long jaco_launch_count(JacoHandle* h) {
  if (!h) return JACO_EINVAL;
  const long long n = h->nlaunch;
  h->nlaunch = 0;
  return n;
}
extern "C" int jaco_debug_dump_floats(void) { return JDBG_SIZE; }
// Diagnostic: the tier queues' control words of the last launch (JQ_* layout above), copied to host; synchronises.
extern "C" int jaco_debug_queue_words(JacoHandle* h, int32_t* out_host, int n) {
  if (!h || !out_host || n < JQ_WORDS) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(out_host, h->qctl + (h->qsel ^ 1) * JQ_WORDS, JQ_WORDS * sizeof(int), hipMemcpyDeviceToHost));   // (qsel: the NEXT launch's buffer)
  return JQ_WORDS;
}
extern "C" int jaco_physics_step_debug(JacoHandle* h, const float* ctrl_dev, int nsub, int env, float* dump_host, int dump_floats) {
  if (!h || !dump_host || dump_floats < JDBG_SIZE || env < 0 || env >= h->num_envs) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemset(h->dbg, 0, JDBG_SIZE * sizeof(float)));
  int rc = launch_step(h, ctrl_dev, nsub, nullptr, h->dbg, env);
  if (rc) return rc;
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(dump_host, h->dbg, JDBG_SIZE * sizeof(float), hipMemcpyDeviceToHost));
  return JACO_OK;
}

extern "C" int jaco_get_sensordata(JacoHandle* h, float* out, void* stream) {
  if (!h || !out) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemcpyAsync(out, h->sensordata, (size_t)h->num_envs * h->model_host.nsensor * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return JACO_OK;
}
extern "C" int jaco_get_flags(JacoHandle* h, uint32_t* out, void* stream) {
  if (!h || !out) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemcpyAsync(out, h->flags, (size_t)h->num_envs * sizeof(unsigned), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return JACO_OK;
}
extern "C" int jaco_clear_flags(JacoHandle* h, void* stream) {
  if (!h) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemsetAsync(h->flags, 0, (size_t)h->num_envs * sizeof(unsigned), (hipStream_t)stream));
  return JACO_OK;
}
extern "C" int jaco_get_stats(JacoHandle* h, int32_t* out, void* stream) {
  if (!h || !out) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipMemcpyAsync(out, h->stats, (size_t)h->num_envs * 4 * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return JACO_OK;
}

extern "C" int jaco_set_option(JacoHandle* h, const char* name, double v) {
  if (!h || !name) return JACO_EINVAL;
  ENTER(h);
  JacoModelDev& m = h->model_host;
  if (!strcmp(name, "disable_contact")) { h->disable_contact = v != 0; return JACO_OK; }
  if (!strcmp(name, "pair_list")) { h->pair_list = v != 0; return JACO_OK; }
  if (!strcmp(name, "sep_cache")) { h->sep_cache = v != 0; return JACO_OK; }
  if (!strcmp(name, "mpr_pairs")) {   // (an A/B build option of the kernel: collision.h JACO_MPR_PAIRS)
    if (!JACO_MPR_PAIRS && v != 0) { h->err = "jaco_set_option: mpr_pairs needs a library built with -DJACO_MPR_PAIRS=1"; return JACO_EINVAL; }
    h->mpr_pairs = v != 0; return JACO_OK;
  }
  if (!strcmp(name, "arm_kernel")) { h->arm_kernel = v != 0; return JACO_OK; }
  if (!strcmp(name, "schedule")) { h->schedule = v != 0; return JACO_OK; }
  if (!strcmp(name, "concurrent_heavy")) { h->concurrent = v != 0; return JACO_OK; }
  if (!strcmp(name, "tier_return")) { h->tier_return = v != 0; return JACO_OK; }
  if (!strcmp(name, "heavy_workers")) { h->workers = v < 1 ? 1 : (int)v; return JACO_OK; }
  if (!strcmp(name, "handdown")) { h->handdown = v != 0; return JACO_OK; }
  if (!strcmp(name, "merge_prepare")) { h->merge_prepare = v != 0; if (!h->merge_prepare) h->q_ready = false; return JACO_OK; }
  if (!strcmp(name, "auto_reset")) { h->auto_reset = v != 0; return JACO_OK; }
  if (!strcmp(name, "min_nsub_sched")) { h->min_nsub_sched = v < 1 ? 1 : (int)v; return JACO_OK; }
  if (!strcmp(name, "min_nsub_order")) { h->min_nsub_order = v < 1 ? 1 : (int)v; return JACO_OK; }
  if (!strcmp(name, "hints")) { h->use_hints = v < 0 ? 0 : (v > 2 ? 2 : (int)v); return JACO_OK; }   // 0 off, 1 biggest tier of the last step, 2 tier of its last substep
  if (!strcmp(name, "obs_mode")) { if (v != 0 && v != 1) { h->err = "jaco_set_option: obs_mode must be 0 or 1"; return JACO_EINVAL; } h->obs_mode = (int)v; return JACO_OK; }
  else if (!strcmp(name, "iterations")) m.iterations = (int)v;
  else if (!strcmp(name, "tolerance")) m.tolerance = (float)v;
  else if (!strcmp(name, "ls_iterations")) m.ls_iterations = (int)v;
  else if (!strcmp(name, "mpr_iterations")) m.mpr_iterations = (int)v;
  else if (!strcmp(name, "mpr_tolerance")) m.mpr_tolerance = (float)v;
  else if (!strcmp(name, "mpr_output")) m.mpr_output = (int)v;
  else if (!strcmp(name, "compensated")) m.compensated = v != 0;
  else { h->err = std::string("jaco_set_option: unknown option ") + name; return JACO_EINVAL; }
  HIPCHK(h, hipDeviceSynchronize());
  return upload_model(h);
}

// Diagnostic build (-DJACO_PROFILE_STAGES) only: per-env per-stage shader-cycle sums; returns JACO_EINVAL otherwise.
extern "C" int jaco_stage_profile(JacoHandle* h, uint64_t* out_host, int reset) {
#ifdef JACO_PROFILE_STAGES
  if (!h) return JACO_EINVAL;
  ENTER(h);
  size_t n = (size_t)h->num_envs * JPROF_N;
  if (!h->prof) { HIPCHK(h, hipMalloc(&h->prof, n * 8)); HIPCHK(h, hipMemset(h->prof, 0, n * 8)); }
  HIPCHK(h, hipDeviceSynchronize());
  if (out_host) HIPCHK(h, hipMemcpy(out_host, h->prof, n * 8, hipMemcpyDeviceToHost));
  if (reset) HIPCHK(h, hipMemset(h->prof, 0, n * 8));
  return JACO_OK;
#else
  (void)out_host; (void)reset;
  if (h) h->err = "jaco_stage_profile: library built without JACO_PROFILE_STAGES";
  return JACO_EINVAL;
#endif
}

extern "C" int jaco_enable_timing(JacoHandle* h, int enable) {
  if (!h) return JACO_EINVAL;
  h->timing = enable != 0;
  h->events_used = 0;
  return JACO_OK;
}
static int event_mean(JacoHandle* h, const std::vector<std::pair<hipEvent_t, hipEvent_t>>& evs, double* avg_ms) {
  double tot = 0;
  for (size_t i = 0; i < h->events_used; i++) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, evs[i].first, evs[i].second));
    tot += ms;
  }
  *avg_ms = h->events_used ? tot / h->events_used : 0.0;
  return JACO_OK;
}
extern "C" int jaco_kernel_time_ms(JacoHandle* h, double* avg_ms, int* launches) {
  if (!h || !avg_ms) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipDeviceSynchronize());
  int rc = event_mean(h, h->kevents, avg_ms);
  if (rc) return rc;
  if (launches) *launches = (int)h->events_used;
  h->events_used = 0;
  return JACO_OK;
}
extern "C" int jaco_step_time_ms(JacoHandle* h, double* avg_ms) {
  if (!h || !avg_ms) return JACO_EINVAL;
  ENTER(h);
  HIPCHK(h, hipDeviceSynchronize());
  return event_mean(h, h->events, avg_ms);
}
