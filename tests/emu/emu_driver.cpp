// TEST INFRASTRUCTURE: runs the unmodified kernel source under the lockstep wavefront emulator.
// Built by tests/emu/Makefile into libjaco_emu.so; used by CPU-side (-m "not gpu") kernel checks.
#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../../mujoco_jaco_amd/csrc/model_blob.h"
#include "../../mujoco_jaco_amd/csrc/physics_kernel.h"

void emu_run_wave(int block, std::function<void()> body);

extern "C" int emu_dbg_size() { return JDBG_SIZE; }
extern "C" int emu_lds_bytes() { return (int)sizeof(JacoLDS<JacoLight>); }
extern "C" int emu_lds_bytes_heavy() { return (int)sizeof(JacoLDS<JacoHeavy>); }
extern "C" int emu_lds_bytes_medium() { return (int)sizeof(JacoLDS<JacoMedium>); }
extern "C" int emu_lds_bytes_huge() { return (int)sizeof(JacoLDS<JacoHuge>); }

static int g_mpr_output_fwd();
static JacoModelDev g_model;
static std::vector<float> g_hull;
static int g_no_tier_return_fwd();
static std::vector<int> g_hint;
static int g_use_hints = 0;
static int g_obs_mode = 0;
extern "C" void emu_set_obs_mode(int v) { g_obs_mode = v; }
extern "C" void emu_set_hints(int on) { g_use_hints = on; }
long emu_counter[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // (8: constrained Newton solves, 9: their Hessian builds = matrix-core passes)
extern "C" long emu_get_counter(int i, int reset) { long v = emu_counter[i & 15]; if (reset) emu_counter[i & 15] = 0; return v; }
static int g_no_pairlist = 0;
static std::vector<float> g_terminal;
extern "C" const float* emu_last_terminal() { return g_terminal.data(); }   // [nenv][2] (success, wb) latched by terminal steps
static int g_sep_cache = 1;
static std::vector<float> g_sepdir;
extern "C" void emu_set_sep_cache(int on) { g_sep_cache = on; g_sepdir.clear(); }   // 0: every hull pair goes through MPR (option "sep_cache" = 0); either call empties the cache
static int g_mpr_pairs = 1;
extern "C" void emu_set_mpr_pairs(int on) { g_mpr_pairs = on; }   // option "mpr_pairs" of the library (counter 10: passes of the two-pair routine)
extern "C" void emu_set_pair_list(int on) { g_no_pairlist = !on; }   // 0: every pair tested in every substep (option "pair_list" = 0 of the library)
static int g_handdown = 0, g_handed_down = 0;
extern "C" void emu_set_handdown(int on) { g_handdown = on; }
extern "C" int emu_handed_down() { return g_handed_down; }   // envs passed from the heavy drain to the second medium drain so far
static int emu_launch(JacoStepArgs A, int* heavy_envs) {
  A.no_tier_return = g_no_tier_return_fwd();
  A.no_pairlist = g_no_pairlist;
  A.mpr_pairs = g_mpr_pairs;
  if (g_sepdir.size() != (size_t)A.nenv * JMAXPAIR * 4) g_sepdir.assign((size_t)A.nenv * JMAXPAIR * 4, 0.f);
  A.sepdir = g_sep_cache ? g_sepdir.data() : nullptr;
  if (g_terminal.size() != (size_t)A.nenv * 2) g_terminal.assign((size_t)A.nenv * 2, 0.f);
  A.terminal = g_terminal.data();
  A.terminal_obs = nullptr;
  A.obs_mode = g_obs_mode;
  std::vector<int> remaining(A.nenv, 0), lists(6 * (size_t)A.nenv, -1);
  int count[3] = {0, 0, 0}, taken[3] = {0, 0, 0}, light_left = A.nenv;
  A.remaining = remaining.data(); A.light_left = &light_left;
  for (int t = 0; t < 3; t++) { A.q[t].list = lists.data() + (size_t)t * 2 * A.nenv; A.q[t].count = &count[t]; A.q[t].taken = &taken[t]; A.q[t].limit = nullptr; A.q[t].reserve = nullptr; }
  A.routed_mark = nullptr; A.launch_id = 1; A.nslots = nullptr;
  if ((int)g_hint.size() != A.nenv) g_hint.assign(A.nenv, 0);
  A.hint = g_use_hints ? g_hint.data() : nullptr; A.hint_mode = g_use_hints;
  emu_grid = A.nenv;
  // (as jaco_env.hip: the step kernel proper serves modes 0 / 1, every other mode the full-code twin)
  // ... and contact-free steps (disable_contact, models without a collidable pair) the lean kernel, alone
  if ((A.disable_contact || A.model->npair == 0) && A.env_mode <= 1) {
    A.disable_contact = 1; A.hint = nullptr;
    for (int e = 0; e < A.nenv; e++) emu_run_wave(e, [&]() { jaco_physics_kernel_arm(A); });
    if (heavy_envs) *heavy_envs = 0;
    return 0;
  }
  for (int e = 0; e < A.nenv; e++) emu_run_wave(e, [&]() { if (A.env_mode >= 2) jaco_physics_kernel_listed(A); else jaco_physics_kernel(A); });
  emu_grid = 1;
  // (the resident workers of the GPU build leave as soon as the light grid is done: here that is always the case, so the
  // drains serve every queue; they are the same serve functions).  Same launch sequence as jaco_env.hip, grids of one workgroup.
  if (count[0] > 0) emu_run_wave(0, [&]() { jaco_physics_kernel_medium(A); });
  emu_run_wave(0, [&]() { jaco_physics_kernel_medium_drain(A); });
  if (count[1] > 0) emu_run_wave(0, [&]() { jaco_physics_kernel_heavy_workers(A); });
  if (g_handdown && A.env_mode == 1) {
    A.handdown = 1;
    emu_run_wave(0, [&]() { jaco_physics_kernel_heavy_drain(A); });
    A.handdown = 0;
    taken[0] -= 1; taken[1] -= 1;   // (jaco_drain_round2_kernel: each drain workgroup ends one claim beyond the end)
    g_handed_down += count[0] - taken[0];
    emu_run_wave(0, [&]() { jaco_physics_kernel_medium_drain(A); });
  }
  emu_run_wave(0, [&]() { jaco_physics_kernel_heavy_drain(A); });
  if (count[2] > 0) emu_run_wave(0, [&]() { jaco_physics_kernel_huge_workers(A); });
  emu_run_wave(0, [&]() { jaco_physics_kernel_huge_drain(A); });
  if (heavy_envs) *heavy_envs = count[0];
  return 0;
}
// env-level call: mode 1 = step (nsub = frame_skip), mode 2 = forward only
static int g_auto_reset = 0;
static std::vector<float> g_qpos0;
extern "C" void emu_set_auto_reset(int on, const float* qpos0, int nq) { g_auto_reset = on; g_qpos0.assign(qpos0, qpos0 + nq); }
static std::vector<float> g_goal_buf;   // kwarg init_buffer (jaco_set_init_buffer of the library)
static int g_goal_n = 0, g_goal_stride = 0;
extern "C" void emu_set_init_buffer(const float* rows, int nrows, int stride) {
  g_goal_buf.assign(rows, rows + (rows ? (size_t)nrows * stride : 0)); g_goal_n = rows ? nrows : 0; g_goal_stride = rows ? stride : 0;
}
// what jaco_reset_kernel does for one env (jaco_env.hip): sim.reset() + the draws; the forward pass is a mode-2 emu_env_call
extern "C" void emu_reset_env(int task_id, unsigned long long seed, int env, int nq, int nv, const float* qpos0, const float* base, float* qpos, float* qvel,
                              float* qacc_ws, float* task, float* marker, const float* marker_rest) {
  for (int k = 0; k < nq; k++) qpos[(size_t)env * nq + k] = qpos0[k];
  for (int k = 0; k < nv; k++) { qvel[(size_t)env * nv + k] = 0.f; qacc_ws[(size_t)env * nv + k] = 0.f; }
  for (int k = 0; k < 24; k++) marker[(size_t)env * 24 + k] = marker_rest[k];
  reset_draws(task_id, seed, (unsigned)env, nq >= 23, base, qpos + (size_t)env * nq, task + (size_t)env * JTASK_N,
              GoalBuffer{g_goal_buf.empty() ? nullptr : g_goal_buf.data(), g_goal_n, g_goal_stride});
}
extern "C" int emu_env_call(const void* blob, long blob_size, int nenv, int mode, int frame_skip, int task_id, int nact, unsigned long long seed,
                            float* qpos, float* qvel, float* qacc_ws, float* sensordata, unsigned* flags, int* stats, float* task, float* cache,
                            const float* action, const float* noise, float* obs, float* reward, unsigned char* done, float* marker, int* heavy_envs) {
  std::string err;
  if (jaco_model_from_blob(blob, (size_t)blob_size, &g_model, &g_hull, &err)) { fprintf(stderr, "emu: %s\n", err.c_str()); return -1; }
  if (g_mpr_output_fwd() >= 0) g_model.mpr_output = g_mpr_output_fwd();
  JacoStepArgs A{};
  A.model = &g_model; A.hull = g_hull.data(); A.qpos = qpos; A.qvel = qvel; A.qacc_ws = qacc_ws; A.ctrl = qvel; A.sensordata = sensordata;
  A.flags = flags; A.stats = stats; A.nenv = nenv; A.nsub = mode == 2 ? 1 : frame_skip; A.env_mode = mode; A.task_id = task_id; A.nact = nact;
  A.seed = seed; A.task = task; A.cache = cache; A.action = action; A.noise = noise; A.obs = obs; A.reward = reward; A.done = done; A.marker = marker; A.dbg_env = -1;
  A.auto_reset = g_auto_reset && mode == 1 && (task_id == 0 || task_id == 2 || task_id == 4 || task_id == 7); A.qpos0 = g_qpos0.empty() ? nullptr : g_qpos0.data();
  A.goal_buf = g_goal_buf.empty() ? nullptr : g_goal_buf.data(); A.goal_n = g_goal_n; A.goal_stride = g_goal_stride;
  return emu_launch(A, heavy_envs);
}
// rest pose of the two task-layer markers (what jaco_reset_state writes): 24 floats
extern "C" int emu_marker_rest(const void* blob, long blob_size, float* out) {
  std::string err;
  if (jaco_model_from_blob(blob, (size_t)blob_size, &g_model, &g_hull, &err)) { fprintf(stderr, "emu: %s\n", err.c_str()); return -1; }
  for (int k = 0; k < 24; k++) out[k] = g_model.marker_rest[k / 12][k % 12];
  return 0;
}
static int g_no_tier_return = 0;
static int g_mpr_output = -1;   // -1: the model loader's default
extern "C" void emu_set_mpr_output(int v) { g_mpr_output = v; }
extern "C" void emu_set_tier_return(int on) { g_no_tier_return = !on; }
extern "C" int emu_task_floats() { return JTASK_N; }
extern "C" int emu_cache_floats() { return JCACHE_N; }

extern "C" int emu_physics_step(const void* blob, long blob_size, int nenv, int nsub, int disable_contact, float* qpos, float* qvel,
                                float* qacc_ws, const float* ctrl, float* sensordata, unsigned* flags, int* stats, float* dbg, int dbg_env, int* heavy_envs) {
  static JacoModelDev model;
  static std::vector<float> hull;
  std::string err;
  if (jaco_model_from_blob(blob, (size_t)blob_size, &model, &hull, &err)) { fprintf(stderr, "emu: %s\n", err.c_str()); return -1; }
  if (g_mpr_output >= 0) model.mpr_output = g_mpr_output;
  JacoStepArgs A{};
  A.model = &model; A.hull = hull.data(); A.qpos = qpos; A.qvel = qvel; A.qacc_ws = qacc_ws; A.ctrl = ctrl; A.sensordata = sensordata;
  A.flags = flags; A.stats = stats; A.nenv = nenv; A.nsub = nsub; A.disable_contact = disable_contact; A.dbg = dbg; A.dbg_env = dbg_env;
  static JacoModelDev* keep = &model; (void)keep;
  return emu_launch(A, heavy_envs);
}

static int g_no_tier_return_fwd() { return g_no_tier_return; }
static int g_mpr_output_fwd() { return g_mpr_output; }
