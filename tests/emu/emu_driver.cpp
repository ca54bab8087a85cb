// TEST INFRASTRUCTURE: runs the unmodified kernel source under the lockstep wavefront emulator.
// Built by tests/emu/Makefile into libjaco_emu.so; used by CPU-side (-m "not gpu") kernel checks.
#include <functional>
#include <string>
#include <vector>

#include "../../mujoco_jaco_amd/csrc/model_blob.h"
#include "../../mujoco_jaco_amd/csrc/physics_kernel.h"

void emu_run_wave(int block, std::function<void()> body);

extern "C" int emu_dbg_size() { return JDBG_SIZE; }
extern "C" int emu_lds_bytes() { return (int)sizeof(JacoLDS<JacoLight>); }
extern "C" int emu_lds_bytes_heavy() { return (int)sizeof(JacoLDS<JacoHeavy>); }

extern "C" int emu_physics_step(const void* blob, long blob_size, int nenv, int nsub, int disable_contact, float* qpos, float* qvel,
                                float* qacc_ws, const float* ctrl, float* sensordata, unsigned* flags, int* stats, float* dbg, int dbg_env, int* heavy_envs) {
  static JacoModelDev model;
  static std::vector<float> hull;
  std::string err;
  if (jaco_model_from_blob(blob, (size_t)blob_size, &model, &hull, &err)) { fprintf(stderr, "emu: %s\n", err.c_str()); return -1; }
  JacoStepArgs A{};
  A.model = &model; A.hull = hull.data(); A.qpos = qpos; A.qvel = qvel; A.qacc_ws = qacc_ws; A.ctrl = ctrl; A.sensordata = sensordata;
  A.flags = flags; A.stats = stats; A.nenv = nenv; A.nsub = nsub; A.disable_contact = disable_contact; A.dbg = dbg; A.dbg_env = dbg_env;
  std::vector<int> remaining(nenv, 0), list(nenv, 0);
  int count = 0;
  A.remaining = remaining.data(); A.heavy_list = list.data(); A.heavy_count = &count;
  emu_grid = nenv;
  for (int e = 0; e < nenv; e++) emu_run_wave(e, [&]() { jaco_physics_kernel(A); });
  emu_grid = 1;
  if (count > 0) emu_run_wave(0, [&]() { jaco_physics_kernel_heavy(A); });
  if (heavy_envs) *heavy_envs = count;
  return 0;
}
