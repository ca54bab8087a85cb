"""ms/step of the headline workload (bench.py defaults: 65 536 envs, picking, random actions, reset inside jaco_step) as a function of how
long the rollout has been running: a global reset, step counters staggered over [0, 700) as in bench.py, then windows of `win` timed
steps ending at the marks.  Also the per-window workload statistics (what makes an older rollout more expensive).

  python tools/gpu_soak_curve.py [envs] [marks, comma separated] [action scale] [window]
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mujoco_jaco_amd.env import JacoBatchedEnv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
marks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "50,100,200,400,700,1000").split(",")]
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
win = int(sys.argv[4]) if len(sys.argv) > 4 else 20
env = JacoBatchedEnv(num_envs=B, task="picking", seed=1000, auto_reset=True)
for kv in sys.argv[5:]:
    name, val = kv.split("=")
    env.sim.set_option(name, float(val))
env.reset()
gen = torch.Generator(device=env.device); gen.manual_seed(2000)
ts = env.task_state()
ts[:, 1] = torch.randint(0, env.task_max_steps, (B,), device=env.device, generator=gen).float()
env.set_task_state(ts)
nact = env.action_space.shape[0]
print("# tools/gpu_soak_curve.py %s on %s" % (" ".join(sys.argv[1:]), torch.cuda.get_device_name(0)))
print("# step  ms/step(window of %d)  env-steps/s  done/step  mean: contacts rows newton-iters candidates  bigger-tier-share(step)  hints>0  error-flags" % win)
s = 0
for mk in marks:
    while s < mk - win:
        env.step((torch.rand(B, nact, device=env.device, generator=gen) * 2 - 1) * scale); s += 1
    env.sim.clear_flags()
    nd = torch.zeros((), dtype=torch.int64, device=env.device)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while s < mk:
        _, _, d, _ = env.step((torch.rand(B, nact, device=env.device, generator=gen) * 2 - 1) * scale); s += 1
        nd += d.sum()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / win
    st = env.sim.stats()
    st[:, 3] &= 0xffff   # (the upper half of the last word holds line-search iterations)
    st = st.float().mean(0).cpu().numpy()
    fl = env.sim.flags()
    print("%6d  %7.2f  %9.0f  %7.1f  %6.2f %6.2f %5.2f %6.1f  %.4f  %.4f  0x%x" % (
        mk, dt * 1e3, B / dt, float(nd) / win, st[0], st[1], st[2], st[3], float(((fl & 32) != 0).float().mean()),
        float((env.sim.hints() > 0).float().mean()) if hasattr(env.sim, "hints") else -1.0, int(torch.bitwise_or(fl & 31, torch.zeros_like(fl)).max())), flush=True)
