"""Per-step timing and workload statistics of the default bench workload (diagnostic)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from mujoco_jaco_amd.env import JacoBatchedEnv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
genv = JacoBatchedEnv(num_envs=B, device=0, frame_skip=50, seed=1000, task="picking", robot_file="jaco2_curtain_torque")
env = genv.sim
dev = torch.device("cuda:0")
if "--no-schedule" in sys.argv: env.set_option("schedule", 0)
for opt in ("heavy_workers", "tier_return", "concurrent_heavy", "hints"):
    if "--" + opt in sys.argv: env.set_option(opt, float(sys.argv[sys.argv.index("--" + opt) + 1]))
genv.reset()
gen = torch.Generator(device=dev); gen.manual_seed(2000)
scale = float(sys.argv[sys.argv.index("--action-scale") + 1]) if "--action-scale" in sys.argv else 1.0
resets = "--resets" in sys.argv   # bench.py's loop: fresh actions, masked reset of finished envs, staggered episode ages
if resets:
    ts = genv.task_state(); ts[:, 1] = torch.randint(0, 700, (B,), device=dev, generator=gen).float(); genv.set_task_state(ts)
    for _ in range(int(sys.argv[sys.argv.index("--preroll") + 1]) if "--preroll" in sys.argv else 0):
        o, r, d, _ = genv.step((torch.rand(B, 7, device=dev, generator=gen) * 2 - 1) * scale); genv.reset(d)
tot = 0.0
for i in range(n):
    env.clear_flags()
    a = (torch.rand(B, 7, device=dev, generator=gen) * 2 - 1) * scale
    torch.cuda.synchronize(); t = time.perf_counter()
    o, r, d, _ = genv.step(a)
    if resets: genv.reset(d)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    if i >= 4: tot += dt
    import ctypes
    qw = (ctypes.c_int * 32)()
    nq = env.L.jaco_debug_queue_words(env.h, qw, 32)
    if "--queues" in sys.argv:
        print("        queues (medium, heavy, huge): queued %s  claimed by workers %s  workers started %s  reserve %s  queued from hints %s" % (
            list(qw[0:3]), list(qw[3:6]), list(qw[6:9]), list(qw[10:13]), list(qw[14:17])))
    st = env.stats().float()
    fl = env.flags()
    print("step %2d  %.2f ms  done %.4f  ncon mean %.2f max %d  nefc mean %.1f  iters mean %.3f  cand mean %.2f  heavy %.4f  singular %.4f" % (
        i, dt * 1e3, d.float().mean().item(), st[:, 0].mean().item(), int(st[:, 0].max().item()), st[:, 1].mean().item(), st[:, 2].mean().item(),
        (st[:, 3].long() & 0xffff).float().mean().item(), ((fl & 32) != 0).float().mean().item(), ((fl & 64) != 0).float().mean().item()))
print("mean ms/step over steps 4..%d: %.2f" % (n - 1, tot / max(1, n - 4) * 1e3))
