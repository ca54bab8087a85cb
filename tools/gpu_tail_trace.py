"""Per-step tail of the step launch set behind the light kernel (diagnostic): launch-set time minus light-kernel time, with what the
tier queues held and how much of it the resident workers took."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from mujoco_jaco_amd.env import JacoBatchedEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
genv = JacoBatchedEnv(num_envs=B, device=0, frame_skip=50, seed=1000, task="picking")
env = genv.sim
dev = genv.device
genv.reset()
gen = torch.Generator(device=dev); gen.manual_seed(2000)
ts = genv.task_state(); ts[:, 1] = torch.randint(0, 700, (B,), device=dev, generator=gen).float(); genv.set_task_state(ts)
for _ in range(20):
    o, r, d, _ = genv.step(torch.rand(B, 7, device=dev, generator=gen) * 2 - 1); genv.reset(d)
qw = (ctypes.c_int * 32)()
for i in range(n):
    a = torch.rand(B, 7, device=dev, generator=gen) * 2 - 1
    env.enable_timing(True)
    o, r, d, _ = genv.step(a)
    torch.cuda.synchronize()
    st = env.step_time_ms(); km, _ = env.kernel_time_ms()
    env.L.jaco_debug_queue_words(env.h, qw, 32)
    env.enable_timing(False)
    genv.reset(d)
    print("step %2d launch set %.3f ms light kernel %.3f ms tail %.3f | queued %s by workers %s resets %d" % (
        i, st, km, st - km, list(qw[0:3]), list(qw[3:6]), int(d.sum().item())))
