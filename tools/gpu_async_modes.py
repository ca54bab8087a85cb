"""Mean step time of the bench loop (fresh random actions, masked resets, no host sync inside the loop) under different execution
options, with and without the HIP timing events (diagnostic)."""
import os, sys, time
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
def run(k):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k):
        o, r, d, _ = genv.step(torch.rand(B, 7, device=dev, generator=gen) * 2 - 1); genv.reset(d)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
run(30)
for name, opts, timing in (("default, timing off", {}, False), ("default, timing on", {}, True), ("default, timing off", {}, False),
                           ("concurrent_heavy 0, timing off", {"concurrent_heavy": 0}, False), ("hints 0, timing off", {"hints": 0}, False),
                           ("hints 1, timing off", {"hints": 1}, False), ("default, timing on", {}, True)):
    for k, v in (("concurrent_heavy", 1), ("hints", 2)): env.set_option(k, v)
    for k, v in opts.items(): env.set_option(k, v)
    env.enable_timing(timing)
    run(5)
    ms = run(n)
    extra = ""
    if timing:
        st = env.step_time_ms(); km, _ = env.kernel_time_ms(); extra = "  launch set %.2f light kernel %.2f" % (st, km)
    env.enable_timing(False)
    print("%-34s %.2f ms/step%s" % (name, ms, extra), flush=True)
