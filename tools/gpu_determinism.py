"""Run-to-run determinism of the policy-driven regime (diagnostic): the same 8 192-env rollout under the shipped picking policy, three times per
setting of the execution options; reports the first env / step at which two runs differ.  This is the tool that found the round-4 write-back
race (auto_reset + resident tier workers: two XCDs holding dirty copies of one env's observation / cache rows; physics_kernel.h, epilogue `wt`).
   python tools/gpu_determinism.py"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from mujoco_jaco_amd.env import JacoBatchedEnv
from mujoco_jaco_amd.policy import HPCPolicy
root = "/root/repo"
B = 8192
def run(pl, sc, nsteps=12, policy=True, auto=True, opts=()):
    env = JacoBatchedEnv(num_envs=B, task="picking", seed=33, auto_reset=auto)
    env.sim.set_option("pair_list", pl); env.sim.set_option("sep_cache", sc)
    for k, v in opts: env.sim.set_option(k, v)
    obs = env.reset()
    gen = torch.Generator(device=env.device); gen.manual_seed(8)
    pol = HPCPolicy.load(os.path.join(root, "tests", "golden", "policy_picking.npz"), device=env.device)
    ts = env.task_state(); ts[:, 1] = torch.randint(0, 600, (B,), device=env.device, generator=gen).float(); env.set_task_state(ts)
    hist = []
    for s in range(nsteps):
        obs, rew, done, _ = env.step(pol.predict(obs)[0])
        hist.append(env.sim.get_state()[0].clone())
    out = (hist, obs.clone(), env.sim.flags().clone(), env.sim.stats().clone())
    env.close()
    return out

def cmp(name, a, b):
    first = None
    for st, (x, y) in enumerate(zip(a[0], b[0])):
        if not torch.equal(x, y):
            first = st; break
    if first is None:
        print(name, ": identical over all steps", flush=True); return
    d = (a[0][first] - b[0][first]).abs().max(1).values
    bad = torch.nonzero(d > 0).flatten()
    print(name, ": first difference at step", first + 1, "envs differing", len(bad), "max diff %.3e" % d.max().item(), "env ids", bad[:8].tolist(),
          "flags", [hex(int(a[2][i])) for i in bad[:4]], [hex(int(b[2][i])) for i in bad[:4]], flush=True)
for name, opts, auto in (("defaults", (), True), ("auto_reset off", (), False), ("concurrent_heavy 0", (("concurrent_heavy", 0),), True), ("hints 0", (("hints", 0),), True),
                         ("handdown 0", (("handdown", 0),), True), ("schedule 0", (("schedule", 0),), True), ("tier_return 0", (("tier_return", 0),), True),
                         ("concurrent 0 + hints 0 + schedule 0", (("concurrent_heavy", 0), ("hints", 0), ("schedule", 0)), True)):
    a = run(0, 0, opts=opts, auto=auto); b = run(0, 0, opts=opts, auto=auto); c = run(0, 0, opts=opts, auto=auto)
    cmp(name + " run 1 vs 2", a, b); cmp(name + " run 1 vs 3", a, c)
