"""Run-to-run determinism of the batched env (diagnostic): the same rollout twice or three times per regime, comparing states, observations,
rewards, done flags, sticky flags and statistics bit for bit; reports the first env / step at which two runs differ.  Regimes: the shipped
picking / placing policies (grasp contacts, in-kernel or explicit resets), random actions at scale 1 and 0.1 (stick-on-stick contacts, bigger
tiers), other tasks.  This is the tool that found the round-4 write-back race (auto_reset + resident tier workers: two XCDs holding dirty
copies of one env's observation / cache rows; physics_kernel.h, epilogue `wt`).
   python tools/gpu_determinism.py [envs]"""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mujoco_jaco_amd.env import JacoBatchedEnv
from mujoco_jaco_amd.policy import HPCPolicy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192


def run(task, nsteps, policy, scale, auto, opts=()):
    env = JacoBatchedEnv(num_envs=B, task=task, seed=33, auto_reset=auto)
    for k, v in opts: env.sim.set_option(k, v)
    obs = env.reset()
    gen = torch.Generator(device=env.device); gen.manual_seed(8)
    pol = HPCPolicy.load(os.path.join(ROOT, "tests", "golden", "policy_%s.npz" % policy), device=env.device) if policy else None
    ts = env.task_state(); ts[:, 1] = torch.randint(0, env.task_max_steps - 20, (B,), device=env.device, generator=gen).float(); env.set_task_state(ts)
    nact = env.action_space.shape[0]
    hist = []
    for s in range(nsteps):
        a = pol.predict(obs)[0] if pol is not None else (torch.rand(B, nact, device=env.device, generator=gen) * 2 - 1) * scale
        obs, rew, done, _ = env.step(a)
        hist.append(torch.cat([env.sim.get_state()[0], env.sim.get_state()[1], obs, rew[:, None], done[:, None].float()], 1).clone())
        if not env.auto_reset and bool(done.any()):
            obs = env.reset(done)
    out = (hist, env.sim.flags().clone(), env.sim.stats()[:, :3].clone())
    env.close()
    return out


def cmp(name, a, b):
    first = None
    for st, (x, y) in enumerate(zip(a[0], b[0])):
        if not torch.equal(x, y):
            first = st; break
    if first is None:
        same = torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        print("%-58s identical over %d steps%s" % (name, len(a[0]), "" if same else "  (flags / stats differ!)"), flush=True)
        return
    d = (a[0][first] - b[0][first]).abs().max(1).values
    bad = torch.nonzero(d > 0).flatten()
    print("%-58s FIRST DIFFERENCE at step %d: %d envs, max %.3e, env ids %s, flags %s" % (name, first + 1, len(bad), d.max().item(), bad[:6].tolist(),
          [hex(int(a[1][i])) for i in bad[:4]]), flush=True)


for name, kw in (("picking, shipped policy, reset inside jaco_step, 94 steps", dict(task="picking", nsteps=94, policy="picking", scale=1, auto=True)),
                 ("picking, random actions x 0.1, reset inside jaco_step", dict(task="picking", nsteps=24, policy=None, scale=0.1, auto=True)),
                 ("picking, random actions x 1, explicit reset chain", dict(task="picking", nsteps=24, policy=None, scale=1.0, auto=False)),
                 ("placing, shipped policy, explicit reset chain (hold)", dict(task="placing", nsteps=60, policy="placing", scale=1, auto=False)),
                 ("pickAndplace, random actions, reset inside jaco_step", dict(task="pickAndplace", nsteps=24, policy=None, scale=1.0, auto=True)),
                 ("reaching, random actions, reset inside jaco_step", dict(task="reaching", nsteps=24, policy=None, scale=1.0, auto=True)),
                 ("grasping, random actions, explicit reset chain (pre-reach)", dict(task="grasping", nsteps=12, policy=None, scale=1.0, auto=False))):
    a = run(**kw); b = run(**kw)
    cmp(name, a, b)
