"""Success rate of the reference's shipped HPC policies on the batched env (main.py:221-263: success = reward > 100 and done),
one deterministic episode per env.  Usage: gpu_policy_eval.py [task] [num_envs] [relativity_sign] [max_steps]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from mujoco_jaco_amd.env import JacoBatchedEnv
from mujoco_jaco_amd.policy import HPCPolicy


def evaluate(task="picking", B=2048, sign=1.0, max_steps=700, seed=5, verbose=True, noise_free=False):
    # task reaching: the plain SAC policy of models_baseline/policies/reaching reads the reaching goal from obs[17:23] (the
    # rulebased_subgoal = False observation branch, env_mujoco_util.py:255-270), 6-wide action, 500-step episodes
    env = JacoBatchedEnv(num_envs=B, task=task, seed=seed, rulebased_subgoal=(task != "reaching"))
    pol = HPCPolicy.load(os.path.join(ROOT, "tests", "golden", "policy_%s.npz" % task), device=env.device, relativity_sign=sign, nact=env.action_space.shape[0])
    max_steps = min(max_steps, env.task_max_steps)
    if noise_free:
        env.set_noise(torch.full((B, 12), 0.5))
    obs = env.reset()
    active = torch.ones(B, dtype=torch.bool, device=env.device)
    succ = torch.zeros(B, dtype=torch.bool, device=env.device)
    length = torch.zeros(B, dtype=torch.int32, device=env.device)
    ret = torch.zeros(B, device=env.device)
    wsum = torch.zeros(2, device=env.device); wn = 0
    t0 = time.time()
    for s in range(max_steps):
        a, w = pol.predict(obs)
        obs, rew, done, _ = env.step(a)
        ret += torch.where(active, rew, torch.zeros_like(rew))
        length += active.int()
        fin = active & done
        succ |= fin & (rew > 100)
        active &= ~done
        wsum = wsum[:w.shape[1]] + w.mean(0); wn += 1
        if verbose and (s % 100 == 99 or not bool(active.any())):
            print("step %3d: finished %d / %d, successes %d, mean primitive weights %s" % (s + 1, int((~active).sum()), B, int(succ.sum()), (wsum / wn).tolist()), flush=True)
        if not bool(active.any()):
            break
    fl = env.sim.flags()
    res = {"task": task, "envs": B, "success_rate": float(succ.float().mean()), "finished": int((~active).sum()), "mean_length": float(length.float().mean()),
           "mean_return": float(ret.mean()), "nan_flag": int(((fl & 8) != 0).sum()), "overflow_flags": int(((fl & 3) != 0).sum()), "seconds": time.time() - t0,
           "relativity_sign": sign}
    env.close()
    return res


if __name__ == "__main__":
    task = sys.argv[1] if len(sys.argv) > 1 else "picking"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    sign = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    ms = int(sys.argv[4]) if len(sys.argv) > 4 else 700
    print(evaluate(task, B, sign, ms))
