"""GPU box: which execution option owns the closed-loop outliers?  Re-runs the HIP leg of tools/env_drift.py under a list of option settings
against ONE oracle leg (an .npz written earlier) and prints, per setting, the envs beyond 1e-4 at the end -- outliers that vanish when an
option is switched off belong to that option's code path; outliers shared by every setting belong to the arithmetic.
  python tools/gpu_drift_options.py <oracle.npz> [B] [nstep]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import env_drift

ref = dict(np.load(sys.argv[1]))
B = int(sys.argv[2]) if len(sys.argv) > 2 else ref["qpos"].shape[1]
nstep = int(sys.argv[3]) if len(sys.argv) > 3 else ref["qpos"].shape[0]
ref = {k: v[:nstep, :B] for k, v in ref.items()}
sets = {}
for opts in ("", "sep_cache=0", "pair_list=0", "sep_cache=0,pair_list=0", "schedule=0,hints=0,concurrent_heavy=0", "compensated=0"):
    path = os.path.join(ROOT, "gpurun_out", "drift_opt.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    env = dict(os.environ, JACO_DRIFT_OPTS=opts)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "env_drift.py"), "gpu", path, str(B), str(nstep)], check=True, env=env)
    g = dict(np.load(path))
    err = np.abs(g["qpos"].astype(np.float64) - ref["qpos"]).max(2)
    dead = np.cumsum(g["done"].astype(bool) | ref["done"].astype(bool), 0).astype(bool)
    bad = np.where(~dead[-1] & (err[-1] > 1e-4))[0]
    sets[opts] = set(bad.tolist())
    print("%-45s beyond 1e-4: %3d  max %.1e  touch mismatches %d  envs %s" % (opts or "(default)", len(bad), err[-1][~dead[-1]].max(),
          int(((g["obs"][..., 0] != ref["obs"][..., 0]) & ~dead).sum()), sorted(bad.tolist())[:40]), flush=True)
base = sets[""]
for o, s in sets.items():
    if o:
        print("%-45s gone vs default: %s | new: %s" % (o, sorted(base - s), sorted(s - base)))
