"""Extracts the weights of the reference's shipped HPC policies into data fixtures (build container only).

  python tests/golden/make_policy_vectors.py

Reads /root/reference/models_baseline/policies/{picking,placing}/policy.zip (stable-baselines format: JSON `data`, .npz
`parameters`; main.py:221-263 is the reference's evaluation loop for them) and writes tests/golden/policy_<task>.npz:
the actor networks' kernels / biases and each tail's observation / action index lists -- arrays and integers only, no
reference source.  The fixtures travel to the GPU box, the reference does not.  Also writes the logged success rates the
GPU test compares with (logger_csv/SR_*_abalation_GA.csv: mean of the last 100 logged values, max, final).
"""
import csv
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mujoco_jaco_amd import policy  # noqa: E402

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def success_log(path):
    vals = []
    for row in csv.reader(open(path)):
        for x in row:
            try:
                vals.append(float(x))
            except ValueError:
                pass
    return vals


if __name__ == "__main__":
    sr = {}
    for task in ("picking", "placing"):
        P = policy.read_policy_zip(os.path.join(REF, "models_baseline", "policies", task, "policy.zip"))
        out = os.path.join(HERE, "policy_%s.npz" % task)
        policy.save_npz(P, out)
        print(task, [(t["name"], len(t["obs_index"]), t["act_index"]) for t in P["tails"]], "%.0f KB" % (os.path.getsize(out) / 1024))
        Q = policy.load_npz(out)
        for a, b in zip(P["tails"], Q["tails"]):
            assert a["name"] == b["name"] and a["obs_index"] == b["obs_index"] and np.array_equal(a["out"][0], b["out"][0])
        f = os.path.join(REF, "logger_csv", "SR_%s_abalation_GA.csv" % task)
        v = success_log(f)
        sr[task] = {"file": "logger_csv/SR_%s_abalation_GA.csv" % task, "n": len(v), "mean_last_100": float(np.mean(v[-100:])), "max": float(np.max(v)), "final": float(v[-1])}
    # the plain SAC reaching policy (no logged success rate exists for it in the reference tree)
    P = policy.read_policy_zip(os.path.join(REF, "models_baseline", "policies", "reaching", "policy.zip"))
    policy.save_npz(P, os.path.join(HERE, "policy_reaching.npz"))
    Pp = policy.load_npz(os.path.join(HERE, "policy_picking.npz"))
    print("reaching", [(t["name"], t["hidden"][0][0].shape) for t in P["tails"]],
          "identical to the picking zip's frozen reaching primitive:", all(np.array_equal(a[0], b[0]) for a, b in zip(P["tails"][0]["hidden"], Pp["tails"][0]["hidden"])))
    json.dump(sr, open(os.path.join(HERE, "policy_success_rates.json"), "w"), indent=1)
    print(sr)
