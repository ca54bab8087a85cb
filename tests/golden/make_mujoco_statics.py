"""Build-container only: extracts the MuJoCo-produced object heights the reference holds and writes tests/golden/mujoco_rest_heights.json.

The reference has no physics tests, but its recorded expert trajectories
(/root/reference/models_baseline/trajectories/*.npz, observations of older 23-dim env versions, written by the
authors' real MuJoCo 2.0 through `_get_observation`, env_mujoco_util.py:223-271) hold the object's z coordinate in
column 10 -- float32 numbers of MuJoCo's own `sim.step()` (mujoco.py:278) in two situations that do not depend on the
arm at all:

* reaching.npz, rows 0..9: the object is spawned at z = 0.1898 (env_mujoco_util.py:215) in an env version without the
  holder under it, falls freely for three env steps (50 substeps of 1 ms each, env_mujoco.py:24,119-120, the
  observation reading xpos of the start of the 50th substep), hits the floor plane flat
  (jaco2_curtain_torque.xml:49,292) and settles: a ten-value transient of plane-box contacts, then the rest height in
  4 725 of 4 774 rows.
* grasping_trajectory_expert5 / expert6.npz: the object is spawned at 0.1898 = 1 cm inside the holder box
  (xml:300-302) and is pushed out by the box-box contacts: a thirteen-value transient that 12 + 17 episodes repeat
  value for value, then the rest height in 4 973 / 4 169 rows.

Only numbers leave this script (values, row indices, multiplicities, file and column names): data, not source.
"""
import json
import os

import numpy as np

REF = "/root/reference/models_baseline/trajectories"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mujoco_rest_heights.json")
COL = 10   # obs[8:11] = object xyz (env_mujoco_util.py:247 of today's 26-dim layout; same slot in the 23-dim records)


def f32_list(a):
    """float32 values as their shortest round-trip decimal strings' floats (exactly recoverable by np.float32(x))."""
    return [float(np.float32(x)) for x in a]


def most_common(col):
    vals, cnt = np.unique(col, return_counts=True)
    i = int(np.argmax(cnt))
    return float(vals[i]), int(cnt[i])


def main():
    out = {"column": COL, "generator": "tests/golden/make_mujoco_statics.py", "frame_skip": 50, "timestep": 0.001,
           "spawn_z_literal": 0.1898}
    d = np.load(os.path.join(REF, "reaching.npz"))
    o = d["obs"]
    assert o.dtype == np.float32 and d["episode_starts"].sum() == 1
    rest, mult = most_common(o[:, COL])
    # the transient: rows from the spawn to the first row at the rest value; x / y / orientation slots unchanged throughout
    first_rest = int(np.argmax(o[:, COL] == np.float32(rest)))
    assert (o[:first_rest + 1, 8:10] == o[0, 8:10]).all() and (o[:first_rest + 1, 11:14] == 0).all()
    out["floor_drop"] = {"file": "reaching.npz", "rows": [0, first_rest], "z": f32_list(o[:first_rest + 1, COL]),
                         "rest_z": rest, "rest_multiplicity": mult, "rows_total": int(o.shape[0]),
                         "geoms": "object box (.027 .027 .03) flat on the floor plane"}
    hold = {"files": {}, "geoms": "object box flat on the holder box (top at z = 0.17), spawned 1 cm inside it"}
    seq_ref = None
    for f in ("grasping_trajectory_expert5.npz", "grasping_trajectory_expert6.npz"):
        d = np.load(os.path.join(REF, f))
        o = d["obs"]
        st = np.where(d["episode_starts"])[0]
        rest, mult = most_common(o[:, COL])
        z0 = np.float32(0.18984711)
        eps = [int(s) for s in st if o[s, COL] == z0]
        n = 13
        seqs = np.stack([o[s:s + n, COL] for s in eps])
        # the transient shared value for value by the episodes the arm leaves alone for the first 13 steps
        ref = seqs[0] if seq_ref is None else seq_ref
        same = [e for e, q in zip(eps, seqs) if (q == ref).all()]
        seq_ref = ref
        hold["files"][f] = {"rest_z": rest, "rest_multiplicity": mult, "rows_total": int(o.shape[0]),
                            "episodes_starting_at_first_value": len(eps), "episodes_with_identical_transient": len(same),
                            "first_rows": same[:20]}
    hold["z"] = f32_list(seq_ref)
    hold["rest_z"] = float(np.float32(0.19997096))
    out["holder_pushout"] = hold
    with open(OUT, "w") as fp:
        json.dump(out, fp, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
