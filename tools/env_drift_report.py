"""Report of an env-level closed-loop drift run (tools/env_drift.py legs): summary at the marks, then every env beyond 1e-4 at the last
mark with its onset step, the coordinate group that carries the error and what the two sides saw there.
  python tools/env_drift_report.py <gpu.npz> <oracle.npz>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import env_drift

g, r = dict(np.load(sys.argv[1])), dict(np.load(sys.argv[2]))
S = env_drift.summarize(g, r)
print(json.dumps(S))
nstep, B = g["qpos"].shape[:2]
err = np.abs(g["qpos"].astype(np.float64) - r["qpos"])          # [step, env, coord]
e = err.max(2)
dead = np.cumsum(g["done"].astype(bool) | r["done"].astype(bool), 0).astype(bool)
live_end = ~dead[-1]
bad = np.where(live_end & (e[-1] > 1e-4))[0]
groups = (("arm", slice(0, 6)), ("fingers", slice(6, 9)), ("object", slice(9, 16)), ("pedestal", slice(16, 23)))
print("envs beyond 1e-4 after %d env steps: %d of %d live (%d finished earlier on both sides, done flags equal: %s); error flags on the HIP side: 0x%x" % (
    nstep, len(bad), int(live_end.sum()), int((~live_end).sum()), S["done_flags_equal"], int(np.bitwise_or.reduce(g["flags"] & 31))))
print("touch-class mismatches (obs[0]) over all live env steps: %d" % int(((g["obs"][..., 0] != r["obs"][..., 0]) & ~dead).sum()))
cause = {}
for k in bad:
    onset = int(np.argmax(e[:, k] > 1e-5))
    lead = max(groups, key=lambda gs: err[onset, k, gs[1]].max())[0]
    final = max(groups, key=lambda gs: err[-1, k, gs[1]].max())[0]
    # what moves at the onset: object / pedestal speed on the oracle side (a body in motion = an impact or a sliding contact)
    dobj = np.abs(r["qpos"][onset, k, 9:12] - r["qpos"][max(onset - 1, 0), k, 9:12]).max()
    dped = np.abs(r["qpos"][onset, k, 16:19] - r["qpos"][max(onset - 1, 0), k, 16:19]).max()
    kind = ("object in motion (knocked / sliding)" if dobj > 1e-4 else ("pedestal in motion" if dped > 1e-5 else "bodies at rest")) if lead in ("object", "pedestal") else "arm / finger contact"
    cause[(lead, kind)] = cause.get((lead, kind), 0) + 1
    print("  env %5d: onset step %2d (err %.1e), led by %-8s -> final %.1e in %-8s | oracle object moved %.1e, pedestal %.1e in the onset step | %s" % (
        k, onset + 1, e[onset, k], lead, e[-1, k], final, dobj, dped, kind))
f = g["flags"]
badm = np.zeros(B, bool); badm[bad] = True
for bit, name in ((64, "controller took the pseudo-inverse branch (|det| < 1e-3: a discrete switch)"), (32, "stepped by a bigger capacity tier at least once (EE axis sticks on the marker's sticks: 36 more rows that push on the wrist)")):
    print("  %-110s outliers %2d / %d, all live envs %.1f %%" % (name, int(((f & bit) != 0)[badm].sum()), len(bad), 100 * ((f & bit) != 0)[live_end].mean()))
lead_coord = [int(err[int(np.argmax(e[:, k] > 1e-5)), k].argmax()) for k in bad]
print("  coordinate that leads at the onset (0-5 arm joints, 6-8 fingers, 9-15 object, 16-22 pedestal):", dict(zip(*np.unique(lead_coord, return_counts=True))))
print("by cause:", {"%s / %s" % c: n for c, n in sorted(cause.items(), key=lambda x: -x[1])})
