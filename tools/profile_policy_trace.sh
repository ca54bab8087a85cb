#!/bin/bash
# Kernel timeline (rocprofv3 kernel trace) of the policy-driven bench leg: the last steps' launches, ms from each step's first kernel.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ptrace
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --policy picking --task picking --steps 6 --warmup 2 --preroll ${1:-180} --no-cpu-baseline --extra-scales= --policy-leg= > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
step, t0, tlast = -1, 0, -10**9
out = []
for s, e, k in rows:
    short = "round2" if "round2" in k else "prepare" if ("prepare" in k or "route" in k) else "order" if "order" in k else "mdrain" if "medium_drain" in k else "hdrain" if "heavy_drain" in k else "huge" if "huge" in k else "hworkers" if "heavy_workers" in k else "workers" if "medium" in k else "listed" if "listed" in k else "light" if "jaco_physics_kernel" in k else None
    if short is None: continue
    if short == "prepare" and s - tlast > 100000: step += 1; t0 = s   # (a step starts at its first queue kernel: the routing kernel, or a prepare kernel right in front of it)
    if short == "prepare": tlast = s
    if short in ("prepare", "order", "round2"): continue
    if e - s > 100000: out.append("launch %3d %-8s start %8.3f ms  end %8.3f ms" % (step, short, (s - t0) * 1e-6, (e - t0) * 1e-6))
print("\n".join(out[-24:]))
PY
tail -1 $OUT/run.log | cut -c1-200
