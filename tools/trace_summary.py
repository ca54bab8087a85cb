"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: for the last N env steps (a step starts at the first queue kernel --
jaco_route_kernel, or a jaco_prepare_kernel in front of it -- of a launch set that holds jaco_physics_kernel), mean start offset and duration of every kernel in launch order,
GPU-busy time and the step period.  Usage: trace_summary.py <dir with *kernel_trace.csv> [nsteps]"""
import csv, glob, sys, collections
d = sys.argv[1]; nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# step boundaries: the light step kernel (not the _listed reset twin); a step = everything from the prepare kernel preceding it up to the next such prepare
light = [i for i, r in enumerate(rows) if r[2].startswith("jaco_physics_kernel") and "listed" not in r[2] and r[2].rstrip() == "jaco_physics_kernel"]
starts = []
for i in light:
    j = i
    while j > 0 and "jaco_prepare_kernel" not in rows[j][2] and "jaco_route_kernel" not in rows[j][2]:
        j -= 1
    if j > 0 and "jaco_route_kernel" in rows[j][2] and "jaco_prepare_kernel" in rows[j - 1][2]:
        j -= 1
    starts.append(j)
starts = starts[-(nlast + 1):]
agg = collections.OrderedDict()
periods, busy = [], []
for a, b in zip(starts[:-1], starts[1:]):
    t0 = rows[a][0]
    periods.append(rows[b][0] - t0)
    ivs = sorted((s, e) for s, e, _ in rows[a:b])
    tot, cs, ce = 0, ivs[0][0], ivs[0][1]
    for s, e in ivs[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    busy.append(tot + ce - cs)
    seen = collections.Counter()
    for s, e, k in rows[a:b]:
        seen[k] += 1
        key = "%s#%d" % (k, seen[k])
        agg.setdefault(key, []).append((s - t0, e - s))
print("steps %d: period mean %.3f ms, GPU busy (union of kernels) %.3f ms" % (len(periods), sum(periods) / len(periods) * 1e-6, sum(busy) / len(busy) * 1e-6))
for k, v in sorted(agg.items(), key=lambda kv: sum(x[0] for x in kv[1]) / len(kv[1])):
    print("  %-70s n %3d  start %9.1f us  dur %9.1f us" % (k[:70], len(v), sum(x[0] for x in v) / len(v) * 1e-3, sum(x[1] for x in v) / len(v) * 1e-3))
