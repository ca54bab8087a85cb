#!/bin/bash
# rocprofv3 PMC passes (kernel-trace + counters only, one pass per counter group) of a short bench run; writes the light kernel's
# per-launch means (env-step launches only: the reset-time forward passes are told apart by their duration) to
# gpurun_out/pmc_<tag>.json in the layout bench.py reads from profiles/r02_pmc.json.
#   usage: profile_pmc.sh <tag> [level=env] [extra bench args...]
cd /tmp && export TMPDIR=/tmp
TAG=$1; LEVEL=${2:-env}; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
CMD="python3 $GRAFT_REPO_ROOT/bench.py --level $LEVEL --steps 4 --warmup 2 --preroll 20 --no-cpu-baseline --extra-scales= --policy-leg= --config-legs= $@"
i=0
# (PMC_GROUPS="A B;C D" collects only those groups, one pass each)
if [ -n "$PMC_GROUPS" ]; then IFS=';' read -ra GROUPS_ <<< "$PMC_GROUPS"; else GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE"); fi
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
dur = {}
for f in sorted(glob.glob("$OUT/p*/*/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("jaco_physics_kernel(") or r["Kernel_Name"] == "jaco_physics_kernel":
            dur[(f.split("/")[-3], r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    p = f.split("/")[-3]
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if not (kn.startswith("jaco_physics_kernel(") or kn == "jaco_physics_kernel"): continue
        d = dur.get((p, r["Dispatch_Id"]))
        if d is None or d < 5.0: continue      # env-step launches only (the masked forward pass of a reset takes well under a ms)
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg["_ms_" + r["Counter_Name"]].append(d)
out = {k: sum(v) / len(v) for k, v in agg.items() if not k.startswith("_ms_")}
ms = [x for k, v in agg.items() if k.startswith("_ms_") for x in v]
out["kernel_ms"] = sum(ms) / max(1, len(ms))
out["launches_per_counter"] = min((len(v) for k, v in agg.items() if not k.startswith("_ms_")), default=0)
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    # /opt/skills/guides/MI355X_MICROARCH.md, HBM section: the counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B read
    # requests at 64 B (exactly half for wide coalesced reads; narrower widths are uncalibrated) -> doubled; WRITE_SIZE is exact
    out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0
out["note"] = "light-tier kernel jaco_physics_kernel, mean over the env-step launches of the run; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 read correction of the guide; fabric-side bytes, Infinity-Cache hits included; narrow accesses uncalibrated)"
json.dump(out, open("$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
