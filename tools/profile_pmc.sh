#!/bin/bash
# rocprofv3 PMC passes (kernel-trace + counters only, one pass per counter group) of a short ctrl-level bench run.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
rm -rf $OUT && mkdir -p $OUT
CMD="python3 $GRAFT_REPO_ROOT/bench.py --level ${2:-ctrl} --steps 3 --warmup ${3:-1} --no-cpu-baseline"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_WAVES_RESTORED SQ_INST_LEVEL_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:40], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for (kn, cn), (v, n) in sorted(agg.items()):
        if "jaco_physics" in kn: print("%-42s %-26s per-launch %.6g  (n=%d)" % (kn, cn, v / n, n))
PY
