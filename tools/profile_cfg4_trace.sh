#!/bin/bash
# Kernel timeline (rocprofv3 kernel trace) of BASELINE config 4 (65 536 envs, frame_skip 4, termination masking, no reset): where the time outside the
# light kernel goes.  Output: gpurun_out/cfg4trace/summary.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/cfg4trace
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --frame-skip 4 --no-reset --steps 40 --warmup 10 --preroll 60 --no-cpu-baseline --extra-scales= --policy-leg= --config-legs= --flag-census 0 "$@" > $OUT/run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $OUT 30 > $OUT/summary.txt
cat $OUT/summary.txt
tail -1 $OUT/run.log | cut -c1-300
rm -rf $OUT/*/   # (the raw CSVs are big; the summary is what is kept)
