#!/bin/bash
# rocprofv3 kernel-trace of the default bench run; summary copied to profiles/ by the caller.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/rocprof_$1
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py ${@:2} > $OUT/bench.log 2>&1
find $OUT -name "*stats*.csv" | head -3
