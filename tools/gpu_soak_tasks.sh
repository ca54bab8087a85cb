#!/bin/bash
# Soak of every task and the side regimes in one GPU call (tools/gpu_soak.py): -> gpurun_out/soak_tasks.txt
O=$GRAFT_REPO_ROOT/gpurun_out/soak_tasks.txt
echo "# tools/gpu_soak.py on MI355X, round-5 build: <envs> <steps> <task> <action scale> [--auto-reset]; random actions, finished envs reset in the loop" > $O
run() { echo "== $@" >> $O; timeout -k 10 170 python3 $GRAFT_REPO_ROOT/tools/gpu_soak.py "$@" 2>&1 | grep -v amdgpu >> $O || exit 1; }
run 65536 300 picking 1.0 --auto-reset
run 65536 100 picking 0.1 --auto-reset
run 16384 200 placing 1.0
run 16384 200 reaching 1.0 --auto-reset
run 16384 100 grasping 1.0
run 16384 200 pickAndplace 1.0 --auto-reset
run 16384 100 releasing 1.0
run 16384 50 carrying 1.0
run 16384 100 pushing 1.0 --auto-reset
tail -3 $O
