#!/bin/bash
# The round's whole profile set in one GPU call (run on the GPU box; every step bounded): stage profiles (headline, policy-driven and
# contact-free regimes), the ms/step curve over a 1 000-step rollout, rocprofv3 kernel trace of the default bench command, PMC passes, and
# the default bench line itself.  Outputs under gpurun_out/; tools/install_profiles.sh copies them into profiles/ with the round's prefix.
# Needs the diagnostic twin of the library, built here first:  python __graft_entry__.py variant prof
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 200 python3 $R/tools/gpu_stage_profile.py 16384 50 --env > $O/stage_profile.txt 2>&1 || exit 1
timeout -k 10 300 python3 $R/tools/gpu_stage_profile.py 16384 50 --env --policy --presteps 150 --by-rows > $O/stage_profile_policy.txt 2>&1 || exit 1
timeout -k 10 200 python3 $R/tools/gpu_stage_profile.py 4096 50 --arm > $O/stage_profile_arm.txt 2>&1 || exit 1
timeout -k 10 200 python3 $R/tools/gpu_soak_curve.py 65536 50,100,200,400,700,1000 > $O/soak_curve.txt 2>&1 || exit 1
timeout -k 10 700 bash $R/tools/profile_bench.sh final > $O/profile_bench.log 2>&1 || exit 1
timeout -k 10 100 python3 $R/tools/profile_window.py $O/rocprof_final $O/rocprof_final/bench.log $O/bench_kernel_stats.csv $O/bench_kernel_window.txt > /dev/null 2>&1 || exit 1
rm -rf $O/rocprof_final/*/*kernel_trace.csv   # (tens of MB of per-launch rows; the stats and the window summary stay)
timeout -k 10 500 bash $R/tools/profile_pmc.sh final env > $O/pmc_final.log 2>&1 || exit 1
rm -rf $O/pmc_final
cd $R && timeout -k 10 700 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err || exit 1
tail -1 $O/bench_final.json
