#!/bin/bash
# Copy the outputs of tools/gpu_profile_all.sh from gpurun_out/ into profiles/ under the round's prefix (default r04).
# The PMC file gets the layout bench.py replays (commit = HEAD: commit the sources the libraries were built from first).
set -e
cd "$(dirname "$0")/.."
P=${1:-r05}
grep -v amdgpu.ids gpurun_out/stage_profile.txt > profiles/${P}_stage_profile.txt
grep -v amdgpu.ids gpurun_out/stage_profile_policy.txt > profiles/${P}_stage_profile_policy.txt
grep -v amdgpu.ids gpurun_out/stage_profile_arm.txt > profiles/${P}_stage_profile_arm.txt
grep -v amdgpu.ids gpurun_out/soak_curve.txt > profiles/${P}_soak_curve.txt
cp gpurun_out/bench_kernel_stats.csv profiles/${P}_bench_kernel_stats.csv
cp gpurun_out/bench_kernel_window.txt profiles/${P}_bench_kernel_window.txt
tail -1 gpurun_out/bench_final.json > profiles/${P}_bench_env_level.json
python3 - <<PY
import json, subprocess
d = json.load(open("gpurun_out/pmc_final.json"))
out = {"commit": subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip(),
       "command": "tools/profile_pmc.sh final env  (rocprofv3 --kernel-trace --pmc <group>, one pass per group, bench.py --level env --steps 4 --warmup 2 --preroll 20, no side legs)",
       "env_B65536_fs50": d}
json.dump(out, open("profiles/${P}_pmc.json", "w"), indent=1)
print({k: d[k] for k in ("kernel_ms", "SQ_INSTS_VALU", "hbm_bytes_per_launch") if k in d})
PY
