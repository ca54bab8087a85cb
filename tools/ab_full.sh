#!/bin/bash
# same-box A/B of library files over every bench leg: usage ab_full.sh <lib.so> <lib.so> ...
for rep in 1 2; do
for lib in "$@"; do
  JACO_ENV_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1 > gpurun_out/ab_full.json || exit 1
  python tools/bench_summary.py gpurun_out/ab_full.json | sed "s/^/$lib rep $rep: /"
done; done
