#!/bin/bash
# same-box A/B of a library option over every bench leg: usage ab_option.sh <name> <value A> <value B> [reps]   (e.g. ab_option.sh mpr_pairs 0 1)
for rep in $(seq 1 ${4:-2}); do
for v in $2 $3; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 12 --warmup 3 --set-option $1=$v 2>/dev/null | tail -1 > gpurun_out/ab_opt.json || exit 1
  python tools/bench_summary.py gpurun_out/ab_opt.json | sed "s/^/$1=$v rep $rep: /"
done; done
