#!/bin/bash
# usage: ab.sh <variant .so name> ...   (A/B: headline short + policy leg, alternating)
H="--no-cpu-baseline --extra-scales= --policy-leg= --config-legs= --preroll 60 --steps 20 --flag-census 0"
P="--policy picking --task picking --steps 10 --warmup 2 --preroll 180 --no-cpu-baseline --extra-scales= --policy-leg= --config-legs= --flag-census 0"
for rep in 1 2; do
for lib in "$@"; do
  JACO_ENV_LIB=$lib python bench.py $H > gpurun_out/ab_h.json 2> gpurun_out/ab_h.err || exit 1
  JACO_ENV_LIB=$lib python bench.py $P > gpurun_out/ab_p.json 2> gpurun_out/ab_p.err || exit 1
  python - "$lib" <<'PY'
import json, sys
h=json.loads(open("gpurun_out/ab_h.json").read().strip().splitlines()[-1]); p=json.loads(open("gpurun_out/ab_p.json").read().strip().splitlines()[-1])
print(sys.argv[1], "headline %.3f ms (kernel %.3f)  policy %.3f ms" % (h["ms_per_step"], h["roofline"]["kernel_ms"], p["ms_per_step"]), flush=True)
PY
done; done
