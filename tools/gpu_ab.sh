#!/bin/bash
# Time build/libjaco_env_<name>.so variants back to back on the default bench workload (run on the GPU box).
for name in "$@"; do
  JACO_ENV_LIB=$GRAFT_REPO_ROOT/build/libjaco_env_$name.so python bench.py --steps ${AB_STEPS:-6} --warmup 2 --no-cpu-baseline --extra-scales "" --policy-leg "" | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(r['value']), 'env-steps/s', round(r['ms_per_step'], 2), 'ms', 'kernel_ms', round(r['roofline']['kernel_ms'], 2), 'flags', r['config']['flags_or'])"
done
