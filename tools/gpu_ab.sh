#!/bin/bash
# Time build/libjaco_env_<name>.so variants ("main" = the shipped mujoco_jaco_amd/libjaco_env.so) back to back on the default bench
# workload (run on the GPU box).  A/B noise on one box is ~0.3 %.
for name in "$@"; do
  if [ "$name" = main ]; then lib=libjaco_env.so; else lib=$GRAFT_REPO_ROOT/build/libjaco_env_$name.so; fi
  JACO_ENV_LIB=$lib python bench.py --steps ${AB_STEPS:-12} --warmup 3 --no-cpu-baseline --extra-scales "" --policy-leg "" --config-legs "" | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(r['value']), 'env-steps/s', round(r['ms_per_step'], 2), 'ms', 'kernel_ms', round(r['roofline']['kernel_ms'], 3), 'error flags', r['config']['error_flags_or'])"
done
