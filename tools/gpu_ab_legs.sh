#!/bin/bash
# Like gpu_ab.sh, with the side legs: default workload, action scale 0.3 / 0.05 and the policy-driven leg, per library variant.
for name in "$@"; do
  if [ "$name" = main ]; then lib=libjaco_env.so; else lib=$GRAFT_REPO_ROOT/build/libjaco_env_$name.so; fi
  JACO_ENV_LIB=$lib python bench.py --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline --config-legs "" 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = r['config']
print('$name', round(r['value']), 'env-steps/s', round(r['ms_per_step'], 2), 'ms; small-action', c.get('small_action_env_steps_per_s'), 'policy', c.get('policy_driven'), 'flags', c['error_flags_or'])"
done
