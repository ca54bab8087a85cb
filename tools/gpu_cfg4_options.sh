C="python bench.py --level env --frame-skip 4 --no-reset --steps 60 --warmup 10 --preroll 60 --no-cpu-baseline --extra-scales= --policy-leg= --config-legs= --flag-census 0"
for o in "" "--set-option min_nsub_order=8" "--set-option min_nsub_order=8 --set-option min_nsub_sched=8" "" "--set-option min_nsub_order=8"; do
  $C $o 2>/dev/null | python -c "
import sys, json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$o]', round(r['value']/1e6,3), 'M', round(r['ms_per_step'],3), 'ms kernel', round(r['roofline']['kernel_ms'],3), 'launches', r['config']['launches_per_step'], 'done', round(r['config']['done_fraction'],3))"
done
