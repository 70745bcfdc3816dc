#!/bin/bash
# GPU box: value and ms per iteration of the three single-GPU configs, short.
#   tools/quick_bench.sh [STEPS]
STEPS=${1:-20}
for c in stromgren stromgren_diffuse lexington; do
  python bench.py --config $c --steps $STEPS --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', '%.4g' % d['value'], '%.2f ms' % d['ms_per_step'], 'first generation %.2f ms' % d['roofline']['kernel_avg_ms'])"
done
