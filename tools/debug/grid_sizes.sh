#!/bin/bash
# GPU box: the headline configuration on grids of N^3 cells.
#   tools/debug/grid_sizes.sh OUT
OUT=$1; mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for N in 64 128 256 384 512; do
  python3 bench.py --config stromgren --ncell $N --steps 20 --no-cpu-baseline --no-also 2>/dev/null | tail -n 1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
n = $N
steps = d['dda_steps_per_packet']
k = d['roofline'].get('kernel_avg_ms')
print('stromgren %4d^3: %.3e packets/s, %.2f ms per iteration of 1e8 packets, first generation %.2f ms, %.1f steps/packet, %.2e steps/s' % (n, d['value'], d['ms_per_step'], k, steps, steps * 1e8 / (k * 1e-3)))
" >> "$OUT"
done
