#!/bin/bash
# GPU box: bench.py of one config with every library in cmacionize_amd/variants
# (make -C cmacionize_amd/csrc variant NAME=.. DEFS=..) and with the product
# build.  usage: tools/variant_scan.sh CONFIG [STEPS]
CFG=${1:-lexington}
STEPS=${2:-10}
REPO=$(cd "$(dirname "$0")/.." && pwd)
run() {
  python "$REPO/bench.py" --config $CFG --steps $STEPS --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g' % d['value'], '%.2f ms' % d['ms_per_step'], 'transport %.2f ms' % d['transport_kernels_ms_per_step'])"
}
unset CMI_GPU_LIBRARY
run product
for L in "$REPO"/cmacionize_amd/variants/*.so; do
  export CMI_GPU_LIBRARY=$L
  run $(basename $L .so | sed 's/libcmi_gpu_//')
done
unset CMI_GPU_LIBRARY
run product
