#!/bin/bash
# CPU-only sanitizer pass (GPU AddressSanitizer is not available on the pool):
#  - the oracle (oracle/*.c) rebuilt with -fsanitize=address,undefined and run
#    through its pinning / physics / replica tests;
#  - the C++ host side (.param parser, plugin lowering) rebuilt with the same
#    flags and run as `cmi-gpu --dry-run --describe` on the three benchmark
#    parameter files (no GPU needed).
# usage (from the repo root): tools/sanitize_cpu.sh
set -eu
REPO=$(pwd)
TMP=$(mktemp -d)
trap 'cp "$TMP/libcmio.so.orig" "$REPO/oracle/libcmio.so" 2>/dev/null || true; rm -rf "$TMP"' EXIT
make -C oracle >/dev/null
cp oracle/libcmio.so "$TMP/libcmio.so.orig"
gcc -O1 -g -fPIC -std=gnu11 -ffp-contract=off -fopenmp \
    -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o oracle/libcmio.so oracle/cmio_*.c -lm
ASAN=$(gcc -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python -m pytest \
    tests/test_oracle_pinning.py tests/test_oracle_physics.py \
    tests/test_replica_distributed.py -x -q 2>&1 | tee "$TMP/oracle.log" | tail -2
if grep -q "runtime error\|AddressSanitizer" "$TMP/oracle.log"; then
  echo "sanitizer reports in the oracle run"; exit 1
fi
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer \
    -fopenmp -o "$TMP/cmi-gpu-asan" cmacionize_amd/host/cmi_gpu_main.cpp \
    -Lcmacionize_amd -lcmi_gpu -Wl,-rpath,"$REPO/cmacionize_amd" \
    -Wl,-rpath,/opt/rocm/lib
for f in stromgren stromgren_diffuse lexingtonHII40; do
  cp benchmarks/$f.param benchmarks/lexingtonHII40.yml "$TMP/"
  (cd "$TMP" && ASAN_OPTIONS=detect_leaks=0 ./cmi-gpu-asan --params $f.param \
      --dry-run --describe > $f.json 2> $f.err)
  if [ -s "$TMP/$f.err" ]; then cat "$TMP/$f.err"; exit 1; fi
  echo "$f: host dry run clean"
done
echo "sanitizer pass clean"
