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
    tests/test_oracle_emissivity.py tests/test_oracle_subgrid.py \
    tests/test_oracle_reemission_branches.py \
    tests/test_replica_distributed.py -x -q 2>&1 | tee "$TMP/oracle.log" | tail -2
if grep -q "runtime error\|AddressSanitizer" "$TMP/oracle.log"; then
  echo "sanitizer reports in the oracle run"; exit 1
fi
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer \
    -fopenmp -o "$TMP/cmi-gpu-asan" cmacionize_amd/host/cmi_gpu_main.cpp \
    -Lcmacionize_amd -lcmi_gpu -lz -Wl,-rpath,"$REPO/cmacionize_amd" \
    -Wl,-rpath,/opt/rocm/lib
for f in stromgren stromgren_diffuse lexingtonHII40; do
  cp benchmarks/$f.param benchmarks/lexingtonHII40.yml "$TMP/"
  (cd "$TMP" && ASAN_OPTIONS=detect_leaks=0 ./cmi-gpu-asan --params $f.param \
      --dry-run --describe > $f.json 2> $f.err)
  if [ -s "$TMP/$f.err" ]; then cat "$TMP/$f.err"; exit 1; fi
  echo "$f: host dry run clean"
done
# the HDF5 writer and reader: a snapshot written and read back (a run started
# from it), then the reader on truncated and bit-flipped copies of it - every
# one must end in a clean exit or an error message, never in a sanitizer report
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer \
    -Icmacionize_amd/host -o "$TMP/hdf5cli" tests/support/hdf5_reader_cli.cpp -lz
(cd "$TMP" && sed -e 's/\[64, 64, 64\]/[9, 9, 9]/' -e 's/NumberDensity: 0/NumberDensity: 1/' \
    lexingtonHII40.param > snap.param &&
  ASAN_OPTIONS=detect_leaks=0 ./cmi-gpu-asan --params snap.param --dry-run \
    --dry-run-snapshot > snap.out 2> snap.err &&
  python3 - <<'PY'
import re
t = open("snap.param").read()
t = t.replace("type: BlockSyntax\n  filename: lexingtonHII40.yml",
              "type: CMacIonizeSnapshot\n  filename: first.hdf5")
t = t.replace("prefix: lexingtonHII40_", "prefix: again_")
open("again.param", "w").write(t)
PY
  mv lexingtonHII40_000.hdf5 first.hdf5 &&
  ASAN_OPTIONS=detect_leaks=0 ./cmi-gpu-asan --params again.param --dry-run \
    --dry-run-snapshot > again.out 2> again.err)
if [ -s "$TMP/snap.err" ] || [ -s "$TMP/again.err" ]; then
  cat "$TMP/snap.err" "$TMP/again.err"; exit 1
fi
python3 - "$TMP" <<'PY'
import random, subprocess, sys, os
tmp = sys.argv[1]
data = open(os.path.join(tmp, "first.hdf5"), "rb").read()
rng = random.Random(5)
paths = ["/", "/Parameters", "/PartType0", "/PartType0/Temperature",
         "/PartType0/Coordinates", "/Units"]
bad = 0
for trial in range(300):
    b = bytearray(data)
    if trial % 3 == 0:
        b = b[:rng.randrange(8, len(b))]
    else:
        for _ in range(rng.randrange(1, 6)):
            # the metadata sits at the start of the file
            at = rng.randrange(0, min(len(b), 20000))
            b[at] = rng.randrange(256)
    name = os.path.join(tmp, "fuzz.hdf5")
    open(name, "wb").write(b)
    for path in paths:
        r = subprocess.run([os.path.join(tmp, "hdf5cli"), name, path],
                           capture_output=True, text=True, errors="replace",
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:"
                                    "allocator_may_return_null=1"))
        if "Sanitizer" in r.stderr or "runtime error" in r.stderr or \
                r.returncode not in (0, 1):
            bad += 1
            print("trial", trial, path, "rc", r.returncode)
            print(r.stderr[-1500:])
            break
    if bad:
        break
print("hdf5 reader: 300 damaged files, %s" % ("clean" if not bad else "FAILED"))
sys.exit(1 if bad else 0)
PY
# the readers of the other snapshot formats (Fortran unformatted dumps of
# Phantom and SPHNG, task-based CMacIonize snapshots) on damaged copies of the
# reference's fixtures, through the DensityFunction factory
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer \
    -Icmacionize_amd/host -Iinclude -o "$TMP/dfcli" \
    tests/support/density_function_cli.cpp -lz
python3 - "$TMP" "$REPO" <<'PY'
import random, subprocess, sys, os
tmp, repo = sys.argv[1], sys.argv[2]
golden = os.path.join(repo, "tests", "golden")
cases = [("PhantomSnapshot", "Phantomtest.dat", ""),
         ("SPHNGSnapshot", "SPHNGtest.dat", ""),
         ("SPHNGSnapshot", "SPHNGtest_notags.dat", ""),
         ("BufferedCMacIonizeSnapshot", "taskbased.hdf5",
          "SimulationBox:\n  anchor: [-5. pc, -5. pc, -5. pc]\n"
          "  sides: [10. pc, 10. pc, 10. pc]\n"
          "DensityGrid:\n  number of cells: [8, 8, 8]\n")]
rng = random.Random(7)
bad = 0
for kind, fixture, extra in cases:
    data = open(os.path.join(golden, fixture), "rb").read()
    for trial in range(120):
        b = bytearray(data)
        if trial == 0:
            pass  # the intact file
        elif trial % 3 == 0:
            b = b[:rng.randrange(1, len(b))]
        else:
            for _ in range(rng.randrange(1, 6)):
                at = rng.randrange(0, min(len(b), 3000))
                b[at] = rng.randrange(256)
        name = os.path.join(tmp, "fuzz.dat")
        open(name, "wb").write(b)
        param = os.path.join(tmp, "fuzz.param")
        open(param, "w").write(extra + "DensityFunction:\n  type: %s\n"
                               "  filename: %s\n" % (kind, name))
        r = subprocess.run([os.path.join(tmp, "dfcli"), param],
                           input="0.001 0.002 0.003\n", capture_output=True,
                           text=True, errors="replace",
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:"
                                    "allocator_may_return_null=1"))
        if trial == 0 and r.returncode != 0:
            bad += 1
            print(kind, fixture, "intact file rejected:", r.stderr[-500:])
        if "Sanitizer" in r.stderr or "runtime error" in r.stderr or \
                r.returncode not in (0, 1):
            bad += 1
            print(kind, fixture, "trial", trial, "rc", r.returncode)
            print(r.stderr[-1500:])
            break
    if bad:
        break
print("snapshot readers: 4 x 120 damaged files, %s" %
      ("clean" if not bad else "FAILED"))
sys.exit(1 if bad else 0)
PY
# the generic lowering of third-party plugins (tabulate_spectrum /
# tabulate_ions, the registries): cmi-gpu built with tests/support/
# third_party_plugins.cpp on a .param file that names its types
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer \
    -fopenmp -DTP_WITH_MAIN -o "$TMP/tp-asan" \
    tests/support/third_party_plugins.cpp -Lcmacionize_amd -lcmi_gpu -lz \
    -Wl,-rpath,"$REPO/cmacionize_amd" -Wl,-rpath,/opt/rocm/lib
for SPEC in ThirdPartyFalling Uniform; do
  sed -e "s/type: Monochromatic/type: $SPEC/" \
      -e "/^CrossSections:/,/^  type:/s/type: FixedValue/type: ThirdPartyPowerLaw/" \
      -e "/^RecombinationRates:/,/^  type:/s/type: FixedValue/type: ThirdPartyPowerLaw/" \
      benchmarks/stromgren.param > "$TMP/tp.param"
  (cd "$TMP" && ASAN_OPTIONS=detect_leaks=0 ./tp-asan --params tp.param \
      --dry-run --describe > tp.json 2> tp.err)
  if [ -s "$TMP/tp.err" ]; then cat "$TMP/tp.err"; exit 1; fi
  grep -q '"lowering": "scripted"' "$TMP/tp.json"
  echo "generic lowering ($SPEC): clean"
done
echo "sanitizer pass clean"
