#!/bin/bash
# GPU box: average duration of some kernels in `bench.py --config CFG --steps N`
# for the product and every library in cmacionize_amd/variants.
#   tools/debug/kernel_avg.sh CFG STEPS "regex of kernel names"
CFG=${1:-lexington}; STEPS=${2:-10}; PAT=${3:-emission_key}
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/ka_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka_$1 -- python3 $REPO/bench.py --config $CFG --steps $STEPS --no-cpu-baseline --no-also > /tmp/ka_$1.log 2>&1
  python3 - "$1" "$PAT" <<'PY'
import csv, glob, re, sys, json
name, pat = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/ka_%s/**/*kernel_stats.csv" % name, recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if re.search(pat, r["Name"])]
line = [l for l in open("/tmp/ka_%s.log" % name) if l.startswith("{")]
val = json.loads(line[-1])["ms_per_step"] if line else -1
print("%-8s ms/step %.2f  " % (name, val) + "  ".join("%s %.3f ms x%s" % (re.sub(r"\(.*", "", r["Name"])[:40], float(r["AverageNs"]) / 1e6, r["Calls"]) for r in rows))
PY
}
unset CMI_GPU_LIBRARY
run product
for L in "$REPO"/cmacionize_amd/variants/*.so; do
  export CMI_GPU_LIBRARY=$L
  run $(basename $L .so | sed 's/libcmi_gpu_//')
done
