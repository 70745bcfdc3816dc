#!/bin/bash
# Build container: recreate the tree the episode of DESIGN_LOG.md 9 was studied
# on - commit 3e4ff5c (the parent of the side-step f3f5c93) exported to
# gpurun_in/old (git-ignored; it travels to the GPU box with gpurun), its
# library, its oracle, and the variant libraries named on the command line.
#
#   tools/debug/episode_setup.sh [variant ...]
#     init     Packet<FULL> p = {}
#     w3 w4    the H-only kernels built for 3 / 4 waves per SIMD (no spills)
#     o1 o2    -O1 / -O2
#     fK aK    field K of the packet initialised / all fields but K (K = 1..13)
#     x1       -mllvm -amdgpu-spill-sgpr-to-vgpr=0
#     x4       -mllvm -amdgpu-waitcnt-forcezero=1
#     x5       -mllvm -enable-post-misched=0
#     x6       -mllvm -amdgpu-snop-padding=4
#     x7       -mllvm -amdgpu-prealloc-sgpr-spill-vgprs
#     swap     the base library with the spill store and the s_or_b64 exec of
#              the faulty join block in the other order (12 bytes)
# then, on the GPU box:  tools/debug/episode.sh base <variants>
#                        tools/debug/episode_trace.sh base x1 w3
#                        python3 tools/check_exec_spills.py gpurun_in/old/cmacionize_amd/libcmi_gpu.so
set -eu
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OLD=$REPO/gpurun_in/old
if [ ! -d "$OLD" ]; then
  mkdir -p "$REPO/gpurun_in"
  git -C "$REPO" archive --prefix=old/ 3e4ff5c | tar -x -C "$REPO/gpurun_in"
  rm -rf "$OLD/profiles" "$OLD"/BENCH* "$OLD"/GPUTEST*
  mkdir -p "$OLD/tools/debug"
  cp "$REPO/tools/debug/cont_decomposed.py" "$OLD/tools/debug/"
  python3 - "$OLD" "$REPO" <<'PY'
import sys
old, repo = sys.argv[1], sys.argv[2]
# the trace script: cont_decomposed.py's set-up, then what every block hands
# over after every stage
s = open(repo + "/tools/debug/cont_decomposed.py").read()
head = s[:s.index("group = EngineGroup([b.engine for b in backends])")]
open(old + "/tools/debug/cont_trace.py", "w").write(head + '''group = EngineGroup([b.engine for b in backends])
out = {}
def dump(stage):
    for r, b in enumerate(backends):
        b.synchronize()
        rows = b.take_exports().cpu().numpy().copy()
        ids = rows[:, 13].view(np.uint64) & np.uint64(0xffffffff)
        out["s%d_r%d_rows" % (stage, r)] = rows[np.argsort(ids, kind="stable")]
        t, c, ns = b.get_counters()
        out["s%d_r%d_counters" % (stage, r)] = np.array([t] + list(c) + [ns])
for b in backends:
    b.reset_grid()
    b.shoot(21, 0, 0, npacket)
dump(0)
stage = 0
while group.exchange_flights(21, 0):
    stage += 1
    dump(stage)
dump(stage + 1)
np.savez(os.environ.get("CMI_TRACE_OUT", "trace.npz"), **out)
print("stages", stage + 1)
''')
# debug switches in the old kernels.h
p = old + "/cmacionize_amd/csrc/kernels.h"
s = open(p).read()
s = s.replace("#define CMI_BLOCK 256\n",
              "#ifndef DBG_WAVES\n#define DBG_WAVES 6\n#endif\n"
              "#ifndef DBG_ALLBUT\n#define DBG_ALLBUT 0\n#endif\n"
              "#ifndef DBG_FIELD\n#define DBG_FIELD 0\n#endif\n"
              "#define CMI_BLOCK 256\n", 1)
s = s.replace(": ((PAD && !HEAT) ? CMI_PAD_WAVES : 6)))\n        shoot_kernel",
              ": ((PAD && !HEAT) ? CMI_PAD_WAVES : DBG_WAVES)))\n        shoot_kernel", 1)
s = s.replace("  Packet<FULL> p;\n  /* Lanes without a packet take part", '''#ifdef DBG_INIT
  Packet<FULL> p = {};
#else
  Packet<FULL> p;
#endif
#if DBG_FIELD || DBG_ALLBUT
#define DBG_SET(k, stmt) if (DBG_FIELD == k || (DBG_ALLBUT != 0 && DBG_ALLBUT != k)) { stmt; }
  for (int ax = 0; ax < 3; ++ax) {
    DBG_SET(1, p.pos[ax] = 0.)
    DBG_SET(2, p.dir[ax] = 0.)
    DBG_SET(3, p.inv_dir[ax] = 0.)
    DBG_SET(4, p.tmax[ax] = 0.)
    DBG_SET(5, p.tdelta[ax] = 0.)
    DBG_SET(6, p.cstep[ax] = 0)
    DBG_SET(7, p.rem[ax] = 0)
    DBG_SET(8, p.index[ax] = 0)
  }
  DBG_SET(9, p.tau = 0.)
  DBG_SET(10, p.sigma_He = 0.)
  DBG_SET(11, p.t = 0.)
  DBG_SET(12, p.cell = 0)
  DBG_SET(13, p.type = 0)
#endif
  /* Lanes without a packet take part''', 1)
open(p, "w").write(s)
PY
  make -C "$OLD/cmacionize_amd/csrc"
  make -C "$OLD/oracle"
fi
cd "$OLD/cmacionize_amd/csrc"
BASEF="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function"
for v in "$@"; do
  [ -f "../variants/libcmi_gpu_$v.so" ] && continue
  case $v in
    init) make variant NAME=$v DEFS="-DDBG_INIT" ;;
    w3|w4) make variant NAME=$v DEFS="-DDBG_WAVES=${v#w}" ;;
    o1|o2) make variant NAME=$v HIPFLAGS="${BASEF/-O3/-O${v#o}}" ;;
    f*) make variant NAME=$v DEFS="-DDBG_FIELD=${v#f}" ;;
    a*) make variant NAME=$v DEFS="-DDBG_ALLBUT=${v#a}" ;;
    x1) make variant NAME=$v HIPFLAGS="$BASEF -mllvm -amdgpu-spill-sgpr-to-vgpr=0" ;;
    x4) make variant NAME=$v HIPFLAGS="$BASEF -mllvm -amdgpu-waitcnt-forcezero=1" ;;
    x5) make variant NAME=$v HIPFLAGS="$BASEF -mllvm -enable-post-misched=0" ;;
    x6) make variant NAME=$v HIPFLAGS="$BASEF -mllvm -amdgpu-snop-padding=4" ;;
    x7) make variant NAME=$v HIPFLAGS="$BASEF -mllvm -amdgpu-prealloc-sgpr-spill-vgprs" ;;
    swap) mkdir -p ../variants && python3 - <<'PY'
b = open("../libcmi_gpu.so", "rb").read()
old = bytes.fromhex("30407CDC00167F00" "7E12FE87")  # scratch_store_dwordx4 v[22:25] off:48; s_or_b64 exec, exec, s[18:19]
assert b.count(old) == 1, b.count(old)
open("../variants/libcmi_gpu_swap.so", "wb").write(
    b.replace(old, bytes.fromhex("7E12FE87" "30407CDC00167F00")))
PY
      ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
ls ../variants 2>/dev/null || true
