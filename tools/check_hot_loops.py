#!/usr/bin/env python3
"""After `make -C cmacionize_amd/csrc asm`: for every shoot_kernel variant,
find the march loop (the innermost loop that holds the combining table's
ds_cmpst) in engine.s and report its size and - the point - whether the
register allocator left scratch (spill) accesses inside it. The kernels are
built with a forced occupancy; the cold code (emission, end of flight) is
allowed to spill, the march loop is not.

    python tools/check_hot_loops.py [cmacionize_amd/csrc/engine.s]
exit code 1 if a variant the benchmark configs launch has scratch accesses in
its march loop.
"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "cmacionize_amd/csrc/engine.s"
text = open(path).read().split("\n")
starts = [i for i, l in enumerate(text)
          if re.match(r"^_Z12shoot_kernelI(Lb[01]E){6,9}Ev9ShootArgs:", l)]
bad = 0
print("%-22s %6s %6s %6s %6s %8s" % ("variant <F,H,R,X,T,P,PAD,TRK,Q>", "lines", "valu",
                                     "salu", "lds", "scratch"))
for s in starts:
    e = next(i for i in range(s, len(text)) if "s_endpgm" in text[i])
    body = text[s:e]
    flags = re.findall(r"Lb([01])E", text[s].split("Ev9ShootArgs")[0])[:9]
    cas = [i for i, l in enumerate(body) if "ds_cmpst" in l]
    if not cas:
        continue
    # all loops of the kernel: (label line, line of the last branch back to it)
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if not m:
            continue
        back = [j for j in range(i + 1, len(body))
                if re.search(r"s_c?branch\w*\s+" + re.escape(m.group(1)) + r"\b",
                             body[j])]
        if back:
            loops.append((i, back[-1]))
    # the march loop: the smallest loop around the table's compare-and-swap
    # that is more than the probing loop itself
    around = [(e2 - h2, h2, e2) for h2, e2 in loops
              if h2 < cas[0] < e2 and e2 - h2 > 100]
    if not around:
        continue
    _, h, end = min(around)
    loop = body[h:end + 1]
    count = lambda p: sum(1 for l in loop if l.strip().startswith(p))
    scratch = sum(1 for l in loop if "scratch_" in l)
    name = "<%s>" % ",".join(flags)
    print("%-22s %6d %6d %6d %6d %8d" % (name, len(loop), count("v_"),
                                         count("s_"), count("ds_"), scratch))
    # fail for the variants the benchmark configs launch (hydrogen-only, and
    # multi-ion with heating; no inline re-emission, incremental marcher);
    # elsewhere a spill in the loop is only reported
    if scratch and tuple(flags[:5]) in (("0", "0", "0", "0", "1"),
                                        ("1", "1", "0", "0", "1")):
        bad = 1
sys.exit(bad)
