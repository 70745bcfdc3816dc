#!/usr/bin/env python3
"""After `make -C cmacionize_amd/csrc asm`: the loops of every tile_kernel
variant that hold its LDS adds (the march loop and the loops around it), their
sizes and the scratch (spill) accesses inside them.

    python tools/check_tile_loops.py [cmacionize_amd/csrc/engine.s]
"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "cmacionize_amd/csrc/engine.s"
text = open(path).read().split("\n")
starts = [i for i, l in enumerate(text) if re.match(r"^_Z11tile_kernelI\S*:", l)]
for s in starts:
    e = next(i for i in range(s, len(text)) if "s_endpgm" in text[i])
    body = text[s:e]
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
    adds = [i for i, l in enumerate(body) if "ds_add_f64" in l]
    print(text[s].split(":")[0], "lines", len(body), "scratch",
          sum("scratch_" in l for l in body))
    for h, en in loops:
        inner = [a for a in adds if h < a < en]
        if inner:
            loop = body[h:en + 1]
            count = lambda p: sum(1 for l in loop if l.strip().startswith(p))
            print("   loop of %5d lines: valu %4d salu %4d lds %3d global %3d "
                  "scratch %3d" % (len(loop), count("v_"), count("s_"),
                                   count("ds_"), count("global_"),
                                   sum("scratch_" in l for l in loop)))
