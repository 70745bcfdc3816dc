#!/usr/bin/env python3
"""Static check of ONE kernel's ISA (llvm -S output): is every SGPR-spill lane
(v_readlane_b32 sX, vN, k) and every scratch slot (scratch_load ... offset:o)
written on EVERY path from the kernel's entry before it is read?

Forward "definitely written" dataflow over the basic blocks (labels, s_branch,
s_cbranch_*): in-state = intersection of the predecessors' out-states. A read
that is not covered is reported with its line - a value the allocator considers
undefined on some path (harmless if nothing depends on it there), or a bug.

usage: spill_dataflow.py file.s kernel_symbol"""
import re
import sys

path, symbol = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(symbol + ":"))
end = next(i for i in range(start, len(lines))
           if lines[i].startswith(".Lfunc_end"))
body = lines[start + 1:end]

label_re = re.compile(r"^(\.LBB\d+_\d+):")
blocks = []          # (name, [(lineno, text)])
cur = ("entry", [])
for k, l in enumerate(body):
    m = label_re.match(l)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [])
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur[1].append((start + 2 + k, t))
blocks.append(cur)
index = {name: i for i, (name, _) in enumerate(blocks)}

succ = [[] for _ in blocks]
for i, (name, ins) in enumerate(blocks):
    fall = True
    for _, t in ins:
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", t)
        if m:
            succ[i].append(index[m.group(1)])
        m = re.match(r"s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            succ[i].append(index[m.group(1)])
            fall = False
        if t.startswith("s_endpgm"):
            fall = False
    if fall and i + 1 < len(blocks):
        succ[i].append(i + 1)
pred = [[] for _ in blocks]
for i, ss in enumerate(succ):
    for s in ss:
        pred[s].append(i)

wl = re.compile(r"v_writelane_b32\s+(v\d+),\s*\S+,\s*(\d+)")
rl = re.compile(r"v_readlane_b32\s+\S+,\s*(v\d+),\s*(\d+)")
st = re.compile(r"scratch_store_(dword\w*)\s+off,\s*\S+,\s*off(?:\s+offset:(\d+))?")
ld = re.compile(r"scratch_load_(dword\w*)\s+\S+,\s*off,\s*off(?:\s+offset:(\d+))?")
width = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}


def effects(t):
    """(writes, reads) as sets of keys"""
    m = wl.match(t)
    if m:
        return {("lane", m.group(1), int(m.group(2)))}, set()
    m = rl.match(t)
    if m:
        return set(), {("lane", m.group(1), int(m.group(2)))}
    m = st.match(t)
    if m:
        o = int(m.group(2) or 0)
        return {("scratch", o + 4 * k) for k in range(width[m.group(1)])}, set()
    m = ld.match(t)
    if m:
        o = int(m.group(2) or 0)
        return set(), {("scratch", o + 4 * k) for k in range(width[m.group(1)])}
    return set(), set()


universe = set()
for _, ins in blocks:
    for _, t in ins:
        w, r = effects(t)
        universe |= w | r
out = [set(universe) for _ in blocks]
out[0] = set()
changed = True
while changed:
    changed = False
    for i, (name, ins) in enumerate(blocks):
        state = set() if i == 0 else (
            set.intersection(*[out[p] for p in pred[i]]) if pred[i]
            else set(universe))
        for _, t in ins:
            w, _ = effects(t)
            state |= w
        if state != out[i]:
            out[i] = state
            changed = True
bad = 0
for i, (name, ins) in enumerate(blocks):
    state = set() if i == 0 else (
        set.intersection(*[out[p] for p in pred[i]]) if pred[i]
        else set(universe))
    for no, t in ins:
        w, r = effects(t)
        for key in sorted(r - state):
            print("%s line %d: %s   <- %s not written on every path" %
                  (name, no, t, key))
            bad += 1
        state |= w
print("%d blocks, %d spill locations, %d reads not covered" %
      (len(blocks), len(universe), bad))
