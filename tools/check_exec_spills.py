#!/usr/bin/env python3
"""Scan the device code of a built libcmi_gpu.so for the miscompile behind the
"result that changed with one more kernel argument" (DESIGN_LOG.md 9):

    <join block of a divergent if / else>:
        scratch_store_dwordx4 off, v[22:25], off offset:48   ; spill
        s_or_b64 exec, exec, s[18:19]                        ; lanes of the
                                                             ; other side back on

A spill (or reload) that the register allocator put at the top of a join block
AHEAD of the `s_or_b64 exec, exec, sN` that re-enables the lanes of the other
branch runs for the lanes of the fall-through predecessor only; the other
lanes' copy of the value never reaches the slot, and the reload (under the full
mask) hands them whatever the slot held. ROCm 7.2.0's clang did this once, in
`shoot_kernel<false, true, false, false, true>` of commit 3e4ff5c (p.pos[1] of
the packets of an isotropic continuous source).

The library's gfx950 code object is extracted (llvm-objcopy,
clang-offload-bundler) and disassembled with branch targets as labels
(llvm-objdump --symbolize-operands); every label whose block starts with
scratch_store / scratch_load instructions followed by `s_or_b64 exec, exec`
(the join) or `s_or_saveexec_b64` (the start of an else side: spill code of
that block is right where the compiler normally puts it, between this
instruction and the `s_xor_b64 exec` that narrows the mask again) is reported.
Exit status 1 if there is one.

usage: check_exec_spills.py path/to/libcmi_gpu.so"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        obj = os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary",
                        "--only-section=.hip_fatbin", lib, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"),
                        "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--output=" + obj], check=True,
                       stderr=subprocess.DEVNULL)
        out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d",
                              "--symbolize-operands", "--no-show-raw-insn",
                              obj], check=True, capture_output=True,
                             text=True)
    return out.stdout.split("\n")


def scan(lines):
    hits, function = [], None
    for i, line in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
        if not m:
            continue
        if not re.match(r"^L\d+$", m.group(1)):
            function = m.group(1)
            continue
        pending, j = [], i + 1
        while j < len(lines):
            t = lines[j].split("//")[0].strip()
            if not t:
                j += 1
                continue
            if t.startswith(("scratch_store", "scratch_load")):
                pending.append(t)
                j += 1
                continue
            if t.startswith(("s_waitcnt", "s_nop")):
                j += 1
                continue
            # (the join of an if / else: s_or_b64 exec, exec, sN; the start of
            # an else side: s_or_saveexec_b64 sA, sB - there the spill code of
            # the block belongs BETWEEN it and the s_xor_b64 that narrows the
            # mask to the else side, where all lanes of the if / else are on)
            if t.startswith(("s_or_b64 exec, exec,", "s_or_saveexec_b64")) \
                    and pending:
                hits.append((function, m.group(1), pending, t))
            break
    return hits


def main():
    lib = sys.argv[1]
    lines = disassemble(lib)
    hits = scan(lines)
    labels = sum(1 for l in lines if re.match(r"^[0-9a-f]+ <L\d+>:", l))
    for function, label, pending, restore in hits:
        print("%s, block %s: %s BEFORE %s" % (function, label,
                                              "; ".join(pending), restore))
    print("%s: %d branch targets, %d with spill code ahead of the exec "
          "restore" % (os.path.basename(lib), labels, len(hits)))
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
