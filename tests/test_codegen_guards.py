"""Guards on the code the compiler emitted for the shipped library.

DESIGN_LOG.md 9: round 5's "result that changed with one more kernel argument"
was a miscompile - ROCm 7.2.0's clang put a register spill at the top of a
control-flow join block AHEAD of the `s_or_b64 exec, exec, sN` that re-enables
the lanes of the other branch, so only the fall-through side's lanes spilled
their copy of `p.pos[1]`. tools/check_exec_spills.py finds that pattern in the
disassembly of the built library (it flags the failing build of commit
3e4ff5c and none of the builds that gave the right result); the product
library must be free of it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "cmacionize_amd", "libcmi_gpu.so")

# the block of shoot_kernel<false, true, false, false, true> of the failing
# build (llvm-objdump -d --symbolize-operands), and the same with the two
# instructions in the right order
FAILING = """
000000000020f000 <_Z12shoot_kernelILb0ELb1ELb0ELb0ELb1ELb0ELb0ELb0EEv9ShootArgs>:
	v_mov_b64_e32 v[24:25], v[2:3]                             // 00000020F55C: 7E307102

000000000020f560 <L9698>:
	scratch_store_dwordx4 off, v[22:25], off offset:48         // 00000020F560: DC7C4030 007F1600
	s_or_b64 exec, exec, s[18:19]                              // 00000020F568: 87FE127E
	scratch_store_dwordx2 v27, v[32:33], off                   // 00000020F56C: DC746000 007F201B
"""
RIGHT = FAILING.replace(
    "\tscratch_store_dwordx4 off, v[22:25], off offset:48         "
    "// 00000020F560: DC7C4030 007F1600\n"
    "\ts_or_b64 exec, exec, s[18:19]                              "
    "// 00000020F568: 87FE127E\n",
    "\ts_or_b64 exec, exec, s[18:19]                              "
    "// 00000020F560: 87FE127E\n"
    "\tscratch_store_dwordx4 off, v[22:25], off offset:48         "
    "// 00000020F564: DC7C4030 007F1600\n")


def test_scanner_flags_the_failing_block_and_not_the_repaired_one():
    import check_exec_spills as c
    hits = c.scan(FAILING.split("\n"))
    assert len(hits) == 1
    function, label, pending, restore = hits[0]
    assert function.startswith("_Z12shoot_kernelILb0ELb1ELb0ELb0ELb1")
    assert label == "L9698" and "v[22:25]" in pending[0]
    assert restore.startswith("s_or_b64 exec, exec, s[18:19]")
    assert RIGHT != FAILING and c.scan(RIGHT.split("\n")) == []
    # waits between the two (a build with -amdgpu-waitcnt-forcezero) do not
    # hide it
    padded = FAILING.replace("007F1600\n", "007F1600\n\ts_waitcnt vmcnt(0)\n")
    assert len(c.scan(padded.split("\n"))) == 1


def test_product_library_has_no_spill_ahead_of_an_exec_restore():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "cmacionize_amd",
                                                   "csrc")], check=True)
    r = subprocess.run([sys.executable,
                        os.path.join(ROOT, "tools", "check_exec_spills.py"),
                        LIB], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 with spill code ahead of the exec restore" in r.stdout
