"""wino3x3 UR form: the U fragments are loaded by inline-asm buffer_load_dwordx4 into registers two K-steps ahead of their use.  The compiler does not
know that these registers are in flight, so nothing but the MFMAs may touch them between the first load and the last MFMA of a wave's loop -- a
v_mov / spill of such a register would read whatever the register held before its load landed.  This script disassembles the kernel (or reads a
.s file) and checks exactly that, per wave-row instance.  Used by tests/test_wino_cpu.py; prints the register sets."""
import re
import subprocess
import sys


KERNEL = "_ZN6lspf2f7wino3x3ILi1ELi3ELb1ELb1ELb1ELb0E"      # wino3x3<1, 3, true, true, UR = true, WT = false>
KERNEL_NS4 = "_ZN6lspf2f7wino3x3ILi1ELi4ELb1ELb1ELb1ELb0E"  # four register sets, three steps ahead (tune key wino_ureg=2)
KERNEL_WT = "_ZN6lspf2f7wino3x3ILi1ELi3ELb1ELb1ELb1ELb1E"   # ... with write-through output stores (tune key out_wt)


def compile_to_asm(src, hipcc="hipcc"):
    """device assembly of one source with the flags csrc/Makefile builds it with"""
    return subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-S", "--cuda-device-only", src, "-o", "-"],
                          check=True, capture_output=True, text=True).stdout


def kernel_text(asm, mangled_prefix):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and l.rstrip().endswith(tuple(":")) or (l.startswith(mangled_prefix) and ":" in l[:len(mangled_prefix) + 80]))
    end = next((i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines))
    return lines[start:end]


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(lines):
    """regions checked: every K loop (blocks the asm printer marks as part of a loop that holds the MFMAs) and each prologue (first asm load of a
    wave-row instance .. its loop header).  Behind a loop every load has landed (the last step waits vmcnt(0)), so the epilogue may reuse the registers."""
    load_re = re.compile(r"\s*buffer_load_dwordx4 (v\[\d+:\d+\]), v\d+, s\[\d+:\d+\], s\d+ offen offset:")
    label_re = re.compile(r"^(\.LBB\d+_\d+):(.*)$")
    loop_of = [None] * len(lines)
    cur = None
    for i, l in enumerate(lines):
        m = label_re.match(l)
        if m:
            c = m.group(2)
            if "Loop Header" in c:
                cur = m.group(1)[2:]
            elif "in Loop: Header=" in c:
                cur = re.search(r"Header=(BB\d+_\d+)", c).group(1)
            else:
                cur = None
        loop_of[i] = cur
    kloops = {loop_of[i] for i, l in enumerate(lines) if "v_mfma_f32_32x32x2_f32" in l and loop_of[i]}
    region = [loop_of[i] in kloops for i in range(len(lines))]
    loads = [i for i, l in enumerate(lines) if load_re.match(l)]
    if not loads or not kloops:
        raise AssertionError("no register-form U loads / K loops in this kernel")
    for i in loads:                                          # prologues: from a load outside the loops to the next K-loop block
        if not region[i]:
            j = i
            while j < len(lines) and not region[j] and j < i + 200:
                region[j] = True
                if re.match(r"\s*(s_branch|s_endpgm|s_setpc)", lines[j]):          # the block chain of this prologue ends here (its tail blocks sit by the loop)
                    break
                j += 1
    ureg = set()
    for i in loads:
        ureg |= regs_of(load_re.match(lines[i]).group(1))
    nm = sum("v_mfma_f32_32x32x2_f32" in l for l in lines)
    bad = []
    for i, l0 in enumerate(lines):
        if not region[i]:
            continue
        l = l0.split(";")[0].strip()
        if not l or l.startswith((".", "s_")) or load_re.match(l0):
            continue
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", l)
        touched = set().union(*[regs_of(t) for t in toks]) if toks else set()
        if not (touched & ureg):
            continue
        if l.startswith("v_mfma_f32_32x32x2_f32"):
            ops = [o.strip() for o in l[len("v_mfma_f32_32x32x2_f32"):].split(",")]      # D, A, B, C: only B may be a U register
            if not (regs_of(ops[0]) | regs_of(ops[1]) | regs_of(ops[3])) & ureg:
                continue
        bad.append((i, l0))
    return sorted(ureg), len(loads), nm, bad


KERNELS = {                      # file -> [(mangled prefix, U registers, MFMAs, loads)]
    "wino.hip": [(KERNEL, 48, 4 * 3 * 16, 4 * (2 * 4 + 3 * 4)), (KERNEL_WT, 48, 4 * 3 * 16, 4 * (2 * 4 + 3 * 4)),
                 (KERNEL_NS4, 64, 4 * 4 * 16, 4 * (3 * 4 + 4 * 4))],
}


def main():
    import argparse
    import os
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--hipcc", default="hipcc")
    ap.add_argument("--quiet", action="store_true", help="print only failures (the Makefile step)")
    a = ap.parse_args()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "livespeechportraits_amd", "csrc")
    rc = 0
    for f, kernels in KERNELS.items():
        asm = compile_to_asm(os.path.join(root, f), a.hipcc)
        for prefix, nreg, nmfma, nloads in kernels:
            ureg, nl, nm, bad = check(kernel_text(asm, prefix))
            fail = bool(bad) or len(ureg) != nreg or nm != nmfma or nl != nloads
            if fail or not a.quiet:
                print("%s %s: U registers v%d..v%d (%d, expected %d), %d loads (%d), %d MFMAs (%d), %d foreign touches" % (
                    f, prefix[12:], ureg[0], ureg[-1], len(ureg), nreg, nl, nloads, nm, nmfma, len(bad)))
            for i, l in bad[:20]:
                print("  line %d: %s" % (i, l.strip()))
            rc |= 1 if fail else 0
    if rc:
        print("check_ureg_asm: this toolchain's wino3x3 touches a U register in flight -- do NOT ship this build (csrc/wino.hip, UR form)", file=sys.stderr)
    return rc


if __name__ == "__main__":
    sys.exit(main())
