#!/usr/bin/env python3
"""Static check of the distance between a matrix instruction and the first touch of its result registers.

MI355X has no interlock between v_mfma_f32_32x32x16_f16 and a vector instruction that reads (or overwrites) its
result: the 16 result registers are written four at a time while the instruction runs, and a read that comes too
early returns the old contents.  Measured with tools/mfma_hazard_probe.hip (one wave per SIMD, so nothing
stretches the distance), in wait states = instructions issued in between, `s_nop N` counting N + 1:

    result register   0   1-3   4   5-7   8   9-11   12   13-15
    needs at least    5    6    7    8    9    10    11    12

The ROCm 7.2 compiler keeps every such read 11 wait states behind the matrix instruction, one short for the last
three registers.  The library's own code reads the registers in ascending order, which keeps it safe; this script
proves that on the compiled ISA so that a compiler or source change that breaks it is caught without a GPU.

    tools/check_mfma_distance.py [ray-tracing_amd/csrc/rt_trace.hip more .hip files]      (default: every file with matrix code)

Exit status 1 and one line per violation if any register of a matrix result is touched too early.  Only the
fall-through path is followed (a taken branch inside the 12-instruction window is not).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]
NEED = [5, 6, 6, 6, 7, 8, 8, 8, 9, 10, 10, 10, 11, 12, 12, 12]
WINDOW = max(NEED)

REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs_of(text):
    """Set of (file, index) registers named in an operand string."""
    out = set()
    for m in REG.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(3)), int(m.group(4)) + 1):
                out.add((m.group(1), i))
    return out


def instructions(path):
    """(function, line number, mnemonic, operand text) of every instruction of an assembly file."""
    func = None
    for no, line in enumerate(open(path), 1):
        code = line.split(";", 1)[0].rstrip()
        if not code:
            continue
        if not code[0].isspace():
            if code.endswith(":") and not code.startswith(".L"):
                func = code[:-1]
            continue
        parts = code.split(None, 1)
        if parts[0].startswith("."):
            continue
        yield func, no, parts[0], parts[1] if len(parts) > 1 else ""


def check(asm_path):
    ins = list(instructions(asm_path))
    bad = []
    for i, (func, no, op, args) in enumerate(ins):
        if not op.startswith("v_mfma_f32_32x32x16"):
            continue
        dest = args.split(",", 1)[0]
        m = REG.search(dest)
        if not m or m.group(3) is None:
            continue
        file_, lo = m.group(1), int(m.group(3))
        waited = 0
        for func2, no2, op2, args2 in ins[i + 1:]:
            if func2 != func or waited >= WINDOW:
                break
            if op2.startswith("v_mfma") or op2.startswith("v_smfma"):
                waited += 1  # a dependent matrix instruction is interlocked by the hardware
                continue
            for f, r in regs_of(args2):
                if f == file_ and lo <= r < lo + 16 and waited < NEED[r - lo]:
                    bad.append("%s: line %d: %s %s touches register %d of the result of line %d after %d wait states (needs %d)"
                               % (func, no2, op2, args2, r - lo, no, waited, NEED[r - lo]))
            waited += int(args2.split()[0]) + 1 if op2 == "s_nop" else 1
    return bad


def main(argv):
    sources = argv[1:] or [os.path.join(ROOT, "ray-tracing_amd", "csrc", f) for f in ("rt_trace.hip", "rt_trace_dense.hip", "rt_wavefront.hip")]
    # the extra flags the build gives a source file (csrc/FLAGS: file name, then the flags)
    extra = {}
    for line in open(os.path.join(ROOT, "ray-tracing_amd", "csrc", "FLAGS")):
        if line.strip() and not line.startswith("#"):
            extra[line.split()[0]] = line.split()[1:]
    bad, count = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in sources:
            if src.endswith(".s"):
                asm = src
            else:
                asm = os.path.join(tmp, os.path.basename(src) + ".s")
                subprocess.run([HIPCC] + FLAGS + extra.get(os.path.basename(src), []) + ["-o", asm, src], check=True, stderr=subprocess.DEVNULL)
            count += sum(1 for _, _, op, _ in instructions(asm) if op.startswith("v_mfma_f32_32x32x16"))
            bad += check(asm)
    for b in bad:
        print(b)
    print("%d matrix instructions checked, %d early touches" % (count, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
