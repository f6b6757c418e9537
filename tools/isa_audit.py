#!/usr/bin/env python3
"""ISA audit of the step kernel's substep loops.

    python tools/isa_audit.py [--kernel <prefix of the mangled name, e.g. _ZN4taco16taco_step_kernelILi64ELi4ELb0ELb0ELb0ELb0ELb0ELb0EEE>] [--out profiles/rNN_isa_audit.txt]

Compiles taco_capi.hip to gfx950 assembly with the product flags (taco_amd/build.py FLAGS), cuts out one kernel, finds its natural
loops (a backward branch to a label), and prints for every loop that contains VALU work an instruction census priced with the issue
costs measured by tools/ubench/issue_mix (profiles/r01_e_ubench_issue_mix.txt, ns per wave-instruction at 4 waves per SIMD):

    (the table is PRICE below; SALU / branch / waitcnt are listed, not priced: they issue from the scalar port)

The census is static: a loop body's branches over rare blocks (ballot-guarded) are listed as separate inner regions.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# ns per wave-instruction on a SHARED SIMD (4 wavefronts per SIMD), measured by tools/ubench/issue_mix, bank and bank2 on MI355X
# (profiles/r01_e_ubench_issue_mix.txt, profiles/r02_a_ubench_bank.txt, profiles/r02_a_ubench_bank2.txt)
PRICE = {"vop2": 1.05,     # VOP1 / VOP2 with VGPR, literal or inline-constant sources (v_mul 1.04-1.06, v_fmaak 1.05, v_fmac 1.08-1.10)
         "vop3": 1.2,      # the same operands in a 64-bit encoding (v_fma 1.20, v_mul_e64 1.15, modifiers 1.21-1.24)
         "sgpr": 2.05,     # ANY VALU instruction with an SGPR source operand (v_mul s 2.10, v_fmac s 2.07, v_fma ..s 1.89, v_mov s 2.04)
         "med3": 1.9,      # v_med3 / v_max3 / v_min3 (1.78-2.03)
         "cmp": 1.9,       # v_cmp* (to VCC 2.02, to an SGPR pair 1.85)
         "cndmask": 2.0,   # v_cndmask (reads VCC or an SGPR pair: 2.02)
         "trans": 3.7,     # v_rcp / v_sqrt / v_rsq / v_exp / v_log (3.5-3.9)
         "dpp": 1.2}
TRANS = ("v_rcp_", "v_sqrt_", "v_rsq_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
VREG = re.compile(r"\bv(\d+|\[\d+:\d+\])")
SREG = re.compile(r"(?<![a-z_])(s\d+|s\[\d+:\d+\]|vcc|exec)\b")


def compile_asm():
    from taco_amd import build as b
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "step.s")
    flags = [f for f in b.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([b.HIPCC] + flags + ["--cuda-device-only", "-S", "-o", out, os.path.join(b.CSRC, "taco_capi.hip")])
    return out


def cut_kernel(path, name):
    lines = open(path).read().split("\n")
    # (a PREFIX of the mangled name is enough: the argument list behind the template arguments changes with the kernel's signature)
    start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1]


def classify(ins):
    op = ins.split()[0]
    if op.startswith("s_"):
        if op.startswith("s_cbranch") or op == "s_branch":
            return "branch"
        if op.startswith("s_waitcnt") or op == "s_nop":
            return "wait"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_"):
        return "vmem"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "cmp"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_"):
        args = ins[len(op):]
        srcs = args.split(",")[1:]  # first operand is the destination
        if op.startswith(("v_div_scale", "v_readlane", "v_readfirstlane")):
            srcs = srcs[1:] if op.startswith("v_div_scale") else srcs  # v_div_scale has two destinations
        if "dpp" in ins or "quad_perm" in ins or "row_" in ins:
            return "dpp"
        if any(SREG.search(x) for x in srcs) or op.startswith("v_div_fmas"):  # (v_div_fmas reads VCC implicitly)
            return "sgpr"
        if op.startswith(("v_med3", "v_max3", "v_min3")):
            return "med3"
        three = op.endswith("_e64") or op.startswith(("v_fma_", "v_mad_", "v_bfi", "v_lshl_add", "v_lshl_or", "v_add3", "v_div_fixup", "v_div_scale", "v_mul_hi", "v_mul_lo",
                                                       "v_and_or", "v_or3", "v_xad", "v_alignbit", "v_perm", "v_cvt_pk", "v_ldexp", "v_mbcnt", "v_readlane", "v_writelane"))
        return "vop3" if three else "vop2"
    return "other"


def audit(lines):
    # labels and instructions
    label_at = {}
    prog = []  # (kind, text)
    for l in lines:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                label_at[m.group(1)] = len(prog)
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            label_at[m.group(1)] = len(prog)
            continue
        prog.append(s.split(";")[0].strip())
    loops = []
    for i, ins in enumerate(prog):
        if ins.startswith("s_cbranch") or ins.startswith("s_branch"):
            tgt = ins.split()[-1]
            if tgt in label_at and label_at[tgt] <= i:
                loops.append((label_at[tgt], i, tgt))
    return prog, loops


def census(prog, lo, hi):
    c = collections.Counter()
    ops = collections.Counter()
    for ins in prog[lo:hi + 1]:
        k = classify(ins)
        c[k] += 1
        ops[(k, ins.split()[0])] += 1
    return c, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="_ZN4taco16taco_step_kernelILi64ELi1ELb0ELb1ELb0ELb0ELb0ELb0EEE")
    ap.add_argument("--asm", default=None, help="use this assembly file instead of compiling")
    ap.add_argument("--out", default=None)
    ap.add_argument("--min-valu", type=int, default=150, help="only report loops with at least this many VALU instructions")
    ap.add_argument("--ops", action="store_true", help="list every opcode of the reported loops")
    args = ap.parse_args()
    path = args.asm or compile_asm()
    lines = cut_kernel(path, args.kernel)
    prog, loops = audit(lines)
    out = []
    out.append(f"kernel {args.kernel}: {len(prog)} instructions, {len(loops)} backward branches")
    tot, _ = census(prog, 0, len(prog) - 1)
    out.append("whole kernel (static): " + ", ".join(f"{k} {v}" for k, v in sorted(tot.items())))
    # innermost-first: report loops that are not strictly containing another reported loop with the same VALU mass
    for lo, hi, tgt in sorted(loops, key=lambda t: t[1] - t[0]):
        c, ops = census(prog, lo, hi)
        valu = sum(c[k] for k in PRICE)
        if valu < args.min_valu:
            continue
        ns = sum(c[k] * PRICE[k] for k in PRICE)
        out.append("")
        out.append(f"loop {tgt}: instructions {lo}..{hi} ({hi - lo + 1}), VALU {valu}, priced {ns:.0f} ns per iteration per wavefront")
        out.append("  " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())))
        inner = [(a, b, t) for a, b, t in loops if a >= lo and b <= hi and (a, b) != (lo, hi)]
        if inner:
            out.append("  inner loops: " + ", ".join(f"{t}[{a}..{b}]" for a, b, t in inner))
        # forward branches inside the loop = rare-form regions
        fw = [ins for ins in prog[lo:hi + 1] if ins.startswith("s_cbranch")]
        out.append(f"  conditional branches inside: {len(fw)}")
        top = collections.Counter()
        for (k, op), v in ops.items():
            if k in PRICE:
                top[f"{op}[{k}]"] += v
        out.append("  top VALU opcodes: " + ", ".join(f"{o} {v}" for o, v in top.most_common(40)))
        if args.ops:
            for ins in prog[lo:hi + 1]:
                out.append("      " + ins)
    text = "\n".join(out)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
