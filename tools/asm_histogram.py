#!/usr/bin/env python3
"""Instruction histogram of a stretch of a kernel's device assembly (build/asm/*.s from `make -C kifs_raymarching_amd/csrc
asm`): by class -- fma (v_fma / v_fmac / v_fmaak / v_fmamk: the polynomial and Newton chains), mul/add, packed, select +
compare, quarter-rate (v_rcp / v_sqrt / v_rsq / v_exp / v_log / v_sin / v_cos), convert / round, integer and bit moves,
scalar, branch, wait states.  Used for the generalised Julia set's orbit step.

    python tools/asm_histogram.py KERNEL_SUBSTRING FIRST_LABEL LAST_LABEL        (labels as in the .s file, inclusive)
    python tools/asm_histogram.py --genjulia                                      (finds the orbit step's blocks itself)
"""
import collections
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ASM = ROOT / "build" / "asm" / "kifs_kernels.s"


def classify(op):
    if op.startswith("v_pk_"):
        return "packed f32"
    if re.match(r"v_(fma|fmac|fmaak|fmamk)_", op):
        return "fma"
    if re.match(r"v_(mul|add|sub|subrev|max|min)_f32", op):
        return "mul / add / max"
    if re.match(r"v_(rcp|sqrt|rsq|exp|log|sin|cos)_", op):
        return "quarter rate"
    if re.match(r"v_(cmp|cmpx|cndmask)", op):
        return "select + compare"
    if re.match(r"v_(cvt|rndne|trunc|floor|ceil|fract)", op):
        return "convert / round"
    if op.startswith("v_"):
        return "integer / bit / move"
    if op.startswith("s_nop") or op.startswith("s_waitcnt") or op.startswith("s_sleep"):
        return "wait state"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "scalar"
    if op.startswith("ds_") or op.startswith("global_") or op.startswith("buffer_"):
        return "memory"
    return "other"


def function_blocks(lines, kernel):
    start = next(i for i, l in enumerate(lines) if kernel in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\w+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ["entry", []]
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %bb\.(\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1) if l.startswith(".L") else "%bb." + m.group(1), []]
            continue
        t = l.strip()
        if t and not t.startswith(";") and not t.startswith("."):
            cur[1].append(t.split()[0])
    blocks.append(cur)
    return blocks


def main():
    lines = ASM.read_text().split("\n")
    if sys.argv[1:] == ["--genjulia"]:
        blocks = function_blocks(lines, "render_group_kernelILi2ELi0ELi2E")
        # the orbit step: the block with the two exp2 and the sin/cos range reductions (three v_rndne) and a reciprocal, and its neighbours
        # back to the loop header and on to the back edge
        k = next(i for i, b in enumerate(blocks) if sum(op.startswith("v_rndne") for op in b[1]) >= 3
                 and any(op.startswith("v_rcp") for op in b[1]) and len(b[1]) > 100)
        first = k
        while first > 0 and not any(op.startswith("s_cbranch_vccnz") for op in blocks[first][1][:3]):
            first -= 1
        last = k
        while not any(op == "s_cbranch_execnz" for op in blocks[last][1]):
            last += 1
        chosen = blocks[first:last + 1]
    else:
        kernel, a, b = sys.argv[1:4]
        blocks = function_blocks(lines, kernel)
        names = [x[0] for x in blocks]
        chosen = blocks[names.index(a):names.index(b) + 1]
    total = collections.Counter()
    print("| block | instructions | " + " |")
    for name, ops in chosen:
        c = collections.Counter(classify(op) for op in ops)
        total.update(c)
        print(f"  {name:12s} {len(ops):4d}  " + ", ".join(f"{k} {v}" for k, v in c.most_common()))
    n = sum(total.values())
    print(f"  {'TOTAL':12s} {n:4d}")
    for k, v in total.most_common():
        print(f"    {k:24s} {v:4d}  {100.0 * v / n:5.1f} %")


if __name__ == "__main__":
    main()
