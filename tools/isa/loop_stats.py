#!/usr/bin/env python3
"""Innermost loops of every function in a gfx950 assembly listing (hipcc -S --cuda-device-only), with the VALU instructions
of each classified the way docs/VALU_COSTS.md prices them: how many f32 add/mul/fma carry an SGPR operand (normal rate instead
of fast), how many lane moves (v_readlane / v_writelane: spilled SGPRs), f64, conversions, compares/selects ...

    python3 tools/isa/loop_stats.py /tmp/c0.s [min_instructions]
"""
import re
import sys
from collections import Counter

FAST_F32 = ("v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_subrev_f32", "v_mac_f32")


def classify(op, args):
    if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
        return "lane"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.endswith("_f64") and not op.startswith("v_cvt") and not op.startswith("v_cmp"):
        return "f64"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "cmp/sel"
    if op in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64"):
        return "trans"
    if op.split("_e")[0] in FAST_F32 or op in FAST_F32:
        srcs = args.split(",")[1:]
        if any(re.match(r"\s*-?\|?s\d+|\s*-?\|?s\[", s) or s.strip() in ("vcc_lo", "vcc_hi") for s in srcs):
            return "f32+sgpr"
        return "f32 fast"
    if "u64" in op or "_co_" in op or "addc" in op or op.endswith("_b64"):
        return "int64"
    if op.startswith("v_"):
        return "other valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("ds_", "global_", "buffer_", "scratch_", "flat_")):
        return "mem"
    return "?"


def main():
    path = sys.argv[1]
    min_len = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    lines = open(path).read().split("\n")
    fn, labels, body = None, {}, []
    out = []

    def flush():
        if fn is None:
            return
        loops = []
        for i, (ln, text) in enumerate(body):
            m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", text) or re.match(r"\s*s_branch\s+(\.LBB\d+_\d+)", text)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                loops.append((labels[m.group(1)], i))
        inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
        for a, b in inner:
            ops = Counter()
            n = 0
            for ln, text in body[a:b + 1]:
                t = text.strip()
                if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                    continue
                parts = t.split(None, 1)
                op, args = parts[0], (parts[1] if len(parts) > 1 else "")
                args = args.split(";")[0]
                ops[classify(op, args)] += 1
                n += 1
            if n >= min_len:
                out.append((fn, body[a][0], body[b][0], n, ops))

    for ln, text in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):", text)
        if m:
            flush()
            fn, labels, body = m.group(1), {}, []
            continue
        if fn is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", text)
        if m:
            labels[m.group(1)] = len(body)
        body.append((ln, text))
        if text.startswith(".Lfunc_end"):
            flush()
            fn = None
    total = Counter()
    for fn, a, b, n, ops in out:
        valu = sum(v for k, v in ops.items() if k not in ("salu", "mem", "?"))
        print(f"{fn[:70]:70s} L{a}-{b} n={n:4d} valu={valu:4d} " + " ".join(f"{k}={v}" for k, v in sorted(ops.items())))
        total.update(ops)
    print("TOTAL", dict(total))


main()
