#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing (tuning aid).

usage: isa_blocks.py listing.s <substring of the mangled kernel name> [min instructions per block]
Prints every basic block with its VALU / SALU / LDS / VMEM / scratch counts, the blocks that branch backwards (loops)
marked, and a total.  Half-rate VALU forms (v_cvt_*, DPP, v_pk_*, 64-bit) are counted in `slow` as well.
"""
import re
import sys


def classify(op):
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if key in l and not l.startswith("\t") and re.match(r"^\S+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, name = [], [], "entry"
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((";", ".")) and not re.match(r"\.LBB\d+_\d+:", s):
            continue
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        cur.append(s)
    blocks.append((name, cur))
    order = {n: i for i, (n, _) in enumerate(blocks)}
    tot = {}
    for i, (n, ins) in enumerate(blocks):
        c = {}
        back = []
        for s in ins:
            op = s.split()[0]
            k = classify(op)
            c[k] = c.get(k, 0) + 1
            if k == "valu" and (op.startswith(("v_cvt", "v_pk_", "v_lshl_add_u64", "v_mul_hi", "v_mad_u64", "v_mul_lo")) or "dpp" in s or "_f64" in op or "_b64" in op or "_u64" in op):
                c["slow"] = c.get("slow", 0) + 1
            if op in ("v_readlane_b32", "v_writelane_b32"):
                c["lane"] = c.get("lane", 0) + 1
            if op.startswith(("s_cbranch", "s_branch")):
                t = s.split()[-1]
                if t in order and order[t] <= i:
                    back.append(t)
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
        if len(ins) >= min_n or back:
            print(f"{n:12s} n={len(ins):5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + (f"  <- loop to {back}" if back else ""))
    print("total", tot)


if __name__ == "__main__":
    main()
