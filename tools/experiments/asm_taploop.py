#!/usr/bin/env python3
"""What hipcc made of the tap loop of one msda_fwd_group instantiation: how many ds_read_b128 are issued back to back before
a wait (LDS reads in flight), scratch traffic inside the level loop, instruction mix between the level loop's barriers.

    python tools/experiments/asm_taploop.py <file.s> <substring of the mangled kernel name> [more substrings]
(make the .s with: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=fast -S --cuda-device-only)
"""
import collections
import sys


def main():
    path, keys = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if l.startswith("_ZN") and l.split(":")[0] and all(k in l for k in keys)
              and l.rstrip().split(";")[0].rstrip().endswith(":")]
    for st in starts:
        end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = [l.strip() for l in lines[st:end]]
        ins = [l for l in body if l and not l.startswith((";", ".")) and not l.split()[0].endswith(":")]
        bars = [i for i, l in enumerate(ins) if l.startswith("s_barrier")]
        print(lines[st].split(":")[0][:150])
        print(f"  instructions {len(ins)}; s_barrier at {bars}")
        if len(bars) < 2:
            continue
        # the level loop: from its second barrier to the next backward branch region -- take the longest barrier-free stretch
        spans = [(bars[i], bars[i + 1]) for i in range(len(bars) - 1)] + [(bars[-1], len(ins))]
        a, b = max(spans, key=lambda ab: ab[1] - ab[0])
        loop = ins[a:b]
        c = collections.Counter(l.split()[0] for l in loop)
        print(f"  longest barrier-free stretch: {len(loop)} instructions; " + ", ".join(f"{k} {v}" for k, v in c.most_common(14)))
        runs, cur = collections.Counter(), 0
        for l in loop:
            op = l.split()[0]
            if op.startswith("ds_read"):
                cur += 1
            elif op.startswith("s_waitcnt") and "lgkmcnt" in l:
                if cur:
                    runs[cur] += 1
                cur = 0
        print("  ds_read issued between lgkmcnt waits (run length: count):", dict(sorted(runs.items())))
        print("  scratch ops in that stretch:", sum(1 for l in loop if l.startswith("scratch_")),
              " v_readlane/v_writelane:", sum(1 for l in loop if l.startswith(("v_readlane", "v_writelane"))))


if __name__ == "__main__":
    main()
