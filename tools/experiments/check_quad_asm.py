#!/usr/bin/env python3
"""Audit of the hand-counted s_waitcnt vmcnt(N) scheme in msda_forward_quad.hip (see the comment above make_desc there).

Compiles the file to gfx950 assembly with the Makefile's flags and, for every msda_fwd_quad instantiation, walks the
steady-state level loop (twice, to carry state over the back edge) with a model of the in-order VMEM queue:

  * every VMEM instruction enters the queue; `s_waitcnt vmcnt(N)` retires all but the N youngest;
  * no instruction may name a VGPR that is the destination of a load still in the queue;
  * the loop must contain no scratch (spill) traffic and no compiler-issued global/flat loads.

Exit status 1 on any violation.  Run by tests/test_cabi.py (no GPU needed) and by hand after touching the kernel.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mvdetr_amd", "csrc")
HIPFLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=fast"]

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
VMEM = re.compile(r"^\s*(buffer_|global_|scratch_|flat_)(load|store|atomic)")


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels(asm):
    cur, name = None, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN6mvdetr13msda_fwd_quad\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                yield name, cur
                cur = None
            else:
                cur.append(line)


def level_loop(lines):
    """[start, end) of the level loop: from its header label to the barrier that ends a level."""
    barriers = [i for i, l in enumerate(lines) if re.match(r"^\s*s_barrier", l)]
    if len(barriers) < 2:
        return None
    end = barriers[1]
    first_dma = next(i for i in range(barriers[0], end) if re.search(r"buffer_load_dwordx4 .* lds", lines[i]))
    start = max(i for i in range(barriers[0], first_dma) if re.match(r"^\.LBB\d+_\d+:", lines[i]))
    return start, end + 1


def audit(name, lines):
    span = level_loop(lines)
    if span is None:
        return [f"{name}: level loop not found"]
    body = [l for l in lines[span[0]:span[1]] if l.strip() and not l.strip().startswith(";")]
    problems, queue = [], []          # queue of (is_load_to_vgpr, dest regs)
    stats = {"vmem": 0, "waits": [], "instrs": len(body)}
    for rep in range(2):
        for l in body:
            code = l.split(";")[0]
            if re.match(r"^\s*scratch_", code):
                problems.append(f"{name}: scratch access inside the level loop: {code.strip()}")
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", code)
            if m:
                n = int(m.group(1))
                if rep == 1:
                    stats["waits"].append(n)
                queue = queue[len(queue) - n:] if n < len(queue) else queue
                if n == 0:
                    queue = []
                continue
            pending = set().union(*[d for _, d in queue]) if queue else set()
            is_vmem = bool(VMEM.match(code))
            used = regs_of(code)
            dest = set()
            if is_vmem:
                if re.match(r"^\s*(global_|flat_)load", code):
                    problems.append(f"{name}: compiler-issued load inside the level loop: {code.strip()}")
                if re.match(r"^\s*buffer_load", code) and " lds" not in code:
                    first = code.split(",")[0]
                    dest = regs_of(first)
                    used = regs_of(",".join(code.split(",")[1:]))
            hit = used & pending
            if hit:
                problems.append(f"{name}: v{sorted(hit)} touched before its load was waited for: {code.strip()}")
            if is_vmem:
                queue.append((bool(dest), dest))
                if rep == 1:
                    stats["vmem"] += 1
    return problems, stats


def main():
    src = os.path.join(CSRC, "msda_forward_quad.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "q.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *HIPFLAGS, "-S", "--cuda-device-only", "-o", out, src], check=True,
                       stderr=subprocess.DEVNULL)
        asm = open(out).read()
    bad = 0
    for name, lines in kernels(asm):
        res = audit(name, lines)
        if isinstance(res, list):
            print("\n".join(res))
            bad += len(res)
            continue
        problems, stats = res
        tag = re.search(r"quadILi(\d+)ELi(\d+)ELi(\d+)E", name)
        print(f"msda_fwd_quad<D={tag.group(1)}, NG={tag.group(2)}, FUSED={tag.group(3)}>: {stats['instrs']} instructions, "
              f"{stats['vmem']} VMEM per level, waits {sorted(set(stats['waits']))}: {'OK' if not problems else 'FAIL'}")
        for p in problems[:10]:
            print("   ", p)
        bad += len(problems)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
