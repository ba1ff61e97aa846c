#!/usr/bin/env python3
"""Phase stamps of EVERY workgroup of the fused camera-grouped forward (msda_fwd_group) -- where a launch's time goes:
ramp (kernel entry of the first to the last workgroup), window copies, tap phases, far taps + stores, tail.

Needs the stamp build of the library:  make -C mvdetr_amd/csrc libmvdetr_ops_trace.so
    MVDETR_OPS_LIB=$PWD/mvdetr_amd/csrc/libmvdetr_ops_trace.so python tools/experiments/group_trace.py [--noise 1.0]
Slots (msda_group_kernel.h): 0 entry, 1 shift known, 2+2l window l resident, 3+2l taps of level l done, 34 stored; 100 MHz.
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))
from fwd_variants import inputs, time_us  # noqa: E402
from mvdetr_amd import _lib  # noqa: E402
import MultiScaleDeformableAttention as MSDA  # noqa: E402

SLOTS, END = 40, 34


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="wildtrack")
    ap.add_argument("--noise", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--slice-outer", action="store_true")
    a = ap.parse_args()
    d = inputs(a.config, a.noise, a.batch)
    lo = not a.slice_outer
    rows = torch.tensor(MSDA.slice_major_rows(d["M"], d["L"], d["P"], d["D"], level_outer=lo), device="cuda")
    raw = d["plain"].index_select(-1, rows).contiguous()
    fn = lambda: MSDA.ms_deform_attn_forward_fused(d["value"], d["shapes"], d["lsi"], d["ref_lm"], None, None, raw=raw,  # noqa: E731
                                                   ref_level_major=True, raw_level_outer=lo)
    avg, med, mn = time_us(fn, 20)
    lib = _lib.lib()
    arm = lib.mvdetr_debug_group_trace_arm
    arm.argtypes, arm.restype = [ctypes.c_void_p], ctypes.c_int
    nwg = 1024
    table = torch.zeros(nwg * 4 * SLOTS, dtype=torch.int64, device="cuda")
    assert arm(table.data_ptr()) == 0
    fn()
    torch.cuda.synchronize()
    arm(None)
    t = table.cpu().view(nwg, 4, SLOTS)
    L = d["L"]
    live = [w for w in range(nwg) if t[w, 0, 0] != 0]
    ran = [w for w in live if t[w, 0, END] != 0]
    t0 = min(int(t[w, 0, 0]) for w in live)
    us = lambda x: (int(x) - t0) / 100.0  # noqa: E731
    ends = sorted(us(t[w, 0, END]) for w in ran)
    starts = sorted(us(t[w, 0, 0]) for w in live)
    print(f"# {a.config} B={a.batch} noise {a.noise}  events: avg {avg:.1f} med {med:.1f} min {mn:.1f} us   JOBMAP={os.environ.get('MVDETR_MSDA_JOBMAP', '-')}")
    print(f"workgroups launched {len(live)}, with a job {len(ran)};  kernel entry: first 0.00, median {starts[len(starts) // 2]:.2f}, last {starts[-1]:.2f} us")
    print(f"job end (first job of each workgroup): first {ends[0]:.2f}  p10 {ends[len(ends) // 10]:.2f}  median {ends[len(ends) // 2]:.2f}  p90 {ends[len(ends) * 9 // 10]:.2f}  last {ends[-1]:.2f} us")
    # per-phase means over the workgroups with a job (wave 0's stamps; the waves leave barriers together)
    def mean(xs):
        xs = list(xs)
        return sum(xs) / max(1, len(xs))
    print(f"entry -> shift known: {mean((int(t[w, 0, 1]) - int(t[w, 0, 0])) / 100 for w in ran):.2f} us")
    tot_copy = tot_taps = 0.0
    for l in range(L):
        prev = 1 if l == 0 else 3 + 2 * (l - 1)
        for wave in (0, 3):
            copy = mean((int(t[w, wave, 2 + 2 * l]) - int(t[w, wave, prev])) / 100 for w in ran)
            taps = mean((int(t[w, wave, 3 + 2 * l]) - int(t[w, wave, 2 + 2 * l])) / 100 for w in ran)
            if wave == 0:
                tot_copy += copy
                tot_taps += taps
            print(f"  level {l} wave {wave}: wait+copy {copy:6.2f}  taps {taps:6.2f} us")
    tail = mean((int(t[w, 0, END]) - int(t[w, 0, 3 + 2 * (L - 1)])) / 100 for w in ran)
    print(f"sum over levels (wave 0): copies {tot_copy:.2f}  taps {tot_taps:.2f};  far taps + stores {tail:.2f} us")
    # slowest / fastest jobs, and per XCD
    dur = {w: (int(t[w, 0, END]) - int(t[w, 0, 0])) / 100 for w in ran}
    order = sorted(ran, key=lambda w: dur[w])
    print("fastest jobs:", " ".join(f"wg{w}:{dur[w]:.1f}" for w in order[:6]), "  slowest:", " ".join(f"wg{w}:{dur[w]:.1f}" for w in order[-6:]))
    for k in range(8):
        ws = [w for w in ran if w % 8 == k]
        xcc = sorted(set(int(t[w, 0, SLOTS - 1]) >> 32 & 0xf for w in ws))
        print(f"  blockIdx%8={k}: XCC_ID {xcc}  jobs {len(ws)}  mean duration {mean(dur[w] for w in ws):.1f}  last end {max(us(t[w, 0, END]) for w in ws):.1f}")
    # who shares a CU?  HW_ID: wave [3:0], SIMD [5:4], pipe [7:6], CU [11:8], SH [12], SE [15:13]
    def cu_key(w):
        hw = int(t[w, 0, SLOTS - 1])
        return (hw >> 32 & 0xf, hw >> 13 & 7, hw >> 12 & 1, hw >> 8 & 0xf)
    by_cu = {}
    for w in ran:
        by_cu.setdefault(cu_key(w), []).append(w)
    alone = [ws[0] for ws in by_cu.values() if len(ws) == 1]
    paired = [w for ws in by_cu.values() if len(ws) == 2 for w in ws]
    more = [w for ws in by_cu.values() if len(ws) > 2 for w in ws]
    print(f"CUs with a job: {len(by_cu)}; jobs alone on their CU {len(alone)} (mean {mean(dur[w] for w in alone):.1f} us), "
          f"two per CU {len(paired)} (mean {mean(dur[w] for w in paired):.1f} us), more {len(more)}")
    tcols = (d["W"] + 15) // 16
    # (jobs in the last tile column are a quarter full at 180 columns)
    simd0 = {}
    for w in ran:
        s4 = tuple(int(t[w, v, SLOTS - 1]) >> 4 & 3 for v in range(4))
        simd0.setdefault(s4, []).append(dur[w])
    print("duration by the SIMDs of waves 0..3:", "  ".join(f"{k}:{len(v)}x{mean(v):.1f}" for k, v in sorted(simd0.items())))
    gaps = []
    for ws in by_cu.values():
        if len(ws) == 2:
            gaps.append(abs(int(t[ws[0], 0, 2 + 2 * 3]) - int(t[ws[1], 0, 2 + 2 * 3])) / 100)
    gaps.sort()
    if gaps:
        print(f"phase distance of a CU's two workgroups at level 3: median {gaps[len(gaps) // 2]:.2f} us, p90 {gaps[len(gaps) * 9 // 10]:.2f} us")
    # which of a CU's two workgroups is the slower one: the one in the higher wave slots (= dispatched second)?
    hi_slow = lo_slow = 0
    for ws in by_cu.values():
        if len(ws) == 2:
            a, b2 = ws
            sa, sb = int(t[a, 0, SLOTS - 1]) & 0xf, int(t[b2, 0, SLOTS - 1]) & 0xf
            ea, eb = int(t[a, 0, 0]), int(t[b2, 0, 0])
            if sa != sb:
                slower_is_hi = (dur[a] > dur[b2]) == (sa > sb)
                hi_slow += slower_is_hi
                lo_slow += not slower_is_hi
    print(f"pairs where the workgroup in the HIGHER wave slot is the slower one: {hi_slow}, the lower: {lo_slow}")
    later = sum(1 for ws in by_cu.values() if len(ws) == 2 and (dur[ws[0]] > dur[ws[1]]) == (int(t[ws[0], 0, 0]) > int(t[ws[1], 0, 0])))
    print(f"pairs where the workgroup that ENTERED later is the slower one: {later} of {sum(1 for ws in by_cu.values() if len(ws) == 2)}")
    slow = sorted(ran, key=lambda w: -dur[w])[:12]
    print("slowest jobs (wg: duration, CU key, partner's duration):")
    for w in slow:
        ws = by_cu[cu_key(w)]
        print(f"   wg{w}: {dur[w]:.1f} {cu_key(w)} partners {[round(dur[x], 1) for x in ws if x != w]}")
    # the two workgroups of a CU: how far apart are they in phase at level 3?
    cus = {}
    for w in ran:
        hw = int(t[w, 0, SLOTS - 1])
        cus.setdefault((hw >> 32 & 0xf, hw & 0xffffffff & ~0xf0), []).append(w)       # (XCC, HW_ID without the wave slot bits)
    print(f"distinct (XCC, HW_ID sans wave slot) keys: {len(cus)}")
    hist = {}
    for w in ran:
        key = round(us(t[w, 0, 2 + 2 * 3]) / 2) * 2
        hist[key] = hist.get(key, 0) + 1
    print("level-3 window resident at (us, 2-us bins):", " ".join(f"{k}:{v}" for k, v in sorted(hist.items())))


if __name__ == "__main__":
    main()
