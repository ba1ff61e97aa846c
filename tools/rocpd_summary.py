#!/usr/bin/env python3
"""Text summary of rocprofv3 rocpd (.db) outputs: per-kernel durations (from --kernel-trace) and
per-kernel PMC counter averages (from --pmc passes).

    python tools/rocpd_summary.py out.db [more.db ...] [--filter mvdetr] [--per-dispatch]
"""
import argparse
import sqlite3
from collections import defaultdict


def short(name, n=110):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--filter", default="")
    ap.add_argument("--per-dispatch", action="store_true")
    a = ap.parse_args()
    for path in a.dbs:
        con = sqlite3.connect(path)
        cur = con.cursor()
        print(f"== {path}")
        rows = cur.execute("select name, duration, start, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count "
                           "from kernels order by start").fetchall()
        agg = defaultdict(list)
        meta = {}
        for name, dur, start, gx, wx, lds, vg, sg in rows:
            if a.filter in name:
                agg[name].append(dur / 1e3)
                meta[name] = (gx, wx, lds, vg, sg)
        if agg:
            print(f"{'kernel':112s} {'calls':>5s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}  grid/wg/lds/vgpr/sgpr")
            for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                print(f"{short(name):112s} {len(d):5d} {sum(d) / len(d):10.1f} {min(d):10.1f} {max(d):10.1f}  "
                      + "/".join(str(x) for x in meta[name]))
                if a.per_dispatch:
                    print("      per-dispatch us:", " ".join(f"{x:.0f}" for x in d))
        try:
            prow = cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection "
                               "order by dispatch_id").fetchall()
        except sqlite3.Error:
            prow = []
        pm = defaultdict(lambda: defaultdict(list))
        for name, cname, val, did in prow:
            if a.filter in name:
                pm[name][cname].append(val)
        for name, cs in pm.items():
            print(f"-- PMC {short(name)}")
            for cname, vals in sorted(cs.items()):
                line = f"   {cname:32s} n={len(vals):3d} avg={sum(vals) / len(vals):16.1f} min={min(vals):16.1f} max={max(vals):16.1f}"
                print(line)
                if a.per_dispatch:
                    print("      per-dispatch:", " ".join(f"{x:.0f}" for x in vals))


if __name__ == "__main__":
    main()
