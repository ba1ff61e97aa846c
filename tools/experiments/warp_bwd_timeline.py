#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace db of tools/experiments/warp_bwd_sweep.py: span from the start of warp_bwd_scans to the end of
the following warp_bwd_gather, and the gap between the two kernels."""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
spans, gaps, between = [], [], []
prev_end = None
for i, (name, st, en) in enumerate(rows[:-1]):
    if "warp_bwd_scans" in name and "warp_bwd_gather" in rows[i + 1][0]:
        g = rows[i + 1]
        spans.append((g[2] - st) / 1e3)
        gaps.append((g[1] - en) / 1e3)
        if prev_end is not None:
            between.append((st - prev_end) / 1e3)
        prev_end = g[2]
n = len(spans)
print(f"{n} calls: scans.start -> gather.end avg {sum(spans) / n:.1f} us; gap scans.end -> gather.start avg {sum(gaps) / n:.1f} us; "
      f"gather.end -> next scans.start avg {sum(between) / max(1, len(between)):.1f} us")
