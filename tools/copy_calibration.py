#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this box for the access widths of the backward kernels.

MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports exactly half the bytes of a wide (16 bytes per lane) streaming read; other
widths and WRITE_SIZE are uncalibrated.  This tool measures them: tools/experiments/copy_widths.hip copies 1 GiB with 4, 8 and
16 bytes per lane and reads / writes sparse 8- / 16-byte pieces, under separate --pmc passes (FETCH_SIZE, WRITE_SIZE and the
request counters they derive from), and prints counter value / known bytes per kernel.  Run on the GPU box:
    bash tools/gpu.sh copycal > gpurun_out/copycal.log
"""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "_copycal")
GIB = float(1 << 30)
# known traffic per launch: (bytes read from memory, bytes written to memory); sparse reads fetch whole 128-byte lines at best
KNOWN = {
    "copy_w<float>": (GIB, GIB), "copy_w<float2>": (GIB, GIB), "copy_w<float4>": (GIB, GIB),
}


def run_pass(exe, counters, tag):
    d = os.path.join(OUT, tag)
    subprocess.run(["rocprofv3", "--pmc", *counters, "-d", d, "-o", "p", "--", exe], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    return os.path.join(d, "p_results.db")


def counter_sums(db):
    """[(dispatch id, kernel name, {counter: value summed over the dispatch's rows})] in launch order"""
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection "
                       "group by dispatch_id, counter_name order by dispatch_id").fetchall()
    out = {}
    for did, name, cname, val in rows:
        out.setdefault(did, (name, {}))[1][cname] = val
    return [(did, n, v) for did, (n, v) in sorted(out.items())]


# the launches of one repetition, in order (tools/experiments/copy_widths.hip)
LAUNCHES = ["copy 4 B/lane (1 GiB read, 1 GiB written)", "copy 8 B/lane (1 GiB, 1 GiB)", "copy 16 B/lane (1 GiB, 1 GiB)",
            "read 8 B of every 256 B (4 Mi lines of 128 B = 512 MiB; 32 MiB useful)", "read 8 B of every 64 B (1 GiB of lines; 128 MiB useful)",
            "write 16 B of every 128 B (8 Mi pieces = 128 MiB useful)", "write 16 B of every 48 B (21.3 Mi pieces = 341 MiB useful)"]


def main():
    exe = os.path.join(OUT, "copy_widths")
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", os.path.join(ROOT, "tools/experiments/copy_widths.hip"), "-o", exe], check=True)
    print(subprocess.run([exe], capture_output=True, text=True).stdout.strip())
    passes = [("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
              ("rdreq", ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"]), ("wrreq", ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]),
              ("hit", ["TCC_HIT_sum", "TCC_MISS_sum"])]
    merged = {}
    for tag, ctrs in passes:
        try:
            ks = [(n, v) for _, n, v in counter_sums(run_pass(exe, ctrs, tag)) if "copy_w" in n or "sparse" in n]
            for i, (n, v) in enumerate(ks[-len(LAUNCHES):]):       # the third repetition
                merged.setdefault(i, {}).update(v)
        except Exception as e:  # a counter this rocprofv3 does not know: the other passes still count
            print(f"# pass {tag} ({' '.join(ctrs)}): {type(e).__name__}: {e}")
    print(f"{'launch (third repetition)':72s} {'FETCH_SIZE KB':>14s} {'WRITE_SIZE KB':>14s} {'RDREQ':>10s} {'RDREQ_32B':>10s} {'WRREQ':>10s} {'WRREQ_64B':>10s} {'TCC_HIT':>10s} {'TCC_MISS':>10s}")
    for i, what in enumerate(LAUNCHES):
        v = merged.get(i, {})
        g = lambda k: v.get(k, float("nan"))  # noqa: E731
        print(f"{what:72s} {g('FETCH_SIZE'):14.0f} {g('WRITE_SIZE'):14.0f} {g('TCC_EA0_RDREQ_sum'):10.0f} {g('TCC_EA0_RDREQ_32B_sum'):10.0f} "
              f"{g('TCC_EA0_WRREQ_sum'):10.0f} {g('TCC_EA0_WRREQ_64B_sum'):10.0f} {g('TCC_HIT_sum'):10.0f} {g('TCC_MISS_sum'):10.0f}")
    print("# 1 GiB = 1048576 KB.  FETCH_SIZE / 1048576 for copy_w<float4> = the guide's 0.5; the same ratio for the narrower copies and the "
          "sparse readers is what profiles/README.md records.")


if __name__ == "__main__":
    sys.exit(main())
