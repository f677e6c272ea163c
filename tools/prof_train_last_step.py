#!/usr/bin/env python3
"""Summarise ONE steady-state training step of a `rocprofv3 --kernel-trace --output-format csv` run of
tools/bench_train_step.py.  Whole-run --stats are dominated by MIOpen's one-off solver search in the first step; a
step-long window is delimited by the last WKV backward kernel of two consecutive steps (24 per step: 12 layers x 2
directions).  usage: prof_train_last_step.py <kernel_trace.csv> [n_top]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [r for r in rows if "wkv6_bwd_gw_kernel" in r["Kernel_Name"]]
    per_step = 24
    t1 = int(marks[-1]["End_Timestamp"])
    t0 = int(marks[-1 - per_step]["End_Timestamp"])
    last = [r for r in rows if t0 < int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= t1]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in last:
        a = agg[r["Kernel_Name"][:120]]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"one step: {len(last)} kernel launches, busy {tot / 1e6:.3f} ms, span {(t1 - t0) / 1e6:.3f} ms")
    print(f"{'total ms':>10} {'%':>6} {'calls':>6} {'avg us':>10}  kernel")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{v[0] / 1e6:10.3f} {100 * v[0] / tot:6.1f} {v[1]:6d} {v[0] / v[1] / 1e3:10.1f}  {k}")


if __name__ == "__main__":
    main()
