#!/usr/bin/env python3
"""Summarise the LAST bench step of a `rocprofv3 --kernel-trace --output-format csv` run.

Whole-run --stats are dominated by MIOpen's one-off solver search during warm-up (naive_conv_* kernels); the
steady state is the last step, delimited by the CTC log-softmax kernel that ends each step.
usage: prof_last_step.py <kernel_trace.csv> [n_top] [batches_per_step]   (windowed / c2 runs: a step is that many batches,
each closed by its own log-softmax)"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [r for r in rows if "SoftMaxForward" in r["Kernel_Name"] or "log_softmax_kernel" in r["Kernel_Name"]]
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    t0 = int(ends[-1 - nb]["End_Timestamp"]) if len(ends) > nb else 0
    t1 = int(ends[-1]["End_Timestamp"]) if ends else int(rows[-1]["End_Timestamp"])
    last = [r for r in rows if t0 < int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= t1]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in last:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg[r["Kernel_Name"][:120]]
        a[0] += d
        a[1] += 1
    tot = sum(v[0] for v in agg.values())
    span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
    print(f"last step: {len(last)} kernel launches, busy {tot / 1e6:.3f} ms, span {span / 1e6:.3f} ms")
    print(f"{'total ms':>10} {'%':>6} {'calls':>6} {'avg us':>10}  kernel")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{v[0] / 1e6:10.3f} {100 * v[0] / tot:6.1f} {v[1]:6d} {v[0] / v[1] / 1e3:10.1f}  {k}")


if __name__ == "__main__":
    main()
