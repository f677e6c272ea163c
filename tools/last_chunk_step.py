"""Kernel table of ONE streamed chunk step from a rocprofv3 --kernel-trace CSV of tools/bench_streaming.py (the last full
step before the end: from one first-subsampling-convolution launch to the next).  Usage: last_chunk_step.py trace.csv"""
import csv
import statistics
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = [i for i, r in enumerate(rows) if "conv3x3s2_c1" in r["Kernel_Name"]]
a, b = key[-3], key[-2]
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print(f"one chunk step: {len(step)} launches, span {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
agg = defaultdict(lambda: [0, 0])
for r in step:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"]][0] += d
    agg[r["Kernel_Name"]][1] += 1
print("  total us  calls  avg us  kernel")
for n, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
    print(f"{d / 1e3:10.1f} {c:6d} {d / c / 1e3:7.1f}  {n[:130]}")
gaps = [int(step[i + 1]["Start_Timestamp"]) - int(step[i]["End_Timestamp"]) for i in range(len(step) - 1)]
print(f"gaps between launches: mean {statistics.mean(gaps) / 1e3:.2f} us, sum {sum(gaps) / 1e3:.1f} us")
