#!/usr/bin/env python3
"""GPU busy time vs span of the tail of a rocprofv3 kernel trace: how much of the wall time no kernel was running (host-bound
share).  usage: trace_busy.py <kernel_trace.csv> [fraction of the trace to analyse from its end, default 0.5]"""
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for _, e in rows) - rows[0][0]
print(f"{len(rows)} launches: span {span / 1e6:.1f} ms, some kernel running {busy / 1e6:.1f} ms ({100.0 * busy / span:.1f} %), "
      f"idle {(span - busy) / 1e6:.1f} ms; sum of kernel durations {sum(e - s for s, e in rows) / 1e6:.1f} ms")
