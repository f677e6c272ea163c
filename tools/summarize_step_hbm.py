#!/usr/bin/env python3
"""HBM bytes per kernel over the LAST bench step, from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, as
MI355X_MICROARCH.md prescribes) and a kernel trace of the same command.

  python tools/summarize_step_hbm.py <dir with fetch/ write/ trace/> <out.txt>

Units and corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports KB; on gfx950 FETCH_SIZE tallies the 128-byte read
requests of wide coalesced loads (16 B per lane, global_load and buffer_load ... lds alike -- every kernel of this step reads
that way) at 64 bytes, so reads are DOUBLED; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Infinity-Cache hits
are counted, not excluded: a tensor written by one kernel and read by the next (<= 256 MiB between the two uses) shows up in
both although it may never have reached the DRAM."""
import collections
import csv
import glob
import sys


def last_step(rows, key_name, key_order):
    rows = sorted(rows, key=lambda r: int(r[key_order]))
    ends = [i for i, r in enumerate(rows) if "log_softmax_kernel" in r[key_name] or "SoftMaxForward" in r[key_name]]
    lo = ends[-2] + 1 if len(ends) >= 2 else 0
    hi = ends[-1] + 1 if ends else len(rows)
    return rows[lo:hi]


def short(name):
    n = name.replace("pafc::(anonymous namespace)::", "").replace("void ", "")
    return n[:100]


def main():
    src, out = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: dict(calls=0, rd=0.0, wr=0.0, ns=0))
    for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        path = glob.glob(f"{src}/{which}/**/*counter_collection.csv", recursive=True)[0]
        rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
        for r in last_step(rows, "Kernel_Name", "Dispatch_Id"):
            a = agg[short(r["Kernel_Name"])]
            if which == "fetch":
                a["calls"] += 1
                a["rd"] += 2.0 * float(r["Counter_Value"]) * 1024      # doubled: see the header
            else:
                a["wr"] += float(r["Counter_Value"]) * 1024
    path = glob.glob(f"{src}/trace/**/*kernel_trace.csv", recursive=True)[0]
    for r in last_step(list(csv.DictReader(open(path))), "Kernel_Name", "Start_Timestamp"):
        agg[short(r["Kernel_Name"])]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot_rd = sum(a["rd"] for a in agg.values())
    tot_wr = sum(a["wr"] for a in agg.values())
    tot_ns = sum(a["ns"] for a in agg.values())
    with open(out, "w") as f:
        f.write(__doc__.split("\n\n")[0] + "\n\n")
        f.write(f"last step: {sum(a['calls'] for a in agg.values())} launches, {tot_ns / 1e6:.3f} ms of kernels, "
                f"{tot_rd / 1e9:.2f} GB read + {tot_wr / 1e9:.2f} GB written = {(tot_rd + tot_wr) / 1e9:.2f} GB "
                f"({(tot_rd + tot_wr) / max(tot_ns, 1):.2f} GB/s x 1e0 = {(tot_rd + tot_wr) / max(tot_ns, 1) * 1e0:.0f} B/ns average)\n")
        f.write(f"{'MB read':>10} {'MB written':>11} {'calls':>6} {'ms':>8} {'GB/s':>8}  kernel\n")
        for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["rd"] + kv[1]["wr"])):
            gbs = (a["rd"] + a["wr"]) / a["ns"] if a["ns"] else 0.0
            f.write(f"{a['rd'] / 1e6:10.1f} {a['wr'] / 1e6:11.1f} {a['calls']:6d} {a['ns'] / 1e6:8.3f} {gbs:8.0f}  {k}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
