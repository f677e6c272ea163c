#!/usr/bin/env python3
"""Matrix-core utilisation per kernel from one rocprofv3 counter pass over bench.py:

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python bench.py --steps 2 \\
      --warmup 1 --no-cpu-baseline
  python tools/summarize_mfma_pmc.py <dir>/.../*_counter_collection.csv [n_top]

MfmaUtil as rocprofiler-sdk defines it for gfx950 (counter_defs.yaml): sum over SIMDs of the cycles the MFMA pipe is
busy / (GPU-active cycles x number of SIMDs).  The CSV carries each counter summed over its instances:
SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md: divide by 8), so
    util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024).
Counter passes serialise kernels and run at a lower clock: read the ratio, not the durations.  Kernels of the last bench
step only (delimited by the CTC log-softmax kernel, as in prof_last_step.py)."""
import collections
import csv
import sys

SIMDS, XCDS = 1024, 8


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    disp = collections.OrderedDict()
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]),
                                                    "t1": int(r["End_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ds = sorted(disp.values(), key=lambda d: d["t0"])
    ends = [i for i, d in enumerate(ds) if "log_softmax_kernel" in d["name"]]
    last = ds[ends[-2] + 1:ends[-1] + 1] if len(ends) >= 2 else ds
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for d in last:
        a = agg[d["name"][:110]]
        a[0] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[1] += d.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        a[2] += 1
    tot_busy = sum(a[0] for a in agg.values())
    tot_cyc = sum(a[1] for a in agg.values())
    print(f"last step: {len(last)} dispatches, GPU-active cycles {tot_cyc:.3e}, MfmaUtil over the step "
          f"{100 * tot_busy / (tot_cyc * SIMDS):.1f} %")
    print(f"{'MfmaUtil %':>10} {'share of cycles %':>18} {'calls':>6}  kernel")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        util = 100 * a[0] / (a[1] * SIMDS) if a[1] else 0.0
        print(f"{util:10.1f} {100 * a[1] / tot_cyc:18.1f} {a[2]:6d}  {k}")


if __name__ == "__main__":
    main()
