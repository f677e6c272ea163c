"""Per-launch averages of the counters tools/prof_gemm_pmc.sh collected, one table per shape, with the derived figures the
DESIGN cites: MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * CUs * 4 SIMDs) as rocprofiler-sdk defines it
for gfx950; L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS); bytes = requests x 64 B (TCC_EA0_RDREQ: 128-B requests tallied at
64 B on gfx950 for wide reads -- MI355X_MICROARCH.md -- so HBM read bytes = 2 x 64 x RDREQ).
   python tools/summarize_gemm_pmc.py gpurun_out/gemm_pmc profiles/r03d_gemm_ph_pmc.txt"""
import collections, csv, glob, os, sys

src, dst = sys.argv[1], sys.argv[2]
names = {0: "ffn w_1 + SiLU 512->2048", 1: "ffn w_2 + residual 2048->512", 2: "pointwise_conv1 + GLU", 3: "pointwise_conv2 + residual",
         5: "r,k,v stack 6 x 512->512", 6: "CTC head 512->5000"}
out = []
for c in sorted({int(os.path.basename(f).split("_")[0][1:]) for f in glob.glob(os.path.join(src, "c*_*.csv"))}):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, f"c{c}_*.csv")):
        for r in csv.DictReader(open(f)):
            if "gemm_ph_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # rocprofv3 emits one row per (dispatch, counter[, dimension]); sum the dimension rows of a dispatch, then average
    avg = {k: sum(v) / 8.0 for k, v in vals.items()}
    out.append(f"== {names.get(c, c)} (M = 44 998, bf16, 8 launches, per-launch averages)")
    for k in sorted(avg):
        out.append(f"  {k:38s} {avg[k]:16.4e}")
    g = avg.get("GRBM_GUI_ACTIVE")
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        out.append(f"  -> MfmaUtil {100 * avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (g / 8 * 256 * 4):.1f} %   (kernel ~{g / 8 / 2.1e3:.1f} us at 2.1 GHz)")
    if "TCC_HIT_sum" in avg:
        out.append(f"  -> L2 hit rate {100 * avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']):.1f} %;  "
                   f"L2 requests {avg['TCC_REQ_sum']:.3e};  fabric reads {avg['TCC_EA0_RDREQ_sum']:.3e} requests")
    if "SQ_WAVE_CYCLES" in avg:
        w = avg["SQ_WAVE_CYCLES"]
        out.append(f"  -> of the wave cycles: waiting (s_waitcnt / barrier) {100 * avg['SQ_WAIT_ANY'] / w:.1f} %, issue stalls "
                   f"{100 * avg['SQ_WAIT_INST_ANY'] / w:.1f} % (LDS issue {100 * avg['SQ_WAIT_INST_LDS'] / w:.1f} %), issuing "
                   f"{100 * avg['SQ_ACTIVE_INST_ANY'] / w:.1f} %")
    out.append("")
open(dst, "w").write("\n".join(out))
print("\n".join(out))
