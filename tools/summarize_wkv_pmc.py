"""Summarise the rocprofv3 passes of the bidirectional scan into the files bench.py and DESIGN.md cite.

  python tools/summarize_wkv_pmc.py gpurun_out/prof_wkv_r01f profiles/r01f_wkv6_bidir_T44998_bf16

Input directory: stats/ (--kernel-trace --stats), fetch/ (--pmc FETCH_SIZE), write/ (--pmc WRITE_SIZE), sq/ (SQ_*
counters), each from `python tools/bench_wkv6_one.py 1 44998 bf16`.  HBM bytes follow MI355X_MICROARCH.md: rocprofv3
reports KB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so reads are doubled; WRITE_SIZE is exact.
"""
import csv, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (scan_source_sha: the kernels the counters were taken on)
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
B, T, C, H, ndir, eb = 1, 44998, 512, 8, 2, 2
PASS = {"wkv6_mfma_kernel<unsigned short, false>": "pass_A_chunk_state", "wkv6_pass_a_cl_kernel": "pass_A_chunk_state",
        "wkv6_scan_kernel": "pass_B_state_scan",
        "wkv6_mfma_kernel<unsigned short, true>": "pass_C_output", "wkv6_pass_c_cl_kernel": "pass_C_output"}


def which(name):
    for k, v in PASS.items():
        if k in name:
            return v
    return None


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        p = which(r["Kernel_Name"])
        if p:
            acc[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {p: {c: sum(v) / len(v) for c, v in d.items()} for p, d in acc.items()}


fetch, write, sq = counters(f"{src}/fetch/run_counter_collection.csv"), counters(f"{src}/write/run_counter_collection.csv"), \
    counters(f"{src}/sq/run_counter_collection.csv")
raw = {p: {"FETCH_SIZE": round(fetch[p]["FETCH_SIZE"], 2), "WRITE_SIZE": round(write[p]["WRITE_SIZE"], 2)} for p in fetch}
hbm = sum(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024 for v in raw.values())
alg = B * T * C * 5 * eb * ndir
dur = {}
for r in csv.DictReader(open(f"{src}/stats/run_kernel_stats.csv")):
    p = which(r["Name"])
    if p:
        dur[p] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1)}
out = {
    "command": "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_* (separate passes) -- "
               "python tools/bench_wkv6_one.py 1 44998 bf16",
    "scan_source_sha": bench.scan_source_sha(),
    "shape": {"B": B, "T": T, "C": C, "H": H, "ndir": ndir, "elem_bytes": eb},
    "units": "rocprofv3 reports KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at "
             "64 B); WRITE_SIZE exact",
    "per_launch_KB_raw": raw,
    "hbm_bytes_per_launch_corrected": int(hbm),
    "algorithmic_bytes_per_launch": alg,
    "kernel_avg_duration": dur,
    "op_us_sum_of_kernels": round(sum(d["avg_us"] for d in dur.values()), 1),
    "sq_counters_per_launch": {p: {c: float(f"{v:.4e}") for c, v in d.items()} for p, d in sq.items()},
}
json.dump(out, open(dst + "_hbm_traffic.json", "w"), indent=1)
shutil.copy(f"{src}/stats/run_kernel_stats.csv", dst + "_kernel_stats.csv")
print(json.dumps(out, indent=1))
