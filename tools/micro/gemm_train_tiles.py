"""Forward / input-gradient GEMM shapes of the training step (M rows) by tile variant of the hand-written kernels:
the 128-wide kernel at 128 x 128 / 128 x 64 / 64 x 64 (PAFC_GEMM_TILE) and the phase-pipelined 256-wide one at 256 / 192 / 128 / 64 rows.
  python tools/micro/gemm_train_tiles.py [M]            event-timed (includes the host side of a call)
  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/micro/gemm_train_tiles.py [M]
  python tools/micro/gemm_train_tiles.py --trace DIR [M]   kernel durations of that run, grouped in launch order"""
import csv, glob, os, sys
REPS, WARM = 20, 3
SHAPES = [(512, 512), (2048, 512), (1024, 512), (512, 1024), (512, 2048), (512, 128), (128, 512)]       # (K, N)
VARIANTS = ["auto", "128x128", "128x64", "64x64", "ph256", "ph192", "ph128", "ph64"]


def variants_for(K, N):
    return [v for v in VARIANTS if not (v.startswith("ph") and (N < 256 or K % 128))]


if len(sys.argv) > 1 and sys.argv[1] == "--trace":
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 15392
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "gemm_bf16_kernel" in r["Kernel_Name"] or "gemm_ph_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    n, i = REPS + WARM, 0
    print("rows %d; kernel us by variant" % M)
    for K, N in SHAPES:
        out = []
        for v in variants_for(K, N):
            out.append("%s %.1f" % (v, sum(d[i * n + WARM:(i + 1) * n]) / REPS / 1e3))
            i += 1
        print("K=%4d N=%4d: " % (K, N) + "  ".join(out))
    sys.exit(0)

import torch
from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16, gemm_bf16_ph
M = int(sys.argv[1]) if len(sys.argv) > 1 else 15392
for K, N in SHAPES:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    out = []
    for v in variants_for(K, N):
        os.environ.pop("PAFC_GEMM_TILE", None)
        if v == "auto":
            fn = lambda: gemm_bf16(x, w)
        elif v.startswith("ph"):
            tm = int(v[2:])
            fn = lambda: gemm_bf16_ph(x, w, tile_m=tm)
        else:
            os.environ["PAFC_GEMM_TILE"] = v
            fn = lambda: gemm_bf16(x, w)
        if v in ("128x128", "128x64", "64x64") and N >= 256 and K % 128 == 0 and M * N > 2 * 256 * 128 * 128:
            pass       # (auto may route such a shape to the 256-wide kernel; the forced tile only applies inside the 128-wide one)
        for _ in range(WARM):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REPS):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append("%s %.1f" % (v, a.elapsed_time(b) / REPS * 1e3))
    os.environ.pop("PAFC_GEMM_TILE", None)
    print("K=%4d N=%4d: " % (K, N) + "  ".join(out), flush=True)
