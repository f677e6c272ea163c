"""The weight-gradient GEMM by split count S (PAFC_GEMM_TN_S) at the training step's shapes: is plan()'s cost model picking well?
Event-timed pairs of launches (main + reduce), us.  python tools/micro/gemm_tn_splits.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from paper_accurate_fast_cheap_amd import _lib
from paper_accurate_fast_cheap_amd.hip_ops import gemm_tn
_lib.lib()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 15392


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for Z, M, N, cands in ((1, 512, 512, (8, 12, 16, 20, 24, 32)), (3, 512, 512, (3, 4, 5, 6, 8, 10)), (1, 2048, 512, (2, 3, 4, 5, 6, 8)),
                       (1, 512, 2048, (2, 3, 4, 5, 6, 8)), (1, 1024, 512, (4, 6, 8, 10, 12, 16)), (1, 512, 128, (16, 24, 32, 48, 64)),
                       (1, 128, 2048, (8, 12, 16, 24, 32)), (1, 512, 64, (16, 32, 48, 64)), (1, 5000, 512, (1, 2, 3, 4))):
    dy = torch.randn((Z, R, M) if Z > 1 else (R, M), device="cuda").to(torch.bfloat16)
    x = torch.randn((Z, R, N) if Z > 1 else (R, N), device="cuda").to(torch.bfloat16)
    os.environ.pop("PAFC_GEMM_TN_S", None)
    out = ["plan %.1f" % timed(lambda: gemm_tn(dy, x))]
    for s_ in cands:
        os.environ["PAFC_GEMM_TN_S"] = str(s_)
        out.append("S=%d %.1f" % (s_, timed(lambda: gemm_tn(dy, x))))
    os.environ.pop("PAFC_GEMM_TN_S", None)
    print("Z=%d %4d x %4d: " % (Z, M, N) + " | ".join(out), flush=True)
