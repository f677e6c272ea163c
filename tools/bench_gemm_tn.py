"""Weight-gradient GEMM dw = dy^T x at the c4 training shapes: hand-written split-K TN kernel vs the library (torch.mm).
  python tools/bench_gemm_tn.py [rows]"""
import json, sys, torch
from paper_accurate_fast_cheap_amd import _lib
from paper_accurate_fast_cheap_amd.hip_ops import gemm_tn

_lib.lib()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
dev = torch.device("cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for M, N in ((2048, 512), (512, 2048), (512, 512), (1024, 512), (512, 1024), (5000, 512), (512, 128), (128, 512), (512, 64), (64, 512)):
    dy = torch.randn(R, M, device=dev).to(torch.bfloat16)
    x = torch.randn(R, N, device=dev).to(torch.bfloat16)
    t_own = timed(lambda: gemm_tn(dy, x))
    t_lib = timed(lambda: (dy.t() @ x).float())
    fl = 2.0 * R * M * N
    print(json.dumps({"R": R, "M": M, "N": N, "own_us": round(t_own, 1), "own_tflops": round(fl / t_own / 1e6, 1),
                      "library_us": round(t_lib, 1), "library_tflops": round(fl / t_lib / 1e6, 1)}))
