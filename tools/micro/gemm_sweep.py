"""Every layer GEMM shape at a few row counts through the hand-written dispatcher and the library, 100 launches each, under
rocprofv3 --kernel-trace: tools/prof_gemm_mid2.sh turns the trace into kernel durations per (rows, shape, path)."""
import sys
import torch
from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16, linear_bias_act
bf = torch.bfloat16
rows = [int(v) for v in sys.argv[1].split(",")]
path = sys.argv[2]
SHAPES = [(512, 2048, 1, "silu", False), (2048, 512, 1, "none", True), (512, 512, 1, "none", True), (1024, 512, 1, "none", True),
          (512, 512, 6, "none", False), (512, 128, 2, "tanh", False)]
for M in rows:
    for K, N, Z, act, res in SHAPES:
        shp = (lambda *s: (Z,) + s) if Z > 1 else (lambda *s: s)
        x = torch.randn(shp(M, K), device="cuda", dtype=bf); w = torch.randn(shp(N, K), device="cuda", dtype=bf) * 0.05
        b = None if Z > 1 else torch.randn(N, device="cuda", dtype=bf)
        r = torch.randn(shp(M, N), device="cuda", dtype=bf) if res else None
        wt = w.transpose(-1, -2).contiguous() if Z > 1 else None
        if path == "own":
            f = lambda: gemm_bf16(x, w, b, act, residual=r)
        elif Z > 1:
            f = lambda: torch.bmm(x, wt)
        else:
            f = lambda: linear_bias_act(x, w, b, act, residual=r)
        torch.cuda.synchronize()
        # a marker kernel between groups: a fill of a distinctive size
        torch.empty(1000 + len(SHAPES), device="cuda").fill_(1.0)
        for _ in range(100):
            f()
torch.cuda.synchronize()
