"""Mid-to-large row counts (window batches, c2 decode batches): the 256-wide phase-pipelined kernel at tile heights 256 / 192 /
128 / 64 against the 128-wide kernel's pick, 50 launches each, under rocprofv3 --kernel-trace (tools/prof_gemm_ph_rows.sh)."""
import os, sys
import torch
from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16, gemm_bf16_ph
bf = torch.bfloat16
rows = [int(v) for v in sys.argv[1].split(",")]
variant = sys.argv[2]       # "auto" | "small" (run with PAFC_PH_MIN_FILL=100000) | "256" | "192" | "128" | "64"
SHAPES = [(512, 2048, 1, "silu", False), (2048, 512, 1, "none", True), (512, 512, 1, "none", True), (1024, 512, 1, "none", True),
          (512, 512, 6, "none", False)]
for M in rows:
    for K, N, Z, act, res in SHAPES:
        shp = (lambda *s: (Z,) + s) if Z > 1 else (lambda *s: s)
        x = torch.randn(shp(M, K), device="cuda", dtype=bf); w = torch.randn(shp(N, K), device="cuda", dtype=bf) * 0.05
        b = None if Z > 1 else torch.randn(N, device="cuda", dtype=bf)
        r = torch.randn(shp(M, N), device="cuda", dtype=bf) if res else None
        if variant in ("auto", "small"):
            f = lambda: gemm_bf16(x, w, b, act, residual=r)
        else:
            f = lambda: gemm_bf16_ph(x, w, b, act, residual=r, tile_m=int(variant))
        torch.cuda.synchronize()
        torch.empty(1000 + len(SHAPES), device="cuda").fill_(1.0)
        for _ in range(50):
            f()
torch.cuda.synchronize()
