"""The mid-size regime (1k - 8k rows: 16-64 streams of a chunk step, the paper's 2 000-frame windows, short c2 batches):
the 128 x 128 hand-written kernel, the phase-pipelined one with short tiles, the few-rows kernel and the library."""
import sys, time
import torch
from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16, gemm_bf16_ph, gemm_skinny, linear_bias_act
dev, bf = "cuda", torch.bfloat16


def bench(fn):
    try:
        for _ in range(3): fn()
    except Exception as e:
        return float("nan")
    torch.cuda.synchronize(); t0 = time.time(); n = 50
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e6


for M in (1024, 2048, 4096, 6144):
    for (K, N, act, res) in [(512, 2048, "silu", False), (2048, 512, "none", True), (512, 512, "none", True), (1024, 512, "none", True)]:
        x = torch.randn(M, K, device=dev, dtype=bf); w = torch.randn(N, K, device=dev, dtype=bf) * 0.05
        b = torch.randn(N, device=dev, dtype=bf); r = torch.randn(M, N, device=dev, dtype=bf) if res else None
        t128 = bench(lambda: gemm_bf16(x, w, b, act, residual=r))
        tph = {tm: bench(lambda: gemm_bf16_ph(x, w, b, act, residual=r, tile_m=tm)) for tm in (128, 256)}
        tsk = bench(lambda: gemm_skinny(x, w, b, act, residual=r))
        tlib = bench(lambda: linear_bias_act(x, w, b, act, residual=r))
        print(f"M={M:5d} K={K:4d} N={N:4d} {act:5s} res={int(res)}: 128x128 {t128:6.1f}  ph tm128 {tph[128]:6.1f}  ph tm256 {tph[256]:6.1f}  "
              f"few-rows {tsk:6.1f}  library {tlib:6.1f} us", flush=True)
