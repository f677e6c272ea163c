"""Split-operand projections of the headline precision at few rows: the small tiles of csrc/gemm_bf16.hip (pafc_gemm_bf16_f32out)
against the 256-wide phase-pipelined kernel (pafc_gemm_ph_ex) and exact fp32 products (pafc_gemm_f32), microseconds per launch --
the numbers behind hip_ops.DISPATCH["split_small_max_rows"] / ["split_layers_min_rows"].  python tools/bench_split_small_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paper_accurate_fast_cheap_amd import hip_ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
FORMS = [("w_1 planes+SiLU", 512, 2048, "silu", "planes", False), ("w_2 +residual", 2048, 512, "none", "f32", True),
         ("pointwise_conv2 +residual", 512, 512, "none", "f32", True), ("pointwise_conv1 plain", 512, 1024, "none", "f32", False)]


def timed(f, n=40):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("us per launch: small tiles (gemm_bf16_f32out) | 256-wide tiles (gemm_ph_ex) | exact fp32 (gemm_f32); max |small - fp64|")
for name, K, N, act, kind, res in FORMS:
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g) * 0.1
    w3 = hip_ops.split_planes(w, triple=True)
    for M in (249, 499, 999, 1536, 1996, 2500, 3992, 7984):
        a = torch.randn(M, K, device=dev, generator=g)
        ap = hip_ops.split_planes(a)
        r = torch.randn(M, N, device=dev, generator=g) if res else None
        call = lambda: hip_ops.gemm_ph_ex(ap, w3, b, act, alpha=0.5 if res else 1.0, residual=r, a_split=True, out_kind=kind)
        hip_ops._SPLIT_SMALL_MAX_ROWS = 1 << 30
        got = call()
        t_small = timed(call)
        hip_ops._SPLIT_SMALL_MAX_ROWS = 0
        t_big = timed(call)
        t_f32 = timed(lambda: hip_ops.gemm_f32(a, w, b, act, alpha=0.5 if res else 1.0, residual=r))
        want = (0.5 if res else 1.0) * (a.double() @ w.double().t()) + b.double()
        if act == "silu":
            want = torch.nn.functional.silu(want)
        if res:
            want = want + r.double()
        val = (got[:, :N].float() + got[:, N:].float()) if kind == "planes" else got
        err = float((val.double() - want).abs().max())
        print(f"{name:28s} rows {M:5d}: {t_small:7.1f} | {t_big:7.1f} | {t_f32:7.1f}   err {err:.2e}", flush=True)
