"""Tile-shape sweep of the phase-pipelined GEMM on the encoder's long-form shapes against the 128 x 128 kernel and the
library's measured pick: variants interleaved round by round in ONE process, medians reported (device-to-device and
run-to-run spread is larger than the differences of interest).   python tools/bench_gemm_tiles.py [M]"""
import statistics, sys
import torch
from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16, gemm_bf16_ph, linear_bias_act
M = int(sys.argv[1]) if len(sys.argv) > 1 else 44998
dev, bf = "cuda", torch.bfloat16


def race(variants, rounds=9, reps=6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    times = {k: [] for k in variants}
    for k, fn in variants.items():
        fn(); fn()
    for _ in range(rounds):
        for k, fn in variants.items():
            ev[0].record()
            for _ in range(reps): fn()
            ev[1].record(); torch.cuda.synchronize()
            times[k].append(ev[0].elapsed_time(ev[1]) / reps * 1e3)
    return {k: (statistics.median(v), min(v)) for k, v in times.items()}


for (K, N, Z, act, res, name) in [(512, 2048, 1, "silu", False, "ffn w_1 + SiLU"), (2048, 512, 1, "none", True, "ffn w_2 + residual"),
                                  (512, 1024, 1, "none", False, "pointwise_conv1"),
                                  (512, 512, 1, "none", True, "pointwise_conv2 + residual"), (1024, 512, 1, "none", True, "slot output + residual"),
                                  (512, 512, 6, "none", False, "r,k,v stack (6 problems)")]:
    shp = (lambda *s: (Z, *s)) if Z > 1 else (lambda *s: s)
    x = torch.randn(shp(M, K), device=dev, dtype=bf); w = torch.randn(shp(N, K), device=dev, dtype=bf) * 0.05
    b = torch.randn(N, device=dev, dtype=bf) if Z == 1 else None
    r = torch.randn(shp(M, N), device=dev, dtype=bf) if res else None
    fl = 2.0 * M * K * N * Z
    al = 0.5 if res else 1.0
    v = {f"ph {tm}x{tn}": (lambda tn=tn, tm=tm: gemm_bf16_ph(x, w, b, act, alpha=al, residual=r, tile_n=tn, tile_m=tm))
         for tn in (256,) for tm in (256, 192)}
    v["dispatch"] = lambda: gemm_bf16(x, w, b, act, alpha=al, residual=r)
    if Z == 1:
        v["library"] = lambda: linear_bias_act(x, w, b, act, alpha=al, residual=r)
    else:
        wt = w.transpose(1, 2).contiguous()
        v["library"] = lambda: torch.bmm(x, wt)
    out = race(v)
    best = min((t[0], k) for k, t in out.items() if k.startswith("p"))
    print(f"{name}: M={M} K={K} N={N} Z={Z}: best tile {best[1]} {best[0]:.1f} us = {fl / best[0] / 1e6:.0f} TF/s; "
          f"library {out['library'][0]:.1f} us = {fl / out['library'][0] / 1e6:.0f} TF/s", flush=True)
    print("   " + "  ".join(f"{k}: {t[0]:.1f} (min {t[1]:.1f})" for k, t in out.items()), flush=True)
