"""Hand-written GEMM (pafc_gemm_bf16) vs the library paths on the encoder's shapes (30-minute sequence)."""
import os, sys, time
import torch
import torch.nn.functional as F
from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16, gemm_bf16_ph, glu_interleave, linear_bias_act
M = int(sys.argv[1]) if len(sys.argv) > 1 else 44998
dev, bf = "cuda", torch.bfloat16


def bench(fn, name, flops):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time(); n = 20
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print(f"  {name:34s} {dt*1e6:8.1f} us  {flops/dt/1e12:6.0f} TF/s", flush=True)


def check(got, want, what):
    err = (got.float() - want.float()).abs().max().item()
    ref = want.float().abs().max().item()
    print(f"  check {what}: max err {err:.3g} (max |ref| {ref:.3g})", flush=True)


for (K, N, act, res, name) in [(512, 2048, "silu", False, "ffn w_1 + SiLU"), (2048, 512, "none", True, "ffn w_2 + residual"),
                               (512, 1024, "none", False, "pointwise_conv1"), (512, 512, "none", True, "pointwise_conv2 + residual"),
                               (1024, 512, "none", True, "slot output + residual"), (512, 128, "tanh", False, "maa LoRA down + tanh")]:
    print(f"{name}: M={M} K={K} N={N}")
    x = torch.randn(M, K, device=dev, dtype=bf); w = torch.randn(N, K, device=dev, dtype=bf) * 0.05
    b = torch.randn(N, device=dev, dtype=bf); r = torch.randn(M, N, device=dev, dtype=bf) if res else None
    fl = 2 * M * K * N
    a = {"silu": F.silu, "tanh": torch.tanh, "none": lambda t: t}[act]
    want = a(F.linear(x.float(), w.float(), b.float())) * 1.0
    if res: want = r.float() + 0.5 * F.linear(x.float(), w.float()) + b.float()
    got = gemm_bf16(x, w, b, act, alpha=0.5 if res else 1.0, residual=r)
    check(got, want, "gemm_bf16")
    bench(lambda: gemm_bf16(x, w, b, act, alpha=0.5 if res else 1.0, residual=r), "pafc gemm_bf16 (dispatch)", fl)
    for tn in (256, 128):
        got = gemm_bf16_ph(x, w, b, act, alpha=0.5 if res else 1.0, residual=r, tile_n=tn)
        check(got, want, f"phase-pipelined 256x{tn}")
        bench(lambda: gemm_bf16_ph(x, w, b, act, alpha=0.5 if res else 1.0, residual=r, tile_n=tn), f"phase-pipelined 256x{tn}", fl)
    if act in ("silu", "none"):
        bench(lambda: linear_bias_act(x, w, b, act, alpha=0.5 if res else 1.0, residual=r), "hipBLASLt fused epilogue", fl)
    bench(lambda: a(F.linear(x, w, b)), "torch linear (+act)", fl)

print(f"pointwise_conv1 + GLU: M={M} K=512 N=1024 -> 512")
x = torch.randn(M, 512, device=dev, dtype=bf); w = torch.randn(1024, 512, device=dev, dtype=bf) * 0.05; b = torch.randn(1024, device=dev, dtype=bf)
want = F.glu(F.linear(x.float(), w.float(), b.float()), dim=-1)
for tn in (256, 128):
    wi, bi = glu_interleave(w, tn // 8), glu_interleave(b, tn // 8)
    check(gemm_bf16_ph(x, wi, bi, "glu", tile_n=tn), want, f"phase-pipelined 256x{tn} glu")
    bench(lambda: gemm_bf16_ph(x, wi, bi, "glu", tile_n=tn), f"phase-pipelined 256x{tn} glu", 2 * M * 512 * 1024)
wi, bi = glu_interleave(w), glu_interleave(b)
bench(lambda: gemm_bf16(x, wi, bi, "glu"), "pafc gemm_bf16 128x128 glu", 2 * M * 512 * 1024)

print("batched r,k,v of both directions: 6 x (M, 512) x (512, 512)")
z = torch.randn(6, M, 512, device=dev, dtype=bf); w6 = torch.randn(6, 512, 512, device=dev, dtype=bf) * 0.05
check(gemm_bf16(z, w6), torch.bmm(z.float(), w6.float().transpose(1, 2)), "batched")
bench(lambda: gemm_bf16(z, w6), "pafc gemm_bf16 batched", 6 * 2 * M * 512 * 512)
for tn in (256, 128):
    check(gemm_bf16_ph(z, w6, tile_n=tn), torch.bmm(z.float(), w6.float().transpose(1, 2)), f"batched phase-pipelined 256x{tn}")
    bench(lambda: gemm_bf16_ph(z, w6, tile_n=tn), f"phase-pipelined 256x{tn} batched", 6 * 2 * M * 512 * 512)
w6t = w6.transpose(1, 2).contiguous()
bench(lambda: torch.bmm(z, w6t), "torch bmm", 6 * 2 * M * 512 * 512)
