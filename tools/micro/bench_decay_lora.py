"""Interleaved race at the 30-minute shape: the decay LoRA as two GEMMs against the one-pass kernel (weights resident in LDS).
python tools/micro/bench_tmix_front.py"""
import sys
import torch
sys.path.insert(0, '.')
from paper_accurate_fast_cheap_amd import hip_ops
M, C, nd = 44998, 512, 2
bf = torch.bfloat16
zw = torch.randn(nd, M, C, device='cuda', dtype=bf)
d1n = torch.randn(nd, 64, C, device='cuda', dtype=bf) * 0.05
d2n = torch.randn(nd, C, 64, device='cuda', dtype=bf) * 0.3
separate = lambda: hip_ops.gemm_bf16(hip_ops.gemm_bf16(zw, d1n, act="tanh"), d2n)
fused = lambda: hip_ops.decay_lora(zw, d1n, d2n)
ws, wf = separate(), fused()
print("w max diff", (wf.float() - ws.float()).abs().max().item(), "mean", (wf.float() - ws.float()).abs().mean().item())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
bias = (torch.randn(nd, C, device='cuda') * 0.5).to(bf)
runs = {"two GEMMs": separate, "one pass": fused, "one pass + bias": lambda: hip_ops.decay_lora(zw, d1n, d2n, bias)}
res = {k: [] for k in runs}
for _ in range(7):
    for k, f in runs.items():
        f()
        ev[0].record()
        for _ in range(5):
            f()
        ev[1].record(); torch.cuda.synchronize(); res[k].append(ev[0].elapsed_time(ev[1]) / 5 * 1e3)
for k, ts in res.items():
    ts.sort(); print(f"{k:12s} median us {ts[3]:7.1f}  min {ts[0]:7.1f}")
