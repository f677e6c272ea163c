"""Interleaved race at the 30-minute shape: token shift + lerp pass followed by the tanh-epilogue GEMM against the one-pass
kernel (W1 resident in LDS).  python tools/micro/bench_lora_down.py"""
import sys
import torch
sys.path.insert(0, '.')
from paper_accurate_fast_cheap_amd import hip_ops
M, C, nd = 44998, 512, 2
bf = torch.bfloat16
x = torch.randn(1, M, C, device='cuda', dtype=bf)
maa = torch.rand(nd, C, device='cuda', dtype=bf)
w1n = torch.randn(nd, 128, C, device='cuda', dtype=bf) * 0.05
separate = lambda: hip_ops.gemm_bf16(hip_ops.tmix_shift_mix(x, maa[0], maa[1]).view(nd, M, C), w1n, act="tanh")
fused = lambda: hip_ops.tmix_lora_down(x, maa, w1n)
ts_, tf = separate(), fused()
print("t max diff", (tf.float() - ts_.float()).abs().max().item(), "mean", (tf.float() - ts_.float()).abs().mean().item())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
import os
def fw(n):
    def f():
        os.environ["PAFC_LORA_DOWN_WAVES"] = n
        return fused()
    return f
runs = {"two kernels": separate, "one pass 8w": fw("8"), "one pass 16w": fw("16")}
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
