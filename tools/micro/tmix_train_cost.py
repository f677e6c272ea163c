"""How much of the c4 training step is the time-mix block's element-wise chain?  Times mix_project (token shift, LoRA,
lerps, r/k/v/w projections) forward + backward at the c4 shape, and the same with the projections alone."""
import torch
from paper_accurate_fast_cheap_amd import _lib
from paper_accurate_fast_cheap_amd.rwkv_v6.tmix import RWKV_Tmix_x060c

_lib.lib()
dev = torch.device("cuda")
torch.manual_seed(0)
blk = RWKV_Tmix_x060c(64, 12, 512, 512, 1).to(torch.bfloat16).to(dev)
x = torch.randn(32, 500, 512, device=dev, dtype=torch.bfloat16, requires_grad=True)


def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def full():
    r, k, v, w = blk.mix_project(x)
    (r.float().sum() + k.float().sum() + v.float().sum() + w.float().sum()).backward()


def proj_only():
    from paper_accurate_fast_cheap_amd.hip_ops import linear
    r = linear(x, blk.receptance.weight, None); k = linear(x, blk.key.weight, None); v = linear(x, blk.value.weight, None)
    (r.float().sum() + k.float().sum() + v.float().sum()).backward()


t_full, t_proj = timed(full), timed(proj_only)
print(f"mix_project fwd+bwd {t_full:.3f} ms; r,k,v projections alone {t_proj:.3f} ms; "
      f"element-wise + LoRA part {(t_full - t_proj):.3f} ms per direction and layer = {(t_full - t_proj) * 24:.1f} ms per c4 step")
