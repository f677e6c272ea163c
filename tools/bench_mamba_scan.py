"""The SSD scan alone at the 30-minute shape: forward / reverse direction, fp32 / bf16(+skip) output.
  python tools/bench_mamba_scan.py"""
import json, torch
from paper_accurate_fast_cheap_amd import _lib, hip_ops

_lib.lib()
dev = torch.device("cuda")
B, L, H = 1, 44998, 16
torch.manual_seed(0)
xbc = (0.5 * torch.randn(B, L, H * 64 + 256, device=dev)).to(torch.bfloat16)
dt = 0.02 + 0.1 * torch.rand(B, L, H, device=dev)
log_a = (-dt * (1 + 3 * torch.rand(H, device=dev))).contiguous()
D = torch.randn(H, device=dev)


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
    return round(a.elapsed_time(b) / reps * 1e3, 1)


out = {}
for rev in (False, True):
    out[f"fp32_y{'_reverse' if rev else ''}_us"] = timed(lambda: hip_ops.mamba2_scan(xbc, dt, log_a, H, rev))
    out[f"bf16_skip_y{'_reverse' if rev else ''}_us"] = timed(lambda: hip_ops.mamba2_scan(xbc, dt, log_a, H, rev, D=D))
print(json.dumps(out))
