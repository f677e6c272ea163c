"""The bidirectional scan at c2's shapes (decode batch 64, T' = 25 ... 500): time per launch against the chunk length."""
import sys, time
import torch
from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward_bidir
from paper_accurate_fast_cheap_amd import _lib


def mk(B, T, C, H, dt):
    r, k, v = (torch.randn(B, T, C, device='cuda').mul_(0.5).to(dt) for _ in range(3))
    w = (torch.randn(B, T, C, device='cuda') - 3).to(dt)
    u = (torch.randn(H, C // H, device='cuda') * 0.3).to(dt)
    return r, k, v, w, u


dt = torch.bfloat16
for (B, T) in [(64, 64), (64, 128), (64, 262), (64, 400), (64, 499), (32, 262), (8, 500)]:
    f, b = mk(B, T, 512, 8, dt), mk(B, T, 512, 8, dt)
    auto = _lib.lib().pafc_wkv6_pick_chunk_len(B, T, 512, 8, 2)
    line = [f"B={B} T={T} auto L={auto}:"]
    for chunk in (0, 32, 64, 96, 128, 192, 256, 10 ** 6):
        for _ in range(3):
            wkv6_forward_bidir(f, b, chunk_len=chunk)
        torch.cuda.synchronize(); t0 = time.time(); n = 20
        for _ in range(n):
            wkv6_forward_bidir(f, b, chunk_len=chunk)
        torch.cuda.synchronize(); dtm = (time.time() - t0) / n
        line.append(f"{'auto' if chunk == 0 else ('serial' if chunk > 10**5 else chunk)} {dtm * 1e6:.0f}us")
    print("  ".join(line), flush=True)
