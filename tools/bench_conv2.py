"""conv2 of the subsampling (pafc_conv3x3s2_nhwc_bf16): tile-shape variants (PAFC_CONV_TILE) on the 30-minute shape."""
import os, sys, time
import torch
from paper_accurate_fast_cheap_amd.hip_ops import conv3x3s2_nhwc
T1 = int(sys.argv[1]) if len(sys.argv) > 1 else 89998
B, F1, C = 1, 39, 512
x = torch.randn(B, T1, F1, C, device="cuda", dtype=torch.bfloat16)
w = (torch.randn(9, C, C, device="cuda", dtype=torch.bfloat16) * 0.02)
b = torch.randn(C, device="cuda", dtype=torch.bfloat16)
T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
fl = 2 * B * T2 * F2 * C * C * 9
ref = None
for v in sys.argv[2:] or ["1", "2", "3", "4"]:
    os.environ["PAFC_CONV_TILE"] = v
    y = conv3x3s2_nhwc(x, w, b, relu=True)
    torch.cuda.synchronize()
    if ref is None:
        ref = y
    err = (y.float() - ref.float()).abs().max().item()
    for _ in range(2): conv3x3s2_nhwc(x, w, b, relu=True)
    torch.cuda.synchronize(); t0 = time.time(); n = 5
    for _ in range(n): conv3x3s2_nhwc(x, w, b, relu=True)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print(f"tile variant {v}: {dt*1e3:.3f} ms  {fl/dt/1e12:.0f} TF/s   max diff vs variant {sys.argv[2] if len(sys.argv)>2 else 1}: {err:.3g}", flush=True)

# the phase-pipelined implicit GEMM (csrc/gemm_ph.hip), interleaved with the 128 x 128 kernel of this file's variant 1
from paper_accurate_fast_cheap_amd.hip_ops import conv3x3s2_nhwc_ph
os.environ["PAFC_CONV_TILE"] = "1"
variants = {"128x128 (conv_sub.hip)": lambda: conv3x3s2_nhwc(x, w, b, relu=True)}
for tm in (256, 192):
    variants[f"phase-pipelined {tm}x256"] = (lambda tm=tm: conv3x3s2_nhwc_ph(x, w, b, relu=True, tile_m=tm))
    err = (conv3x3s2_nhwc_ph(x, w, b, relu=True, tile_m=tm).float() - ref.float()).abs().max().item()
    print(f"phase-pipelined {tm}x256: max diff vs variant 1: {err:.3g}")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
times = {k: [] for k in variants}
for _ in range(5):
    for k, fn in variants.items():
        fn(); ev[0].record(); fn(); fn(); ev[1].record(); torch.cuda.synchronize()
        times[k].append(ev[0].elapsed_time(ev[1]) / 2)
for k, v in times.items():
    v.sort()
    print(f"{k}: median {v[len(v) // 2]:.3f} ms = {fl / v[len(v) // 2] / 1e9:.0f} TF/s (min {v[0]:.3f})", flush=True)
