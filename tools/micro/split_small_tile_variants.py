"""The split-operand projections of a window batch (fp32 model + bf16 slot) on the small tiles of csrc/gemm_bf16.hip by TILE and
K-SPLIT variant (PAFC_F32OUT_TILE / PAFC_F32OUT_KSPLIT), us per launch, event-timed on an otherwise idle GPU:
  python tools/micro/split_small_tile_variants.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from paper_accurate_fast_cheap_amd import hip_ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
FORMS = [("w_2 +residual", 2048, 512, "none", "f32", True), ("pointwise_conv2 +residual", 512, 512, "none", "f32", True),
         ("pointwise_conv1 plain", 512, 1024, "none", "f32", False), ("w_1 planes+SiLU", 512, 2048, "silu", "planes", False)]
VARIANTS = [("auto", None, None), ("64x64", "64x64", "1"), ("128x64", "128x64", "1"), ("128x128", "128x128", "1"),
            ("128x64 k2", "128x64", "2"), ("128x128 k2", "128x128", "2"), ("128x128 k3", "128x128", "3"), ("128x128 k4", "128x128", "4")]


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


rows = [int(v) for v in sys.argv[1:]] or [1996, 3992]
hip_ops._SPLIT_SMALL_MAX_ROWS = 1 << 30
hip_ops._SPLIT_SMALL_MAX_OUT = 1 << 40
for name, K, N, act, kind, res in FORMS:
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g) * 0.1
    w3 = hip_ops.split_planes(w, triple=True)
    for M in rows:
        a = torch.randn(M, K, device=dev, generator=g)
        ap = hip_ops.split_planes(a)
        r = torch.randn(M, N, device=dev, generator=g) if res else None
        ref = a.double() @ w.double().t() + b.double()
        out = []
        for label, tile, ks in VARIANTS:
            for k_, v_ in (("PAFC_F32OUT_TILE", tile), ("PAFC_F32OUT_KSPLIT", ks)):
                if v_ is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v_
            if label == "auto":          # what the package does today (the 256-wide kernel beyond DISPATCH's small-tile bounds)
                hip_ops._SPLIT_SMALL_MAX_ROWS, hip_ops._SPLIT_SMALL_MAX_OUT = hip_ops.DISPATCH["split_small_max_rows"], 1 << 22
                out.append("auto %.1f" % timed(lambda: hip_ops.gemm_ph_ex(ap, w3, b, act, alpha=0.5 if res else 1.0, residual=r, a_split=True, out_kind=kind)))
                hip_ops._SPLIT_SMALL_MAX_ROWS, hip_ops._SPLIT_SMALL_MAX_OUT = 1 << 30, 1 << 40
                continue
            call = lambda: hip_ops.gemm_ph_ex(ap, w3, b, act, alpha=0.5 if res else 1.0, residual=r, a_split=True, out_kind=kind)
            t = timed(call)
            got = call()
            if kind == "f32" and act == "none":
                want = (0.5 * (a.double() @ w.double().t()) + b.double() + r.double()) if res else ref
                err = float((got.double() - want).abs().max())
                assert err < 1e-3, (label, err)
            out.append("%s %.1f" % (label, t))
        for k_ in ("PAFC_F32OUT_TILE", "PAFC_F32OUT_KSPLIT"):
            os.environ.pop(k_, None)
        print("%-26s rows %5d: " % (name, M) + " | ".join(out), flush=True)
