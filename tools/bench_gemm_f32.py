"""pafc_gemm_f32 (csrc/gemm_f32.hip) by kernel variant and row count: microseconds per launch of the fp32 projections of a layer
(w_1 512 -> 2048 + SiLU, w_2 2048 -> 512 + residual, pointwise 512 -> 512, pointwise_conv1 512 -> 1024, CTC head 512 -> 5000) from a
single 2 000-frame window (499 rows) to a c2 batch (16 000 rows), against the framework's F.linear on ONE stream.

  python tools/bench_gemm_f32.py            runs itself once per variant (PAFC_GEMM_F32_KERNEL = 0 auto, 1 = 128 x 128 staged tiles,
                                            2 = 64 x 64 staged tiles, 3 = few-rows split-K kernel with 32 x 32 tiles)
                                            and prints one table; a correctness check of every variant against float64 rides along."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("w_1", 512, 2048, "silu", False), ("w_2", 2048, 512, "none", True), ("pw", 512, 512, "none", True),
          ("pw1", 512, 1024, "none", False), ("head", 512, 5000, "none", False)]
ROWS = [249, 499, 996, 1992, 3992, 8000, 16000]


def child():
    import torch
    import torch.nn.functional as F
    from paper_accurate_fast_cheap_amd import hip_ops
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    out = {}
    torch_too = os.environ.get("PAFC_GEMM_F32_KERNEL", "0") == "0"
    for name, K, N, act, res in SHAPES:
        w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        b = torch.randn(N, device=dev, generator=g) * 0.1
        for M in ROWS:
            a = torch.randn(M, K, device=dev, generator=g)
            r = torch.randn(M, N, device=dev, generator=g) if res else None
            fn = lambda: hip_ops.gemm_f32(a, w, b, act, alpha=0.5 if res else 1.0, residual=r)
            got = fn()
            want = (0.5 if res else 1.0) * (a.double() @ w.double().t()) + b.double()
            if res:
                want = want + r.double()
            if act == "silu":
                want = F.silu(want)
            err = float((got.double() - want).abs().max())
            assert err <= 1e-4 * max(1.0, K ** 0.5 / 16), (name, M, err)

            def timed(f, n=30):
                for _ in range(5):
                    f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    f()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / n
            out[f"{name} {M}"] = round(timed(fn), 1)
            if torch_too:
                def tf():
                    y = F.linear(a, w, b)
                    return F.silu(y) if act == "silu" else (y * 0.5 + r if res else y)
                out[f"torch {name} {M}"] = round(timed(tf), 1)
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if os.environ.get("PAFC_GEMM_F32_CHILD") == "1":
        return child()
    res = {}
    for v in ("0", "1", "2", "3"):
        env = dict(os.environ, PAFC_GEMM_F32_CHILD="1", PAFC_GEMM_F32_KERNEL=v, PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=280)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(f"variant {v} failed:\n{r.stdout[-500:]}\n{r.stderr[-1500:]}")
            continue
        res[v] = json.loads(line[0][7:])
    print("us per launch, fp32; columns: auto | 128x128 staged | 64x64 staged | few-rows 32x32 | torch F.linear (+ separate epilogue ops), one stream")
    for name, K, N, act, resd in SHAPES:
        for M in ROWS:
            key = f"{name} {M}"
            cells = [res.get(v, {}).get(key) for v in ("0", "1", "2", "3")] + [res.get("0", {}).get("torch " + key)]
            print(f"{name:5s} {K:4d}->{N:4d} rows {M:6d}: " + " | ".join(f"{c:8.1f}" if c is not None else "       -" for c in cells))


if __name__ == "__main__":
    main()
