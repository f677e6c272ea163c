"""Training-step GEMM shapes (M = 16 000 rows = 32 utterances): F.linear / matmul through the library against pafc_gemm_bf16
(forward y = x W^T and the input gradient dx = dy W as the same kernel on a transposed copy of W).  python tools/bench_gemm_train.py [M]"""
import sys, time
import torch
import torch.nn.functional as F
from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
dev, bf = "cuda", torch.bfloat16


def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time(); n = 20
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6


for (K, N, name) in [(512, 2048, "ffn w_1"), (2048, 512, "ffn w_2"), (512, 1024, "pointwise_conv1"), (512, 512, "r/k/v, pointwise_conv2"),
                     (512, 128, "LoRA down"), (9728, 512, "subsampling out"), (512, 5000, "ctc head")]:
    x = torch.randn(M, K, device=dev, dtype=bf); w = torch.randn(N, K, device=dev, dtype=bf) * 0.05
    b = torch.randn(N, device=dev, dtype=bf); dy = torch.randn(M, N, device=dev, dtype=bf)
    wt = w.t().contiguous()
    e1 = (gemm_bf16(x, w, b).float() - F.linear(x, w, b).float()).abs().max().item()
    e2 = (gemm_bf16(dy, wt).float() - (dy @ w).float()).abs().max().item()
    t = [bench(lambda: F.linear(x, w, b)), bench(lambda: gemm_bf16(x, w, b)), bench(lambda: dy @ w), bench(lambda: gemm_bf16(dy, wt)),
         bench(lambda: w.t().contiguous())]
    print(f"{name:24s} K={K:5d} N={N:5d}: fwd library {t[0]:7.1f} us, ours {t[1]:7.1f} us | dgrad library {t[2]:7.1f} us, ours {t[3]:7.1f} us "
          f"(+ transpose {t[4]:5.1f} us) | max diff {e1:.3g} {e2:.3g}", flush=True)
