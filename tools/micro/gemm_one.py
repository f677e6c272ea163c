"""One GEMM shape through one path, 200 launches (for rocprofv3 --kernel-trace --stats): gemm_one.py M K N path [res] [act]"""
import sys
import torch
from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16, linear_bias_act
M, K, N, path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
res = len(sys.argv) > 5 and sys.argv[5] == "1"
act = sys.argv[6] if len(sys.argv) > 6 else "none"
bf = torch.bfloat16
x = torch.randn(M, K, device="cuda", dtype=bf); w = torch.randn(N, K, device="cuda", dtype=bf) * 0.05
b = torch.randn(N, device="cuda", dtype=bf); r = torch.randn(M, N, device="cuda", dtype=bf) if res else None
f = (lambda: gemm_bf16(x, w, b, act, residual=r)) if path == "own" else (lambda: linear_bias_act(x, w, b, act, residual=r))
for _ in range(200):
    f()
torch.cuda.synchronize()
