"""FFN w_1 / w_2 at the 30-minute shape in the layouts the library offers (which operand is K-contiguous, which result
is transposed): is any of them faster than the nn.Linear layout?  Timing only."""
import torch
M, dev, bf = 44998, "cuda", torch.bfloat16
x = torch.randn(M, 512, device=dev, dtype=bf)
w1 = torch.randn(2048, 512, device=dev, dtype=bf) * 0.05
w2 = torch.randn(512, 2048, device=dev, dtype=bf) * 0.05
h = torch.randn(M, 2048, device=dev, dtype=bf)
ht = h.t().contiguous()          # (2048, M)
xt = x.t().contiguous()          # (512, M)
w1t = w1.t().contiguous()        # (512, 2048)
w2t = w2.t().contiguous()        # (2048, 512)


def bench(fn, name, flops):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): fn()
    b.record(); torch.cuda.synchronize()
    dt = a.elapsed_time(b) / 30 * 1e-3
    print(f"  {name:58s} {dt*1e6:8.1f} us  {flops/dt/1e12:6.0f} TF/s", flush=True)


f1 = 2 * M * 512 * 2048
print("w_1: (M,512) -> (M,2048)")
bench(lambda: torch.mm(x, w1.t()), "x @ w1^T          (NT, out (M, 2048))", f1)
bench(lambda: torch.mm(x, w1t), "x @ w1t           (NN, out (M, 2048))", f1)
bench(lambda: torch.mm(w1, x.t()), "w1 @ x^T          (NT, out (2048, M))", f1)
bench(lambda: torch.mm(w1, xt), "w1 @ xt           (NN, out (2048, M))", f1)
bench(lambda: torch.mm(xt.t(), w1t), "xt^T @ w1t        (TN, out (M, 2048))", f1)
print("w_2: (M,2048) -> (M,512)")
bench(lambda: torch.mm(h, w2.t()), "h @ w2^T          (NT, out (M, 512))", f1)
bench(lambda: torch.mm(h, w2t), "h @ w2t           (NN, out (M, 512))", f1)
bench(lambda: torch.mm(ht.t(), w2t), "ht^T @ w2t        (TN: hidden-major h, out (M, 512))", f1)
bench(lambda: torch.mm(w2, ht), "w2 @ ht           (NN, out (512, M))", f1)
