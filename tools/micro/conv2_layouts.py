"""Subsampling conv2 (512 -> 512, 3x3, stride 2) forward + backward through the library at the c4 training shape: NCHW
(contiguous) vs channels_last input -- does the NHWC path avoid the layout transposes?  And conv1 (1 -> 512) forward."""
import torch
import torch.nn.functional as F
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
B, T, Fq = 32, 999, 39
w2 = (torch.randn(512, 512, 3, 3, device=dev) * 0.02).to(bf).requires_grad_()
b2 = torch.zeros(512, device=dev, dtype=bf, requires_grad=True)
x_nchw = torch.randn(B, 512, T, Fq, device=dev, dtype=bf).requires_grad_()
x_cl = x_nchw.detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_()
w2_cl = w2.detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_()


def timed(fn, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def run(x, w):
    y = F.relu(F.conv2d(x, w, b2, stride=2))
    y.float().sum().backward()
    return y


print(f"conv2 fwd+bwd NCHW          {timed(lambda: run(x_nchw, w2)):.2f} ms")
print(f"conv2 fwd+bwd channels_last {timed(lambda: run(x_cl, w2_cl)):.2f} ms   (out strides {run(x_cl, w2_cl).stride()})")
x1 = torch.randn(B, 1, 2000, 80, device=dev, dtype=bf)
w1 = (torch.randn(512, 1, 3, 3, device=dev) * 0.1).to(bf).requires_grad_()
b1 = torch.zeros(512, device=dev, dtype=bf, requires_grad=True)
print(f"conv1 fwd+bwd NCHW          {timed(lambda: F.relu(F.conv2d(x1, w1, b1, stride=2)).float().sum().backward()):.2f} ms")
with torch.no_grad():
    print(f"conv1 fwd only              {timed(lambda: F.relu(F.conv2d(x1, w1, b1, stride=2))):.2f} ms")
