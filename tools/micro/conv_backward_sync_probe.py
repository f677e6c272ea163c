"""Does the library's convolution backward (the last library call of the training step: conv2 of Conv2dSubsampling4, NHWC bf16) block
the host until the GPU has drained?  Host time of the call on an idle queue against the same call behind ~20 ms of queued kernels.
  python tools/micro/conv_backward_sync_probe.py"""
import time
import torch
dev = "cuda"
B, C, T1, F1 = 32, 512, 962, 39
a = torch.randn(B, T1, F1, C, device=dev, dtype=torch.bfloat16)
w = torch.randn(C, C, 3, 3, device=dev, dtype=torch.bfloat16) * 0.02
g = torch.randn(B, (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1, C, device=dev, dtype=torch.bfloat16)
big = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)


def conv_bwd():
    return torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), a.permute(0, 3, 1, 2), w, [C], [2, 2], [0, 0], [1, 1], False, [0, 0], 1,
                                               [True, True, True])


for _ in range(3):
    conv_bwd()
torch.cuda.synchronize()
for label, load in (("idle queue", 0), ("behind queued kernels", 12)):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(load):
            big @ big
        e1.record()
        t0 = time.perf_counter()
        conv_bwd()
        host = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        ts.append((host, e0.elapsed_time(e1)))
    print(f"{label}: host time of convolution_backward {min(t[0] for t in ts):.2f}-{max(t[0] for t in ts):.2f} ms; queued work in front {ts[-1][1]:.1f} ms")
