"""Time Conv2dSubsampling4 variants on the 30-min shape (B=1, T=179998) in bf16."""
import time, torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = "cuda"
dt = torch.bfloat16
B, T, Fd, C = 1, 179998, 80, 512
x = torch.randn(B, T, Fd, device=dev, dtype=dt)
w1 = torch.randn(C, 1, 3, 3, device=dev, dtype=dt) * 0.3; b1 = torch.randn(C, device=dev, dtype=dt) * 0.1
w2 = torch.randn(C, C, 3, 3, device=dev, dtype=dt) * 0.02; b2 = torch.randn(C, device=dev, dtype=dt) * 0.1
wl = torch.randn(C, C * 19, device=dev, dtype=dt) * 0.01; bl = torch.randn(C, device=dev, dtype=dt) * 0.1

def ref():
    y = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2))
    y = F.relu(F.conv2d(y, w2, b2, stride=2))
    b, c, t, f = y.shape
    return F.linear(y.transpose(1, 2).contiguous().view(b, t, c * f), wl, bl)

w1c = w1.contiguous(memory_format=torch.channels_last); w2c = w2.contiguous(memory_format=torch.channels_last)
wl_perm = wl.view(C, C, 19).permute(0, 2, 1).reshape(C, 19 * C).contiguous()
def cl():
    y = F.relu(F.conv2d(x.unsqueeze(1).contiguous(memory_format=torch.channels_last), w1c, b1, stride=2))
    y = F.relu(F.conv2d(y, w2c, b2, stride=2))
    b, c, t, f = y.shape
    return F.linear(y.permute(0, 2, 3, 1).reshape(b, t, f * c), wl_perm, bl)

# conv1 as unfold-GEMM producing NHWC directly, conv2 channels_last
def gemm1():
    p = x.unsqueeze(1).unfold(2, 3, 2).unfold(3, 3, 2)          # (B,1,T1,F1,3,3) view
    p = p.reshape(B, -1, 9)                                        # copy: (B, T1*F1, 9)
    y = F.relu(F.linear(p, w1.view(C, 9), b1))                     # (B, T1*F1, C) NHWC
    T1 = (T - 3) // 2 + 1; F1 = (Fd - 3) // 2 + 1
    y = y.view(B, T1, F1, C).permute(0, 3, 1, 2)                   # logical NCHW, channels_last memory
    y = F.relu(F.conv2d(y, w2c, b2, stride=2))
    b, c, t, f = y.shape
    return F.linear(y.permute(0, 2, 3, 1).reshape(b, t, f * c), wl_perm, bl)

def bench(fn, name):
    for _ in range(2): out = fn()
    torch.cuda.synchronize(); t0 = time.time(); n = 3
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.time()-t0)/n*1e3:.2f} ms", flush=True)
    return out
with torch.no_grad():
    a = bench(ref, "nchw reference layout")
    b = bench(cl, "channels_last")
    print("max diff cl vs ref", (a.float()-b.float()).abs().max().item(), a.float().abs().max().item())
    try:
        c = bench(gemm1, "unfold-gemm conv1 + channels_last conv2")
        print("max diff gemm1 vs ref", (a.float()-c.float()).abs().max().item())
    except Exception as e:
        print("gemm1 failed", repr(e)[:200])
