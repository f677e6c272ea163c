import torch, time, sys
sys.path.insert(0, '.')
from paper_accurate_fast_cheap_amd import hip_ops
M, C, nd = 44998, 512, 2
bf = torch.bfloat16
x = torch.randn(1, M, C, device='cuda', dtype=bf)
t = torch.randn(nd, M, 128, device='cuda', dtype=bf).tanh()
w2t = (torch.randn(nd, 4, C, 32, device='cuda', dtype=bf) * 0.1)
maa = torch.rand(nd, 4, C, device='cuda', dtype=bf)
import os
os.environ['PAFC_LORA_LDSW'] = '0'
z = hip_ops.tmix_lora_mix4(x, t, w2t, maa)
# reference through the unfused pair
m = torch.empty(nd, 4, M, C, device='cuda', dtype=bf)
for d in range(nd):
    torch.bmm(t[d].view(M, 4, 32).transpose(0, 1), w2t[d].transpose(1, 2).contiguous(), out=m[d])
zr = hip_ops.tmix_mix4(x, m, maa)
print("max diff vs two-step", (z.float() - zr.float()).abs().max().item())
import os
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
res = {k: [] for k in ("0", "ws64", "ws128", "ws64c4", "ws256c4", "ws512c4", "ws128c8", "ws256c8", "ws512c8")}
def setenv(v):
    for k in ("PAFC_LORA_LDSW", "PAFC_LORA_WS_BLOCKS", "PAFC_LORA_WS_WCOLS"): os.environ.pop(k, None)
    if v.startswith("ws"):
        b, _, c = v[2:].partition("c")
        os.environ["PAFC_LORA_WS_BLOCKS"] = b
        if c: os.environ["PAFC_LORA_WS_WCOLS"] = c
    else: os.environ["PAFC_LORA_LDSW"] = v
setenv("ws128c8"); print("ws equal", torch.equal(hip_ops.tmix_lora_mix4(x, t, w2t, maa), z))
for _ in range(7):
    for v in res:
        setenv(v)
        hip_ops.tmix_lora_mix4(x, t, w2t, maa)
        ev[0].record()
        for _ in range(5): hip_ops.tmix_lora_mix4(x, t, w2t, maa)
        ev[1].record(); torch.cuda.synchronize(); res[v].append(ev[0].elapsed_time(ev[1]) / 5 * 1e3)
for v, ts in res.items():
    ts.sort(); print("tmix_lora_mix4 variant", v, ": median us", round(ts[3], 1), "min", round(ts[0], 1))
