import torch, time, sys, os
from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward_bidir
B, T = int(sys.argv[1]), int(sys.argv[2]); dt = torch.bfloat16 if sys.argv[3] == "bf16" else torch.float32
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
def mk():
    r,k,v=(torch.randn(B,T,512,device='cuda').mul_(0.5).to(dt) for _ in range(3))
    w=(torch.randn(B,T,512,device='cuda')-3).to(dt); u=(torch.randn(8,64,device='cuda')*0.3).to(dt)
    return r,k,v,w,u
f, b = mk(), mk()
for _ in range(3): wkv6_forward_bidir(f, b, chunk_len=chunk)
torch.cuda.synchronize(); t0=time.time()
for _ in range(10): wkv6_forward_bidir(f, b, chunk_len=chunk)
torch.cuda.synchronize(); print("us per op", (time.time()-t0)/10*1e6)
