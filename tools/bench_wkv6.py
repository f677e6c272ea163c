import torch, time, sys
from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward, wkv6_forward_bidir
def mk(B,T,C,H,dt):
    r,k,v=(torch.randn(B,T,C,device='cuda').mul_(0.5).to(dt) for _ in range(3))
    w=(torch.randn(B,T,C,device='cuda')-3).to(dt); u=(torch.randn(H,C//H,device='cuda')*0.3).to(dt)
    return r,k,v,w,u
for (B,T) in [(1,44998),(8,2500),(8,10000),(64,400)]:
  for dt in (torch.bfloat16, torch.float32):
    f=mk(B,T,512,8,dt); b=mk(B,T,512,8,dt)
    for chunk in (0, 64, 128, 256, 512, 10**6):
        for _ in range(2): wkv6_forward_bidir(f,b,chunk_len=chunk)
        torch.cuda.synchronize(); t0=time.time(); n=5
        for _ in range(n): wkv6_forward_bidir(f,b,chunk_len=chunk)
        torch.cuda.synchronize(); dtm=(time.time()-t0)/n
        es=2 if dt==torch.bfloat16 else 4
        byts=B*T*512*10*es
        print(f"B={B} T={T} {str(dt)[6:]} chunk={chunk}: {dtm*1e6:.0f} us  alg {byts/dtm/1e9:.0f} GB/s  flops {B*T*512*896/dtm/1e12:.1f} TF", flush=True)
