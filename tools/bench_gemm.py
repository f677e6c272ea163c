import torch, time
import torch.nn.functional as F
from paper_accurate_fast_cheap_amd.hip_ops import linear_bias_act
M=44998
def bench(fn, name, flops):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.time(); n=20
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt=(time.time()-t0)/n
    print(f"{name}: {dt*1e6:.1f} us  {flops/dt/1e12:.0f} TF/s", flush=True)
for (K,N) in [(512,2048),(2048,512),(512,1024),(512,512),(1024,512)]:
    x=torch.randn(M,K,device='cuda',dtype=torch.bfloat16); w=torch.randn(N,K,device='cuda',dtype=torch.bfloat16)*0.05; b=torch.randn(N,device='cuda',dtype=torch.bfloat16)
    fl=2*M*K*N
    bench(lambda: F.linear(x,w,b), f"torch linear {K}->{N}", fl)
    bench(lambda: F.silu(F.linear(x,w,b)), f"torch linear+silu {K}->{N}", fl)
    bench(lambda: linear_bias_act(x,w,b,"silu"), f"pafc linear_bias_silu {K}->{N}", fl)
    bench(lambda: linear_bias_act(x,w,b,"none"), f"pafc linear_bias {K}->{N}", fl)
