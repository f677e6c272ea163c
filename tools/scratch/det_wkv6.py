import torch, sys
sys.path.insert(0, '.')
from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward_bidir
torch.manual_seed(0)
B,T,dt=1,44998,torch.bfloat16
def mk():
    r,k,v=(torch.randn(B,T,512,device='cuda').mul_(0.5).to(dt) for _ in range(3))
    w=(torch.randn(B,T,512,device='cuda')-3).to(dt); u=(torch.randn(8,64,device='cuda')*0.3).to(dt)
    return r,k,v,w,u
f,b=mk(),mk()
y0=[t.clone() for t in wkv6_forward_bidir(f,b)]
for i in range(5):
    y=wkv6_forward_bidir(f,b)
    for d in range(2):
        ne=(y[d]!=y0[d]).sum().item()
        print(i,d,'differing elements',ne, 'max abs', (y[d].float()-y0[d].float()).abs().max().item())
print('sum', y0[0].float().abs().sum().item())
