"""Host-side cost of one training step by Python function (cProfile over 5 steps; the step is host-bound once the GPU work is ~55 ms).
  python tools/micro/train_cprofile.py [n_top]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from paper_accurate_fast_cheap_amd import _lib
from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
device = torch.device("cuda", 0)
_lib.lib()
model, _ = B.build_model("fp32", device)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
g = torch.Generator().manual_seed(777)
lens = torch.randint(100, 2001, (32,), generator=g)
fb = torch.randn(32, int(lens.max()), 80, device=device)
tl = torch.randint(1, 100, (32,), generator=g)
tl = torch.minimum(tl, ((lens - 1) // 2 - 1) // 2 // 2).clamp(min=1)
target = torch.randint(1, 4999, (32, int(tl.max())), generator=g)
batch = {"feats": fb, "feats_lengths": lens.to(device), "target": target.to(device), "target_lengths": tl.to(device)}
for i in range(3):
    train_step(model, batch, opt, device, amp_dtype=torch.bfloat16, step_index=i)
torch.cuda.synchronize()
pr = cProfile.Profile()
# (backward on the calling thread, so that the profile sees the Python side of the custom Functions' backward too)
with torch.autograd.set_multithreading_enabled(False):
    pr.enable()
    for i in range(5):
        train_step(model, batch, opt, device, amp_dtype=torch.bfloat16, step_index=3 + i)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
