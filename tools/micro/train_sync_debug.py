"""Host synchronisations inside one training step (torch.cuda.set_sync_debug_mode("warn") with the package frames of each warning):
round 6 found `float(coef)` of clip_grad_norm_ this way.  python tools/micro/train_sync_debug.py"""
import os, sys, warnings, traceback
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
torch.cuda.set_sync_debug_mode("warn")
def showwarning(message, category, filename, lineno, file=None, line=None):
    print("SYNC WARNING:", message)
    for fr in traceback.extract_stack()[:-1]:
        if "paper_accurate_fast_cheap_amd" in fr.filename or "torch/optim" in fr.filename or "torch/nn/utils" in fr.filename:
            print("   ", os.path.relpath(fr.filename), fr.lineno, fr.name)
warnings.showwarning = showwarning
warnings.simplefilter("always")
train_step(model, batch, opt, device, amp_dtype=torch.bfloat16, step_index=3)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("done")
