"""Gradients of one training step's loss with an A/B switch (an environment variable read by the path under test) on and off: same
model, same batch, same dropout masks (the seed is reset before each pass).  python tools/micro/train_grad_ab.py PAFC_TRAIN_LINEAR_GROUP"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from paper_accurate_fast_cheap_amd import _lib, hip_ops
var = sys.argv[1] if len(sys.argv) > 1 else "PAFC_TRAIN_LINEAR_GROUP"
device = torch.device("cuda", 0)
_lib.lib()
model, _ = B.build_model("fp32", device)
model.train()
g = torch.Generator().manual_seed(777)
lens = torch.randint(100, 2001, (32,), generator=g)
fb = torch.randn(32, int(lens.max()), 80, device=device)
tl = torch.randint(1, 100, (32,), generator=g)
tl = torch.minimum(tl, ((lens - 1) // 2 - 1) // 2 // 2).clamp(min=1)
target = torch.randint(1, 4999, (32, int(tl.max())), generator=g)
batch = {"feats": fb, "feats_lengths": lens.to(device), "target": target.to(device), "target_lengths": tl.to(device)}
out = {}
for rep in range(2):
    for mode in ("1", "0"):
        os.environ[var] = mode
        torch.manual_seed(5)
        model.zero_grad(set_to_none=True)
        with hip_ops.train_shadows(), torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(batch, device)["loss"]
        loss.backward()
        out[(rep, mode)] = (float(loss), {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None})
print("loss", {k: round(v[0], 4) for k, v in out.items()})


def cmp(a, b, label):
    worst = []
    for n in a:
        da, db = a[n].double().flatten(), b[n].double().flatten()
        rel = float((da - db).norm() / (da.norm() + 1e-30))
        worst.append((rel, n))
    worst.sort(reverse=True)
    print(label, "largest relative gradient differences:", [(round(r, 5), n) for r, n in worst[:6]])


cmp(out[(0, "1")][1], out[(1, "1")][1], "on vs on (run-to-run noise):")
cmp(out[(0, "0")][1], out[(1, "0")][1], "off vs off (run-to-run noise):")
cmp(out[(0, "1")][1], out[(0, "0")][1], "on vs off:")
