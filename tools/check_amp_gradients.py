import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench as B
from paper_accurate_fast_cheap_amd import _lib
_lib.lib()
dev = torch.device("cuda", 0)
model, _ = B.build_model("fp32", dev)
g = torch.Generator().manual_seed(777)
nb = 8
lens = torch.randint(100, 801, (nb,), generator=g)
feats, _ = B.front_end(B.synthetic_waveform(60.0, 777), dev)
src = feats[0]
fb = torch.zeros(nb, int(lens.max()), 80, device=dev)
for j, n in enumerate(lens.tolist()):
    off = (j * 7919) % (src.shape[0] - 2001)
    fb[j, :n] = src[off:off + n]
tl = torch.randint(1, 61, (nb,), generator=g)
tl = torch.minimum(tl, ((lens - 1) // 2 - 1) // 2 // 2).clamp(min=1)
tgt = torch.randint(1, B.VOCAB, (nb, int(tl.max())), generator=g)
batch = {"feats": fb, "feats_lengths": lens.to(dev), "target": tgt.to(dev), "target_lengths": tl.to(dev)}
model.eval()   # no dropout: compare gradients only
def grads(amp):
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        loss = model(batch, dev)["loss"]
    loss.backward()
    return float(loss), {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None}
l0, g0 = grads(False)
l0b, g0b = grads(False)
l1, g1 = grads(True)
print("loss fp32", l0, l0b, "amp", l1)
tot = lambda gg: float(torch.sqrt(sum((v.double() ** 2).sum() for v in gg.values())))
print("gnorm fp32", tot(g0), tot(g0b), "amp", tot(g1))
rows = []
for n in g0:
    a, b = g0[n].flatten().double(), g1[n].flatten().double()
    c = float((a @ b) / (a.norm() * b.norm() + 1e-30))
    a2 = g0b[n].flatten().double()
    c2 = float((a @ a2) / (a.norm() * a2.norm() + 1e-30))
    rows.append((c, c2, float(a.norm()), float(b.norm()), n))
rows.sort()
for r in rows[:25]:
    print("cos amp %.4f  cos rerun %.4f  |g32| %.3e |gamp| %.3e  %s" % r)
print("median cos", rows[len(rows) // 2][0], "n", len(rows))
big = sorted(rows, key=lambda r: -r[2])[:10]
for r in big:
    print("BIG cos amp %.4f  cos rerun %.4f  |g32| %.3e |gamp| %.3e  %s" % r)
