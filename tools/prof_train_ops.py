"""Host-side view of one training step (torch.profiler): which ATen ops are called how often and what they cost on the CPU --
the step is launch-bound on a slow host, so the op COUNT is what to cut.  python tools/prof_train_ops.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from paper_accurate_fast_cheap_amd import _lib
from paper_accurate_fast_cheap_amd.utils.train_utils import train_step

device = torch.device("cuda", 0)
_lib.lib()
model, _ = B.build_model("fp32", device)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
g = torch.Generator().manual_seed(777)
lens = torch.randint(100, 2001, (32,), generator=g)
feats, _ = B.front_end(B.synthetic_waveform(60.0, 777), device)
src = feats[0]
fb = torch.zeros(32, int(lens.max()), 80, device=device)
for j, n in enumerate(lens.tolist()):
    off = (j * 7919) % (src.shape[0] - 2001)
    fb[j, :n] = src[off:off + n]
tl = torch.randint(1, 161, (32,), generator=g)
tl = torch.minimum(tl, ((lens - 1) // 2 - 1) // 2 // 2).clamp(min=1)
target = torch.randint(1, 4999, (32, int(tl.max())), generator=g)
batch = {"feats": fb, "feats_lengths": lens.to(device), "target": target.to(device), "target_lengths": tl.to(device)}
for i in range(3):
    train_step(model, batch, opt, device, amp_dtype=torch.bfloat16, step_index=i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
STACKS = "--stacks" in sys.argv        # attribute the small ops (casts, copies, adds) to the package line that calls them
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=STACKS) as prof:
    train_step(model, batch, opt, device, amp_dtype=torch.bfloat16, step_index=3)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::mul", "aten::mul_", "aten::add_", "aten::add", "aten::sum", "aten::contiguous")]
rows.sort(key=lambda e: -e.count)
for e in rows[:60]:
    print(f"{e.key:16s} {e.count:5d}  cpu {e.cpu_time_total / 1e3:7.2f} ms  {str(e.input_shapes)[:150]}")

if STACKS:
    # (Python frames are not recorded by this build's profiler; the op tree is: attribute every small op to the nearest enclosing
    # event that is not an aten:: op -- a custom Function's forward / backward, an autograd node, an optimizer section)
    import collections
    small = ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::mul", "aten::sum",
             "aten::cat", "aten::stack", "aten::div", "aten::fill_", "aten::zero_", "aten::zeros_like", "aten::sub", "aten::neg",
             "aten::mul_", "aten::masked_fill", "aten::masked_fill_", "aten::where", "aten::bitwise_not", "aten::flip", "aten::glu",
             "aten::silu", "aten::tanh", "aten::sigmoid", "aten::eq", "aten::lt", "aten::ge", "aten::arange")
    launches = {}
    for e in prof.events():
        if e.device_type.name != "CPU":
            continue
    by = collections.Counter()
    for e in prof.events():
        if e.name not in small or e.device_type.name != "CPU":
            continue
        par = e.cpu_parent
        if par is not None and par.name in small:
            continue                      # the outer aten op is counted (contiguous -> clone -> copy_)
        where = "(top level)"
        while par is not None:
            if not par.name.startswith("aten::"):
                where = par.name
                break
            par = par.cpu_parent
        shape = str(e.input_shapes)[:70] if e.input_shapes else ""
        by[(e.name, where[:70], shape)] += 1
    print("\nsmall ops by enclosing event (count >= 12):")
    for (name, where, shape), n in sorted(by.items(), key=lambda kv: -kv[1]):
        if n >= 12:
            print(f"{n:5d}  {name:18s} {where:70s} {shape}")

    # every aten::copy_ (casts and device-to-device copies) by its chain of enclosing ops
    byc = collections.Counter()
    for e in prof.events():
        if e.name != "aten::copy_" or e.device_type.name != "CPU":
            continue
        chain, par, where = [], e.cpu_parent, "(top level)"
        while par is not None:
            if par.name.startswith("aten::"):
                chain.append(par.name[6:])
            else:
                where = par.name
                break
            par = par.cpu_parent
        byc[("<-".join(chain) or "-", where[:60], str(e.input_shapes)[:48])] += 1
    print("\naten::copy_ by enclosing chain (count >= 6):")
    for (chain, where, shape), n in sorted(byc.items(), key=lambda kv: -kv[1]):
        if n >= 6:
            print(f"{n:5d}  {chain:34s} {where:60s} {shape}")
    agg = collections.Counter()
    for (chain, where, shape), n in byc.items():
        agg[(chain, where)] += n
    print("\naten::copy_ by (chain, enclosing event), all:")
    for (chain, where), n in sorted(agg.items(), key=lambda kv: -kv[1])[:30]:
        print(f"{n:5d}  {chain:34s} {where}")
    # device-to-device copies issued below the op level (hipMemcpyAsync) by enclosing event
    mc = collections.Counter()
    for e in prof.events():
        if e.name not in ("hipMemcpyAsync", "hipMemcpyWithStream") or e.device_type.name != "CPU":
            continue
        chain, par = [], e.cpu_parent
        while par is not None and len(chain) < 4:
            chain.append(par.name[:48])
            par = par.cpu_parent
        mc[" <- ".join(chain)] += 1
    print("\nhipMemcpyAsync by enclosing events:")
    for k, n in sorted(mc.items(), key=lambda kv: -kv[1])[:25]:
        print(f"{n:5d}  {k}")
