"""Config c4: the training step.  GPU: autograd through the HIP WKV backward inside the whole encoder equals the CPU
oracle's gradients.  CPU (gloo, 2 ranks): the DDP step semantics -- averaged gradients equal the single-process
gradients of the concatenated batch, clipping, skip-on-nonfinite."""
import os
import subprocess
import sys

import pytest
import torch

from oracle import encoder_oracle as EO
from oracle import wkv6_oracle as WO
from tests import synth
from tests.conftest import ROOT, load_golden


_ORIG_FORWARD = WO.forward


class _OracleWKV(torch.autograd.Function):
    """Autograd wrapper around the C restatement (forward wkv6_cuda.cu:8-63, backward :65-263) for reference grads."""

    @staticmethod
    def forward(ctx, r, k, v, w, u):
        ctx.save_for_backward(r, k, v, w, u)
        return _ORIG_FORWARD(r.contiguous(), k.contiguous(), v.contiguous(), w.contiguous(), u.contiguous())

    @staticmethod
    def backward(ctx, gy):
        r, k, v, w, u = ctx.saved_tensors
        return WO.backward(r.contiguous(), k.contiguous(), v.contiguous(), w.contiguous(), u.contiguous(), gy.contiguous())


@pytest.mark.gpu
def test_encoder_ctc_gradients_match_oracle(hip, monkeypatch):
    from paper_accurate_fast_cheap_amd.transformer.asr_model import ASRModel
    from paper_accurate_fast_cheap_amd.transformer.ctc import CTC
    from paper_accurate_fast_cheap_amd.transformer.encoder import ConformerEncoder
    g = load_golden("encoder_reduced_f32")
    conf = dict(g["conf"], dropout_rate=0.0, positional_dropout_rate=0.0)
    sd = {k: v for k, v in synth.synth_state_dict(g["spec"], g["seed"]).items() if not k.startswith("global_cmvn")}
    csd = synth.synth_state_dict(g["ctc_spec"], g["ctc_seed"])
    xs = synth.randn((2, 83, 80), 5, 2.0)
    lens = torch.tensor([83, 61])
    ys = torch.tensor([[3, 7, 7, 9, 1], [4, 2, 8, 0, 0]])
    ylens = torch.tensor([5, 3])

    # reference gradients on CPU: oracle graph with autograd through the C backward
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    cleaves = {"ctc." + k: v.clone().requires_grad_() for k, v in csd.items()}
    monkeypatch.setattr(EO.wkv6_oracle, "forward", lambda r, k, v, w, u: _OracleWKV.apply(r, k, v, w, u))
    out, masks = EO.encoder_forward(xs, lens, leaves, conf, env={})
    logp = EO.ctc_log_softmax(out, cleaves).transpose(0, 1)
    enc_lens = masks.squeeze(1).sum(1)
    ref_loss = torch.nn.functional.ctc_loss(logp, ys, enc_lens, ylens, blank=0, reduction="sum", zero_infinity=True) / 2
    ref_loss.backward()

    enc = ConformerEncoder(80, **conf)
    enc.load_state_dict(sd)
    ctc = CTC(50, 128)
    ctc.load_state_dict(csd)
    model = ASRModel(50, enc, ctc).cuda().train()
    batch = {"feats": xs, "feats_lengths": lens, "target": ys, "target_lengths": ylens}
    loss = model(batch, torch.device("cuda"))["loss"]
    loss.backward()
    assert float(loss) == pytest.approx(float(ref_loss), rel=1e-4)
    checked = 0
    for name, p in model.encoder.named_parameters():
        rg = leaves[name].grad
        assert p.grad is not None, name
        scale = max(float(rg.abs().max()), 1e-3)
        err = float((p.grad.cpu() - rg).abs().max())
        assert err <= 2e-3 * scale, f"{name}: {err:.3e} vs scale {scale:.3e}"
        checked += 1
    assert checked == len(list(model.encoder.parameters())) and checked > 100


@pytest.mark.gpu
def test_train_step_updates_and_skips_nonfinite(hip):
    from paper_accurate_fast_cheap_amd.utils.init_model import init_model
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    g = load_golden("encoder_reduced_f32")   # fp32 slot parameters: an Adam step of 1e-4 is below bf16 resolution
    cfg = dict(encoder="conformer", encoder_conf=dict(g["conf"]), input_dim=80, output_dim=50, ctc="ctc",
               ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})

    class A:
        checkpoint = None
    torch.manual_seed(3)
    model, _ = init_model(A(), cfg)
    model = model.cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    batch = {"feats": synth.randn((4, 120, 80), 1, 2.0), "feats_lengths": torch.tensor([120, 100, 90, 64]),
             "target": torch.randint(1, 50, (4, 6), generator=torch.Generator().manual_seed(2)),
             "target_lengths": torch.tensor([6, 5, 4, 3])}
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    info = train_step(model, batch, opt, torch.device("cuda"), grad_clip=0.1)
    assert info["updated"] and torch.isfinite(info["loss"]) and torch.isfinite(info["grad_norm"])
    same = [n for n, p in model.named_parameters() if torch.equal(before[n], p.detach())]
    # the LoRA output matrices see tanh(x @ 0) = 0 at the reference's init (src/model.py:244,251): zero gradient;
    # time_maa_* start at 1 - (i/C)^p with entries exactly 1.0 / near 1 where an Adam step of 1e-4 can round away
    assert all(n.endswith(("time_maa_rkvw_w2", "time_decay_w2")) or ".time_maa_" in n for n in same), same
    for d in ("rwkv_wrapper_forward", "rwkv_wrapper_backward"):      # both directions of the slot are trained
        assert not any(d in n and n.endswith(("receptance.weight", "time_faaaa", "time_decay")) for n in same)
    bad = dict(batch, feats=batch["feats"] * float("nan"))
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    info = train_step(model, bad, opt, torch.device("cuda"), grad_clip=0.1)
    assert not info["updated"]
    assert all(torch.equal(before[n], p.detach()) for n, p in model.named_parameters())


@pytest.mark.gpu
def test_train_step_with_a_fused_optimizer_never_synchronises_the_host(hip):
    """Round 6: rounds 3-5 read the clip coefficient back (`float(coef)`) between backward and the update -- one host synchronisation
    per step, found with torch.cuda.set_sync_debug_mode.  Now the clip rides in the fused optimizer's `grad_scale`; a step under
    sync-debug "error" must not raise (bf16 autocast: the CTC objective on the hand-written kernels -- torch's ctc_loss copies the
    lengths to the host)."""
    from paper_accurate_fast_cheap_amd.utils.init_model import init_model
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    g = load_golden("encoder_reduced_f32")
    cfg = dict(encoder="conformer", encoder_conf=dict(g["conf"], dropout_rate=0.0, positional_dropout_rate=0.0), input_dim=80,
               output_dim=56, ctc="ctc", ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})

    class A:
        checkpoint = None
    dev = torch.device("cuda")
    batch = {"feats": synth.randn((4, 120, 80), 1, 2.0).cuda(), "feats_lengths": torch.tensor([120, 100, 90, 64]).cuda(),
             "target": torch.randint(1, 50, (4, 6), generator=torch.Generator().manual_seed(2)).cuda(),
             "target_lengths": torch.tensor([6, 5, 4, 3]).cuda()}
    torch.manual_seed(3)
    model, _ = init_model(A(), cfg)
    model = model.cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    from paper_accurate_fast_cheap_amd.hip_ops import ctc_head_loss_eligible
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert ctc_head_loss_eligible(torch.zeros(4, 29, model.ctc.ctc_lo.in_features, device="cuda"), model.ctc.ctc_lo.weight, batch["target"])
    for i in range(2):
        train_step(model, batch, opt, dev, grad_clip=0.1, step_index=i, amp_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        info = train_step(model, batch, opt, dev, grad_clip=0.1, step_index=2, amp_dtype=torch.bfloat16)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert isinstance(info["updated"], torch.Tensor) and bool(info["updated"]) and float(info["grad_norm"]) > 0.1     # clipping was active
    assert getattr(opt, "grad_scale", None) is None and getattr(opt, "found_inf", None) is None


@pytest.mark.gpu
def test_inference_after_a_training_step_that_moved_parameters(hip):
    """Round 6: the first bf16-autocast training step moves the r / k / v weights and the lerp coefficients of every time-mix block
    into shared buffers (`p.data = group[i]`).  Inference plans and captured graphs made BEFORE hold the old addresses: the move
    bumps the parameter epoch, so the next inference pass -- eager and from the graph cache -- reads the moved, updated parameters
    and equals a fresh model loaded from the trained state_dict."""
    from paper_accurate_fast_cheap_amd.utils.init_model import init_model
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    g = load_golden("encoder_reduced_bf16slot")        # fp32 model around the bf16 time-mix slot (the YAML default)
    cfg = dict(encoder="conformer", encoder_conf=dict(g["conf"], dropout_rate=0.0, positional_dropout_rate=0.0), input_dim=80,
               output_dim=56, ctc="ctc", ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})

    class A:
        checkpoint = None
    dev = torch.device("cuda")
    torch.manual_seed(3)
    model, _ = init_model(A(), cfg)
    model = model.cuda()
    feats = synth.randn((12, 120, 80), 1, 2.0).cuda()          # 12 x 29 = 348 rows: the grouped projections want >= 256
    lens = torch.tensor([120] * 12).cuda()
    batch = {"feats": feats, "feats_lengths": lens, "target": torch.randint(1, 50, (12, 6), generator=torch.Generator().manual_seed(2)).cuda(),
             "target_lengths": torch.tensor([6, 5, 4, 3] * 3).cuda()}

    def infer(m):
        m.eval()
        with torch.no_grad():
            return m.ctc_logprobs(m._forward_encoder(feats, lens)[0]).float().clone()
    enc = model.encoder
    enc.graph_cache_size = 2
    before = [infer(model) for _ in range(4)]                     # the shape is replayed from a captured graph by now
    assert all(torch.equal(before[0], b) for b in before[1:])
    tm = model.encoder.encoders[0].self_attn.rwkv_wrapper_forward.tmix_block
    ptr0 = tm.receptance.weight.data_ptr()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, fused=True)
    train_step(model, batch, opt, dev, grad_clip=0.1, amp_dtype=torch.bfloat16)
    assert tm.receptance.weight.data_ptr() != ptr0 and tm.key.weight.data_ptr() - tm.receptance.weight.data_ptr() == tm.key.weight.numel() * 2
    after = [infer(model) for _ in range(4)]
    assert all(torch.equal(after[0], a) for a in after[1:])
    assert float((after[0] - before[0]).abs().max()) > 1e-3       # the step changed the model (lr 1e-2: above bf16 resolution)
    torch.manual_seed(4)
    fresh, _ = init_model(A(), cfg)
    fresh = fresh.cuda()
    fresh.load_state_dict(model.state_dict())
    torch.testing.assert_close(infer(fresh), after[0], rtol=0, atol=0)


@pytest.mark.gpu
def test_train_step_fused_optimizer_skips_nonfinite_on_the_device(hip):
    """With a fused optimizer the skip-on-inf / nan decision (train_utils.py:702-711) is taken on the device (the optimizer's
    `found_inf` flag): no host synchronisation in the step, info["updated"] is a 0-dim bool tensor; a NaN batch leaves the
    parameters, both Adam moments and the step counts exactly as they were, a finite batch updates as the unfused path does."""
    from paper_accurate_fast_cheap_amd.utils.init_model import init_model
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    g = load_golden("encoder_reduced_f32")
    cfg = dict(encoder="conformer", encoder_conf=dict(g["conf"], dropout_rate=0.0, positional_dropout_rate=0.0), input_dim=80,
               output_dim=50, ctc="ctc", ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})

    class A:
        checkpoint = None
    batch = {"feats": synth.randn((4, 120, 80), 1, 2.0), "feats_lengths": torch.tensor([120, 100, 90, 64]),
             "target": torch.randint(1, 50, (4, 6), generator=torch.Generator().manual_seed(2)),
             "target_lengths": torch.tensor([6, 5, 4, 3])}
    runs = {}
    for fused in (True, False):
        torch.manual_seed(3)
        model, _ = init_model(A(), cfg)
        model = model.cuda()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=fused)
        info = train_step(model, batch, opt, torch.device("cuda"), grad_clip=0.1)
        assert bool(info["updated"]) and (isinstance(info["updated"], torch.Tensor) == fused)
        runs[fused] = {n: p.detach().clone() for n, p in model.named_parameters()}
        if fused:
            snap = {n: p.detach().clone() for n, p in model.named_parameters()}
            st = {i: {k: v.detach().clone() for k, v in s_.items() if torch.is_tensor(v)} for i, s_ in enumerate(opt.state.values())}
            bad = dict(batch, feats=batch["feats"] * float("nan"))
            info = train_step(model, bad, opt, torch.device("cuda"), grad_clip=0.1)
            assert isinstance(info["updated"], torch.Tensor) and not bool(info["updated"])
            assert all(torch.equal(snap[n], p.detach()) for n, p in model.named_parameters())
            for i, s_ in enumerate(opt.state.values()):
                for k, v in s_.items():
                    if torch.is_tensor(v):
                        assert torch.equal(st[i][k], v), k            # exp_avg, exp_avg_sq and step untouched
            assert getattr(opt, "found_inf", None) is None
    for n in runs[True]:       # the first Adam step is lr * sign-like: elements whose gradient is ~0 may differ by a fraction of lr = 1e-4
        torch.testing.assert_close(runs[True][n], runs[False][n], rtol=1e-5, atol=2e-5)


_DDP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["PAFC_ROOT"])
from paper_accurate_fast_cheap_amd.transformer.ctc import CTC
from paper_accurate_fast_cheap_amd.transformer.positionwise_feed_forward import PositionwiseFeedForward
from paper_accurate_fast_cheap_amd.utils.train_utils import train_step, wrap_model_ddp, reduce_seen_frames

class Tiny(torch.nn.Module):          # the CPU-runnable tail of the path: FFN + CTC head (the WKV slot is GPU-only)
    def __init__(self):
        super().__init__()
        self.ff = PositionwiseFeedForward(16, 32, 0.0, torch.nn.SiLU())
        self.ctc = CTC(11, 16)
    def forward(self, batch, device):
        h = self.ff(batch["feats"])
        loss, _ = self.ctc(h, batch["feats_lengths"], batch["target"], batch["target_lengths"])
        return {"loss": loss}

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
ref = Tiny()
model = Tiny(); model.load_state_dict(ref.state_dict())
ddp = wrap_model_ddp(model)
g = torch.Generator().manual_seed(1)
feats = torch.randn(4, 12, 16, generator=g); lens = torch.tensor([12, 10, 9, 7])
tgt = torch.randint(1, 11, (4, 3), generator=g); tl = torch.tensor([3, 2, 3, 1])
sl = slice(rank * 2, rank * 2 + 2)
batch = {"feats": feats[sl], "feats_lengths": lens[sl], "target": tgt[sl], "target_lengths": tl[sl]}
opt = torch.optim.SGD(ddp.parameters(), lr=0.0)      # lr 0: look at the gradients themselves
ddp.train(); out = ddp(batch, torch.device("cpu")); out["loss"].backward()
# single-process reference: mean over ranks of per-rank (sum / B_rank) losses
tot = 0.0
for r in range(world):
    s2 = slice(r * 2, r * 2 + 2)
    tot = tot + ref({"feats": feats[s2], "feats_lengths": lens[s2], "target": tgt[s2], "target_lengths": tl[s2]}, None)["loss"]
(tot / world).backward()
for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
    assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-6), n
opt.zero_grad()
info = train_step(ddp, batch, torch.optim.Adam(ddp.parameters(), lr=1e-3), torch.device("cpu"), grad_clip=0.1)
assert info["updated"] and float(info["grad_norm"]) > 0
w = [p.detach().clone() for p in model.parameters()]
flat = torch.cat([x.flatten() for x in w]); gathered = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
assert all(torch.equal(gathered[0], t) for t in gathered)      # replicas stay identical after the step
assert reduce_seen_frames(int(lens[sl].sum()), torch.device("cpu")) == (int(lens.sum()) if rank == 0 else int(lens[sl].sum()))
if rank == 0: print("DDP_OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_ddp_step(tmp_path):
    script = tmp_path / "ddp_worker.py"
    script.write_text(_DDP_WORKER)
    env = dict(os.environ, PAFC_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29627", str(script)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "DDP_OK" in out.stdout


class _TinyCtc(torch.nn.Module):
    """The CPU-runnable tail of the path (FFN + CTC head) -- enough to exercise the step's control flow."""

    def __init__(self):
        super().__init__()
        from paper_accurate_fast_cheap_amd.transformer.ctc import CTC
        from paper_accurate_fast_cheap_amd.transformer.positionwise_feed_forward import PositionwiseFeedForward
        self.ff = PositionwiseFeedForward(16, 32, 0.0, torch.nn.SiLU())
        self.ctc = CTC(11, 16)
        self.seen_dtype = None

    def forward(self, batch, device):
        h = self.ff(batch["feats"])
        self.seen_dtype = h.dtype
        loss, _ = self.ctc(h.float(), batch["feats_lengths"], batch["target"], batch["target_lengths"])
        return {"loss": loss}


def _tiny_batch():
    g = torch.Generator().manual_seed(1)
    return {"feats": torch.randn(4, 12, 16, generator=g), "feats_lengths": torch.tensor([12, 10, 9, 7]),
            "target": torch.randint(1, 11, (4, 3), generator=g), "target_lengths": torch.tensor([3, 2, 3, 1])}


def test_train_step_hard_clip_drops_the_update_after_warmup():
    """train_utils.py:683-684,712-716: grad_norm above grad_clip_hard_maxvalue => no update once past the warm-up."""
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    torch.manual_seed(0)
    model, batch, cpu = _TinyCtc(), _tiny_batch(), torch.device("cpu")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in model.parameters()]
    info = train_step(model, batch, opt, cpu, step_index=5, clip_hard_maxvalue=1e-6, clip_hard_warmup=3)
    assert not info["updated"] and float(info["grad_norm"]) > 1e-6
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    assert all(p.grad is None for p in model.parameters())                      # gradients are dropped all the same
    info = train_step(model, batch, opt, cpu, step_index=2, clip_hard_maxvalue=1e-6, clip_hard_warmup=3)
    assert info["updated"]                                                      # still inside the warm-up
    info = train_step(model, batch, opt, cpu, step_index=9, clip_hard_maxvalue=1e6, clip_hard_warmup=3)
    assert info["updated"]


def test_train_step_accumulates_over_accum_grad_batches():
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    torch.manual_seed(0)
    model, batch, cpu = _TinyCtc(), _tiny_batch(), torch.device("cpu")
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    before = [p.detach().clone() for p in model.parameters()]
    info = train_step(model, batch, opt, cpu, accum_grad=2, step_index=0)
    assert not info["updated"] and info["grad_norm"] is None
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    g1 = [p.grad.clone() for p in model.parameters()]
    info = train_step(model, batch, opt, cpu, accum_grad=2, step_index=1, grad_clip=1e9)
    assert info["updated"]
    # two identical half-weighted batches accumulate to the gradient of one: the first call left exactly half of it
    model2 = _TinyCtc()
    model2.load_state_dict({k: v for k, v in zip(model.state_dict().keys(), before)})
    model2(batch, cpu)["loss"].backward()
    for h, p in zip(g1, model2.parameters()):
        assert torch.allclose(2 * h, p.grad, rtol=1e-5, atol=1e-7)


def test_train_step_bf16_autocast_on_cpu():
    """`dtype: bf16` (train_utils.py:614-626): the forward runs under autocast, master weights and grads stay fp32."""
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    torch.manual_seed(0)
    model, batch, cpu = _TinyCtc(), _tiny_batch(), torch.device("cpu")
    ref = _TinyCtc()
    ref.load_state_dict(model.state_dict())
    ref(batch, cpu)["loss"].backward()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    grads = {}
    hooks = [p.register_hook(lambda g, n=n: grads.__setitem__(n, g.clone())) for n, p in model.named_parameters()]
    info = train_step(model, batch, opt, cpu, grad_clip=1e9, amp_dtype=torch.bfloat16)
    for h in hooks:
        h.remove()
    assert model.seen_dtype == torch.bfloat16 and info["updated"]
    for n, p in ref.named_parameters():
        assert grads[n].dtype == torch.float32
        assert float((grads[n] - p.grad).abs().max()) <= 0.05 * float(p.grad.abs().max()) + 1e-4, n


@pytest.mark.gpu
def test_train_step_amp_scaler_semantics(hip):
    """`--use_amp` (train_utils.py:635,655-657,698-709): fp16 autocast, scaled backward, unscale -> clip -> scaler.step;
    an overflowing batch is skipped by the scaler and halves the scale."""
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    torch.manual_seed(0)
    dev = torch.device("cuda")
    model = _TinyCtc().to(dev)
    batch = {k: v.to(dev) for k, v in _tiny_batch().items()}
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    before = [p.detach().clone() for p in model.parameters()]
    info = train_step(model, batch, opt, dev, scaler=scaler, grad_clip=0.1)
    assert model.seen_dtype == torch.float16 and info["updated"] and torch.isfinite(info["grad_norm"])
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    before = [p.detach().clone() for p in model.parameters()]
    bad = dict(batch, feats=batch["feats"] * float("inf"))
    info = train_step(model, bad, opt, dev, scaler=scaler, grad_clip=0.1)
    assert not info["updated"] and scaler.get_scale() == 512.0
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))


@pytest.mark.gpu
def test_train_step_bf16_autocast_through_the_encoder(hip):
    """The whole model under bf16 autocast (fp32 master weights, bf16 time-mix slot): loss close to the fp32 step's,
    fp32 gradients for the fp32 parameters, every parameter reached, an update made."""
    from paper_accurate_fast_cheap_amd.utils.init_model import init_model
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    for golden in ("encoder_reduced_bf16slot", "encoder_reduced_f32"):
        g = load_golden(golden)
        cfg = dict(encoder="conformer", encoder_conf=dict(g["conf"], dropout_rate=0.0, positional_dropout_rate=0.0),
                   input_dim=80, output_dim=50, ctc="ctc", ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})

        class A:
            checkpoint = None
        torch.manual_seed(3)
        model, _ = init_model(A(), cfg)
        model = model.cuda()
        batch = {"feats": synth.randn((4, 120, 80), 1, 2.0), "feats_lengths": torch.tensor([120, 100, 90, 64]),
                 "target": torch.randint(1, 50, (4, 6), generator=torch.Generator().manual_seed(2)),
                 "target_lengths": torch.tensor([6, 5, 4, 3])}
        dev = torch.device("cuda")
        model.train()
        ref = model(batch, dev)["loss"]
        ref.backward()
        ref_loss = float(ref.detach())
        ref_grads = {n: p.grad.float().clone() for n, p in model.named_parameters()}
        model.zero_grad(set_to_none=True)
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        grads = {}
        hooks = [p.register_hook(lambda gr, n=n: grads.__setitem__(n, gr)) for n, p in model.named_parameters()]
        info = train_step(model, batch, opt, dev, grad_clip=0.1, amp_dtype=torch.bfloat16)
        for h in hooks:
            h.remove()
        assert info["updated"] and torch.isfinite(info["grad_norm"])
        assert float(info["loss"]) == pytest.approx(ref_loss, rel=3e-2), golden
        for n, p in model.named_parameters():
            assert n in grads, n
            assert grads[n].dtype == p.dtype and torch.isfinite(grads[n]).all(), n
            a, b = ref_grads[n].flatten().double(), grads[n].flatten().double()
            if float(a.norm()) > 1e-6 * max(1.0, a.numel() ** 0.5):      # LoRA outputs start with exactly zero gradient
                assert float(a @ b / (a.norm() * b.norm())) > 0.97, (golden, n)
                assert 0.8 < float(b.norm() / a.norm()) < 1.25, (golden, n)


@pytest.mark.gpu
def test_conv_module_under_fp16_autocast(hip):
    """`--use_amp` is fp16 autocast (train_utils.py:635): the conv module must run (the depthwise kernels in fp32, the
    result rounded back to fp16) and hand fp32 gradients to its fp32 parameters."""
    from paper_accurate_fast_cheap_amd.transformer.convolution import ConvolutionModule
    torch.manual_seed(0)
    m = ConvolutionModule(128, 15, torch.nn.SiLU(), "layer_norm", causal=False, bias=True).cuda()
    x = synth.randn((2, 40, 128), 3).cuda().requires_grad_()
    with torch.autocast("cuda", dtype=torch.float16):
        y, _ = m(x)
    y.float().square().sum().backward()
    ref, _ = m(x.detach())
    assert torch.isfinite(y).all() and float((y.float() - ref.float()).abs().max()) < 0.05 * float(ref.abs().max()) + 1e-2
    assert all(p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("golden", ["encoder_reduced_f32"])
def test_train_step_fp16_amp_through_the_encoder(hip, golden):
    """The reference's `--use_amp` (fp16 autocast + GradScaler, train_utils.py:635-709) end to end: the step runs, the
    loss is close to the fp32 loss, the update is made (or skipped by the scaler on overflow, never half-applied).
    fp32 slot only: GradScaler cannot un-scale the bf16 gradients of a bf16 slot (a framework limit the reference shares);
    the mixed-precision mode of the bf16 slot is bf16 autocast."""
    from paper_accurate_fast_cheap_amd.utils.init_model import init_model
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    g = load_golden(golden)
    cfg = dict(encoder="conformer", encoder_conf=dict(g["conf"], dropout_rate=0.0, positional_dropout_rate=0.0),
               input_dim=80, output_dim=50, ctc="ctc", ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})

    class A:
        checkpoint = None
    torch.manual_seed(3)
    model, _ = init_model(A(), cfg)
    model = model.cuda()
    batch = {"feats": synth.randn((4, 120, 80), 1, 2.0), "feats_lengths": torch.tensor([120, 100, 90, 64]),
             "target": torch.randint(1, 50, (4, 6), generator=torch.Generator().manual_seed(2)),
             "target_lengths": torch.tensor([6, 5, 4, 3])}
    dev = torch.device("cuda")
    model.train()
    ref_loss = float(model(batch, dev)["loss"].detach())
    model.zero_grad(set_to_none=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=256.0)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    info = train_step(model, batch, opt, dev, grad_clip=0.1, scaler=scaler)
    assert float(info["loss"]) == pytest.approx(ref_loss, rel=3e-2)
    changed = any(not torch.equal(before[n], p.detach()) for n, p in model.named_parameters())
    assert changed == bool(info["updated"])
    assert all(torch.isfinite(p).all() for p in model.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("C,T,B", [(128, 67, 2), (512, 203, 3)])
def test_subsampling_training_path_equals_framework_autograd(hip, monkeypatch, C, T, B):
    """Conv2dSubsampling4 under bf16 autocast + autograd: the NHWC kernel path (conv1 forward + weight-gradient kernels,
    conv2 forward kernel + library backward on channels_last views, permuted `out` weight) gives the outputs and the
    parameter gradients of the module chain through the library (PAFC_TRAIN_KERNELS=0)."""
    from paper_accurate_fast_cheap_amd.transformer.embedding import RelPositionalEncoding
    from paper_accurate_fast_cheap_amd.transformer.subsampling import Conv2dSubsampling4
    torch.manual_seed(8)
    sub = Conv2dSubsampling4(80, C, 0.0, RelPositionalEncoding(C, 0.0)).cuda().train()
    x = synth.randn((B, T, 80), 71, 1.5).cuda()
    mask = torch.ones(B, 1, T, dtype=torch.bool, device="cuda")
    Tp = ((T - 1) // 2 - 1) // 2
    gy = synth.randn((B, Tp, C), 72, 1.0).cuda()

    def run(amp=True):
        sub.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            y, _, m = sub(x, mask)
        (y.float() * gy).sum().backward()
        return y.detach().float(), {n: p.grad.float().clone() for n, p in sub.named_parameters()}
    y_k, g_k = run()
    monkeypatch.setenv("PAFC_TRAIN_KERNELS", "0")
    y_f, g_f = run()
    y_r, g_r = run(amp=False)                       # fp32 module chain: the yardstick for both bf16 paths
    assert y_k.shape == y_f.shape == (B, Tp, C)
    assert float((y_k - y_r).abs().max()) <= 3e-2 * max(1.0, float(y_r.abs().max()))
    assert set(g_k) == set(g_r)
    for n in g_r:
        s = max(float(g_r[n].abs().max()), 1e-6)
        assert g_k[n].shape == g_r[n].shape
        err_k, err_f = float((g_k[n] - g_r[n]).abs().max()), float((g_f[n] - g_r[n]).abs().max())
        # within bf16 noise of the fp32 gradient, and no further from it than the library's own bf16 path plus slack
        assert err_k <= 0.2 * s, (n, err_k, s)           # both bf16 paths sit 6-12 % (max-norm) from the fp32 gradient here
        assert err_k <= 1.5 * err_f + 1e-2 * s, (n, err_k, err_f, s)


def _mixed_grads(device, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [((7, 5), torch.float32), ((33,), torch.bfloat16), ((4, 4, 4), torch.bfloat16), ((3,), torch.float32),
              ((129, 64), torch.bfloat16), ((1, 1, 64), torch.bfloat16)]
    return [(torch.randn(s, generator=g) * 3).to(dt).to(device) for s, dt in shapes]


def _check_clip_equals_torch(device):
    from paper_accurate_fast_cheap_amd.utils.train_utils import clip_grad_norm_
    for max_norm in (0.1, 7.0, 1e4):
        grads = _mixed_grads(device, 5)
        ours = [torch.nn.Parameter(torch.zeros_like(g)) for g in grads]
        theirs = [torch.nn.Parameter(torch.zeros_like(g)) for g in grads]
        for p, q, g in zip(ours, theirs, grads):
            p.grad, q.grad = g.clone(), g.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(theirs, max_norm)
        n = clip_grad_norm_(ours, max_norm)
        assert torch.equal(n.cpu(), n_ref.cpu())
        for p, q in zip(ours, theirs):
            assert torch.equal(p.grad, q.grad), (max_norm, p.dtype, p.shape)


def test_clip_grad_norm_equals_torch_cpu():
    """train_utils.clip_grad_norm_ (one multi-tensor launch per dtype group) is torch.nn.utils.clip_grad_norm_ bit for bit on
    fp32 + bf16 gradients, clipping or not."""
    _check_clip_equals_torch(torch.device("cpu"))


@pytest.mark.gpu
def test_clip_grad_norm_equals_torch_gpu():
    _check_clip_equals_torch(torch.device("cuda"))


@pytest.mark.gpu
def test_bf16_weight_shadows_follow_the_parameter():
    """The bf16 copies the training projections use instead of a cast per step: only inside train_shadows(), refreshed on
    entry whatever changed the parameter (a fused optimizer does not touch Tensor._version); outside, a fresh cast."""
    from paper_accurate_fast_cheap_amd import hip_ops
    w = torch.nn.Parameter(torch.randn(64, 64, device="cuda"))
    w.grad = torch.randn_like(w)
    opt = torch.optim.Adam([w], lr=0.1, fused=True)
    a = hip_ops._bf16_shadow(w)
    assert a.dtype == torch.bfloat16 and torch.equal(a, w.detach().bfloat16()) and id(w) not in hip_ops._shadows
    with hip_ops.train_shadows():
        s0 = hip_ops._bf16_shadow(w)
        assert torch.equal(s0, w.detach().bfloat16()) and hip_ops._bf16_shadow(w) is s0     # one cast, then the copy
    opt.step()                                                                               # in place, version untouched
    assert not torch.equal(s0, w.detach().bfloat16())
    with hip_ops.train_shadows():
        assert hip_ops._bf16_shadow(w) is s0 and torch.equal(s0, w.detach().bfloat16())      # refreshed on entry
    assert hip_ops._bf16_shadow(w) is not s0                                                 # outside: a fresh cast
    b = torch.nn.Parameter(torch.randn(64, device="cuda").bfloat16())
    with hip_ops.train_shadows():
        assert hip_ops._bf16_shadow(b) is b                                                  # bf16 parameters as they are


@pytest.mark.gpu
def test_transposed_weight_shadows_and_the_training_linear_on_own_gemms(hip, monkeypatch):
    """The training Linear's forward and input gradient on the hand-written GEMMs: dX = dY W runs against bf16 copies of W^T kept
    beside the parameters -- made on first use, ALL refreshed by one launch of pafc_multi_transpose_bf16 when train_shadows() is
    entered (fp32 and bf16 parameters, shapes that are not multiples of the 64 x 64 tile, a 1 x 1 convolution's weight seen
    through `squeeze(-1)`), re-made on the spot when their stamp is stale.  Gradients equal the library path's to bf16 round-off."""
    from paper_accurate_fast_cheap_amd import hip_ops
    torch.manual_seed(3)
    lin = torch.nn.Linear(512, 2048).cuda()
    odd = torch.nn.Linear(200, 72, bias=False).cuda()
    slot = torch.nn.Linear(512, 512, bias=False).cuda().to(torch.bfloat16)
    conv = torch.nn.Conv1d(512, 1024, 1).cuda()
    mats = [lin.weight, odd.weight, slot.weight, conv.weight.squeeze(-1)]
    for m in mats:                                                     # first use: made from the parameter
        t = hip_ops._bf16_shadow_t(m)
        assert t.dtype == torch.bfloat16 and torch.equal(t, m.detach().to(torch.bfloat16).t())
    with torch.no_grad():                                              # a fused-optimizer style update: versions untouched
        for m in (lin.weight, odd.weight, conv.weight):
            m.data.mul_(-1.5)
        slot.weight.data.mul_(-0.5)
    hip_ops.bump_param_epoch()                                         # (train_step does this after every update)
    with hip_ops.train_shadows():                                      # entry: ONE launch refreshes all four copies
        for m in mats:
            p = hip_ops._param_of(m)
            ent = hip_ops._shadows_t[id(p)]
            assert torch.equal(ent[1], m.detach().to(torch.bfloat16).t()), tuple(m.shape)
            assert hip_ops._bf16_shadow_t(m) is ent[1]
    with torch.no_grad():
        lin.weight.add_(1.0)                                           # in place outside train_step: the stamp is stale ...
    assert torch.equal(hip_ops._bf16_shadow_t(lin.weight), lin.weight.detach().to(torch.bfloat16).t())   # ... re-made on use

    def grads(own):
        monkeypatch.setenv("PAFC_TRAIN_OWN_GEMMS", "1" if own else "0")
        x = (torch.randn(3, 700, 512, device="cuda") * 0.5).requires_grad_()
        torch.manual_seed(11)
        for m in (lin, conv):
            m.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = hip_ops.linear(x, lin.weight, lin.bias)
            z = hip_ops.linear(x, conv.weight.squeeze(-1), conv.bias)
        g1, g2 = torch.randn_like(y), torch.randn_like(z)
        (y * g1).sum().backward(retain_graph=True)
        (z * g2).sum().backward()
        return y.detach().float(), z.detach().float(), x.grad.clone(), lin.weight.grad.clone(), conv.weight.grad.clone()
    torch.manual_seed(5)
    a = grads(True)
    torch.manual_seed(5)
    b = grads(False)
    for u, v, name in zip(a, b, ("y", "z", "dx", "dW lin", "dW conv")):
        scale = float(v.abs().max())
        assert float((u - v).abs().max()) <= 2 ** -6 * scale, (name, float((u - v).abs().max()), scale)


@pytest.mark.gpu
def test_training_linear_in_a_manual_loop_never_reads_a_stale_transposed_copy(hip):
    """A training loop that does NOT go through train_step: forward + backward outside train_shadows(), the weights updated
    through `.data` (Tensor._version untouched, as fused optimizers do) and no parameter-epoch bump.  The input gradient must be
    dY W for the weights the forward pass used -- the kept W^T copies are only trusted inside train_shadows(), whose entry
    refreshes them; outside it the backward pass transposes the forward's own bf16 copy.  Same for the (K, N) LoRA matrices
    of matmul_param, whose FORWARD reads the transposed copy."""
    from paper_accurate_fast_cheap_amd import hip_ops
    torch.manual_seed(9)
    lin = torch.nn.Linear(512, 512).cuda()
    lora = torch.nn.Parameter((torch.randn(512, 128, device="cuda") * 0.05).bfloat16())
    x0 = torch.randn(2, 300, 512, device="cuda") * 0.5
    dy = torch.randn(2, 300, 512, device="cuda").bfloat16()
    with hip_ops.train_shadows():                                     # a train_step-style pass first: registers kept W^T copies
        x = x0.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = hip_ops.linear(x, lin.weight, lin.bias)
        xb = x0.bfloat16().requires_grad_()
        z = hip_ops.matmul_param(xb, lora)
    y.backward(dy)
    z.sum().backward()
    assert id(lin.weight) in hip_ops._shadows_t and id(lora) in hip_ops._shadows_t
    with torch.no_grad():
        lin.weight.data.mul_(-2.0)                                    # behind torch's back: no version change, no epoch bump
        lora.data.mul_(-3.0)
    for _ in range(2):                                                # the manual loop, outside train_shadows()
        x = x0.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = hip_ops.linear(x, lin.weight, lin.bias)
        y.backward(dy)
        wb = lin.weight.detach().bfloat16()
        want_dx = (dy.reshape(-1, 512).float() @ wb.float()).view_as(x0)
        assert float((x.grad - want_dx).abs().max()) <= 2 ** -6 * float(want_dx.abs().max())
        xb = x0.bfloat16().requires_grad_()
        z = hip_ops.matmul_param(xb, lora)
        want_z = xb.detach().float() @ lora.detach().float()
        assert float((z.detach().float() - want_z).abs().max()) <= 2 ** -6 * float(want_z.abs().max())


@pytest.mark.gpu
def test_c4_full_size_training_step(hip):
    """Config c4 at FULL model size (12 layers, 512 d, 8 x 64 heads, CTC over 5000 tokens; 32 ragged utterances): one
    training step under bf16 autocast through the training kernels -- finite loss and gradients on every parameter, the
    parameters move, and the same step taken twice from the same state gives the same loss (dropout off) and the same update
    up to the reductions that are not bit-reproducible."""
    import copy
    import bench as B
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    device = torch.device("cuda", 0)
    model, _ = B.build_model("fp32", device)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout_rate"):
            m.dropout_rate = 0.0
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(100, 601, (32,), generator=g)
    fb = torch.zeros(32, int(lens.max()), 80, device=device)
    for j, n in enumerate(lens.tolist()):
        fb[j, :n] = torch.randn(n, 80, generator=g).to(device)
    tl = torch.minimum(torch.randint(1, 41, (32,), generator=g), ((lens - 1) // 2 - 1) // 2 // 2).clamp(min=1)
    tgt = torch.randint(1, B.VOCAB, (32, int(tl.max())), generator=g)
    batch = {"feats": fb, "feats_lengths": lens.to(device), "target": tgt.to(device), "target_lengths": tl.to(device)}
    state = copy.deepcopy(model.state_dict())
    runs = []
    for _ in range(2):
        model.load_state_dict(state)
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
        # keep the gradients for inspection: step with a huge clip threshold and read them before zero_grad via a hook
        grads = {}
        hooks = [p.register_hook(lambda gr, n=n: grads.__setitem__(n, gr.detach().float().norm().item()))
                 for n, p in model.named_parameters() if p.requires_grad]
        info = train_step(model, batch, opt, device, amp_dtype=torch.bfloat16, step_index=0)
        for h_ in hooks:
            h_.remove()
        runs.append((float(info["loss"]), float(info["grad_norm"]), grads,
                     {n: p.detach().clone() for n, p in model.named_parameters()}))
        assert info["updated"]
    loss, gn, grads, params = runs[0]
    assert loss == loss and abs(loss) < 1e6 and gn == gn and gn > 0
    assert len(grads) == sum(1 for p in model.parameters() if p.requires_grad)
    assert all(v == v and v < 1e9 for v in grads.values())
    moved = sum(int(not torch.equal(params[n], state[n])) for n in params)
    assert moved >= 0.7 * len(params)      # (bf16 slot parameters of magnitude ~1 do not see an Adam step of 1e-4: < 1 ulp)
    assert abs(runs[1][0] - loss) <= 1e-3 * abs(loss)
    assert abs(runs[1][1] - gn) <= 2e-2 * gn


@pytest.mark.gpu
@pytest.mark.parametrize("V,C", [(5000, 512), (40, 128)])
def test_ctc_head_loss_kernels_equal_torch_ctc_loss(hip, V, C):
    """hip_ops.ctc_head_loss (csrc/ctc_loss.hip: head GEMM + loss from the logits + gradient through the log-softmax, no (B, T, V)
    log-probabilities, no host round trip) against the reference's own sequence -- Linear, log_softmax, torch.nn.CTCLoss(sum,
    zero_infinity) / B, ctc.py:53-82 -- in fp32 on the same bf16-valued operands: ragged input lengths, targets with repeated
    labels (which need a blank between them), an empty target, a target that exactly fills its frames, and an utterance with NO
    alignment (more labels than frames: loss 0 and zero gradient under zero_infinity).  Loss to 1e-3 relative, gradients wrt the
    encoder output, the weight and the bias to bf16 round-off of the dense gradient."""
    from paper_accurate_fast_cheap_amd import hip_ops
    torch.manual_seed(5)
    B, T = 6, 57
    x = (torch.randn(B, T, C, device="cuda") * 0.7).bfloat16().float().requires_grad_()
    lin = torch.nn.Linear(C, V).cuda()
    with torch.no_grad():
        lin.weight.copy_(lin.weight.bfloat16().float() * 3)
        lin.bias.copy_(lin.bias.bfloat16().float())
    hlens = torch.tensor([57, 40, 33, 12, 5, 9], device="cuda")
    ylens = torch.tensor([20, 13, 0, 12, 7, 4], device="cuda")            # #3 fills its frames exactly, #4 has no alignment
    ys = torch.full((B, 20), -1, dtype=torch.long, device="cuda")
    g = torch.Generator().manual_seed(3)
    for b, n in enumerate(ylens.tolist()):
        if n:
            ys[b, :n] = torch.randint(1, V, (n,), generator=g)
    ys[0, 3] = ys[0, 2]                                                    # repeated labels
    ys[1, 1] = ys[1, 0]
    ys[3, :12] = torch.arange(1, 13)                                       # 12 distinct labels in 12 frames: one alignment
    # reference: the module's own steps in fp32
    logits = torch.nn.functional.linear(x, lin.weight, lin.bias)
    lp = logits.transpose(0, 1).log_softmax(2)
    want = torch.nn.functional.ctc_loss(lp, ys.clamp_min(0), hlens, ylens, blank=0, reduction="sum", zero_infinity=True) / B
    want.backward()
    gx, gw, gb = x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()
    x.grad = None
    lin.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert hip_ops.ctc_head_loss_eligible(x, lin.weight, ys)
        got = hip_ops.ctc_head_loss(x, lin.weight, lin.bias, hlens, ys, ylens, 0)
    (got * 1.0).backward()
    assert torch.isfinite(got) and abs(float(got) - float(want)) <= 2e-3 * abs(float(want)) + 1e-3, (float(got), float(want))
    for a, b_, name in ((x.grad, gx, "dx"), (lin.weight.grad, gw, "dW"), (lin.bias.grad, gb, "db")):
        scale = float(b_.abs().max())
        assert float((a - b_).abs().max()) <= 2 ** -6 * scale + 1e-6, (name, float((a - b_).abs().max()), scale)
    assert float(x.grad[4].abs().max()) == 0.0 and float(gx[4].abs().max()) == 0.0      # no alignment: zero gradient
    assert float(x.grad[1, 40:].abs().max()) == 0.0                                       # frames beyond the utterance
    # the module entry point takes this path in the training step and the framework's otherwise
    from paper_accurate_fast_cheap_amd.transformer.ctc import CTC
    head = CTC(V, C).cuda()
    head.ctc_lo.load_state_dict(lin.state_dict())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l1 = head.loss(x, hlens, ys, ylens)
    l2 = head(x, hlens, ys.clamp_min(0), ylens)[0]
    assert abs(float(l1) - float(got)) <= 1e-6 and abs(float(l2) - float(want)) <= 1e-5 * abs(float(want))
