"""RNN-T loss (PARITY UNPINNED: the reference's implementation is a third-party package absent from the tree): the
restatement against brute-force enumeration of every alignment, its gradient against finite differences, the joint's
concatenated layout against the dense joint, and the Transducer training forward end to end."""
import itertools
import math

import pytest
import torch

from paper_accurate_fast_cheap_amd.transducer.joint import TransducerJoint
from paper_accurate_fast_cheap_amd.transducer.loss import transducer_loss


def _brute(logp, y, blank):
    """-log sum over all monotone alignments of T blanks and U labels (blank moves t, label moves u)."""
    T, U1, _ = logp.shape
    U = U1 - 1
    total = []
    for pos in itertools.combinations(range(T + U), U):     # positions of the label emissions in the path
        t = u = 0
        s = 0.0
        ok = True
        for step in range(T + U):
            if step in pos:
                if t >= T:
                    ok = False; break
                s += float(logp[t, u, y[u]]); u += 1
            else:
                s += float(logp[t, u, blank]); t += 1
        if ok and t == T and u == U:
            # the path must END with a blank from (T-1, U): enforced because t reaches T only by a blank at t = T-1
            total.append(s)
    m = max(total)
    return -(m + math.log(sum(math.exp(v - m) for v in total)))


@pytest.mark.parametrize("T,U,V", [(1, 0, 3), (3, 2, 4), (4, 3, 5), (5, 1, 3), (2, 4, 6)])
def test_loss_equals_brute_force_enumeration(T, U, V):
    g = torch.Generator().manual_seed(T * 10 + U)
    logits = torch.randn(T * (U + 1), V, generator=g, dtype=torch.float64)
    y = torch.randint(1, V, (1, max(U, 1)), generator=g)
    got = transducer_loss(logits, y, torch.tensor([T]), torch.tensor([U]), blank=0, reduction="none")
    # label emissions at t = T are impossible in the lattice; the enumeration above only counts paths inside it
    want = _brute(logits.log_softmax(-1).view(T, U + 1, V), y[0].tolist(), 0)
    assert float(got[0]) == pytest.approx(want, rel=1e-10)


def test_batched_layout_reductions_and_gradient():
    g = torch.Generator().manual_seed(5)
    Ts, Us, V = [4, 2, 3], [2, 0, 3], 5
    n = sum(t * (u + 1) for t, u in zip(Ts, Us))
    logits = torch.randn(n, V, generator=g, dtype=torch.float64, requires_grad=True)
    y = torch.randint(1, V, (3, 3), generator=g)
    TL, UL = torch.tensor(Ts), torch.tensor(Us)
    none = transducer_loss(logits, y, TL, UL, 0, reduction="none")
    off, each = 0, []
    for i, (t, u) in enumerate(zip(Ts, Us)):
        each.append(transducer_loss(logits[off:off + t * (u + 1)], y[i:i + 1], TL[i:i + 1], UL[i:i + 1], 0, "none")[0])
        off += t * (u + 1)
    assert torch.allclose(none, torch.stack(each))
    assert float(transducer_loss(logits, y, TL, UL, 0, "sum").detach()) == pytest.approx(float(none.detach().sum()))
    assert float(transducer_loss(logits, y, TL, UL, 0, "mean").detach()) == pytest.approx(float(none.detach().sum()) / sum(Ts))
    assert torch.autograd.gradcheck(lambda z: transducer_loss(z, y, TL, UL, 0, "sum"), (logits,), eps=1e-6, atol=1e-6)
    # from_log_softmax: same value when the caller normalises
    lsm = logits.detach().log_softmax(-1)
    assert torch.allclose(transducer_loss(lsm, y, TL, UL, 0, "none", from_log_softmax=True), none.detach())
    with pytest.raises(ValueError):
        transducer_loss(logits[:-1], y, TL, UL, 0)


def test_joint_forward_optimized_is_the_valid_part_of_the_dense_joint():
    torch.manual_seed(1)
    j = TransducerJoint(11, enc_output_size=6, pred_output_size=5, join_dim=7).eval()
    enc, pred = torch.randn(3, 8, 6), torch.randn(3, 5, 5)
    el, pl = torch.tensor([8, 5, 3]), torch.tensor([4, 2, 0])
    dense = j(enc, pred)                                             # (3, 8, 5, 11)
    flat = j.forward_optimized(enc, pred, el, pl)
    want = torch.cat([dense[i, :int(el[i]), :int(pl[i]) + 1].reshape(-1, 11) for i in range(3)])
    torch.testing.assert_close(flat, want)


def test_transducer_training_forward_backward_cpu():
    """Transducer.forward: encoder -> predictor -> joint -> RNN-T loss (+ CTC), gradients reach every sub-module.
    (The recurrent slot's kernels are GPU-only; a linear front end and zero Conformer blocks keep this on the CPU.)"""
    from paper_accurate_fast_cheap_amd.transducer.predictor import RNNPredictor
    from paper_accurate_fast_cheap_amd.transducer.transducer import Transducer, add_blank
    from paper_accurate_fast_cheap_amd.transformer.ctc import CTC

    class TinyEncoder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = torch.nn.Linear(8, 16)

        def output_size(self):
            return 16

        def forward(self, x, lens, *a, **k):
            mask = (torch.arange(x.shape[1])[None, :] < lens[:, None]).unsqueeze(1)
            return self.proj(x), mask

    torch.manual_seed(0)
    V = 9
    model = Transducer(V, 0, TinyEncoder(), RNNPredictor(V, 8, 12, 0.0, 12, 1, dropout=0.0),
                       TransducerJoint(V, 16, 12, 10), ctc=CTC(V, 16), ctc_weight=0.3, transducer_weight=0.7)
    batch = {"feats": torch.randn(2, 7, 8), "feats_lengths": torch.tensor([7, 5]),
             "target": torch.tensor([[3, 4, 2], [5, -1, -1]]), "target_lengths": torch.tensor([3, 1])}
    assert add_blank(batch["target"], 0, -1).tolist() == [[0, 3, 4, 2], [0, 5, 0, 0]]
    out = model(batch, torch.device("cpu"))
    assert torch.isfinite(out["loss"]) and out["loss_rnnt"] > 0 and out["loss_ctc"] is not None
    assert float(out["loss"]) == pytest.approx(0.7 * float(out["loss_rnnt"]) + 0.3 * float(out["loss_ctc"].sum()), rel=1e-6)
    out["loss"].backward()
    for name, prm in model.named_parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [False, True])
def test_transducer_training_step_on_gpu(hip, amp):
    """The hybrid objective (transducer_weight RNN-T + ctc_weight CTC, transducer.py:70-148) over the accelerated
    encoder, one DDP-style step on the GPU, in fp32 and under bf16 autocast: finite loss, every parameter reached, the
    update made."""
    from tests.conftest import load_golden
    from paper_accurate_fast_cheap_amd.transducer.predictor import RNNPredictor
    from paper_accurate_fast_cheap_amd.transducer.transducer import Transducer
    from paper_accurate_fast_cheap_amd.transformer.ctc import CTC
    from paper_accurate_fast_cheap_amd.transformer.encoder import ConformerEncoder
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step
    g = load_golden("encoder_reduced_bf16slot")
    conf = dict(g["conf"], dropout_rate=0.0, positional_dropout_rate=0.0)
    torch.manual_seed(1)
    enc = ConformerEncoder(80, **conf)
    V, D = 30, enc.output_size()
    model = Transducer(V, 0, enc, RNNPredictor(V, 32, 48, 0.0, 48, 1, dropout=0.0), TransducerJoint(V, D, 48, 64),
                       ctc=CTC(V, D), ctc_weight=0.3, transducer_weight=0.7).cuda()
    gen = torch.Generator().manual_seed(5)
    batch = {"feats": torch.randn(3, 90, 80, generator=gen), "feats_lengths": torch.tensor([90, 71, 50]),
             "target": torch.randint(1, V, (3, 5), generator=gen), "target_lengths": torch.tensor([5, 4, 2])}
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    grads = {}
    hooks = [p.register_hook(lambda gr, n=n: grads.__setitem__(n, gr)) for n, p in model.named_parameters()]
    info = train_step(model, batch, opt, torch.device("cuda"), grad_clip=5.0,
                      amp_dtype=torch.bfloat16 if amp else None)
    for h in hooks:
        h.remove()
    assert torch.isfinite(info["loss"]) and info["updated"] and torch.isfinite(info["grad_norm"])
    missing = [n for n, _ in model.named_parameters() if n not in grads]
    assert not missing, missing
    assert all(torch.isfinite(v).all() for v in grads.values())
