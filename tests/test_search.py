"""Config c5: CTC prefix beam search and the CTC-fused RNN-T prefix beam search.  The product's implementations are
device-agnostic host logic over torch ops, so on CPU tensors they must reproduce the reference's tokens and scores
bit for bit (goldens from the reference modules); the GPU variant runs the same code on the MI355X."""
import pytest
import torch

from oracle import search_oracle as SO
from tests import synth
from tests.conftest import load_golden

torch.set_num_threads(4)


def _build(g, device="cpu"):
    from paper_accurate_fast_cheap_amd.transducer.joint import TransducerJoint
    from paper_accurate_fast_cheap_amd.transducer.predictor import RNNPredictor
    from paper_accurate_fast_cheap_amd.transducer.search.prefix_beam_search import PrefixBeamSearch
    from paper_accurate_fast_cheap_amd.transformer.ctc import CTC
    V, D = 50, 128
    ctc = CTC(V, D).eval()
    ctc.load_state_dict(synth.synth_state_dict(g["ctc_spec"], g["ctc_seed"]))
    pred = RNNPredictor(V, embed_size=64, output_size=64, embed_dropout=0.1, hidden_size=64, num_layers=2, bias=True,
                        rnn_type="lstm", dropout=0.1).eval()
    pred.load_state_dict(synth.synth_state_dict(g["pred_spec"], g["pred_seed"]))
    joint = TransducerJoint(V, enc_output_size=D, pred_output_size=64, join_dim=64, prejoin_linear=True,
                            postjoin_linear=False, joint_mode="add", activation="tanh").eval()
    joint.load_state_dict(synth.synth_state_dict(g["joint_spec"], g["joint_seed"]))
    ctc, pred, joint = ctc.to(device), pred.to(device), joint.to(device)
    return ctc, pred, joint, PrefixBeamSearch(None, pred, joint, ctc, 0)


def _same(res, gold, score_tol=0.0):
    assert len(res) == len(gold)
    for r, g in zip(res, gold):
        assert list(r.tokens) == g["tokens"]
        assert [list(n) for n in r.nbest] == g["nbest"]
        assert r.score == pytest.approx(g["score"], abs=score_tol) and r.nbest_scores == pytest.approx(g["nbest_scores"], abs=score_tol)


def test_ctc_prefix_beam_search_matches_reference_golden():
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_prefix_beam_search
    g = load_golden("search_c5")
    _same(ctc_prefix_beam_search(g["logp"], g["enc_lens"], 8, None, 0), g["ctc_prefix"], 1e-9)
    ora = SO.ctc_prefix_beam_search(g["logp"], g["enc_lens"], 8, 0)
    assert [o["tokens"] for o in ora] == [c["tokens"] for c in g["ctc_prefix"]]
    assert [o["nbest"] for o in ora] == [c["nbest"] for c in g["ctc_prefix"]]


def test_rnnt_prefix_beam_search_matches_reference_golden_on_cpu():
    g = load_golden("search_c5")
    ctc, pred, joint, bs = _build(g)
    with torch.no_grad():
        logp = ctc.log_softmax(g["enc_out"])
        assert torch.equal(logp, g["logp"])
        jt = joint(g["enc_out"][:, :5], pred(torch.tensor([[0, 3, 7], [0, 9, 9], [0, 1, 2]])))
        torch.testing.assert_close(jt, g["joint_sample"], rtol=1e-6, atol=1e-6)
        res = bs.prefix_beam_search_decode(g["enc_out"], g["enc_lens"], logp, beam_size=8, ctc_weight=0.3,
                                           transducer_weight=0.7)
    _same(res, g["rnnt"], 1e-4)


def test_search_oracle_matches_reference_golden():
    g = load_golden("search_c5")
    sd = {}
    for pre, spec, seed in (("predictor.", g["pred_spec"], g["pred_seed"]), ("joint.", g["joint_spec"], g["joint_seed"])):
        sd.update({pre + k: v for k, v in synth.synth_state_dict(spec, seed).items()})
    res = SO.rnnt_prefix_beam_search_batch(g["enc_out"], g["enc_lens"], g["logp"], sd, 8)
    assert [r["tokens"] for r in res] == [c["tokens"] for c in g["rnnt"]]
    assert [r["nbest"] for r in res] == [c["nbest"] for c in g["rnnt"]]


def test_search_edge_cases():
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_prefix_beam_search
    logp = torch.log_softmax(synth.randn((2, 6, 7), 3), -1)
    r = ctc_prefix_beam_search(logp, torch.tensor([6, 0]), 3, None, 0)
    assert list(r[1].tokens) == [] and len(r[0].nbest) == 3          # empty utterance -> empty hypothesis
    allblank = torch.full((1, 5, 4), -20.0)
    allblank[..., 0] = 0.0
    assert list(ctc_prefix_beam_search(allblank, torch.tensor([5]), 2, None, 0)[0].tokens) == []


@pytest.mark.gpu
def test_rnnt_prefix_beam_search_on_gpu(hip):
    g = load_golden("search_c5")
    ctc, pred, joint, bs = _build(g, "cuda")
    with torch.no_grad():
        enc = g["enc_out"].cuda()
        logp = ctc.log_softmax(enc)
        res = bs.prefix_beam_search_decode(enc, g["enc_lens"].cuda(), logp, beam_size=8, ctc_weight=0.3,
                                           transducer_weight=0.7)
    # GPU GEMMs round differently from the CPU's: demand the same best hypothesis unless the reference's own top-2
    # scores are within 1e-3 of each other
    for r, c in zip(res, g["rnnt"]):
        if list(r.tokens) != c["tokens"]:
            assert abs(c["nbest_scores"][0] - c["nbest_scores"][1]) < 1e-3
        assert r.score == pytest.approx(c["score"], abs=5e-3)


@pytest.mark.gpu
def test_rnnt_prefix_beam_search_resident_matches_host_loop(hip):
    """Device-resident candidate walk (pafc_rnnt_beam_*) vs the host loop on the same GPU tensors: the same n-best
    token lists and scores to float32 round-off, ragged lengths included; and vs the reference's golden."""
    g = load_golden("search_c5")
    ctc, pred, joint, bs = _build(g, "cuda")
    with torch.no_grad():
        enc, lens = g["enc_out"].cuda(), g["enc_lens"].cuda()
        logp = ctc.log_softmax(enc)
        for beam in (8, 3, 1):
            kw = dict(beam_size=beam, ctc_weight=0.3, transducer_weight=0.7)
            bs.device_resident = True
            res = bs.prefix_beam_search_decode(enc, lens, logp, **kw)
            bs.device_resident = False
            host = bs.prefix_beam_search_decode(enc, lens, logp, **kw)
            for r, h in zip(res, host):
                assert len(r.nbest) == len(h.nbest)
                if [list(n) for n in r.nbest] == [list(n) for n in h.nbest]:
                    assert r.nbest_scores == pytest.approx(h.nbest_scores, abs=2e-3)
                else:
                    # the two paths batch the predictor / joint GEMMs differently (B x beam slots vs live beams only): a
                    # different order is only acceptable between hypotheses the host loop itself scores within 1e-3
                    gaps = [abs(a - b) for i, a in enumerate(h.nbest_scores) for b in h.nbest_scores[i + 1:]]
                    assert gaps and min(gaps) < 1e-3
        bs.device_resident = True
        res = bs.prefix_beam_search_decode(enc, lens, logp, beam_size=8, ctc_weight=0.3, transducer_weight=0.7)
    for r, c in zip(res, g["rnnt"]):
        if list(r.tokens) != c["tokens"]:
            assert abs(c["nbest_scores"][0] - c["nbest_scores"][1]) < 1e-3
        assert r.score == pytest.approx(c["score"], abs=5e-3)
