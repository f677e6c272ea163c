"""Config c5: decode tail on the GPU -- CTC greedy, CTC prefix beam search and the CTC-fused RNN-T prefix
beam search, device-resident vs host bookkeeping, on a DEV-shaped batch (B = 8, T' ~ 250, V = 5000, LSTM 2 x 640,
joint 640, beam 8, weights 0.3 / 0.7).  Random weights and peaky synthetic posteriors; prints one JSON line."""
import json, time
import torch
from paper_accurate_fast_cheap_amd.transducer.joint import TransducerJoint
from paper_accurate_fast_cheap_amd.transducer.predictor import RNNPredictor
from paper_accurate_fast_cheap_amd.transducer.search.prefix_beam_search import PrefixBeamSearch
from paper_accurate_fast_cheap_amd.transformer.ctc import CTC
from paper_accurate_fast_cheap_amd.transformer import search as S

torch.manual_seed(777)
dev = "cuda"
B, T, D, V, beam = 8, 250, 512, 5000, 8
ctc = CTC(V, D).eval().to(dev)
pred = RNNPredictor(V, embed_size=640, output_size=640, embed_dropout=0.1, hidden_size=640, num_layers=2).eval().to(dev)
joint = TransducerJoint(V, enc_output_size=D, pred_output_size=640, join_dim=640).eval().to(dev)
bs = PrefixBeamSearch(None, pred, joint, ctc, 0)
enc = torch.randn(B, T, D, device=dev) * 3
lens = torch.tensor([250, 240, 231, 200, 180, 150, 120, 100], device=dev)
audio_s = float(lens.sum()) * 0.04


def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n, r


out = {"workload": f"c5: B={B}, T'<={T}, V={V}, beam {beam}, predictor LSTM 2x640, joint 640; {audio_s:.0f} s of audio"}
with torch.no_grad():
    logp = ctc.log_softmax(enc)
    dt, g = timed(lambda: S.ctc_greedy_search(logp, lens, 0), 10)
    out["ctc_greedy_ms"] = round(dt * 1e3, 2)
    dt_h, _ = timed(lambda: S.ctc_greedy_search(logp.cpu(), lens.cpu(), 0), 3)
    out["ctc_greedy_host_collapse_ms"] = round(dt_h * 1e3, 2)
    dt, r1 = timed(lambda: S.ctc_prefix_beam_search(logp, lens, beam), 5)
    out["ctc_prefix_beam_resident_ms"] = round(dt * 1e3, 2)
    dt_h, r2 = timed(lambda: S.ctc_prefix_beam_search(logp.cpu(), lens.cpu(), beam), 1)
    out["ctc_prefix_beam_host_ms"] = round(dt_h * 1e3, 2)
    out["ctc_prefix_beam_same_tokens"] = [tuple(a.tokens) for a in r1] == [tuple(b.tokens) for b in r2]
    kw = dict(beam_size=beam, ctc_weight=0.3, transducer_weight=0.7)
    bs.device_resident = True
    dt, r1 = timed(lambda: bs.prefix_beam_search_decode(enc, lens, logp, **kw), 2)
    out["rnnt_prefix_beam_resident_ms"] = round(dt * 1e3, 1)
    bs.device_resident = False
    dt_h, r2 = timed(lambda: bs.prefix_beam_search_decode(enc, lens, logp, **kw), 1)
    out["rnnt_prefix_beam_host_loop_ms"] = round(dt_h * 1e3, 1)
    out["rnnt_same_best"] = sum(list(a.tokens) == list(b.tokens) for a, b in zip(r1, r2))
    out["rnnt_audio_sec_per_sec_resident"] = round(audio_s / dt, 1)
print(json.dumps(out))
