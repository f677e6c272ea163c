"""Where a whole-file window of decode_windows (chunk >= the file: the batcher pads it to the chunk size) spends its time against the
one-sequence pass: batcher, encoder on the padded window, CTC greedy kernel, token fetch.  python tools/micro/decode_tail_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from paper_accurate_fast_cheap_amd import _lib
from paper_accurate_fast_cheap_amd.utils.longform import decode_windows, feats_batcher
from paper_accurate_fast_cheap_amd.hip_ops import ctc_greedy
_lib.lib()
dev = torch.device("cuda", 0)
model, _ = bench.build_model("bf16slot", dev)
feats, _ = bench.front_end(bench.synthetic_waveform(bench.AUDIO_SECONDS, 777), dev)
T = feats.shape[1]
lens = torch.tensor([T], dtype=torch.int32, device=dev)
def sync(): torch.cuda.synchronize()
def t(fn, n=5):
    fn(); sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    sync(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print("encoder + ctc_logprobs (one sequence): %.2f ms" % t(lambda: model.ctc_logprobs(model._forward_encoder(feats, lens)[0])))
    print("decode_windows chunk 200000 bs 1: %.2f ms" % t(lambda: decode_windows(model, feats, 200000, 1)))
    print("feats_batcher only: %.2f ms" % t(lambda: list(feats_batcher(feats, 200000, 1, dev))))
    fb, fl = list(feats_batcher(feats, 200000, 1, dev))[0]
    print("batch shape", tuple(fb.shape), fl)
    enc, mask = model._forward_encoder(fb, fl)
    logp = model.ctc_logprobs(enc)
    print("encoder on the batcher's window: %.2f ms" % t(lambda: model.ctc_logprobs(model._forward_encoder(fb, fl)[0])))
    print("ctc_greedy: %.2f ms" % t(lambda: ctc_greedy(logp.contiguous(), mask.squeeze(1).sum(1), 0, want_frames=True)))
    tk, nt, fr = ctc_greedy(logp.contiguous(), mask.squeeze(1).sum(1), 0, want_frames=True)
    def fetch():
        a, b, c = tk.cpu(), nt.cpu(), fr.cpu()
        n = int(b[0]); return a[0, :n].tolist(), c[0, :n].tolist()
    print("fetch + tolist: %.2f ms" % t(fetch), "tokens", int(nt[0]))
