"""Round-4 additions to the goldens captured from the reference's own Python modules (same recipe and rules as
make_goldens.py: run ONCE in the build container where /root/reference is mounted; fixtures hold data only).

    python tests/golden/make_goldens_r4.py

The causal convolution + cnn_cache path (`causal: true`: wenet/transformer/convolution.py:49-60,113-126; threaded by
BaseEncoder.forward_chunk, wenet/transformer/encoder.py:311-337) -- the configuration the streaming leg (BASELINE
configs[2]) runs and that no earlier golden covered.

  conv_module_causal.pt   -- ConvolutionModule(causal=True), kernel 15 and 31, fp32 and bf16:
      a ragged batch with its pad mask and no cache; one stream cut in three pieces with the cache handed on
      (outputs and every new_cache), and the same stream in one piece.
  encoder_causal_uni.pt   -- the reduced ConformerEncoder with selfattention_layer_type rwkv_tmix60 (uni) + causal: true,
      kernel 15, in fp32 / bf16-slot / whole-model bf16:
      forward() of a ragged batch (per-layer outputs, masks, CTC log-probs and greedy tokens), forward() of one long
      utterance (the whole-sequence yardstick of the state-carrying stream), forward_chunk() twice with the cnn_cache of
      the first call handed to the second (outputs and both caches), forward_chunk_by_chunk() at two chunk sizes.
"""
import os
import sys

import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from tests import synth  # noqa: E402
from tests.golden.make_goldens import REDUCED, YAML, load_synth, save  # noqa: E402


def main():
    ref_shim.install()
    torch.set_grad_enabled(False)
    torch.set_num_threads(4)
    from wenet.transformer.convolution import ConvolutionModule
    from wenet.transformer.ctc import CTC
    from wenet.transformer.encoder import ConformerEncoder
    from wenet.transformer.search import ctc_greedy_search

    # ---- the module alone ----------------------------------------------------------------------------------------
    cases = {}
    for k in (15, 31):
        for prec in ("f32", "bf16"):
            m = ConvolutionModule(128, k, torch.nn.SiLU(), "layer_norm", True).eval()
            spec, cs = load_synth(m, 71)
            dt = torch.bfloat16 if prec == "bf16" else torch.float32
            m = m.to(dt)
            xb = synth.randn((3, 57, 128), 72).to(dt)
            lens = torch.tensor([57, 40, 9])
            mask = (torch.arange(57)[None, :] < lens[:, None]).unsqueeze(1)
            yb, cb = m(xb.clone(), mask)                 # (the module's masked_fill_ is in place on a view of its input)
            xs = synth.randn((1, 83, 128), 73).to(dt)
            cuts = [0, 20, 20 + k + 3, 83]
            pieces, cache = [], torch.zeros((0, 0, 0), dtype=dt)
            for a, b in zip(cuts[:-1], cuts[1:]):
                y, cache = m(xs[:, a:b].clone(), torch.ones((0, 0, 0), dtype=torch.bool), cache)
                pieces.append(dict(y=y, new_cache=cache.clone()))
            whole, cw = m(xs.clone())
            cases[f"k{k}_{prec}"] = dict(spec=spec, seed=71, checksum=cs, kernel=k, xb=xb, lens=lens, yb=yb, cb=cb.clone(),
                                        xs=xs, cuts=cuts, pieces=pieces, whole=whole, whole_cache=cw.clone())
            d = (torch.cat([p["y"] for p in pieces], 1).float() - whole.float()).abs().max()
            print("conv causal", k, prec, tuple(yb.shape), tuple(cb.shape), "pieces vs whole", float(d))
    save("conv_module_causal", dict(cases=cases))

    # ---- the reduced uni-directional encoder with the causal conv module ----------------------------------------
    cfg = yaml.safe_load(open(os.path.join(ref_shim.REFERENCE_ROOT, YAML)))
    xs = synth.randn((3, 203, 80), 74, 2.0)
    lens = torch.tensor([203, 150, 67])
    long = synth.randn((1, 4 * 8 * 14 + 3, 80), 75, 2.0)       # 14 chunks of 8 output frames
    enc_cases = {}
    for prec in ("f32", "bf16slot", "bf16model"):
        conf = dict(cfg["encoder_conf"])
        conf.update(REDUCED)
        conf.update(selfattention_layer_type="rwkv_tmix60", rnn_att_direction="uni", causal=True, cnn_module_kernel=15)
        conf["rwkv_do_bfloat16"] = prec != "f32"
        enc = ConformerEncoder(80, **conf).eval()
        spec, cs = load_synth(enc, 76)
        ctc = CTC(50, 128).eval()
        ctc_spec, ctc_cs = load_synth(ctc, 77)
        x, lg = xs, long
        if prec == "bf16model":
            enc, ctc = enc.to(torch.bfloat16), ctc.to(torch.bfloat16)
            x, lg = xs.to(torch.bfloat16), long.to(torch.bfloat16)
        layer_outs = []
        hooks = [l.register_forward_hook(lambda mod, i, o: layer_outs.append(o[0])) for l in enc.encoders]
        out, masks = enc(x, lens)
        for h in hooks:
            h.remove()
        logp = ctc.log_softmax(out)
        enc_lens = masks.squeeze(1).sum(1)
        hyps = [r.tokens for r in ctc_greedy_search(logp.float(), enc_lens, 0)]
        whole, _ = enc(lg, torch.tensor([lg.size(1)]))
        # forward_chunk twice: windows of chunk 8 (35 input frames, stride 32), the second with the first's cnn_cache
        y0, a0, c0 = enc.forward_chunk(lg[:, 0:35], 0, -1)
        y1, a1, c1 = enc.forward_chunk(lg[:, 32:67], 8, -1, a0, c0)
        cbc = {}
        for chunk in (8, 16):
            ys, m = enc.forward_chunk_by_chunk(lg, chunk, -1)
            cbc[chunk] = dict(ys=ys, masks=m)
        enc_cases[prec] = dict(
            spec=spec, seed=76, checksum=cs, conf=conf, ctc_spec=ctc_spec, ctc_seed=77, ctc_checksum=ctc_cs,
            out=out, masks=masks, layer0=layer_outs[0], layer1=layer_outs[1], logp_full=logp.clone(), greedy=hyps,
            enc_lens=enc_lens, whole=whole,
            chunk0=dict(y=y0, att_shape=tuple(a0.shape), cnn=c0.clone()),
            chunk1=dict(y=y1, att_shape=tuple(a1.shape), cnn=c1.clone()), chunks=cbc)
        print(prec, tuple(out.shape), "cnn_cache", tuple(c0.shape), tuple(c1.shape), "whole", tuple(whole.shape),
              "cbc-vs-whole (no state carry in the reference)", float((cbc[8]["ys"].float() - whole.float()).abs().max()))
    save("encoder_causal_uni", dict(xs=xs, lens=lens, long=long, cases=enc_cases))


if __name__ == "__main__":
    main()
