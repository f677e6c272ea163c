"""Capture golden vectors from the reference's own Python modules.

Run ONCE in the build container (where /root/reference is mounted):
    python tests/golden/make_goldens.py
It imports the reference through oracle/ref_shim.py,
loads deterministic synthetic weights (tests/synth.py) into the reference's
modules, runs them on seeded inputs on CPU and stores {spec, seed, conf,
inputs, outputs} as small .pt files next to this script.  The reference source
never leaves this container: fixtures hold data only.

The WKV op inside the reference modules is oracle/wkv6_oracle.c (the reference
has no CPU implementation of its own), so these goldens pin everything AROUND
the op -- time-mix algebra, wrapper rounding points, flips, padding semantics,
layer wiring, parameter names, subsampling, CTC -- and not the recurrence itself.
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

REDUCED = dict(output_size=128, attention_heads=2, linear_units=256, num_blocks=2)
YAML = "examples/gigaspeech/s0/conf/rwkv/giga.rwkvbi_ds4k31nc_12le.trans.shortform.yaml"


def save(name, obj):
    path = os.path.join(HERE, name + ".pt")
    torch.save(obj, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def load_synth(module, seed):
    spec = synth.spec_of(module.state_dict())
    sd = synth.synth_state_dict(spec, seed)
    module.load_state_dict(sd)
    return spec, synth.checksum(sd)


def main():
    ref_shim.install()
    torch.set_grad_enabled(False)
    torch.set_num_threads(4)

    from wenet.rwkv_v6.rwkv_wrapper import RWKV_TmixWrapper
    from wenet.rwkv_v6.rwkv_wrapper_bidirectional import RWKV_TmixWrapper_bidirectional
    from wenet.rwkv_v6.rwkv_wrapper_bidirectional2 import \
        RWKV_TmixWrapper_bidirectional as RWKV_TmixWrapper_bidirectional2
    from wenet.rwkv_v6.rwkv_wrapper_bidirectional_direction_dropout import \
        RWKV_TmixWrapper_bidirectional_direction_dropout
    from wenet.rwkv_v6.rwkv_wrapper_bidirectional_direction_dropout_both import \
        RWKV_TmixWrapper_bidirectional_direction_dropout_both
    from wenet.transformer.encoder import ConformerEncoder
    from wenet.transformer.ctc import CTC
    from wenet.transformer.search import ctc_greedy_search

    # ---- (i) time-mix block through the uni wrapper, fp32 and bf16 -------------------------
    for do_bf16 in (False, True):
        for (C, H, T, B, tag) in ((128, 2, 23, 2, "reduced"), (512, 8, 9, 1, "full")):
            m = RWKV_TmixWrapper(C // H, C, 12 if tag == "full" else 2, "rwkv", "uni", 2048, do_bf16, 1).eval()
            spec, cs = load_synth(m, 11)
            x = synth.randn((B, T, C), 12)
            y, cache = m(x, x, x)
            # the raw block in its own dtype (what tmix_x060c restates)
            xin = x.to(torch.bfloat16) if do_bf16 else x
            yb = m.tmix_block(xin)
            save(f"uni_wrapper_{tag}_{'bf16' if do_bf16 else 'f32'}",
                 dict(spec=spec, seed=11, checksum=cs, head_size=C // H, do_bfloat16=do_bf16,
                      x=x, y=y, block_y=yb, cache_shape=tuple(cache.shape)))

    # ---- (ii) bidirectional wrappers ---------------------------------------------------------
    for do_bf16 in (False, True):
        m = RWKV_TmixWrapper_bidirectional(64, 128, 2, "rwkv", "bi", 2048, do_bf16, 1).eval()
        spec, cs = load_synth(m, 21)
        x = synth.randn((3, 29, 128), 22)
        y, cache = m(x, x, x)
        m2 = RWKV_TmixWrapper_bidirectional2(64, 128, 2, "rwkv", "bi", 2048, do_bf16, 1).eval()
        m2.load_state_dict(m.state_dict())
        y2 = None
        if do_bf16:  # bidirectional2 keeps a bf16 flip buffer (bidirectional2.py:106), bf16-only by construction
            y2, _ = m2(x, x, x)
            assert torch.equal(y, y2), "bidirectional2 differs from bidirectional"
        save(f"bi_wrapper_{'bf16' if do_bf16 else 'f32'}",
             dict(spec=spec, seed=21, checksum=cs, head_size=64, do_bfloat16=do_bf16, x=x, y=y,
                  bi2_equal=y2 is not None, cache_shape=tuple(cache.shape)))

    # ---- direction-dropout wrappers, eval branches (env read at construction) ---------------
    cases = []
    for cls, cname in ((RWKV_TmixWrapper_bidirectional_direction_dropout, "rwkv_tmix60_dir_layer_drop"),
                       (RWKV_TmixWrapper_bidirectional_direction_dropout_both, "rwkv_tmix60_dir_layer_drop_both")):
        for env in ({}, {"RWKV_BIDIRECTIONAL_LAYERS": "0"}, {"RWKV_BIDIRECTIONAL_LAYERS": "0", "RWKV_ALT_DECODING": "1"},
                    {"RWKV_BIDIRECTIONAL_LAYERS": "1,3"}):
            for layer_id in (0, 1, 2):
                for k in ("RWKV_BIDIRECTIONAL_LAYERS", "RWKV_ALT_DECODING"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                m = cls(64, 128, 4, "rwkv", "bi", 2048, True, layer_id).eval()
                spec, cs = load_synth(m, 31)
                x = synth.randn((2, 17, 128), 32)
                y, _ = m(x, x, x)
                cases.append(dict(kind=cname, env=dict(env), layer_id=layer_id, y=y))
    for k in ("RWKV_BIDIRECTIONAL_LAYERS", "RWKV_ALT_DECODING"):
        os.environ.pop(k, None)
    save("dir_dropout_eval", dict(spec=spec, seed=31, checksum=cs, head_size=64, do_bfloat16=True, x=x, cases=cases))

    # ---- (iii)+(iv) reduced encoder: forward (ragged), per-layer outs, forward_chunk, CTC ---
    cfg = yaml.safe_load(open(os.path.join(ref_shim.REFERENCE_ROOT, YAML)))
    for variant in ("bf16slot", "f32", "uni_bf16slot", "uni_bf16model"):
        conf = dict(cfg["encoder_conf"])
        conf.update(REDUCED)
        if variant == "f32":
            conf["rwkv_do_bfloat16"] = False
        if variant.startswith("uni_"):
            conf["selfattention_layer_type"] = "rwkv_tmix60"
            conf["rnn_att_direction"] = "uni"
        mean = synth.randn((80,), 40)
        istd = torch.rand(80, generator=torch.Generator().manual_seed(41)) + 0.5
        from wenet.transformer.cmvn import GlobalCMVN
        enc = ConformerEncoder(80, global_cmvn=GlobalCMVN(mean, istd), **conf).eval()
        spec, cs = load_synth(enc, 42)
        ctc = CTC(50, 128).eval()
        ctc_spec, ctc_cs = load_synth(ctc, 43)
        xs = synth.randn((3, 203, 80), 44, 2.0)
        lens = torch.tensor([203, 150, 67])
        xc = synth.randn((1, 67, 80), 45, 2.0)
        # `encoder-rtf.py --bf16` (encoder-rtf.py:424-426) casts the whole model.  With the bidirectional slot
        # the reference then fails (the wrapper returns .float(), rwkv_wrapper_bidirectional.py:55-56, and the
        # next LayerNorm has bf16 parameters), so a whole-model-bf16 golden exists for the uni slot only.
        if variant == "uni_bf16model":
            enc = enc.to(torch.bfloat16)
            ctc = ctc.to(torch.bfloat16)
            xs = xs.to(torch.bfloat16)
            xc = xc.to(torch.bfloat16)
        # per-layer outputs: replay BaseEncoder.forward's loop through hooks
        layer_outs = []
        hooks = [l.register_forward_hook(lambda mod, i, o: layer_outs.append(o[0])) for l in enc.encoders]
        out, masks = enc(xs, lens)
        for h in hooks:
            h.remove()
        yc, att_cache, cnn_cache = enc.forward_chunk(xc, 0, -1)
        logp = ctc.log_softmax(out)
        enc_lens = masks.squeeze(1).sum(1)
        hyps = [r.tokens for r in ctc_greedy_search(logp.float(), enc_lens, 0)]
        save(f"encoder_reduced_{variant}",
             dict(spec=spec, seed=42, checksum=cs, conf=conf, ctc_spec=ctc_spec, ctc_seed=43, ctc_checksum=ctc_cs,
                  xs=xs, lens=lens, out=out, masks=masks, layer0=layer_outs[0], layer1=layer_outs[1],
                  chunk_x=xc, chunk_y=yc, att_cache_shape=tuple(att_cache.shape), cnn_cache_shape=tuple(cnn_cache.shape),
                  logp_sample=logp[:, ::7, :].clone(), logp_full=logp.clone(), greedy=hyps, enc_lens=enc_lens))
        print(variant, "greedy lens", [len(h) for h in hyps], "params", sum(p.numel() for p in enc.parameters()))

    # ---- config c5: CTC prefix beam search and the CTC-fused RNN-T prefix beam search ------------------------
    from wenet.transformer.search import ctc_prefix_beam_search
    from wenet.transducer.joint import TransducerJoint
    from wenet.transducer.predictor import RNNPredictor
    from wenet.transducer.search.prefix_beam_search import PrefixBeamSearch
    V, D = 50, 128
    enc_out = synth.randn((3, 37, D), 61)
    enc_lens = torch.tensor([37, 30, 11])
    ctc = CTC(V, D).eval()
    ctc_spec, ctc_cs = load_synth(ctc, 62)
    logp = ctc.log_softmax(enc_out)
    cres = ctc_prefix_beam_search(logp, enc_lens, 8, None, 0)
    pred = RNNPredictor(V, embed_size=64, output_size=64, embed_dropout=0.1, hidden_size=64, num_layers=2, bias=True,
                        rnn_type="lstm", dropout=0.1).eval()
    pred_spec, pred_cs = load_synth(pred, 63)
    joint = TransducerJoint(V, enc_output_size=D, pred_output_size=64, join_dim=64, prejoin_linear=True,
                            postjoin_linear=False, joint_mode="add", activation="tanh").eval()
    joint_spec, joint_cs = load_synth(joint, 64)
    bs = PrefixBeamSearch(None, pred, joint, ctc, 0)
    tres = bs.prefix_beam_search_decode(enc_out, enc_lens, logp, beam_size=8, ctc_weight=0.3, transducer_weight=0.7)
    jt = joint(enc_out[:, :5], pred(torch.tensor([[0, 3, 7], [0, 9, 9], [0, 1, 2]])))
    save("search_c5", dict(
        enc_out=enc_out, enc_lens=enc_lens, ctc_spec=ctc_spec, ctc_seed=62, ctc_checksum=ctc_cs, logp=logp,
        pred_spec=pred_spec, pred_seed=63, joint_spec=joint_spec, joint_seed=64, joint_sample=jt,
        ctc_prefix=[dict(tokens=list(r.tokens), score=r.score, nbest=[list(n) for n in r.nbest],
                         nbest_scores=list(r.nbest_scores)) for r in cres],
        rnnt=[dict(tokens=list(r.tokens), score=r.score, nbest=[list(n) for n in r.nbest],
                   nbest_scores=list(r.nbest_scores)) for r in tres]))
    print("c5 ctc prefix best", [len(r.tokens) for r in cres], "rnnt best", [len(r.tokens) for r in tres])

    # ---- closed-form initialisation of the time-mix parameters (src/model.py:232-260; no RNG involved) ----
    init = {}
    for layer_id in (0, 5, 11):
        blk = RWKV_TmixWrapper(64, 512, 12, "rwkv", "uni", 2048, False, layer_id).tmix_block
        init[layer_id] = {k: v.clone() for k, v in blk.state_dict().items()
                          if k.startswith("time_maa_") and not k.startswith("time_maa_rkvw") or k in ("time_decay", "time_faaaa")}
    save("tmix_init", dict(init=init, n_layers=12, n_embd=512, head_size=64))

    # ---- full-size encoder: parameter names/shapes only (state-dict compatibility contract) --
    conf = dict(cfg["encoder_conf"])
    enc = ConformerEncoder(80, **conf)
    spec = synth.spec_of(enc.state_dict())
    save("encoder_full_spec", dict(spec=spec, conf=conf, n_params=sum(p.numel() for p in enc.parameters())))
    print("full encoder params", sum(p.numel() for p in enc.parameters()))


if __name__ == "__main__":
    main()
