"""Round-2 additions to the goldens captured from the reference's own Python modules (same recipe and rules as
make_goldens.py: run ONCE in the build container where /root/reference is mounted; fixtures hold data only).

    python tests/golden/make_goldens_r2.py

  encoder_chunk_by_chunk.pt -- BaseEncoder.forward_chunk_by_chunk (wenet/transformer/encoder.py:341-402) of the reduced
                               encoder, three slot variants, two chunk sizes;
  dir_dropout_train.pt      -- the TRAIN-time branch of the two direction-dropout wrappers
                               (wenet/rwkv_v6/rwkv_wrapper_bidirectional_direction_dropout{,_both}.py:59-71): outputs under
                               torch.manual_seed(s) for seeds chosen so that every branch occurs.
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
    from wenet.rwkv_v6.rwkv_wrapper_bidirectional_direction_dropout import \
        RWKV_TmixWrapper_bidirectional_direction_dropout
    from wenet.rwkv_v6.rwkv_wrapper_bidirectional_direction_dropout_both import \
        RWKV_TmixWrapper_bidirectional_direction_dropout_both
    from wenet.transformer.cmvn import GlobalCMVN
    from wenet.transformer.encoder import ConformerEncoder

    # ---- forward_chunk_by_chunk ---------------------------------------------------------------------------------
    cfg = yaml.safe_load(open(os.path.join(ref_shim.REFERENCE_ROOT, YAML)))
    cases = {}
    xs = synth.randn((1, 203, 80), 51, 2.0)
    for variant in ("bf16slot", "f32", "uni_bf16slot"):
        conf = dict(cfg["encoder_conf"])
        conf.update(REDUCED)
        if variant == "f32":
            conf["rwkv_do_bfloat16"] = False
        if variant.startswith("uni_"):
            conf["selfattention_layer_type"] = "rwkv_tmix60"
            conf["rnn_att_direction"] = "uni"
        mean = synth.randn((80,), 40)
        istd = torch.rand(80, generator=torch.Generator().manual_seed(41)) + 0.5
        enc = ConformerEncoder(80, global_cmvn=GlobalCMVN(mean, istd), **conf).eval()
        spec, cs = load_synth(enc, 42)
        outs = {}
        for chunk in (16, 5):
            ys, masks = enc.forward_chunk_by_chunk(xs, chunk, -1)
            outs[chunk] = dict(ys=ys, masks=masks)
            print(variant, "chunk", chunk, tuple(ys.shape))
        cases[variant] = dict(spec=spec, seed=42, checksum=cs, conf=conf, outs=outs)
    save("encoder_chunk_by_chunk", dict(xs=xs, cases=cases))

    # ---- train-time direction dropout ------------------------------------------------------------------------------
    x = synth.randn((2, 17, 128), 32)
    out = dict(x=x, head_size=64, do_bfloat16=True, seed=31, cases=[])
    for cls, cname, both in ((RWKV_TmixWrapper_bidirectional_direction_dropout, "rwkv_tmix60_dir_layer_drop", False),
                             (RWKV_TmixWrapper_bidirectional_direction_dropout_both, "rwkv_tmix60_dir_layer_drop_both", True)):
        m = cls(64, 128, 4, "rwkv", "bi", 2048, True, 1)
        spec, cs = load_synth(m, 31)
        out["spec"], out["checksum"] = spec, cs
        m.train()
        # what each branch would give, to label the draws (the labels are a cross-check, the fixture is y)
        a = m.rwkv_wrapper_forward(x, x, x)[0]
        xf = torch.flip(x, [1])
        b = torch.flip(m.rwkv_wrapper_backward(xf, xf, xf)[0], [1])
        label = {"bi": (a + b) / 2, "left": a, "right": b}
        seen = {}
        for seed in range(400):
            torch.manual_seed(seed)
            y, _ = m(x, x, x)
            kind = [k for k, v in label.items() if torch.equal(v, y)]
            assert len(kind) == 1, (cname, seed, kind)
            if seen.get(kind[0], 0) < (3 if kind[0] == "bi" else 2):
                seen[kind[0]] = seen.get(kind[0], 0) + 1
                out["cases"].append(dict(kind=cname, both=both, manual_seed=seed, branch=kind[0], y=y))
        want = {"bi", "left", "right"} if both else {"bi", "left"}
        assert set(seen) == want, (cname, seen)
        print(cname, {k: [c["manual_seed"] for c in out["cases"] if c["kind"] == cname and c["branch"] == k] for k in seen})
    save("dir_dropout_train", out)


if __name__ == "__main__":
    main()
