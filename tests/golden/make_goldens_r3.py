"""Round-3 additions to the goldens captured from the reference's own Python modules (same recipe and rules as
make_goldens.py: run ONCE in the build container where /root/reference is mounted; fixtures hold data only).

    python tests/golden/make_goldens_r3.py

  encoder_postnorm_abspos.pt -- the reduced encoder in two configurations the paper's YAMLs do not use but the
                                reference's classes support:
      "postnorm": normalize_before = False (wenet/transformer/encoder_layer.py:207-208,233-234,246-247,255-256):
                  forward() of a ragged batch, per-layer outputs;
      "abspos":   pos_enc_layer_type = abs_pos (wenet/transformer/embedding.py:58-77): forward() and
                  forward_chunk_by_chunk(), whose windows receive the running output offset (encoder.py:377-399).
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
    from wenet.transformer.encoder import ConformerEncoder

    cfg = yaml.safe_load(open(os.path.join(ref_shim.REFERENCE_ROOT, YAML)))
    xs = synth.randn((3, 151, 80), 61, 2.0)
    lens = torch.tensor([151, 90, 33])
    long = synth.randn((1, 203, 80), 62, 2.0)
    cases = {}
    for name, over in (("postnorm", dict(normalize_before=False)), ("abspos", dict(pos_enc_layer_type="abs_pos"))):
        for prec in ("f32", "bf16slot"):
            conf = dict(cfg["encoder_conf"])
            conf.update(REDUCED)
            conf.update(over)
            conf["rwkv_do_bfloat16"] = prec == "bf16slot"
            enc = ConformerEncoder(80, **conf).eval()
            spec, cs = load_synth(enc, 63)
            out, masks = enc(xs, lens)
            c = dict(spec=spec, seed=63, checksum=cs, conf=conf, out=out, masks=masks)
            if name == "abspos":
                c["chunks"] = {}
                for chunk in (16, 5):
                    ys, m = enc.forward_chunk_by_chunk(long, chunk, -1)
                    c["chunks"][chunk] = dict(ys=ys, masks=m)
            cases[f"{name}_{prec}"] = c
            print(name, prec, tuple(out.shape), float(out.abs().mean()))
    save("encoder_postnorm_abspos", dict(xs=xs, lens=lens, long=long, cases=cases))


if __name__ == "__main__":
    main()
