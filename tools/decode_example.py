"""End-to-end decode on the MI355X path, the shape of the reference's recognize.py for this model family:
waveforms -> HIP fbank -> bidirectional RWKV encoder -> CTC (greedy / prefix beam search on the device) -> SentencePiece
tokens -> text -> WER report.  Random-init weights (no checkpoint can ship here), so the text is noise: the point is the
chain and its interfaces -- pass --checkpoint / --config of a real GigaSpeech model to decode for real.

  python tools/decode_example.py [--config conf.yaml --checkpoint model.pt --bpe_model spm.model --units units.txt]
"""
import argparse, io, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                                                  # noqa: E402
from paper_accurate_fast_cheap_amd.dataset.fbank import fbank                       # noqa: E402
from paper_accurate_fast_cheap_amd.scoring.wer import WerScorer, giga_post_process  # noqa: E402
from paper_accurate_fast_cheap_amd.text import RevBpeTokenizer                      # noqa: E402
from paper_accurate_fast_cheap_amd.utils.init_model import init_model               # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "text")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config"); ap.add_argument("--checkpoint")
    ap.add_argument("--bpe_model", default=os.path.join(G, "spm_tiny.model"))
    ap.add_argument("--units", default=os.path.join(G, "units.txt"))
    ap.add_argument("--mode", default="ctc_greedy_search", choices=["ctc_greedy_search", "ctc_prefix_beam_search"])
    ap.add_argument("--beam_size", type=int, default=8)
    args = ap.parse_args(argv)
    dev = torch.device("cuda")
    tok = RevBpeTokenizer(args.bpe_model, args.units, None)
    if args.config:
        import yaml
        configs = yaml.safe_load(open(args.config))
    else:   # the paper's encoder shape with a vocabulary that matches the tiny tokenizer fixture
        configs = dict(encoder="conformer", encoder_conf=B.encoder_conf(), input_dim=80, output_dim=tok.vocab_size(),
                       ctc="ctc", ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})

    class A:
        checkpoint = args.checkpoint

    torch.manual_seed(777)
    model, _ = init_model(A(), configs)
    model = model.eval().to(torch.bfloat16).to(dev)
    # three synthetic "utterances" (2.5 s, 4 s, 1.2 s) and made-up reference transcripts
    waves = [B.synthetic_waveform(s, 10 + i).to(dev) for i, s in enumerate((2.5, 4.0, 1.2))]
    refs = ["THE STATE-OF-THE-ART", "IT'S UH E-COMMERCE <COMMA> OKAY", "HELLO"]
    feats = [fbank(w, num_mel_bins=80, frame_length=25.0, frame_shift=10.0, dither=0.0, energy_floor=0.0,
                   sample_frequency=16000.0) for w in waves]
    lens = torch.tensor([f.shape[0] for f in feats], device=dev)
    batch = torch.zeros(len(feats), int(lens.max()), 80, dtype=torch.bfloat16, device=dev)
    for i, f in enumerate(feats):
        batch[i, :f.shape[0]] = f.to(torch.bfloat16)
    with torch.no_grad():
        results = model.decode([args.mode], batch, lens, beam_size=args.beam_size)[args.mode]
    scorer = WerScorer()
    out = io.StringIO()
    for i, (r, ref) in enumerate(zip(results, refs)):
        text, pieces = tok.detokenize(list(r.tokens))
        al = scorer.add(f"utt{i}", giga_post_process(ref).split(), giga_post_process(text).split())
        out.write(f"utt{i}: {len(r.tokens)} tokens -> {text[:60]!r}   {al.counts.line()}\n")
    tot = scorer.overall()
    out.write("Overall -> %4.2f %% %s\n" % (tot.wer, tot.line()))
    print(out.getvalue(), end="")
    return results


if __name__ == "__main__":
    main()
