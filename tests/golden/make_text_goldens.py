"""Capture golden vectors for the token -> text -> WER step from the reference, in THIS container.

  python tests/golden/make_text_goldens.py

Writes under tests/golden/text/:
  wer_{ref,hyp}.txt, wer_char_{ref,hyp}.txt   synthetic transcripts (seeded)
  wer_default.out, wer_giga.out, wer_char.out, wer_quiet.out
                                              stdout of the reference's tools/compute-wer{,-giga}.py on them
  spm_tiny.model, units.txt                   a tiny SentencePiece model trained here (seeded corpus) + symbol table
  tokenizer.json                              RevBpeTokenizer / CharTokenizer outputs of the reference on sample lines
The fixtures are data (inputs + expected outputs); the reference's sources stay where they are.
"""
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "text")
sys.path.insert(0, ROOT)

WORDS = ("the of and to in is that for it as was with be by on not he this are or his from at which but have an had "
         "they you were their one all we can her has there been if more when will would who so no e-commerce "
         "state-of-the-art don't it's mother's <unk> uh um <COMMA> <PERIOD> okay o'clock 1990 42 route66 naïve café").split()
CJK = "我是你的不了在人有这中大来上国个到说们为子和地出道也时年得就那要下以生会自着去之过家学对可她里后小么心多天"


def corrupt(rng, words, pool):
    out = []
    for w in words:
        r = rng.random()
        if r < 0.08:
            continue
        if r < 0.18:
            out.append(rng.choice(pool))
        else:
            out.append(w)
        if rng.random() < 0.07:
            out.append(rng.choice(pool))
    return out


def make_transcripts():
    rng = random.Random(777)
    ref, hyp, cref, chyp = [], [], [], []
    for u in range(40):
        n = rng.randint(0, 25) if u % 9 else 0
        words = [rng.choice(WORDS) for _ in range(n)]
        ref.append(f"utt{u:03d} " + " ".join(words))
        if u % 11 != 5:   # some utterances are missing from the hypothesis
            hyp.append(f"utt{u:03d} " + " ".join(corrupt(rng, words, WORDS)))
        n = rng.randint(1, 20)
        chars = [rng.choice(CJK) for _ in range(n)]
        mixed = list(chars)
        if u % 3 == 0:
            mixed.insert(rng.randint(0, n), " hello ")
        if u % 4 == 0:
            mixed.insert(rng.randint(0, n), "<noise>")
        if u % 5 == 0:
            mixed.insert(rng.randint(0, n), "，")
        cref.append(f"c{u:03d} " + "".join(mixed))
        chyp.append(f"c{u:03d} " + "".join(corrupt(rng, mixed, list(CJK))))
    hyp.append("extra_utt not in the reference")
    return ref, hyp, cref, chyp


def run_ref(script, *args):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    return subprocess.run([sys.executable, os.path.join("/root/reference/tools", script), *args], check=True,
                          capture_output=True, text=True, env=env).stdout


def main():
    os.makedirs(OUT, exist_ok=True)
    ref, hyp, cref, chyp = make_transcripts()
    paths = {}
    for name, lines in (("wer_ref", ref), ("wer_hyp", hyp), ("wer_char_ref", cref), ("wer_char_hyp", chyp)):
        paths[name] = os.path.join(OUT, name + ".txt")
        open(paths[name], "w", encoding="utf-8").write("\n".join(lines) + "\n")
    open(os.path.join(OUT, "wer_default.out"), "w", encoding="utf-8").write(
        run_ref("compute-wer.py", "--char=0", "--v=1", paths["wer_ref"], paths["wer_hyp"]))
    open(os.path.join(OUT, "wer_giga.out"), "w", encoding="utf-8").write(
        run_ref("compute-wer-giga.py", "--char=0", "--v=1", paths["wer_ref"], paths["wer_hyp"]))
    open(os.path.join(OUT, "wer_char.out"), "w", encoding="utf-8").write(
        run_ref("compute-wer.py", "--char=1", "--v=1", "--maxw=8", "--padding-symbol=underline",
                paths["wer_char_ref"], paths["wer_char_hyp"]))
    open(os.path.join(OUT, "wer_quiet.out"), "w", encoding="utf-8").write(
        run_ref("compute-wer.py", "--cs=1", "--rt=0", "--v=0", paths["wer_ref"], paths["wer_hyp"]))

    # ---- tokenizer ------------------------------------------------------------------------------------------
    import sentencepiece as spm
    rng = random.Random(778)
    plain = [w for w in WORDS if not w.startswith("<")] + ["<unknown>"]
    corpus = os.path.join(OUT, "_corpus.txt")
    with open(corpus, "w", encoding="utf-8") as f:
        for _ in range(2000):
            f.write(" ".join(rng.choice(plain) for _ in range(rng.randint(3, 12))).upper() + "\n")
    prefix = os.path.join(OUT, "spm_tiny")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=prefix, vocab_size=100, model_type="unigram",
                                   character_coverage=1.0, user_defined_symbols=["<unknown>"], num_threads=1,
                                   shuffle_input_sentence=False)
    os.remove(corpus)
    os.remove(prefix + ".vocab")
    sp = spm.SentencePieceProcessor()
    sp.load(prefix + ".model")
    pieces = [sp.id_to_piece(i) for i in range(sp.get_piece_size())]
    # units.txt as the recipes build it: <blank> 0, <unk> 1, pieces..., <sos/eos> last
    units = ["<blank>", "<unk>"] + [p for p in pieces if p not in ("<unk>", "<s>", "</s>")] + ["<sos/eos>"]
    with open(os.path.join(OUT, "units.txt"), "w", encoding="utf-8") as f:
        for i, u in enumerate(units):
            f.write(f"{u} {i}\n")

    from oracle import ref_shim
    ref_shim.install()
    from wenet.text.rev_bpe_tokenizer import RevBpeTokenizer
    from wenet.text.char_tokenizer import CharTokenizer
    import contextlib, io
    lines = ["THE STATE-OF-THE-ART E-COMMERCE", "  it's  <sw> o'clock  ", "HELLO <unk> WORLD", "", "ROUTE66 IN 1990",
             "CAFÉ NAÏVE ZZZQ", "DON'T <sw><sw> MOTHER'S"]
    with contextlib.redirect_stdout(io.StringIO()):
        tk = RevBpeTokenizer(prefix + ".model", os.path.join(OUT, "units.txt"), None)
        tk2 = RevBpeTokenizer(prefix + ".model", os.path.join(OUT, "units.txt"), None,
                              full_config={"remove_sw": False, "replace_unk_as_unknown": False})
        ck = CharTokenizer(os.path.join(OUT, "units.txt"), None)
        ck2 = CharTokenizer({"A": 1, "B": 2, "▁": 3, "{NOISE}": 4, "<unk>": 0}, ["{NOISE}"], split_with_space=False)
    rec = {"lines": lines, "rev_bpe": [], "rev_bpe_raw": [], "char": [], "char_nls": []}
    for ln in lines:
        for key, t in (("rev_bpe", tk), ("rev_bpe_raw", tk2), ("char", ck)):
            toks, ids = t.tokenize(ln)
            text, toks_back = t.detokenize(ids)
            rec[key].append({"tokens": toks, "ids": ids, "text": text, "tokens_back": toks_back})
    for ln in ["ab {noise} ba", "A B{NOISE}", "xyz"]:
        toks, ids = ck2.tokenize(ln)
        rec["char_nls"].append({"line": ln, "tokens": toks, "ids": ids, "text": ck2.detokenize(ids)[0]})
    rec["vocab_size"] = tk.vocab_size()
    json.dump(rec, open(os.path.join(OUT, "tokenizer.json"), "w", encoding="utf-8"), ensure_ascii=False, indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
