"""token ids -> text -> WER against fixtures captured from the reference
(tests/golden/make_text_goldens.py: its tokenizer classes through the import shim, its tools/compute-wer*.py as
scripts).  Pure host logic: no GPU."""
import io
import json
import os

import pytest

from paper_accurate_fast_cheap_amd.scoring import wer as W
from paper_accurate_fast_cheap_amd.text import CharTokenizer, RevBpeTokenizer, init_tokenizer

G = os.path.join(os.path.dirname(__file__), "golden", "text")


def _report(ref, hyp, **kw):
    buf = io.StringIO()
    W.score_files(os.path.join(G, ref), os.path.join(G, hyp), out=buf, **kw)
    return buf.getvalue()


def _golden(name):
    return open(os.path.join(G, name), encoding="utf-8").read()


def test_wer_report_default_matches_reference_text():
    assert _report("wer_ref.txt", "wer_hyp.txt") == _golden("wer_default.out")


def test_wer_report_giga_post_processing():
    assert _report("wer_ref.txt", "wer_hyp.txt", giga=True) == _golden("wer_giga.out")


def test_wer_report_char_mode_wrapping_and_padding():
    got = _report("wer_char_ref.txt", "wer_char_hyp.txt", tochar=True, max_words_per_line=8, padding_symbol="_")
    assert got == _golden("wer_char.out")


def test_wer_quiet_case_sensitive_keep_tags():
    assert _report("wer_ref.txt", "wer_hyp.txt", case_sensitive=True, remove_tag=False, verbose=0) == _golden("wer_quiet.out")


def test_wer_cli_switches(capsys):
    W.main(["--cs=1", "--rt=0", "--v=0", "--bogus=1", os.path.join(G, "wer_ref.txt"), os.path.join(G, "wer_hyp.txt")])
    assert capsys.readouterr().out == _golden("wer_quiet.out")


def test_align_tie_breaking_and_edges():
    a = W.align([], [])
    assert (a.ops, a.counts.all) == ([], 0) and a.counts.wer == 0.0
    a = W.align(["a", "b"], [])
    assert a.ops == ["del", "del"] and a.rec == ["", ""]
    a = W.align([], ["x"])
    assert a.ops == ["ins"] and a.counts.all == 0 and a.counts.ins == 1
    # two alignments of equal cost: deletion is preferred over insertion over substitution when back-tracing
    a = W.align(["a", "b", "c"], ["b", "c", "d"])
    assert a.counts.sub + a.counts.dele + a.counts.ins == 2
    assert a.ops == ["del", "cor", "cor", "ins"]
    # sums over utterances equal the per-token bookkeeping
    s = W.WerScorer()
    tot = W.ErrorCounts()
    for i, (l, r) in enumerate([("a b c", "a x c"), ("d e", "d e f g"), ("h", "")]):
        tot += s.add(f"u{i}", l.split(), r.split()).counts
    assert s.overall() == tot


def test_characterize_and_normalize():
    # an ASCII run only ends at a space or a non-ASCII character: "hello<unk>" stays one unit, "<noise>ok" splits at ">"
    assert W.characterize("我是<noise>ok 你，的 hello<unk>") == ["我", "是", "<noise>", "ok", "你", "的", "hello<unk>"]
    assert W.normalize(["a<b>c", "<x>", "Ig", "keep"], {"IG"}, False, {"KEEP": ["K", "P"]}) == ["AC", "K", "P"]
    assert W.strip_tags("a<bc") == "a"
    assert W.giga_post_process("uh the state-of-the-art <COMMA> thing") == "THE STATE OF THE ART THING"
    assert W.default_cluster("ROUTE66") == "Other" and W.default_cluster("DON'T") == "English"
    assert W.default_cluster("1990") == "Number" and W.default_cluster("我") == "Mandarin" and W.default_cluster("'") == "Other"


@pytest.fixture(scope="module")
def tok_golden():
    return json.load(open(os.path.join(G, "tokenizer.json"), encoding="utf-8"))


def _check(tk, lines, expected):
    for ln, e in zip(lines, expected):
        toks, ids = tk.tokenize(ln)
        assert toks == e["tokens"] and ids == e["ids"], ln
        text, back = tk.detokenize(ids)
        assert text == e["text"] and back == e["tokens_back"], ln


def test_rev_bpe_tokenizer_matches_reference(tok_golden):
    model, units = os.path.join(G, "spm_tiny.model"), os.path.join(G, "units.txt")
    tk = RevBpeTokenizer(model, units, None)
    _check(tk, tok_golden["lines"], tok_golden["rev_bpe"])
    assert tk.vocab_size() == tok_golden["vocab_size"]
    raw = RevBpeTokenizer(model, units, None, full_config={"remove_sw": False, "replace_unk_as_unknown": False})
    _check(raw, tok_golden["lines"], tok_golden["rev_bpe_raw"])


def test_char_tokenizer_matches_reference(tok_golden):
    ck = CharTokenizer(os.path.join(G, "units.txt"), None)
    _check(ck, tok_golden["lines"], tok_golden["char"])
    ck2 = CharTokenizer({"A": 1, "B": 2, "▁": 3, "{NOISE}": 4, "<unk>": 0}, ["{NOISE}"])
    for e in tok_golden["char_nls"]:
        toks, ids = ck2.tokenize(e["line"])
        assert toks == e["tokens"] and ids == e["ids"] and ck2.detokenize(ids)[0] == e["text"]


def test_init_tokenizer_and_errors(tmp_path):
    conf = {"tokenizer": "rev_bpe", "tokenizer_conf": {"bpe_path": os.path.join(G, "spm_tiny.model"),
            "symbol_table_path": os.path.join(G, "units.txt"), "non_lang_syms_path": None}}
    assert isinstance(init_tokenizer(conf), RevBpeTokenizer)
    conf["tokenizer"] = "char"
    assert type(init_tokenizer(conf)) is CharTokenizer
    conf["tokenizer"] = "whisper"
    with pytest.raises(NotImplementedError):
        init_tokenizer(conf)
    bad = tmp_path / "syms.txt"
    bad.write_text("NOISE\n")
    from paper_accurate_fast_cheap_amd.utils.file_utils import BadSymbolFormat
    with pytest.raises(BadSymbolFormat):
        CharTokenizer({"a": 0}, str(bad))


def test_greedy_tokens_to_text_to_wer_end_to_end(tok_golden):
    """ids as the CTC search returns them -> text -> scorer: the chain a decode run ends with."""
    tk = RevBpeTokenizer(os.path.join(G, "spm_tiny.model"), os.path.join(G, "units.txt"), None)
    ref_text = "THE STATE-OF-THE-ART E-COMMERCE"
    _, ids = tk.tokenize(ref_text)
    hyp_text, _ = tk.detokenize(ids[:-1])   # drop the last piece: one word differs
    s = W.WerScorer()
    r = s.add("u", W.giga_post_process(ref_text).split(), W.giga_post_process(hyp_text).split())
    assert r.counts.all == 7 and r.counts.cor == 6 and r.counts.sub + r.counts.dele == 1
