"""Word error rate with alignments: the scorer behind the paper's WER tables.

Counterpart of ``tools/compute-wer.py`` (generic) and ``tools/compute-wer-giga.py`` (GigaSpeech text post-processing,
``:14-35``) of the reference.  The numbers a recipe reads out of the report -- ``N C S D I`` per utterance, overall
and per word cluster -- depend on how ties between equally cheap alignments are broken, so the dynamic programme
keeps the reference's preference order (deletion, then insertion, then match/substitution; a later candidate wins
only if strictly cheaper, ``compute-wer.py:138-160``) and the same back-trace.  The report text has the same layout.

Library use::

    scorer = WerScorer()
    r = scorer.add("utt1", "the cat sat".split(), "the cat sat down".split())
    scorer.overall()        # ErrorCounts(all=3, cor=3, sub=0, dele=0, ins=1)

Command line (same switches as the reference script; ``--giga`` selects the GigaSpeech post-processing)::

    python -m paper_accurate_fast_cheap_amd.scoring.wer [--giga] [--cs=0|1] [--char=0|1] [--v=N] [--ig=FILE]
        [--splitfile=FILE] [--cluster=FILE] [--maxw=N] [--rt=0|1] [--padding-symbol=space|underline] ref hyp
"""
import sys
import unicodedata
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

SPACES = (" ", "\t", "\r", "\n")
PUNCTS = frozenset("!,?、。！，；？：「」︰『』《》")

# GigaSpeech scoring conventions (compute-wer-giga.py:14-18)
_FILLERS = ["UH", "UHH", "UM", "EH", "MM", "HM", "AH", "HUH", "HA", "ER", "OOF", "HEE", "ACH", "EEE", "EW"]
_NON_SCORING = frozenset(_FILLERS + ["<UNK>", "<unk>", "<COMMA>", "<PERIOD>", "<QUESTIONMARK>", "<EXCLAMATIONPOINT>",
                                     "<SIL>", "<NOISE>", "<MUSIC>", "<OTHER>"])


def giga_post_process(text: str) -> str:
    """Upper-case, split hyphenated words, drop fillers / tags / punctuation tokens (compute-wer-giga.py:20-35)."""
    words = text.upper().replace("-", " ").split()
    return " ".join(w for w in words if w not in _NON_SCORING)


def characterize(string: str) -> List[str]:
    """Split a line into scoring units for ``--char=1``: every CJK-like letter (category Lo) is a unit, ASCII runs
    stay whole words, ``<tag>`` is one unit, listed punctuation and spaces vanish (compute-wer.py:15-46)."""
    out, i, n = [], 0, len(string)
    while i < n:
        ch = string[i]
        if ch in PUNCTS:
            i += 1
            continue
        cat = unicodedata.category(ch)
        if cat in ("Zs", "Cn") or ch in SPACES:
            i += 1
            continue
        if cat == "Lo":
            out.append(ch)
            i += 1
            continue
        stop = ">" if ch == "<" else " "
        j = i + 1
        while j < n and not (ord(string[j]) >= 128 or string[j] in SPACES or string[j] == stop):
            j += 1
        if j < n and string[j] == ">":
            j += 1
        out.append(string[i:j])
        i = j
    return out


def strip_tags(token: str) -> str:
    """Remove ``<...>`` spans (an unterminated ``<`` swallows the rest), compute-wer.py:49-62."""
    out, i, n = [], 0, len(token)
    while i < n:
        if token[i] == "<":
            while i < n and token[i] != ">":
                i += 1
            i += 1
        else:
            out.append(token[i])
            i += 1
    return "".join(out)


def normalize(tokens: Iterable[str], ignore_words=frozenset(), case_sensitive: bool = False,
              split: Optional[Dict[str, List[str]]] = None, remove_tag: bool = True) -> List[str]:
    """compute-wer.py:65-84: upper-case unless case sensitive, drop ignored words, strip tags, expand split words."""
    out = []
    for tok in tokens:
        x = tok if case_sensitive else tok.upper()
        if x in ignore_words:
            continue
        if remove_tag:
            x = strip_tags(x)
        if not x:
            continue
        if split and x in split:
            out.extend(split[x])
        else:
            out.append(x)
    return out


@dataclass
class ErrorCounts:
    all: int = 0
    cor: int = 0
    sub: int = 0
    dele: int = 0
    ins: int = 0

    @property
    def wer(self) -> float:
        return (self.ins + self.sub + self.dele) * 100.0 / self.all if self.all else 0.0

    def __iadd__(self, o: "ErrorCounts"):
        self.all += o.all; self.cor += o.cor; self.sub += o.sub; self.dele += o.dele; self.ins += o.ins
        return self

    def line(self) -> str:
        return "N=%d C=%d S=%d D=%d I=%d" % (self.all, self.cor, self.sub, self.dele, self.ins)


@dataclass
class Alignment:
    lab: List[str] = field(default_factory=list)   # "" where the hypothesis inserted
    rec: List[str] = field(default_factory=list)   # "" where the hypothesis deleted
    ops: List[str] = field(default_factory=list)   # cor / sub / del / ins
    counts: ErrorCounts = field(default_factory=ErrorCounts)


_DEL, _INS, _DIAG = 0, 1, 2


def align(lab: Sequence[str], rec: Sequence[str]) -> Alignment:
    """Levenshtein alignment (unit costs) with the reference's tie-breaking."""
    n, m = len(lab), len(rec)
    prev = list(range(m + 1))
    back = [bytearray([_INS]) * (m + 1)] + [bytearray(m + 1) for _ in range(n)]
    for i in range(1, n + 1):
        cur = [i] + [0] * m
        brow = back[i]
        brow[0] = _DEL
        li = lab[i - 1]
        for j in range(1, m + 1):
            best, how = prev[j] + 1, _DEL
            c = cur[j - 1] + 1
            if c < best:
                best, how = c, _INS
            c = prev[j - 1] + (li != rec[j - 1])
            if c < best:
                best, how = c, _DIAG
            cur[j] = best
            brow[j] = how
        prev = cur
    res = Alignment()
    i, j = n, m
    while i > 0 or j > 0:
        how = back[i][j]
        if how == _DIAG:
            op = "cor" if lab[i - 1] == rec[j - 1] else "sub"
            res.lab.append(lab[i - 1]); res.rec.append(rec[j - 1]); res.ops.append(op)
            i -= 1; j -= 1
        elif how == _DEL:
            res.lab.append(lab[i - 1]); res.rec.append(""); res.ops.append("del")
            i -= 1
        else:
            res.lab.append(""); res.rec.append(rec[j - 1]); res.ops.append("ins")
            j -= 1
    res.lab.reverse(); res.rec.reverse(); res.ops.reverse()
    c = res.counts
    for op in res.ops:
        if op == "cor":
            c.cor += 1; c.all += 1
        elif op == "sub":
            c.sub += 1; c.all += 1
        elif op == "del":
            c.dele += 1; c.all += 1
        else:
            c.ins += 1
    return res


def default_cluster(word: str) -> str:
    """Script class of a word for the per-cluster lines of the report (compute-wer.py:257-296)."""
    ignorable = ("AMPERSAND", "APOSTROPHE", "COMMERCIAL AT", "DEGREE CELSIUS", "EQUALS SIGN", "FULL STOP",
                 "HYPHEN-MINUS", "LOW LINE", "NUMBER SIGN", "PLUS SIGN", "SEMICOLON")
    kinds = []
    for ch in reversed(word):
        name = unicodedata.name(ch)   # raises ValueError for unnamed characters, as the reference does
        if name.startswith("DIGIT"):
            kinds.append("Number")
        elif name.startswith(("CJK UNIFIED IDEOGRAPH", "CJK COMPATIBILITY IDEOGRAPH")):
            kinds.append("Mandarin")
        elif name.startswith(("LATIN CAPITAL LETTER", "LATIN SMALL LETTER")):
            kinds.append("English")
        elif name.startswith("HIRAGANA LETTER"):
            kinds.append("Japanese")
        elif name.startswith(ignorable):
            continue
        else:
            return "Other"
    if not kinds or any(k != kinds[0] for k in kinds):
        return "Other"
    return kinds[0]


def display_width(s: str) -> int:
    return sum(1 + (unicodedata.east_asian_width(c) in "AFW") for c in s)


class WerScorer:
    """Accumulates per-token statistics over utterances (the reference's ``Calculator``, compute-wer.py:87-250)."""

    def __init__(self):
        self.per_token: Dict[str, ErrorCounts] = {}
        self.cluster_of: Dict[str, str] = {}
        self.clusters: Dict[str, Dict[str, int]] = {}

    def _slot(self, tok: str) -> ErrorCounts:
        if tok not in self.per_token:
            self.per_token[tok] = ErrorCounts()
        return self.per_token[tok]

    def add(self, utt: str, lab: Sequence[str], rec: Sequence[str]) -> Alignment:
        for w in list(rec) + list(lab):
            if w not in self.cluster_of:
                name = default_cluster(w)
                self.clusters.setdefault(name, {})[w] = 1
                self.cluster_of[w] = name
        for tok in list(lab) + list(rec):   # registration order of the reference: labels first, then hypothesis
            if tok:
                self._slot(tok)
        res = align(lab, rec)
        for op, l, r in zip(res.ops, res.lab, res.rec):
            if op == "ins":
                self._slot(r).ins += 1
            else:
                s = self._slot(l)
                s.all += 1
                if op == "cor":
                    s.cor += 1
                elif op == "sub":
                    s.sub += 1
                else:
                    s.dele += 1
        return res

    def overall(self) -> ErrorCounts:
        tot = ErrorCounts()
        for c in self.per_token.values():
            tot += c
        return tot

    def cluster(self, words: Iterable[str]) -> ErrorCounts:
        tot = ErrorCounts()
        for w in words:
            if w in self.per_token:
                tot += self.per_token[w]
        return tot


def _flag(v: str) -> bool:
    v = v.lower()
    return v == "true" or v != "0"


def _read_keyword_clusters(path: str):
    """``<Name> w1 w2 ... </Name>`` blocks.  (The reference's loop calls ``.decode`` on a str and cannot run under
    Python 3, compute-wer.py:521; this is the behaviour it documents.)"""
    name, words = "", []
    for line in open(path, "r", encoding="utf-8"):
        for tok in line.rstrip("\n").split():
            if tok[:2] == "</" and tok[-1] == ">" and tok.lstrip("</").rstrip(">") == name:
                yield name, words
                name, words = "", []
            elif tok[0] == "<" and tok[-1] == ">" and name == "":
                name, words = tok.lstrip("<").rstrip(">"), []
            else:
                words.append(tok)


def score_files(ref_file: str, hyp_file: str, *, giga: bool = False, case_sensitive: bool = False,
                tochar: bool = False, verbose: int = 1, ignore_file: Optional[str] = None,
                split_file: Optional[str] = None, cluster_file: str = "", max_words_per_line: int = sys.maxsize,
                remove_tag: bool = True, padding_symbol: str = " ", out=None) -> ErrorCounts:
    """Score ``hyp_file`` against ``ref_file`` (``utt-id word word ...`` per line; utterances missing from the
    hypothesis are skipped) and write the report to ``out`` (default stdout).  Returns the overall counts."""
    out = out or sys.stdout
    w = out.write
    ignore_words = set()
    if ignore_file:
        for line in open(ignore_file, "r", encoding="utf-8"):
            if line.strip():
                ignore_words.add(line.strip())
    if not case_sensitive:
        ignore_words = {x.upper() for x in ignore_words}
    split = None
    if split_file:
        split = {}
        for line in open(split_file, "r", encoding="utf-8"):
            ws = line.strip().split()
            if len(ws) >= 2:
                split[ws[0]] = ws[1:]
        if not case_sensitive:
            split = {k.upper(): [x.upper() for x in v] for k, v in split.items()}

    def units(line: str, for_ref: bool) -> List[str]:
        if tochar:
            return characterize(line)
        return line.rstrip("\n").split() if for_ref else line.strip().split()

    def clean(tokens: List[str]) -> List[str]:
        if giga:
            return giga_post_process(" ".join(tokens)).split()
        return normalize(tokens, ignore_words, case_sensitive, split, remove_tag)

    hyp = {}
    for line in open(hyp_file, "r", encoding="utf-8"):
        arr = units(line, False)
        if arr:
            hyp[arr[0]] = clean(arr[1:])

    scorer = WerScorer()
    for line in open(ref_file, "r", encoding="utf-8"):
        arr = units(line, True)
        if not arr or arr[0] not in hyp:
            continue
        fid = arr[0]
        res = scorer.add(fid, clean(arr[1:]), hyp[fid])
        if not verbose:
            continue
        w("\nutt: %s\n" % fid)
        w("WER: %4.2f %% %s\n" % (res.counts.wer, res.counts.line()))
        pads_l = [max(display_width(a), display_width(b)) - display_width(a) for a, b in zip(res.lab, res.rec)]
        pads_r = [max(display_width(a), display_width(b)) - display_width(b) for a, b in zip(res.lab, res.rec)]
        n, pos = len(res.lab), 0
        while pos < n:
            end = min(n, pos + max_words_per_line)
            tag = "(%s)" % fid.encode("utf-8") if verbose > 1 else ""
            w("lab%s: " % tag + "".join(t + padding_symbol * p + " " for t, p in zip(res.lab[pos:end], pads_l[pos:end])) + "\n")
            w("rec%s: " % tag + "".join(t + padding_symbol * p + " " for t, p in zip(res.rec[pos:end], pads_r[pos:end])) + "\n\n")
            pos = end

    bar = "=" * 75
    if verbose:
        w(bar + "\n\n")
    tot = scorer.overall()
    w("Overall -> %4.2f %% %s\n" % (tot.wer, tot.line()))
    if not verbose:
        w("\n")
        return tot
    for name, words in scorer.clusters.items():
        c = scorer.cluster(words)
        w("%s -> %4.2f %% %s\n" % (name, c.wer, c.line()))
    if cluster_file:
        for name, words in _read_keyword_clusters(cluster_file):
            c = scorer.cluster(words)
            w("%s -> %4.2f %% %s\n" % (name, c.wer, c.line()))
    w("\n" + bar + "\n")
    return tot


def main(argv: List[str]) -> int:
    if not argv:
        print(__doc__)
        return 0
    kw = dict(giga=False)
    args = list(argv)
    while len(args) > 2:
        a = args.pop(0)
        key, _, val = a.partition("=")
        if a == "--giga":
            kw["giga"] = True
        elif key == "--maxw":
            kw["max_words_per_line"] = int(val)
        elif key == "--rt":
            kw["remove_tag"] = _flag(val)
        elif key == "--cs":
            kw["case_sensitive"] = _flag(val)
        elif key == "--cluster":
            kw["cluster_file"] = val
        elif key == "--splitfile":
            kw["split_file"] = val
        elif key == "--ig":
            kw["ignore_file"] = val
        elif key == "--char":
            kw["tochar"] = _flag(val)
        elif key == "--v":
            try:
                kw["verbose"] = int(val)
            except ValueError:
                kw["verbose"] = 1 if _flag(val) else 0
        elif key == "--padding-symbol":
            kw["padding_symbol"] = {"space": " ", "underline": "_"}.get(val.lower(), " ")
        # unknown switches are ignored, as in the reference
    score_files(args[0], args[1], **kw)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
