"""SentencePiece tokenizer of the paper's GigaSpeech recipes (``wenet/text/rev_bpe_tokenizer.py:9-81``).

text -> tokens: strip, drop ``<sw>`` tags (``remove_sw``), rewrite ``<unk>`` as ``<unknown>``
(``replace_unk_as_unknown``), then SentencePiece pieces.  tokens -> text: join, ``▁`` -> space, strip.  The id table is
the ``units.txt`` symbol table (not the SentencePiece ids), exactly as in the reference.  The SentencePiece model is
loaded lazily so that the object can cross a process boundary before first use.
"""
from typing import Dict, List, Optional, Union

from .char_tokenizer import CharTokenizer


class RevBpeTokenizer(CharTokenizer):

    def __init__(self, bpe_model: str, symbol_table: Union[str, Dict], non_lang_syms: Optional[Union[str, List]] = None,
                 split_with_space: bool = False, connect_symbol: str = "", unk: str = "<unk>",
                 full_config: Optional[Dict] = None) -> None:
        super().__init__(symbol_table, non_lang_syms, split_with_space, connect_symbol, unk)
        full_config = full_config or {}
        self.remove_sw = full_config.get("remove_sw", True)
        self.replace_unk_as_unknown = full_config.get("replace_unk_as_unknown", True)
        self._model = bpe_model
        self.bpe_model = None

    def _build_sp(self):
        if self.bpe_model is None:
            import sentencepiece as spm
            self.bpe_model = spm.SentencePieceProcessor()
            self.bpe_model.load(self._model)

    def text2tokens(self, line: str) -> List[str]:
        self._build_sp()
        line = line.strip()
        if self.remove_sw:
            line = line.replace("<sw>", "").replace("  ", " ").strip()
        if self.replace_unk_as_unknown:
            line = line.replace("<unk>", "<unknown>")
        return self.bpe_model.encode(line, out_type=str)

    def tokens2text(self, tokens: List[str]) -> str:
        return self.connect_symbol.join(tokens).replace("▁", " ").strip()
