"""Character tokenizer of the decode tail: symbol-table lookups in both directions plus the character split
(behaviour of ``wenet/text/char_tokenizer.py:9-84``, checked against it through tests/golden/text/).  ``RevBpeTokenizer``
builds on the same tables.

What the behaviour is: with a list of non-linguistic symbols (written ``{x}``, ``<x>`` or ``[x]``) the line is upper-cased
and cut at bracketed spans, and a span that is in the list stays one token; everything else becomes one token per character
(a space is written ``\u2581``), or one token per space-separated word with ``split_with_space``.  Ids come from the symbol
table; a token outside it becomes ``unk`` when the table has that entry and is dropped when it does not.
"""
from typing import Dict, Iterator, List, Optional, Tuple, Union

from ..utils.file_utils import NON_LANG_SYM, read_non_lang_symbols, read_symbol_table
from .base_tokenizer import BaseTokenizer

_SPACE = "\u2581"


class CharTokenizer(BaseTokenizer):

    def __init__(self, symbol_table: Union[str, Dict], non_lang_syms: Optional[Union[str, List]] = None,
                 split_with_space: bool = False, connect_symbol: str = "", unk: str = "<unk>") -> None:
        self._symbol_table = dict(symbol_table) if isinstance(symbol_table, dict) else read_symbol_table(symbol_table)
        self.char_dict = {idx: tok for tok, idx in self._symbol_table.items()}
        self.non_lang_syms_pattern = NON_LANG_SYM if non_lang_syms is not None else None
        self.non_lang_syms = list(non_lang_syms) if isinstance(non_lang_syms, list) else read_non_lang_symbols(non_lang_syms)
        self.split_with_space = split_with_space
        self.connect_symbol = connect_symbol
        self.unk = unk

    def _spans(self, line: str) -> Iterator[Tuple[str, bool]]:
        """(text, is_listed_symbol) pieces of a stripped line, blank pieces dropped."""
        if self.non_lang_syms_pattern is None:
            yield line, line in self.non_lang_syms
            return
        for piece in self.non_lang_syms_pattern.split(line.upper()):
            if piece.strip():
                yield piece, piece in self.non_lang_syms

    def text2tokens(self, line: str) -> List[str]:
        tokens: List[str] = []
        for text, whole in self._spans(line.strip()):
            if whole:
                tokens.append(text)
            elif self.split_with_space:
                tokens.extend(text.split(" "))
            else:
                tokens.extend(_SPACE if ch == " " else ch for ch in text)
        return tokens

    def tokens2text(self, tokens: List[str]) -> str:
        return self.connect_symbol.join(tokens)

    def tokens2ids(self, tokens: List[str]) -> List[int]:
        table = self._symbol_table
        fallback = table.get(self.unk)
        ids = (table.get(tok, fallback) for tok in tokens)
        return [i for i in ids if i is not None]

    def ids2tokens(self, ids: List[int]) -> List[str]:
        return [self.char_dict[i] for i in ids]

    def vocab_size(self) -> int:
        return len(self.char_dict)

    @property
    def symbol_table(self) -> Dict[str, int]:
        return self._symbol_table
