"""Character tokenizer (``wenet/text/char_tokenizer.py:9-84``); also the id <-> token table the BPE tokenizer uses.

Behaviour kept: non-linguistic symbols (``{x}``/``<x>``/``[x]``) are split out of the UPPER-cased line and kept whole
when listed; a space becomes ``▁``; unknown tokens map to ``unk`` if the table has it and are dropped otherwise.
"""
from typing import Dict, List, Optional, Union

from ..utils.file_utils import NON_LANG_SYM, read_non_lang_symbols, read_symbol_table
from .base_tokenizer import BaseTokenizer


class CharTokenizer(BaseTokenizer):

    def __init__(self, symbol_table: Union[str, Dict], non_lang_syms: Optional[Union[str, List]] = None,
                 split_with_space: bool = False, connect_symbol: str = "", unk: str = "<unk>") -> None:
        self.non_lang_syms_pattern = NON_LANG_SYM if non_lang_syms is not None else None
        self._symbol_table = symbol_table if isinstance(symbol_table, dict) else read_symbol_table(symbol_table)
        self.non_lang_syms = non_lang_syms if isinstance(non_lang_syms, list) else read_non_lang_symbols(non_lang_syms)
        self.char_dict = {v: k for k, v in self._symbol_table.items()}
        self.split_with_space = split_with_space
        self.connect_symbol = connect_symbol
        self.unk = unk

    def text2tokens(self, line: str) -> List[str]:
        line = line.strip()
        if self.non_lang_syms_pattern is not None:
            parts = [w for w in self.non_lang_syms_pattern.split(line.upper()) if len(w.strip()) > 0]
        else:
            parts = [line]
        tokens = []
        for part in parts:
            if part in self.non_lang_syms:
                tokens.append(part)
                continue
            pieces = part.split(" ") if self.split_with_space else part
            for ch in pieces:
                tokens.append("▁" if ch == " " else ch)
        return tokens

    def tokens2text(self, tokens: List[str]) -> str:
        return self.connect_symbol.join(tokens)

    def tokens2ids(self, tokens: List[str]) -> List[int]:
        ids = []
        for tok in tokens:
            if tok in self._symbol_table:
                ids.append(self._symbol_table[tok])
            elif self.unk in self._symbol_table:
                ids.append(self._symbol_table[self.unk])
        return ids

    def ids2tokens(self, ids: List[int]) -> List[str]:
        return [self.char_dict[i] for i in ids]

    def vocab_size(self) -> int:
        return len(self.char_dict)

    @property
    def symbol_table(self) -> Dict[str, int]:
        return self._symbol_table
