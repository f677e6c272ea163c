"""Tokenizer interface of the decode path (``wenet/text/base_tokenizer.py:7-44``): ``tokenize`` = text -> (tokens,
ids), ``detokenize`` = ids -> (text, tokens); concrete classes provide the four conversions."""
from typing import Dict, List, Tuple


class BaseTokenizer:

    def tokenize(self, line: str) -> Tuple[List[str], List[int]]:
        tokens = self.text2tokens(line)
        return tokens, self.tokens2ids(tokens)

    def detokenize(self, ids: List[int]) -> Tuple[str, List[str]]:
        tokens = self.ids2tokens(ids)
        return self.tokens2text(tokens), tokens

    def text2tokens(self, line: str) -> List[str]:
        raise NotImplementedError

    def tokens2text(self, tokens: List[str]) -> str:
        raise NotImplementedError

    def tokens2ids(self, tokens: List[str]) -> List[int]:
        raise NotImplementedError

    def ids2tokens(self, ids: List[int]) -> List[str]:
        raise NotImplementedError

    def vocab_size(self) -> int:
        raise NotImplementedError

    @property
    def symbol_table(self) -> Dict[str, int]:
        raise NotImplementedError
