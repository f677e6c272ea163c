from .base_tokenizer import BaseTokenizer
from .char_tokenizer import CharTokenizer
from .rev_bpe_tokenizer import RevBpeTokenizer
from .init_tokenizer import init_tokenizer

__all__ = ["BaseTokenizer", "CharTokenizer", "RevBpeTokenizer", "init_tokenizer"]
