"""``init_tokenizer(configs)`` for the tokenizers the paper's recipes use (``wenet/utils/init_tokenizer.py:26-62``):
``char`` (default) and ``rev_bpe``.  The other reference tokenizers (whisper, paraformer, plain bpe) belong to model
families outside the hot path and raise ``NotImplementedError`` here."""
from .char_tokenizer import CharTokenizer
from .rev_bpe_tokenizer import RevBpeTokenizer


def init_tokenizer(configs):
    kind = configs.get("tokenizer", "char")
    conf = configs["tokenizer_conf"]
    if kind == "char":
        return CharTokenizer(conf["symbol_table_path"], conf["non_lang_syms_path"],
                             split_with_space=conf.get("split_with_space", False),
                             connect_symbol=conf.get("connect_symbol", ""))
    if kind == "rev_bpe":
        return RevBpeTokenizer(conf["bpe_path"], conf["symbol_table_path"], conf["non_lang_syms_path"],
                               split_with_space=conf.get("split_with_space", False), full_config=conf)
    raise NotImplementedError(f"tokenizer '{kind}' is outside the accelerated path (char and rev_bpe are provided)")
