"""Symbol-table / list readers (host side of the token -> text step).

Mirrors ``wenet/utils/file_utils.py:18-68`` of the reference: ``read_lists``,
``read_non_lang_symbols`` (symbols must look like ``{x}``, ``<x>`` or ``[x]``) and
``read_symbol_table`` (``token id`` per line).
"""
import re
from typing import Dict, List, Optional

NON_LANG_SYM = re.compile(r"(\[[^\[\]]+\]|<[^<>]+>|{[^{}]+})")


class BadSymbolFormat(Exception):
    pass


def read_lists(list_file) -> List[str]:
    with open(list_file, "r", encoding="utf8") as fin:
        return [line.strip() for line in fin]


def read_non_lang_symbols(non_lang_sym_path: Optional[str]) -> List[str]:
    if non_lang_sym_path is None:
        return []
    syms = read_lists(non_lang_sym_path)
    for sym in syms:
        if NON_LANG_SYM.fullmatch(sym) is None:
            raise BadSymbolFormat(
                f"non-linguistic symbols must be written {{xxx}}, <xxx> or [xxx]; got '{sym}'")
    return syms


def read_symbol_table(symbol_table_file) -> Dict[str, int]:
    table = {}
    with open(symbol_table_file, "r", encoding="utf8") as fin:
        for line in fin:
            arr = line.strip().split()
            if len(arr) != 2:
                raise ValueError(f"symbol table line must be 'token id': {line!r}")
            table[arr[0]] = int(arr[1])
    return table
