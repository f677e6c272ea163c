"""String -> class registries: the reference's plugin API (wenet/utils/class_utils.py:40-98).

`WENET_ATTENTION_CLASSES[encoder_conf.selfattention_layer_type]` is how a YAML picks the attention slot
(encoder.py:589-602).  The recurrent keys map to the MI355X implementations; the multi-head-attention keys of
the reference ("selfattn", "rel_selfattn", ...) are the baseline being replaced and are out of scope here.
`install_into(wenet_class_utils)` patches a live reference registry instead."""
import torch

from ..rwkv_v6.rwkv_wrapper import RWKV_TmixWrapper
from ..rwkv_v6.rwkv_wrapper_bidirectional import RWKV_TmixWrapper_bidirectional
from ..rwkv_v6.rwkv_wrapper_bidirectional_direction_dropout import (
    RWKV_TmixWrapper_bidirectional_direction_dropout, RWKV_TmixWrapper_bidirectional_direction_dropout_both)
from ..transformer.mamba2 import MambaAttWrapper
from ..transformer.embedding import NoPositionalEncoding, PositionalEncoding, RelPositionalEncoding
from ..transformer.subsampling import Conv2dSubsampling4, LinearNoSubsampling

WENET_ACTIVATION_CLASSES = {
    "hardtanh": torch.nn.Hardtanh,
    "tanh": torch.nn.Tanh,
    "relu": torch.nn.ReLU,
    "selu": torch.nn.SELU,
    "swish": getattr(torch.nn, "SiLU"),
    "gelu": torch.nn.GELU,
}

WENET_SUBSAMPLE_CLASSES = {
    "linear": LinearNoSubsampling,
    "conv2d": Conv2dSubsampling4,
}

WENET_EMB_CLASSES = {
    "embed": PositionalEncoding,
    "abs_pos": PositionalEncoding,
    "rel_pos": RelPositionalEncoding,
    "no_pos": NoPositionalEncoding,
}

WENET_ATTENTION_CLASSES = {
    "rwkv_tmix60": RWKV_TmixWrapper,
    "rwkv_tmix60_bidirectional": RWKV_TmixWrapper_bidirectional,
    "rwkv_tmix60_bidirectional2": RWKV_TmixWrapper_bidirectional,  # same arithmetic (bidirectional2.py:95-150)
    "rwkv_tmix60_dir_layer_drop": RWKV_TmixWrapper_bidirectional_direction_dropout,
    "rwkv_tmix60_dir_layer_drop_both": RWKV_TmixWrapper_bidirectional_direction_dropout_both,
    "mamba_att": MambaAttWrapper,   # Mamba-2 on the same scan kernel; arithmetic third-party -> parity unpinned
}


def install_into(reference_class_utils) -> None:
    """Point a loaded `wenet.utils.class_utils` at the MI355X slot classes (drop-in for the hot path only)."""
    reference_class_utils.WENET_ATTENTION_CLASSES.update(WENET_ATTENTION_CLASSES)
