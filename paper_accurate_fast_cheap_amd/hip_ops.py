"""torch-tensor front ends of the glue kernels in include/pafc_encoder_ops.h (GPU only, no fallback)."""
from typing import Optional

import torch

from . import _lib
from ._lib import c_int, c_void_p


def _bind():
    L = _lib.lib()
    if getattr(L, "_pafc_ops_bound", False):
        return L
    P, I = c_void_p, c_int
    _lib._sig(L.pafc_dwconv1d_cl, I, I, I, I, I, I, I, I, P, P, P, P, I, P, P)
    L._pafc_ops_bound = True
    return L


def depthwise_conv1d_cl(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], left_pad: int,
                        out_len: int, glu: bool = False, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Channels-last depthwise conv1d: x (B, T, C) [or (B, T, 2C) with glu], weight (C, 1, K) -> (B, out_len, C)."""
    _lib.require_gpu(x, weight, bias, lens)
    B, T, Cx = x.shape
    C = Cx // 2 if glu else Cx
    K = weight.shape[-1]
    if weight.shape != (C, 1, K) or weight.dtype != x.dtype or (bias is not None and bias.dtype != x.dtype):
        raise _lib.PafcError("depthwise weight must be (C, 1, K) in the activation dtype")
    if lens is not None and lens.dtype != torch.int32:
        raise _lib.PafcError("lens must be int32")
    y = torch.empty(B, out_len, C, dtype=x.dtype, device=x.device)
    rc = _bind().pafc_dwconv1d_cl(_lib.dtype_code(x.dtype), B, T, C, K, left_pad, out_len, _lib.ptr(x),
                                  _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(y), int(glu), _lib.ptr(lens),
                                  _lib.stream_of(x))
    _lib.check(rc, "pafc_dwconv1d_cl")
    return y
