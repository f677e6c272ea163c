"""torch-tensor front ends of the glue kernels in include/pafc_encoder_ops.h (GPU only, no fallback)."""
import contextlib
import ctypes
import os
import threading
import weakref
from typing import Optional

import torch

from . import _lib
from ._lib import c_int, c_void_p


# ---- dispatch thresholds: ONE table ---------------------------------------------------------------------------------------
# Which kernel family serves a projection depends on how many rows it has.  Every threshold lives here, with the record that
# justifies it; `PAFC_DISPATCH="name=value,name=value"` overrides any of them for A/B runs (the per-name variables of earlier
# rounds -- PAFC_OWN_GEMM_MIN_ROWS, PAFC_LDS_RESIDENT_MIN_ROWS, PAFC_SPLIT_GEMM_MIN_ROWS, PAFC_SKINNY_MAX_ROWS -- are still
# read).  bench.py sets none of them: what it measures is this table.
#
#   name                   default  meaning, and where the number comes from
#   skinny_max_rows            640  bf16 projections of a streaming chunk step with at most this many rows run on
#                                   csrc/gemm_skinny.hip (one launch on the ~4.7 us floor).  8 streams x 78 rows still win over
#                                   tiled kernels, 16 lose: profiles/r03g_streaming_streams_vs_few_rows_limit.txt
#   own_gemm_min_rows            1  bf16 projections with at least this many rows (that are not few-rows launches) run on the
#                                   hand-written tiled GEMMs (gemm_ph.hip 256-wide tiles when they fill the chip, else
#                                   gemm_bf16.hip 128 x 128 / 128 x 64 / 64 x 64).  Was 8192 while the small kernel only had
#                                   128 x 128 tiles; with the tile variants it ties or beats the library's picks from 1 024
#                                   rows up on every layer shape but w_2 (K = 2048: 14.5 vs 10.5 us at 1 024 rows, 20.7 vs
#                                   17.4 at 3 992) and wins the stacked r/k/v product by 5-7 us:
#                                   profiles/r04d_gemm_mid_rows_own_tiles_vs_library.txt.  No library GEMM is left in a bf16 pass.
#   lds_resident_min_rows     8192  the one-pass LoRA kernels (128 KiB of weights staged into LDS per block) from this many
#                                   rows on, below it shift/lerp + small GEMMs; the chunk step always takes the one-pass
#                                   kernels (fewer launches).  At 3 992 rows the one-pass down-projection takes 27.6 us
#                                   against 6.0 + 6.5 for the two small kernels (profiles/r04e_windows_2000x8_kernels_*.txt);
#                                   a c2 pass measured 2048 / 4096 / 8192 within 1 % (profiles/r03c_bench_c2_knobs.txt)
#   split_gemm_min_rows       1024  fp32 activations of a model with the bf16 slot on the bf16 matrix cores as hi + lo planes
#                                   (3 MFMAs per product, ~2^-16 relative) from this many rows on; below it exact fp32
#                                   products on the fp32 matrix cores (csrc/gemm_f32.hip; pure-fp32 models -- rwkv_do_bfloat16:
#                                   False, the 1e-3 parity bar -- always take those; until round 6 these were library GEMMs).  Was 16384 until round 5: at 1 536 - 24 000 rows the split kernel at its best
#                                   tile height takes 28 / 35 / 48 / 78 / 90 / 138 us for w_1 where the library takes 37 / 76 /
#                                   137 / 256 / 273 / 507 (profiles/r05_split_mid_rows.txt; only w_2 below 4 000 rows is faster
#                                   on the library, 32-74 vs 72-74 us): 2 000-frame windows x 8 20 600 -> 33 200 audio-sec/sec,
#                                   c2 in this precision 44 200; DESIGN section 4 "split operands"
#   ln_fold_min_rows         24576  unmasked bf16 inputs of at least this many rows take the schedule with three LayerNorms
#                                   folded into the 256-wide GEMMs either side of them (needs full grids of 256 x 256 tiles:
#                                   3 x the row count at which they fill the chip): profiles/r03d / r03e
#   dwconv_ln_silu_max_rows  24575  up to this many rows the conv module's LayerNorm + SiLU is the depthwise convolution's epilogue
#                                   (one launch and one round trip less per layer: c2 +0.4-2 %, 2 000-frame windows +1.3 %, same box);
#                                   the 30-minute sequence measured the same or 0.2 ms slower with it (the convolution kernel is
#                                   latency-bound at two waves per SIMD and the epilogue's two block reductions add to that), so
#                                   long inputs keep the two kernels
#   split_small_max_rows      8192  split-operand (and bf16 -> fp32) products of at most this many rows AND at most 2^22 outputs
#                                   (w_1, N = 2048: 2 048 rows) run on the small tiles of csrc/gemm_bf16.hip
#                                   (pafc_gemm_bf16_f32out: 64 x 64 / 128 x 64 tiles, three-stage ring) instead of the 256-wide
#                                   phase-pipelined kernel, whose tiles cost ~25 us (K = 512) / ~72 us (K = 2048) each however few
#                                   there are: at 3 992 rows 17 / 21 / 53 us against 27 / 28 / 72 for pointwise_conv2 / pointwise_conv1
#                                   / w_2, w_1 22.7 against 30.5 at 1 996 rows and 37.5 against 30.4 at 2 500.  Round 6, late: 8 192
#                                   (N = 512 products; long-K ones on 128 x 128 tiles with 2-4 K shares per tile: w_2 76.9 -> 49.3 us at
#                                   5 000 rows, 82.5 -> 66.0 at 7 984: profiles/r06z_split_small_tile_variants.txt)
#                                   (profiles/r06n_split_small_rows.txt)
#   split_layers_min_rows      256  the LAYERS of a model with the bf16 slot take the split-operand schedule
#                                   (fused.layer_forward_split) from this many rows on -- since round 6 also below
#                                   split_gemm_min_rows, where their projections run on the small tiles (a single 2 000-frame
#                                   window is 499 rows: exact fp32 products on the fp32 matrix cores cost 12-55 us each there)
# C side (csrc/gemm_bf16.hip): which GEMM family takes a problem is decided by rounds -- the 128-wide kernel while its 128 x 128
# tiles fit one round of two per CU, the 256-wide phase-pipelined kernel beyond (profiles/r04s_gemm_tile_choice_by_rows.txt);
# PAFC_PH_MIN_FILL=<percent> (round 3's rule: 256-wide tiles must cover that share of the CUs) and PAFC_GEMM_TILE (force a tile
# of the small kernel) are A/B switches of the kernels themselves.
DISPATCH = dict(skinny_max_rows=640, own_gemm_min_rows=1, lds_resident_min_rows=8192, split_gemm_min_rows=1024,
                ln_fold_min_rows=24576, dwconv_ln_silu_max_rows=24575, split_small_max_rows=8192, split_layers_min_rows=256)


def _load_dispatch():
    legacy = dict(skinny_max_rows="PAFC_SKINNY_MAX_ROWS", own_gemm_min_rows="PAFC_OWN_GEMM_MIN_ROWS",
                  lds_resident_min_rows="PAFC_LDS_RESIDENT_MIN_ROWS", split_gemm_min_rows="PAFC_SPLIT_GEMM_MIN_ROWS")
    for name, env in legacy.items():
        if os.environ.get(env):
            DISPATCH[name] = int(os.environ[env])
    for item in filter(None, os.environ.get("PAFC_DISPATCH", "").split(",")):
        name, _, value = item.partition("=")
        if name.strip() not in DISPATCH:
            raise ValueError(f"PAFC_DISPATCH: unknown threshold {name!r} (known: {sorted(DISPATCH)})")
        DISPATCH[name.strip()] = int(value)


_load_dispatch()


def train_gemms_own() -> bool:
    """PAFC_TRAIN_OWN_GEMMS=0: forward and input-gradient products of the training step's nn.Linear go to the library (A/B)."""
    return os.environ.get("PAFC_TRAIN_OWN_GEMMS", "1") != "0"


def train_kernels_enabled() -> bool:
    """PAFC_TRAIN_KERNELS=0: the training step differentiates through the framework's own operators (A/B measurements)."""
    import os
    return os.environ.get("PAFC_TRAIN_KERNELS", "1") != "0"


def _bind():
    L = _lib.lib()
    if getattr(L, "_pafc_ops_bound", False):
        return L
    P, I = c_void_p, c_int
    _lib._sig(L.pafc_dwconv1d_cl, I, I, I, I, I, I, I, I, P, P, P, P, I, P, P)
    from ctypes import c_long, c_size_t
    L.pafc_dwconv1d_cl_wgrad_workspace_bytes.restype = c_size_t
    L.pafc_dwconv1d_cl_wgrad_workspace_bytes.argtypes = [I, I, I, I]
    _lib._sig(L.pafc_dwconv1d_cl_wgrad, I, I, I, I, I, I, I, I, P, c_long, P, P, P, P, c_size_t, P)
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


def dwconv_ln_silu_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Does the convolution kernel carry the conv module's LayerNorm + SiLU for this input (bf16, 512 channels, k = 15 / 31)?"""
    return (x.numel() // x.shape[-1] <= DISPATCH["dwconv_ln_silu_max_rows"] and x.dtype == torch.bfloat16 and x.is_cuda
            and x.shape[-1] == 512 and weight.shape[-1] in (15, 31))


def depthwise_conv1d_cl_ln_silu(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], left_pad: int, out_len: int,
                                gamma: torch.Tensor, beta: torch.Tensor, eps: float, lens: Optional[torch.Tensor] = None):
    """SiLU(LayerNorm(depthwise_conv1d_cl(x, ...))) in one pass (convolution.py:131-138 with cnn_module_norm: layer_norm): the
    convolution's epilogue normalises its own rows.  x (B, T, 512) bf16, weight (512, 1, K), K in {15, 31}."""
    _lib.require_gpu(x, weight, bias, gamma, beta, lens)
    B, T, C = x.shape
    K = weight.shape[-1]
    if not dwconv_ln_silu_ok(x, weight) or weight.shape != (C, 1, K) or any(
            t is not None and (t.dtype != x.dtype or not t.is_contiguous()) for t in (weight, bias, gamma, beta)):
        raise _lib.PafcError("depthwise_conv1d_cl_ln_silu: bf16, C = 512, K in (15, 31), contiguous parameters in the activation dtype")
    if x.stride(2) != 1 or x.stride(0) != T * x.stride(1):
        raise _lib.PafcError("depthwise_conv1d_cl_ln_silu: x (B, T, C) with unit channel stride")
    if lens is not None and lens.dtype != torch.int32:
        raise _lib.PafcError("lens must be int32")
    L = _bind()
    if not getattr(L, "_pafc_dwln_bound", False):
        from ctypes import c_float, c_long
        _lib._sig(L.pafc_dwconv1d_cl_ln_silu, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_long, c_void_p,
                  c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p)
        L._pafc_dwln_bound = True
    y = torch.empty(B, out_len, C, dtype=x.dtype, device=x.device)
    rc = L.pafc_dwconv1d_cl_ln_silu(_lib.dtype_code(x.dtype), B, T, C, K, left_pad, out_len, _lib.ptr(x), x.stride(1), _lib.ptr(weight),
                                    _lib.ptr(bias), _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(y), _lib.ptr(lens),
                                    _lib.stream_of(x))
    _lib.check(rc, "pafc_dwconv1d_cl_ln_silu")
    return y


def depthwise_conv1d_cl_wgrad(x: torch.Tensor, dy: torch.Tensor, K: int, left_pad: int, want_bias: bool = True):
    """(dw (C, 1, K), dbias (C) or None) in float32 for y = depthwise_conv1d_cl(x, w, bias, left_pad, dy.shape[1])."""
    _lib.require_gpu(x, dy)
    B, T, C = x.shape
    if dy.shape[0] != B or dy.shape[2] != C or dy.dtype != x.dtype or not dy.is_contiguous() or x.stride(2) != 1 \
            or x.stride(0) != T * x.stride(1):
        raise _lib.PafcError("dy must be contiguous (B, T_out, C) in x's dtype; x (B, T, C) with unit channel stride")
    L = _bind()
    nbytes = L.pafc_dwconv1d_cl_wgrad_workspace_bytes(B, dy.shape[1], C, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = torch.empty(C, 1, K, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device) if want_bias else None
    rc = L.pafc_dwconv1d_cl_wgrad(_lib.dtype_code(x.dtype), B, T, C, K, left_pad, dy.shape[1], _lib.ptr(x), x.stride(1),
                                  _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), nbytes, _lib.stream_of(x))
    _lib.check(rc, "pafc_dwconv1d_cl_wgrad")
    return dw, db


class _DepthwiseConvCL(torch.autograd.Function):
    """Autograd over the channels-last depthwise convolution: dx is the forward kernel on dy with the taps reversed and
    left_pad' = K - 1 - left_pad, dw / dbias the reduction kernel (what autograd derives for nn.Conv1d,
    convolution.py:131)."""

    @staticmethod
    def forward(ctx, x, weight, bias, left_pad, out_len):
        # parameters in the activation dtype (what autocast does for nn.Conv1d); the cast sits inside the Function so that the
        # gradients come back in the parameters' own dtype without a cast node each, and bf16 takes the kept copies
        w = _param_as(weight, x.dtype)
        b = None if bias is None else _param_as(bias, x.dtype)
        ctx.save_for_backward(x, w)
        ctx.left_pad, ctx.has_bias = left_pad, bias is not None
        ctx.w_dtype, ctx.b_dtype = weight.dtype, (None if bias is None else bias.dtype)
        return depthwise_conv1d_cl(x, w, b, left_pad, out_len)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        K = weight.shape[-1]
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = depthwise_conv1d_cl(dy, weight.flip(-1).contiguous(), None, K - 1 - ctx.left_pad, x.shape[1])
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = depthwise_conv1d_cl_wgrad(x, dy, K, ctx.left_pad, want_bias=ctx.has_bias)
            dw = dw.to(ctx.w_dtype)
            db = db.to(ctx.b_dtype) if db is not None else None
        return dx, dw, db, None, None


def depthwise_conv1d_cl_autograd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], left_pad: int,
                                 out_len: int) -> torch.Tensor:
    """Training-side entry: parameters are cast to the activation dtype (what autocast does for nn.Conv1d), gradients
    come back in the parameters' dtype."""
    return _DepthwiseConvCL.apply(x.contiguous(), weight, bias, left_pad, out_len)


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, eps: float, dx_add: Optional[torch.Tensor] = None):
    """(dx in x's dtype, dgamma, dbeta in float32) of y = LayerNorm(x) over the last axis; dx_add (like x): added to dx in the same
    pass (the gradient that reaches x past the norm)."""
    _lib.require_gpu(x, dy, gamma, dx_add)
    if dx_add is not None and (dx_add.dtype != x.dtype or dx_add.shape != x.shape or not dx_add.is_contiguous()):
        raise _lib.PafcError("layernorm_bwd: dx_add contiguous, shaped and typed like x")
    C = x.shape[-1]
    rows = x.numel() // C
    if dy.shape != x.shape or gamma.dtype != x.dtype or gamma.shape != (C,):
        raise _lib.PafcError("layernorm_bwd: dy shaped like x, gamma (C) in x's dtype")
    L = _lib.lib()
    if not getattr(L, "_pafc_lnb_bound", False):
        from ctypes import c_float, c_long, c_size_t
        L.pafc_layernorm_bwd_workspace_bytes.restype = c_size_t
        L.pafc_layernorm_bwd_workspace_bytes.argtypes = [c_long, c_int]
        _lib._sig(L.pafc_layernorm_bwd_add, c_int, c_int, c_int, c_long, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                  c_void_p, c_void_p, c_size_t, c_void_p)
        L._pafc_lnb_bound = True
    nbytes = L.pafc_layernorm_bwd_workspace_bytes(rows, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dx = torch.empty_like(x)
    dgb = torch.empty(2, C, dtype=torch.float32, device=x.device)
    rc = L.pafc_layernorm_bwd_add(_lib.dtype_code(x.dtype), _lib.dtype_code(dy.dtype), rows, C, _lib.ptr(x), _lib.ptr(dy),
                                  _lib.ptr(gamma), float(eps), _lib.ptr(dx_add), _lib.ptr(dx), _lib.ptr(dgb), _lib.ptr(ws), nbytes,
                                  _lib.stream_of(x))
    _lib.check(rc, "pafc_layernorm_bwd")
    return dx, dgb[0], dgb[1]


class _LayerNormTrain(torch.autograd.Function):
    """nn.LayerNorm for the GPU training step: forward = the inference kernel (one pass, optionally straight to bf16 for
    a consumer that would cast anyway), backward = one pass over (x, dy) + a small reduction, instead of the framework's
    fp32 forward, cast and three backward kernels.  with_skip (round 6): x is handed on as a second output -- the residual
    path of a pre-norm branch, x + f(norm(x)) -- so that its gradient comes back HERE and is added inside the backward
    kernel (pafc_layernorm_bwd_add) instead of by autograd's accumulation pass (one (B, T, C) fp32 add per branch)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype, with_skip=False):
        g, b = _param_as(weight, x.dtype), _param_as(bias, x.dtype)
        _, y, _ = add_layernorm(x, None, 1.0, g, b, out_dtype=out_dtype, want_x=False, eps=eps)
        ctx.save_for_backward(x, g)
        ctx.eps, ctx.w_dtype, ctx.b_dtype = eps, weight.dtype, bias.dtype
        return (y, x.view_as(x)) if with_skip else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, g = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros_like(x)
        if dskip is not None and (dskip.dtype != x.dtype or not dskip.is_contiguous()):
            dskip = dskip.to(x.dtype).contiguous()
        dx, dg, db = layernorm_bwd(x, dy.contiguous(), g, ctx.eps, dx_add=dskip)
        if ctx.w_dtype == ctx.b_dtype and ctx.w_dtype != torch.float32:
            dgb = dg._base.to(ctx.w_dtype)          # (2, C): one cast for both (the bf16 ln_x of the slot)
            return dx, dgb[0], dgb[1], None, None, None
        return dx, dg.to(ctx.w_dtype), db.to(ctx.b_dtype), None, None, None


class _LnSiluTrain(torch.autograd.Function):
    """silu(LayerNorm(x)) of the conv module (convolution.py:136-138) for the GPU training step as one kernel each way
    (pafc_layernorm_silu_fwd / _bwd): x -- the depthwise convolution's bf16 output -- in, bf16 out, fp32 arithmetic against the
    norm's own parameters.  Under bf16 autocast the framework chain is cast, LayerNorm (fp32), SiLU, cast -- four passes over
    (B, T, C) forward and four more backward for the same values."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        L = _lib.lib()
        if not getattr(L, "_pafc_lns_bound", False):
            from ctypes import c_float, c_long, c_size_t
            _lib._sig(L.pafc_layernorm_silu_fwd, c_int, c_int, c_int, c_long, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p)
            _lib._sig(L.pafc_layernorm_silu_bwd, c_int, c_int, c_int, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                      c_void_p, c_void_p, c_size_t, c_void_p)
            L.pafc_layernorm_bwd_workspace_bytes.restype = c_size_t
            L.pafc_layernorm_bwd_workspace_bytes.argtypes = [c_long, c_int]
            L._pafc_lns_bound = True
        C = x.shape[-1]
        rows = x.numel() // C
        g, b = weight.detach().contiguous(), bias.detach().contiguous()
        y = torch.empty_like(x)
        _lib.check(L.pafc_layernorm_silu_fwd(_lib.dtype_code(x.dtype), _lib.dtype_code(g.dtype), rows, C, _lib.ptr(x), _lib.ptr(g), _lib.ptr(b),
                                             float(eps), _lib.ptr(y), _lib.stream_of(x)), "pafc_layernorm_silu_fwd")
        ctx.save_for_backward(x, g, b)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b = ctx.saved_tensors
        L = _lib.lib()
        C = x.shape[-1]
        rows = x.numel() // C
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        nbytes = L.pafc_layernorm_bwd_workspace_bytes(rows, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        dx = torch.empty_like(x)
        dgb = torch.empty(2, C, dtype=torch.float32, device=x.device)
        _lib.check(L.pafc_layernorm_silu_bwd(_lib.dtype_code(x.dtype), _lib.dtype_code(g.dtype), rows, C, _lib.ptr(x), _lib.ptr(dy), _lib.ptr(g),
                                             _lib.ptr(b), float(ctx.eps), _lib.ptr(dx), _lib.ptr(dgb), _lib.ptr(ws), nbytes, _lib.stream_of(x)),
                   "pafc_layernorm_silu_bwd")
        if g.dtype != torch.float32:
            dgb = dgb.to(g.dtype)
        return dx, dgb[0], dgb[1], None


def ln_silu_train_eligible(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor]) -> bool:
    """The conv module's LayerNorm + SiLU as one kernel each way: GPU training step, x contiguous bf16 (fp32 or bf16 parameters) or
    fp32 (fp32 parameters), C % 8 == 0, C <= 1024.  PAFC_TRAIN_LN_SILU=0: the separate LayerNorm and SiLU (A/B runs)."""
    if not (x.is_cuda and torch.is_grad_enabled() and weight is not None and bias is not None and train_kernels_enabled()
            and os.environ.get("PAFC_TRAIN_LN_SILU", "1") != "0"):
        return False
    if weight.dtype != bias.dtype or x.shape[-1] % 8 or x.shape[-1] > 1024 or not (x.requires_grad or weight.requires_grad):
        return False
    return ((x.dtype == torch.bfloat16 and weight.dtype in (torch.float32, torch.bfloat16))
            or (x.dtype == torch.float32 and weight.dtype == torch.float32))


def ln_silu_train(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    return _LnSiluTrain.apply(x.contiguous(), weight, bias, eps)


def layer_norm_train_eligible(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor]) -> bool:
    return (x.is_cuda and torch.is_grad_enabled() and weight is not None and bias is not None and train_kernels_enabled()
            and x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] % 8 == 0 and x.shape[-1] <= 1024
            and (x.requires_grad or weight.requires_grad))


def layer_norm_with_skip(x: torch.Tensor, weight, bias, eps: float, bf16_out: bool = False):
    """(LayerNorm(x), x) for a pre-norm residual branch x + f(norm(x)) in the GPU training step: use the SECOND value as the
    branch's residual input and its gradient is added inside the norm's backward kernel (see _LayerNormTrain).  None when the
    kernels do not take the call (the caller then norms and adds as usual)."""
    if not (layer_norm_train_eligible(x, weight, bias) and x.requires_grad and os.environ.get("PAFC_TRAIN_LN_SKIP", "1") != "0"):
        return None
    amp = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
    if amp and x.dtype == torch.bfloat16 and weight.dtype == torch.float32:
        return None                       # (autocast would run this norm in fp32 on a cast copy: not the residual stream itself)
    out_dtype = torch.bfloat16 if (bf16_out and amp) else x.dtype
    return _LayerNormTrain.apply(x.contiguous(), weight, bias, eps, out_dtype, True)


def layer_norm(x: torch.Tensor, weight, bias, eps: float, bf16_out: bool = False) -> torch.Tensor:
    """F.layer_norm over the last axis as the modules call it; in the GPU training step the kernels above.  bf16_out:
    the consumer is a projection that autocast would feed bf16 anyway -- under bf16 autocast the fp32 norm then writes
    bf16 directly (same values as the framework's fp32 result cast by the consumer)."""
    if layer_norm_train_eligible(x, weight, bias):
        amp = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
        if amp and x.dtype == torch.bfloat16 and weight.dtype == torch.float32:
            x = x.float()                 # autocast runs layer_norm in fp32
        out_dtype = torch.bfloat16 if (bf16_out and amp) else x.dtype
        return _LayerNormTrain.apply(x.contiguous(), weight, bias, eps, out_dtype, False)
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, out_dtype: torch.dtype = torch.float32, want_bias: bool = False):
    """dw (M, N) = dy^T @ x for dy (R, M), x (R, N) bf16 with unit column stride: nn.Linear's weight gradient;
    with want_bias also db (M) = dy.sum(0) from the same pass -> (dw, db).  dy (Z, R, M), x (Z, R, N): Z products in one launch
    pair -> dw (Z, M, N) (db (Z, M))."""
    if not (dy.is_cuda and x.is_cuda):
        raise _lib.PafcError("gemm_tn runs on the MI355X only; there is no CPU fallback")
    batched = dy.dim() == 3
    Z = dy.shape[0] if batched else 1
    R, M = dy.shape[-2], dy.shape[-1]
    N = x.shape[-1]
    if (dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or x.dim() != dy.dim() or x.shape[-2] != R or dy.stride(-1) != 1
            or x.stride(-1) != 1 or out_dtype not in (torch.float32, torch.bfloat16) or (batched and x.shape[0] != Z)):
        raise _lib.PafcError("gemm_tn: dy (R, M) and x (R, N) bf16 with unit column stride (or both with a leading batch); fp32 or bf16 output")
    L = _lib.lib()
    if not getattr(L, "_pafc_tn_bound", False):
        from ctypes import c_long, c_size_t
        L.pafc_gemm_tn_batched_workspace_bytes.restype = c_size_t
        L.pafc_gemm_tn_batched_workspace_bytes.argtypes = [c_long, c_int, c_int, c_int]
        _lib._sig(L.pafc_gemm_tn_bf16_batched, c_int, c_long, c_int, c_int, c_int, c_void_p, c_long, c_long, c_void_p, c_long, c_long,
                  c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p)
        L._pafc_tn_bound = True
    nbytes = L.pafc_gemm_tn_batched_workspace_bytes(R, M, N, Z)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    dw = torch.empty((Z, M, N) if batched else (M, N), dtype=out_dtype, device=dy.device)
    db = (torch.empty((Z, M) if batched else (M,), dtype=out_dtype, device=dy.device)) if want_bias else None
    rc = L.pafc_gemm_tn_bf16_batched(R, M, N, Z, _lib.ptr(dy), dy.stride(-2), dy.stride(0) if batched else 0, _lib.ptr(x), x.stride(-2),
                                     x.stride(0) if batched else 0, _lib.ptr(dw), _lib.ptr(db), _lib.dtype_code(out_dtype), _lib.ptr(ws),
                                     nbytes, _lib.stream_of(dy))
    _lib.check(rc, "pafc_gemm_tn_bf16")
    return (dw, db) if want_bias else dw


# ---- element-wise groups of the training step (csrc/train_elementwise.hip) ------------------------------------------------
_dropout_calls = 0       # fallback counter when the device generator exposes no offset


def _next_dropout_stream(device=None):
    """(seed, call index) of the next keep mask, drawn from torch's generator of the device: seed = its seed, index = its Philox
    offset, which is advanced by one step of four like any framework dropout.  `torch.manual_seed` therefore restarts the
    sequence, and an activation-checkpoint recompute -- which restores the device RNG state (`preserve_rng_state`) -- draws the
    masks of the original forward pass again (a process-wide counter would not)."""
    global _dropout_calls
    try:
        idx = device.index if device is not None and device.index is not None else torch.cuda.current_device()
        g = torch.cuda.default_generators[idx]
        off = int(g.get_offset())
        g.set_offset(off + 4)
        return int(g.initial_seed()) & 0xFFFFFFFFFFFFFFFF, off // 4 + 1
    except (RuntimeError, IndexError, AttributeError):
        _dropout_calls += 1
        return int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, _dropout_calls


def _bind_train_elementwise():
    L = _bind()
    if not getattr(L, "_pafc_te_bound", False):
        from ctypes import c_float, c_long, c_ulonglong
        _lib._sig(L.pafc_residual_dropout, c_int, c_int, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p, c_float, c_float,
                  c_ulonglong, c_ulonglong, c_void_p)         # (the first entry is the return type)
        _lib._sig(L.pafc_silu_dropout, c_int, c_int, c_int, c_long, c_void_p, c_void_p, c_void_p, c_float, c_ulonglong, c_ulonglong,
                  c_void_p)
        L._pafc_te_bound = True
    return L


class _ResidualDropout(torch.autograd.Function):
    """out = x + scale * dropout(y, p): one kernel; backward dx = dout (no kernel), dy = one kernel, mask recomputed."""

    @staticmethod
    def forward(ctx, x, y, scale, p):
        L = _bind_train_elementwise()
        x, y = x.contiguous(), y.contiguous()
        seed, off = _next_dropout_stream(x.device) if p > 0 else (0, 0)
        out = torch.empty_like(x)
        _lib.check(L.pafc_residual_dropout(0, _lib.dtype_code(x.dtype), _lib.dtype_code(y.dtype), x.numel(), _lib.ptr(x), _lib.ptr(y),
                                           _lib.ptr(out), float(scale), float(p), seed, off, _lib.stream_of(x)), "pafc_residual_dropout")
        ctx.meta = (float(scale), float(p), seed, off, y.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        scale, p, seed, off, ydt = ctx.meta
        dy = None
        if ctx.needs_input_grad[1]:
            L = _bind_train_elementwise()
            g = g.contiguous()
            dy = torch.empty(g.shape, dtype=ydt, device=g.device)
            _lib.check(L.pafc_residual_dropout(1, _lib.dtype_code(g.dtype), _lib.dtype_code(ydt), g.numel(), _lib.ptr(g), _lib.ptr(None),
                                               _lib.ptr(dy), scale, p, seed, off, _lib.stream_of(g)), "pafc_residual_dropout")
        return (g if ctx.needs_input_grad[0] else None), dy, None, None


class _SiluDropout(torch.autograd.Function):
    """dropout(silu(h), p): one kernel forward, one backward (from h and the recomputed mask; silu(h) is not kept)."""

    @staticmethod
    def forward(ctx, h, p):
        L = _bind_train_elementwise()
        h = h.contiguous()
        seed, off = _next_dropout_stream(h.device) if p > 0 else (0, 0)
        out = torch.empty_like(h)
        _lib.check(L.pafc_silu_dropout(0, _lib.dtype_code(h.dtype), h.numel(), _lib.ptr(h), _lib.ptr(None), _lib.ptr(out), float(p), seed, off,
                                       _lib.stream_of(h)), "pafc_silu_dropout")
        ctx.save_for_backward(h)
        ctx.meta = (float(p), seed, off)
        return out

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        p, seed, off = ctx.meta
        L = _bind_train_elementwise()
        g = g.contiguous()
        dh = torch.empty_like(h)
        _lib.check(L.pafc_silu_dropout(1, _lib.dtype_code(h.dtype), h.numel(), _lib.ptr(h), _lib.ptr(g), _lib.ptr(dh), p, seed, off,
                                       _lib.stream_of(h)), "pafc_silu_dropout")
        return dh, None


def train_elementwise_eligible(x: torch.Tensor, y: Optional[torch.Tensor] = None) -> bool:
    """The fused element-wise training kernels serve this call: GPU training step with the hand-written kernels on, fp32 or
    bf16 contiguous-izable tensors whose size is a multiple of 8 (a 512-wide stream always is)."""
    if not (train_kernels_enabled() and x.is_cuda and torch.is_grad_enabled() and x.numel() % 8 == 0 and x.numel() > 0):
        return False
    # the kernels want 16-byte aligned pointers: a contiguous view at an odd storage offset keeps the framework's operators
    # (a non-contiguous tensor is copied into a fresh, aligned allocation by the Function)
    for t in (x, y):
        if t is not None and t.is_contiguous() and t.data_ptr() % 16:
            return False
    pairs = {(torch.float32, torch.bfloat16), (torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16)}
    if y is not None:
        return (x.dtype, y.dtype) in pairs and y.shape == x.shape and y.is_cuda
    return x.dtype in (torch.float32, torch.bfloat16)


def residual_dropout(x: torch.Tensor, y: torch.Tensor, scale: float, p: float, training: bool) -> torch.Tensor:
    """x + scale * dropout(y, p) (encoder_layer.py:205-255) as one kernel forward and one backward in the GPU training step;
    the framework's operators otherwise."""
    p = float(p) if training else 0.0
    if train_elementwise_eligible(x, y) and (x.requires_grad or y.requires_grad):
        return _ResidualDropout.apply(x, y, float(scale), p)
    d = torch.nn.functional.dropout(y, p, training=training) if p > 0 else y
    return x + (d if scale == 1.0 else scale * d)


def silu_dropout(h: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    """dropout(silu(h), p) (positionwise_feed_forward.py:47-55) as one kernel forward and one backward in the GPU training step."""
    p = float(p) if training else 0.0
    if train_elementwise_eligible(h) and h.requires_grad:
        return _SiluDropout.apply(h, p)
    a = torch.nn.functional.silu(h)
    return torch.nn.functional.dropout(a, p, training=training) if p > 0 else a


# bf16 copies of fp32 master weights for the training step.  Under autocast every projection casts its weight and bias on
# every step (~400 small launches).  Inside `with train_shadows():` (utils.train_utils.train_step wraps the forward pass in
# it) the copy is kept beside the parameter and ALL copies are brought up to date by one multi-tensor copy when the context
# is entered -- unconditionally: fused optimizers update parameters without touching Tensor._version, so nothing cheaper can
# tell a stale copy.  Outside the context every use casts, as before.
_shadows = {}      # id(parameter) -> [weak reference to the parameter, bf16 copy]  (tensors compare element-wise: they cannot
                   # be keys of a WeakKeyDictionary)
_shadows_on = False
_wgroups = {}     # (id(p0), id(p1), ...) -> [weak references, (Z, N, K) bf16 copy]: equally shaped weights as ONE batched operand


def _param_of(p: torch.Tensor) -> torch.Tensor:
    """The parameter behind a reshaping view of it (a 1 x 1 convolution's `weight.squeeze(-1)`: a new tensor object on every
    call, which a registry keyed by object identity would never find again), else the tensor itself."""
    b = getattr(p, "_base", None)
    if b is not None and isinstance(b, torch.nn.Parameter) and b.numel() == p.numel() and b.is_contiguous() and p.is_contiguous():
        return b
    return p


def _bf16_shadow(p: torch.Tensor) -> torch.Tensor:
    if p.dtype == torch.bfloat16:
        return p
    if not _shadows_on:
        return p.to(torch.bfloat16)
    base = _param_of(p)
    if base is not p:
        return _bf16_shadow(base).view(p.shape)
    ent = _shadows.get(id(p))
    if ent is not None and ent[0]() is p and ent[1].device == p.device and ent[1].shape == p.shape:
        return ent[1]                                # refreshed when the context was entered
    sh = p.detach().to(torch.bfloat16)
    _shadows[id(p)] = [weakref.ref(p), sh]
    return sh


def _param_as(p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """A parameter in the dtype a training kernel wants (no autograd through the cast: callers sit inside Functions)."""
    if p.dtype == dtype:
        return p.detach()
    if dtype == torch.bfloat16 and p.dtype == torch.float32 and isinstance(p, torch.nn.Parameter):
        return _bf16_shadow(p).detach()
    return p.detach().to(dtype)


_shadows_t = {}    # id(parameter) -> [weak reference, bf16 copy of the TRANSPOSED matrix (K, N), (param epoch, version) of the copy]
_tr_table = None   # {device: (signature, device table of pafc_multi_transpose_bf16 descriptors, n, total tiles)}


def _bf16_shadow_t(w: torch.Tensor) -> torch.Tensor:
    """bf16 (K, N) copy of a weight (N, K) -- a parameter or a reshaping view of one -- for the input-gradient GEMM dX = dY W.
    The copies live beside their parameters; ALL of them are refreshed by ONE launch when train_shadows() is entered
    (pafc_multi_transpose_bf16) and stamped with the parameter epoch and the parameter's in-place version: a copy whose stamp is
    not the present one (a backward pass outside train_step, an update behind the registry's back) is re-made on the spot."""
    global _tr_table
    N, K = w.shape
    p = _param_of(w)
    stamp = (_param_epoch, p._version)
    ent = _shadows_t.get(id(p))
    if ent is not None and ent[0]() is p and ent[1].device == p.device and tuple(ent[1].shape) == (K, N):
        if ent[2] != stamp:
            with torch.no_grad():
                ent[1].copy_(p.detach().view(N, K).t())
            ent[2] = stamp
        return ent[1]
    if not isinstance(p, torch.nn.Parameter):
        return w.detach().t().to(torch.bfloat16).contiguous()     # not a parameter: nothing to keep it beside
    sh = p.detach().view(N, K).t().to(torch.bfloat16).contiguous()
    _shadows_t[id(p)] = [weakref.ref(p), sh, stamp]
    _tr_table = None
    return sh


def _refresh_transposed_shadows() -> None:
    """One pafc_multi_transpose_bf16 launch per DEVICE that holds registered parameters (descriptor table and launch on that
    device, on its current stream)."""
    global _tr_table
    by_dev = {}
    for key in list(_shadows_t):
        p = _shadows_t[key][0]()
        sh = _shadows_t[key][1]
        if (p is None or sh.device != p.device or sh.numel() != p.numel() or p.dtype not in (torch.float32, torch.bfloat16)
                or not p.is_contiguous()):
            del _shadows_t[key]
            _tr_table = None
        else:
            by_dev.setdefault(p.device, []).append((p, sh))
            _shadows_t[key][2] = (_param_epoch, p._version)
    if not by_dev:
        return
    if _tr_table is None:
        _tr_table = {}
    L = _bind()
    if not getattr(L, "_pafc_mtr_bound", False):
        _lib._sig(L.pafc_multi_transpose_bf16, c_int, c_void_p, c_int, c_int, c_void_p)
        L._pafc_mtr_bound = True
    for dev, live in by_dev.items():
        sig = tuple((p.data_ptr(), sh.data_ptr(), p.dtype) for p, sh in live)
        ent = _tr_table.get(dev)
        if ent is None or ent[0] != sig:
            rows, t0 = [], 0
            for p, sh in live:
                k_, n_ = sh.shape                  # the copy is (K, N); the parameter (N, K) or a reshaping view target of it
                # { src, dst, rows | cols << 32, src_f32 | tile0 << 32 } as four little-endian 64-bit words = the 32-byte descriptor
                rows.append([p.data_ptr(), sh.data_ptr(), n_ | (k_ << 32), int(p.dtype == torch.float32) | (t0 << 32)])
                t0 += ((n_ + 63) // 64) * ((k_ + 63) // 64)
            ent = _tr_table[dev] = (sig, torch.tensor(rows, dtype=torch.int64).to(dev), len(live), t0)
        _, tab, n, tiles = ent
        with torch.cuda.device(dev):
            _lib.check(L.pafc_multi_transpose_bf16(_lib.ptr(tab), n, tiles, _lib.stream_of(tab)), "pafc_multi_transpose_bf16")


def refresh_train_shadows() -> None:
    """Bring every registered bf16 weight copy up to date with its parameter in one multi-tensor copy (and the transposed
    copies in one launch of their own)."""
    _refresh_transposed_shadows()
    live, dead = [], []
    for key, ent in _shadows.items():
        p = ent[0]()
        if p is None or ent[1].device != p.device or ent[1].shape != p.shape:
            dead.append(key)
        else:
            live.append((p, ent[1]))
    for key in dead:
        del _shadows[key]
    for key in list(_wgroups):                      # grouped bf16 copies (r / k / v of a time-mix block as one (3, N, K) operand)
        refs, g = _wgroups[key]
        ps = [r() for r in refs]
        if any(p is None or p.device != g.device or tuple(p.shape) != tuple(g.shape[1:]) for p in ps):
            del _wgroups[key]
        else:
            live.extend((p, g[i]) for i, p in enumerate(ps))
    if live:
        _multi_cast(live)


_cast_table = {}   # device -> (signature, device table of pafc_multi_cast_bf16 descriptors, n, total chunks)


def _multi_cast(pairs) -> None:
    """[(parameter, bf16 copy), ...]: copy <- parameter.  GPU tensors: ONE pafc_multi_cast_bf16 launch per device over a descriptor
    table that is rebuilt only when a pointer changes (torch._foreach_copy_ across dtypes is one launch per tensor on this build:
    108 of the training step's launches); anything else through torch._foreach_copy_."""
    by_dev, rest = {}, []
    for p, sh in pairs:
        if (p.is_cuda and sh.is_cuda and p.device == sh.device and sh.dtype == torch.bfloat16 and p.dtype in (torch.float32, torch.bfloat16)
                and p.is_contiguous() and sh.is_contiguous() and p.numel() == sh.numel() and 0 < p.numel() < (1 << 31)):
            by_dev.setdefault(p.device, []).append((p, sh))
        else:
            rest.append((p, sh))
    if rest:
        with torch.no_grad():
            torch._foreach_copy_([sh for _, sh in rest], [p.detach() for p, _ in rest])
    if not by_dev:
        return
    L = _bind()
    if not getattr(L, "_pafc_mcast_bound", False):
        _lib._sig(L.pafc_multi_cast_bf16, c_int, c_void_p, c_int, c_int, c_void_p)
        L._pafc_mcast_bound = True
    for dev, live in by_dev.items():
        sig = tuple((p.data_ptr(), sh.data_ptr(), p.dtype, p.numel()) for p, sh in live)
        ent = _cast_table.get(dev)
        if ent is None or ent[0] != sig:
            rows, c0 = [], 0
            for p, sh in live:
                # { src, dst, rows | cols << 32, src_f32 | chunk0 << 32 }: the 32-byte descriptor of pafc_multi_transpose_bf16, rows x cols = numel
                rows.append([p.data_ptr(), sh.data_ptr(), p.numel() | (1 << 32), int(p.dtype == torch.float32) | (c0 << 32)])
                c0 += (p.numel() + 4095) // 4096
            ent = _cast_table[dev] = (sig, torch.tensor(rows, dtype=torch.int64).to(dev), len(live), c0)
        _, tab, n, chunks = ent
        with torch.cuda.device(dev):
            _lib.check(L.pafc_multi_cast_bf16(_lib.ptr(tab), n, chunks, _lib.stream_of(tab)), "pafc_multi_cast_bf16")


@contextlib.contextmanager
def train_shadows():
    """The forward pass of one training step: projections take refreshed bf16 copies of their fp32 weights."""
    global _shadows_on
    refresh_train_shadows()
    prev, _shadows_on = _shadows_on, True
    try:
        yield
    finally:
        _shadows_on = prev


def _own_gemm_rows(t2: torch.Tensor) -> bool:
    return t2.dtype == torch.bfloat16 and t2.dim() == 2 and t2.stride(1) == 1 and t2.stride(0) % 8 == 0 and t2.data_ptr() % 16 == 0


class _LinearTrainBf16(torch.autograd.Function):
    """nn.Linear for the bf16 training step, every product on hand-written kernels: forward y = x W^T + b and input gradient
    dX = dY W on the tiled GEMMs (pafc_gemm_bf16; the latter against the bf16 copy of W^T that train_shadows() keeps, one
    multi-tensor transpose per step), the weight gradient through gemm_tn, straight into the weight's dtype -- fp32 master
    weights receive the fp32 sum.  Shapes the kernels do not take (K or N not a multiple of 64 / 8) keep the library."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        wb = _bf16_shadow(weight)
        bb = None if bias is None else _bf16_shadow(bias)
        ctx.save_for_backward(x, wb)
        # the kept W^T copies are only known to be current inside train_shadows() (refreshed on entry, whatever the optimizer did
        # to Tensor._version); a forward pass outside it cast the weight fresh, so its backward transposes THAT copy
        ctx.weight = weight if (_shadows_on and isinstance(_param_of(weight), torch.nn.Parameter)) else None
        ctx.w_dtype = weight.dtype
        ctx.b_dtype = None if bias is None else bias.dtype
        N, K = wb.shape
        x2 = x.reshape(-1, K)
        if (train_gemms_own() and K % 64 == 0 and N % 8 == 0 and _own_gemm_rows(x2) and wb.is_contiguous()
                and (bb is None or bb.dtype == torch.bfloat16)):
            return gemm_bf16(x2, wb.detach(), None if bb is None else bb.detach()).view(x.shape[:-1] + (N,))
        return torch.nn.functional.linear(x, wb, bb)

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        N, K = wb.shape
        dy2 = dy.reshape(-1, N)
        if dy2.stride(1) != 1 or dy2.stride(0) % 8 or dy2.data_ptr() % 16:
            dy2 = dy2.contiguous()
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if train_gemms_own() and N % 64 == 0 and K % 8 == 0 and ctx.weight is not None and _own_gemm_rows(dy2):
                dx = gemm_bf16(dy2, _bf16_shadow_t(ctx.weight)).view(x.shape)      # dY (M, N) x (W^T)(K, N)^T
            elif train_gemms_own() and N % 64 == 0 and K % 8 == 0 and _own_gemm_rows(dy2) and wb.dtype == torch.bfloat16:
                dx = gemm_bf16(dy2, wb.detach().t().contiguous()).view(x.shape)    # the forward's own copy, transposed now
            else:
                dx = (dy2 @ wb).view(x.shape)
        want_b = ctx.b_dtype is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            od = ctx.w_dtype if ctx.w_dtype in (torch.float32, torch.bfloat16) else torch.float32
            if want_b:
                dw, db = gemm_tn(dy2, x2, od, want_bias=True)
                db = db.to(ctx.b_dtype)
            else:
                dw = gemm_tn(dy2, x2, od)
            dw = dw.to(ctx.w_dtype)
        elif want_b:
            db = dy2.sum(0, dtype=torch.float32).to(ctx.b_dtype)
        return dx, dw, db


class _MatmulTrainBf16(torch.autograd.Function):
    """y = act(x @ W + bias) for a parameter stored (K, N) -- the LoRA matrices of the time-mix (src/model.py:277,289) -- on the
    hand-written kernels: forward against the bf16 copy of W^T (N, K) that train_shadows() keeps, input gradient dx = dy W^T against
    W as it lies, weight gradient x^T dy through gemm_tn.  act "tanh" and a bias (the decay's `time_decay + ...`, model.py:289) ride in
    the forward GEMM's epilogue (round 6: one launch instead of three); backward: tanh' from the saved output, the bias gradient a
    column sum."""

    @staticmethod
    def forward(ctx, x, weight, act="none", bias=None):
        K, N = weight.shape
        x2 = x.reshape(-1, K)
        ctx.act, ctx.has_bias = act, bias is not None
        if train_gemms_own() and K % 64 == 0 and N % 8 == 0 and _own_gemm_rows(x2):
            if _shadows_on and isinstance(_param_of(weight), torch.nn.Parameter):
                wt = _bf16_shadow_t(weight)                 # kept beside the parameter, refreshed when train_shadows() was entered
            else:
                wt = weight.detach().t().contiguous()       # outside the step's context nothing vouches for a kept copy
            bb = None if bias is None else bias.detach().reshape(-1).to(torch.bfloat16)
            y = gemm_bf16(x2, wt, bb, act).view(x.shape[:-1] + (N,))                     # (M, K) x (W^T)(N, K)^T
        else:
            y = x @ weight
            if bias is not None:
                y = y + bias.reshape(-1).to(y.dtype)
            if act == "tanh":
                y = torch.tanh(y)
        ctx.save_for_backward(x, weight, y if act == "tanh" else None)
        ctx.bias_shape, ctx.bias_dtype = (None, None) if bias is None else (bias.shape, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        K, N = weight.shape
        if ctx.act == "tanh":
            dy = torch.ops.aten.tanh_backward(dy.contiguous(), y)
        dy2 = dy.reshape(-1, N)
        if dy2.stride(1) != 1 or dy2.stride(0) % 8 or dy2.data_ptr() % 16:
            dy2 = dy2.contiguous()
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if train_gemms_own() and N % 64 == 0 and K % 8 == 0 and weight.is_contiguous() and _own_gemm_rows(dy2):
                dx = gemm_bf16(dy2, weight.detach()).view(x.shape)                       # (M, N) x W (K, N)^T
            else:
                dx = (dy2 @ weight.t()).view(x.shape)
        dw = gemm_tn(x2, dy2, torch.bfloat16) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[3]:
            # (one reduction kernel: fp32 accumulation inside, one rounding to the bias's dtype -- the value of sum(float32).to())
            db = dy2.sum(0, dtype=ctx.bias_dtype).view(ctx.bias_shape)
        return dx, dw, None, db


def matmul_param(x: torch.Tensor, weight: torch.Tensor, act: str = "none", bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(x @ weight + bias) for a 2-D bf16 parameter (act "none" | "tanh"); in the GPU training step on the hand-written kernels
    with the activation and the bias in the GEMM's epilogue."""
    if (x.is_cuda and torch.is_grad_enabled() and weight.requires_grad and weight.dim() == 2 and train_kernels_enabled()
            and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.shape[0] % 8 == 0
            and weight.shape[1] % 8 == 0 and x.numel() // x.shape[-1] >= 256
            and (bias is None or (bias.dtype == torch.bfloat16 and bias.numel() == weight.shape[1]))):
        return _MatmulTrainBf16.apply(x, weight, act, bias)
    y = x @ weight
    if bias is not None:
        y = bias + y
    return torch.tanh(y) if act == "tanh" else y


def linear_train_eligible(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """bf16 activations on the GPU under autograd (autocast(bfloat16) over fp32 master weights, or a bf16 module such
    as the time-mix slot), dims the kernel takes."""
    if not (x.is_cuda and torch.is_grad_enabled() and weight.requires_grad and weight.dim() == 2
            and train_kernels_enabled()):
        return False
    if weight.shape[0] % 8 or weight.shape[1] % 8 or x.numel() // max(1, x.shape[-1]) < 256:
        return False
    if weight.numel() > int(os.environ.get("PAFC_TRAIN_LINEAR_MAX_NUMEL", 2048 * 1024)):      # many output tiles already fill the chip: the library's kernel is as good
        return False
    if x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16:
        return True
    return (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and x.dtype in (torch.bfloat16, torch.float32) and weight.dtype in (torch.float32, torch.bfloat16))


def linear_train(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """F.linear whose weight gradient goes through the hand-written kernel (see linear_train_eligible)."""
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)          # what autocast does to F.linear's input
    return _LinearTrainBf16.apply(x, weight, bias)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """The projections of the encoder layer as the modules call them: F.linear, except in the bf16 training step, where
    the weight gradient takes the hand-written kernel (linear_train)."""
    if linear_train_eligible(x, weight):
        return linear_train(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


def _weight_group(ws) -> Optional[torch.Tensor]:
    """Z equally shaped weights as ONE (Z, N, K) bf16 operand, inside train_shadows() only (None elsewhere).
    bf16 parameters (the time-mix slot): the parameters THEMSELVES are moved into one buffer the first time they are asked for
    together (`p.data = group[i]`: the Parameter objects, their optimizer state and their names stay; load_state_dict copies in
    place; a later `.to()` that breaks the layout is noticed here and mended) -- no copy to keep fresh, no launch per step.
    fp32 parameters: a grouped bf16 copy beside them, refreshed with the other copies when train_shadows() is entered."""
    if not _shadows_on or any(not isinstance(w, torch.nn.Parameter) or w.dim() != 2 or w.shape != ws[0].shape
                              or w.device != ws[0].device or w.dtype != ws[0].dtype for w in ws):
        return None
    if ws[0].dtype == torch.bfloat16:
        g = _as_batch([w.detach() for w in ws])
        if g is None or not g.is_contiguous():
            with torch.no_grad():
                g = torch.stack([w.detach() for w in ws]).contiguous()
                for i, w in enumerate(ws):
                    w.data = g[i]
            bump_param_epoch()        # the parameters moved: plans and captured graphs that hold their old addresses are stale
        return g
    if ws[0].dtype != torch.float32:
        return None
    key = tuple(id(w) for w in ws)
    ent = _wgroups.get(key)
    if ent is not None and all(r() is w for r, w in zip(ent[0], ws)) and ent[1].device == ws[0].device:
        return ent[1]
    g = torch.stack([w.detach().to(torch.bfloat16) for w in ws]).contiguous()
    _wgroups[key] = [[weakref.ref(w) for w in ws], g]
    return g


def _params_as_one(ps) -> Optional[torch.Tensor]:
    """Z equally shaped parameters of one dtype as ONE contiguous (Z, numel) tensor WITHOUT a copy per use: they are moved into one
    buffer the first time they are asked for together (`p.data = buffer[i].view(shape)`, as _weight_group does for the bf16
    projection weights) -- the four lerp coefficients time_maa_r / k / v / w of a time-mix block (src/model.py:236-240), which the
    training step otherwise stacks on every call (four device-to-device copies).  None when they are not such parameters."""
    p0 = ps[0]
    if any(not isinstance(p, torch.nn.Parameter) or p.shape != p0.shape or p.dtype != p0.dtype or p.device != p0.device
           or not p.is_contiguous() for p in ps) or p0.numel() % 8:
        return None
    g = _as_batch([p.detach().view(1, -1) for p in ps])
    if g is None or not g.is_contiguous():
        with torch.no_grad():
            g = torch.stack([p.detach().reshape(1, -1) for p in ps]).contiguous()
            for i, p in enumerate(ps):
                p.data = g[i].view(p.shape)
        bump_param_epoch()            # the parameters moved: plans and captured graphs that hold their old addresses are stale
    return g.view(len(ps), -1)


def _weight_t_group(ws) -> Optional[torch.Tensor]:
    """(Z, K, N) bf16: the transposed copies of _bf16_shadow_t for Z equally shaped weights, as views of ONE tensor (the
    multi-tensor transpose that refreshes the copies writes into the views)."""
    global _tr_table
    if not _shadows_on or any(not isinstance(w, torch.nn.Parameter) or w.dim() != 2 or w.shape != ws[0].shape for w in ws):
        return None
    N, K = ws[0].shape
    ents = [_shadows_t.get(id(w)) for w in ws]
    if all(e is not None and e[0]() is w for e, w in zip(ents, ws)):
        b = ents[0][1]._base
        if (b is not None and tuple(b.shape) == (len(ws), K, N) and b.is_contiguous()
                and all(e[1]._base is b and e[1].data_ptr() == b[i].data_ptr() for i, e in enumerate(ents))):
            for e, w in zip(ents, ws):                       # a copy whose stamp is not the present one is re-made on the spot
                stamp = (_param_epoch, w._version)
                if e[2] != stamp:
                    with torch.no_grad():
                        e[1].copy_(w.detach().t())
                    e[2] = stamp
            return b
    g = torch.empty((len(ws), K, N), dtype=torch.bfloat16, device=ws[0].device)
    with torch.no_grad():
        for i, w in enumerate(ws):
            g[i].copy_(w.detach().t())
            _shadows_t[id(w)] = [weakref.ref(w), g[i], (_param_epoch, w._version)]
    _tr_table = None
    return g


def _as_batch(ts) -> Optional[torch.Tensor]:
    """Z equally shaped 2-d row-major views of ONE storage at equal distances -> the (Z, M, K) strided view over them (no copy),
    else None.  (The lerp kernel leaves z_r, z_k, z_v as slices of one tensor; the WKV backward leaves g_r, g_k, g_v so.)"""
    a = ts[0]
    if a.dim() != 2 or a.stride(1) != 1 or any(t.shape != a.shape or t.stride() != a.stride() or t.dtype != a.dtype
                                                or t.untyped_storage().data_ptr() != a.untyped_storage().data_ptr() for t in ts):
        return None
    d = ts[1].storage_offset() - a.storage_offset()
    if d <= 0 or d % 8 or any(ts[i + 1].storage_offset() - ts[i].storage_offset() != d for i in range(len(ts) - 1)):
        return None
    return torch.as_strided(a, (len(ts),) + tuple(a.shape), (d,) + tuple(a.stride()), a.storage_offset())


class _LinearGroupTrainBf16(torch.autograd.Function):
    """Z bias-free nn.Linear of one shape on Z inputs -- the r / k / v projections of a time-mix block (src/model.py:286-288) -- as
    ONE batched launch each way: forward against the grouped bf16 weight copy, input gradients against the grouped transposed
    copies, the Z weight gradients from one batched gemm_tn pair.  (Three _LinearTrainBf16 nodes: 3 + 3 + 6 launches and three
    Python backward calls per direction and layer.)"""

    @staticmethod
    def forward(ctx, *args):
        Z = len(args) // 2
        xs, ws = args[:Z], args[Z:]
        N, K = ws[0].shape
        x2 = [x.reshape(-1, K) for x in xs]
        xb = _as_batch(x2)
        if xb is None:
            xb = torch.stack(x2)
        y = gemm_bf16(xb, _weight_group(ws))                           # (Z, M, N)
        ctx.save_for_backward(xb)
        ctx.ws, ctx.x_shape = ws, xs[0].shape
        return tuple(y[i].view(xs[0].shape[:-1] + (N,)) for i in range(Z))

    @staticmethod
    def backward(ctx, *dys):
        (xb,) = ctx.saved_tensors
        ws = ctx.ws
        Z = len(ws)
        N, K = ws[0].shape
        d2 = [(torch.zeros(xb.shape[1], N, dtype=xb.dtype, device=xb.device) if g is None else g.reshape(-1, N)) for g in dys]
        db = _as_batch(d2)
        if db is None:
            db = torch.stack(d2)
        dxs = [None] * Z
        if any(ctx.needs_input_grad[:Z]):
            wt = _weight_t_group(ws)
            if wt is None:                                               # (a backward pass outside train_shadows())
                wt = torch.stack([w.detach().t().to(torch.bfloat16) for w in ws]).contiguous()
            dx = gemm_bf16(db, wt)                                       # (Z, M, N) x (Z, K, N)^T -> (Z, M, K)
            dxs = [dx[i].view(ctx.x_shape) for i in range(Z)]
        dws = [None] * Z
        if any(ctx.needs_input_grad[Z:]):
            od = ws[0].dtype if ws[0].dtype in (torch.float32, torch.bfloat16) else torch.float32
            dw = gemm_tn(db, xb, od)                                     # (Z, N, K)
            dws = [dw[i] for i in range(Z)]
        return (*dxs, *dws)


def linear_group_train_eligible(xs, ws) -> bool:
    """Inside train_shadows(), own training GEMMs, bf16 inputs of one shape, equally shaped bias-free weights the tiles take."""
    if not (_shadows_on and train_gemms_own() and os.environ.get("PAFC_TRAIN_LINEAR_GROUP", "1") != "0"):
        return False
    w0, x0 = ws[0], xs[0]
    if not all(isinstance(w, torch.nn.Parameter) and w.requires_grad and w.dim() == 2 and w.shape == w0.shape and w.dtype == w0.dtype
               and w.is_contiguous() for w in ws):
        return False
    N, K = w0.shape
    return (K % 64 == 0 and N % 64 == 0 and w0.dtype in (torch.float32, torch.bfloat16)
            and all(x.is_cuda and x.dtype == torch.bfloat16 and x.shape == x0.shape and x.shape[-1] == K and x.is_contiguous() for x in xs)
            and x0.numel() // K >= 256 and torch.is_grad_enabled())


def linear_group_train(xs, ws):
    """[F.linear(x_i, w_i)] for equally shaped (x_i, w_i): see _LinearGroupTrainBf16."""
    return _LinearGroupTrainBf16.apply(*xs, *ws)


class _CtcHeadLoss(torch.autograd.Function):
    """CTC head + loss of the training step on hand-written kernels only: logits = x W^T + b (pafc_gemm_bf16), the loss from the
    logits (pafc_ctc_loss_forward: row statistics, the alpha / beta lattice over the label columns), and backwards the gradient
    through the log-softmax written once, bf16, into rows padded to a multiple of 64 columns (pafc_ctc_loss_backward), dX = dlogits W
    on pafc_gemm_bf16 against a zero-padded W^T, dW / db through gemm_tn.  The (B, T, V) log-probabilities never exist; no call
    waits for the host (torch's ctc_loss copies its length tensors back: five synchronising calls per step)."""

    @staticmethod
    def forward(ctx, x, weight, bias, hlens, ys, ylens, blank):
        B, T, C = x.shape
        V = weight.shape[0]
        Vp = (V + 63) // 64 * 64
        M = B * T
        xb = x.reshape(M, C).to(torch.bfloat16)
        wb = _bf16_shadow(weight).detach()
        bb = None if bias is None else _bf16_shadow(bias).detach()
        logits = torch.empty((M, Vp), dtype=torch.bfloat16, device=x.device)
        gemm_bf16(xb, wb, bb, out=logits[:, :V])
        hl = hlens.to(torch.int32).contiguous()
        yl = ylens.to(torch.int32).contiguous()
        ysc = ys.to(torch.int64).contiguous()
        Lmax = ysc.shape[1]
        L = _bind_ctc_loss()
        nbytes = L.pafc_ctc_loss_workspace_bytes(B, T, Lmax)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        nll = torch.empty(B, dtype=torch.float32, device=x.device)
        st = _lib.stream_of(x)
        _lib.check(L.pafc_ctc_loss_forward(_lib.PAFC_BF16, B, T, V, _lib.ptr(logits), Vp, _lib.ptr(hl), _lib.ptr(ysc), Lmax, _lib.ptr(yl),
                                           Lmax, int(blank), _lib.ptr(nll), _lib.ptr(ws), nbytes, st), "pafc_ctc_loss_forward")
        ctx.save_for_backward(xb, wb, logits, ws, nll, hl, ysc, yl)
        ctx.dims = (B, T, C, V, Vp, Lmax, int(blank))
        ctx.w_dtype, ctx.b_dtype, ctx.x_dtype = weight.dtype, None if bias is None else bias.dtype, x.dtype
        return nll.sum() / B                          # reduction="sum", then the batch-size average of ctc.py:77

    @staticmethod
    def backward(ctx, g):
        xb, wb, logits, ws, nll, hl, ysc, yl = ctx.saved_tensors
        B, T, C, V, Vp, Lmax, blank = ctx.dims
        M = B * T
        L = _bind_ctc_loss()
        dl = torch.empty((M, Vp), dtype=torch.bfloat16, device=xb.device)
        gf = g.detach().to(torch.float32).reshape(1).contiguous()
        _lib.check(L.pafc_ctc_loss_backward(_lib.PAFC_BF16, B, T, V, _lib.ptr(logits), Vp, _lib.ptr(hl), _lib.ptr(ysc), Lmax, _lib.ptr(yl),
                                            Lmax, blank, _lib.ptr(nll), _lib.ptr(gf), 1.0 / B, _lib.ptr(dl), Vp, _lib.ptr(ws), ws.numel(),
                                            _lib.stream_of(xb)), "pafc_ctc_loss_backward")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = torch.zeros((C, Vp), dtype=torch.bfloat16, device=xb.device)       # W^T, zero beyond V: K = Vp is a multiple of 64
            wt[:, :V].copy_(wb.t())
            dx = gemm_bf16(dl, wt).view(B, T, C).to(ctx.x_dtype)
        want_b = ctx.b_dtype is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            od = ctx.w_dtype if ctx.w_dtype in (torch.float32, torch.bfloat16) else torch.float32
            if want_b:
                dw, db = gemm_tn(dl[:, :V], xb, od, want_bias=True)
                db = db.to(ctx.b_dtype)
            else:
                dw = gemm_tn(dl[:, :V], xb, od)
            dw = dw.to(ctx.w_dtype)
        elif want_b:
            db = dl[:, :V].sum(0, dtype=torch.float32).to(ctx.b_dtype)
        return dx, dw, db, None, None, None, None


def _bind_ctc_loss():
    L = _bind2()
    if not getattr(L, "_pafc_ctcloss_bound", False):
        from ctypes import c_float, c_long, c_size_t
        P, I, G, Z = c_void_p, c_int, c_long, c_size_t
        _lib._sig(L.pafc_ctc_loss_workspace_bytes, Z, I, I, I)
        _lib._sig(L.pafc_ctc_loss_forward, I, I, I, I, I, P, G, P, P, I, P, I, I, P, P, Z, P)
        _lib._sig(L.pafc_ctc_loss_backward, I, I, I, I, I, P, G, P, P, I, P, I, I, P, P, c_float, P, G, P, Z, P)
        L._pafc_ctcloss_bound = True
    return L


def ctc_head_loss_eligible(x: torch.Tensor, weight: torch.Tensor, ys: torch.Tensor) -> bool:
    """The GPU training step under bf16 autocast (or a bf16 model): dims the kernels take."""
    if not (x.is_cuda and torch.is_grad_enabled() and train_kernels_enabled() and x.dim() == 3 and ys.dim() == 2):
        return False
    V, C = weight.shape
    if C % 64 or V % 8 or V * 4 > 150 * 1024 or x.shape[0] > 65535 or ys.shape[1] > 3000 or os.environ.get("PAFC_TRAIN_CTC", "1") == "0":
        return False
    if x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16:
        return True
    return (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and x.dtype in (torch.bfloat16, torch.float32) and weight.dtype in (torch.float32, torch.bfloat16))


def ctc_head_loss(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], hlens: torch.Tensor, ys: torch.Tensor,
                  ylens: torch.Tensor, blank: int = 0) -> torch.Tensor:
    """sum_b CTC(log_softmax(x_b W^T + b), y_b) / B with zero_infinity (ctc.py:53-82), see _CtcHeadLoss."""
    return _CtcHeadLoss.apply(x, weight, bias, hlens, ys, ylens, blank)


def _bind2():
    L = _bind()
    if getattr(L, "_pafc_glue_bound", False):
        return L
    from ctypes import c_float, c_long
    P, I = c_void_p, c_int
    _lib._sig(L.pafc_add_layernorm, I, I, I, I, I, P, P, c_float, P, I, I, P, P, P, P, c_long, I, I, P, P, P, c_long,
              c_float, P)
    # (first entry = return type; the argument lists are checked against include/*.h by tests/test_abi.py)
    _lib._sig(L.pafc_tmix_shift_mix, I, I, I, I, I, I, I, P, P, P, P, P)
    _lib._sig(L.pafc_tmix_mix4, I, I, I, I, I, I, I, P, P, P, P, P)
    _lib._sig(L.pafc_tmix_lora_mix4_bf16, I, I, I, I, I, I, P, P, P, P, P, P)
    _lib._sig(L.pafc_tmix_shift_mix_prev, I, I, I, I, I, I, I, P, P, P, P, P, P)
    _lib._sig(L.pafc_tmix_lora_mix4_bf16_prev, I, I, I, I, I, I, P, P, P, P, P, P, P)
    _lib._sig(L.pafc_tmix_lora_down_bf16_prev, I, I, I, I, I, I, I, P, P, P, P, P, P)
    L._pafc_glue_bound = True
    return L


PAFC_SPLIT_BF16 = 2      # output form: an fp32 value as bf16 planes [hi | lo] (include/pafc_wkv6.h)


def add_layernorm(x: torch.Tensor, y: Optional[torch.Tensor], alpha: float, gamma1, beta1, *, out1: torch.Tensor = None,
                  out_dtype: Optional[torch.dtype] = None, silu: bool = False, zero_rows: bool = False,
                  lens: Optional[torch.Tensor] = None, T: int = 0, mask_y: bool = False, gamma2=None, beta2=None,
                  want_ln: bool = True, want_x: bool = True, eps: float = 1e-5, split1: bool = False, split2: bool = False,
                  stats_x: Optional[torch.Tensor] = None, stats_out1: Optional[torch.Tensor] = None):
    """x_new = x + alpha*y; out1 = LN1(x_new) [silu] [rows >= len zeroed]; out2 = LN2(out1).
    Returns (x_new or x, out1 or None, out2 or None).  `out1` may be a pre-allocated (rows, ld) view (last-dim
    slice of a wider buffer) so that two LayerNorms can land side by side.
    split1 / split2 (fp32 x only): that output comes as bf16 planes (..., 2C) = [hi | lo] of the fp32 result, the A operand
    of gemm_ph_ex(a_split=True); split1 implies split2.
    stats_x / stats_out1: float32 (rows, 8, 2) buffers that receive the row statistics of x_new / of out1 (gemm_bf16_ln)."""
    _lib.require_gpu(x, y, gamma1, beta1, gamma2, beta2, lens)
    C = x.shape[-1]
    rows = x.numel() // C
    out_dtype = out_dtype or x.dtype
    if (split1 or split2) and (x.dtype != torch.float32 or out_dtype != torch.float32 or out1 is not None):
        raise _lib.PafcError("add_layernorm: plane outputs are a form of an fp32 result")
    x_out = torch.empty_like(x) if (y is not None and want_x) else None
    o1 = None
    ld1 = C
    planes = lambda: torch.empty(x.shape[:-1] + (2 * C,), dtype=torch.bfloat16, device=x.device)
    if want_ln:
        if split1:
            o1, ld1 = planes(), 2 * C
        elif out1 is None:
            o1 = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        else:
            o1 = out1
            ld1 = out1.stride(-2)
            if out1.stride(-1) != 1 or out1.dtype != out_dtype:
                raise _lib.PafcError("out1 must be a unit-stride view in the output dtype")
    s2 = split1 or split2
    o2 = None
    if gamma2 is not None:
        o2 = planes() if s2 else torch.empty(x.shape, dtype=out_dtype, device=x.device)
    L = _bind2()
    if not getattr(L, "_pafc_lnex_bound", False):
        from ctypes import c_float, c_long
        P, I = c_void_p, c_int
        _lib._sig(L.pafc_add_layernorm_ex, I, I, I, I, I, I, P, P, c_float, P, I, I, P, P, P, P, c_long, I, I, P, P, P, c_long,
                  c_float, P, P, P)
        L._pafc_lnex_bound = True
    code = _lib.dtype_code(out_dtype)
    rc = L.pafc_add_layernorm_ex(
        _lib.dtype_code(x.dtype), PAFC_SPLIT_BF16 if split1 else code, PAFC_SPLIT_BF16 if s2 else code, rows, C, _lib.ptr(x),
        _lib.ptr(y), float(alpha), _lib.ptr(lens), int(T), int(mask_y), _lib.ptr(x_out), _lib.ptr(gamma1), _lib.ptr(beta1),
        _lib.ptr(o1), ld1, int(silu), int(zero_rows), _lib.ptr(gamma2), _lib.ptr(beta2), _lib.ptr(o2), 2 * C if s2 else C,
        float(eps), _lib.ptr(stats_x), _lib.ptr(stats_out1), _lib.stream_of(x))
    _lib.check(rc, "pafc_add_layernorm")
    return (x_out if y is not None else x), o1, o2  # first item is None when want_x=False


def gemm_bf16_ln(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stats: torch.Tensor, *, act: str = "none",
                 alpha: float = 1.0, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 csum: Optional[torch.Tensor] = None, ln_c: int = 0, eps: float = 1e-5) -> torch.Tensor:
    """The bf16 GEMMs either side of a folded pre-norm LayerNorm (include/pafc_encoder_ops.h: pafc_gemm_bf16_ph_ln).
    csum given: the CONSUMER -- act(rstd (a w^T) - rstd mean csum + bias), a = the un-normalised rows, w = gamma * W, act
    "silu" | "glu", row statistics READ from stats (rows, 8, 2) fp32.  csum None: the PRODUCER -- alpha a w^T + bias + residual
    (N = 512) with the rows' statistics WRITTEN to stats."""
    _lib.require_gpu(bias, stats, csum)
    for t in (a, w, residual, out):
        if t is not None and (not t.is_cuda or t.dtype != torch.bfloat16 or t.stride(-1) != 1 or t.dim() != 2):
            raise _lib.PafcError("gemm_bf16_ln: 2-d bf16 GPU tensors with unit stride in the last dimension")
    M, K = a.shape
    N = w.shape[0]
    No = N // 2 if act == "glu" else N
    if stats.dtype != torch.float32 or stats.numel() != M * 16 or not stats.is_contiguous():
        raise _lib.PafcError("gemm_bf16_ln: stats is a contiguous float32 (M, 8, 2) buffer")
    if out is None:
        out = torch.empty((M, No), dtype=a.dtype, device=a.device)
    L = _bind2()
    if not getattr(L, "_pafc_gemmln_bound", False):
        from ctypes import c_float, c_long
        P, I, G = c_void_p, c_int, c_long
        _lib._sig(L.pafc_gemm_bf16_ph_ln, I, G, I, I, P, G, P, G, P, P, G, P, G, c_float, I, I, P, P, I, c_float, I, P)
        L._pafc_gemmln_bound = True
    from .profiling import op_timer
    with op_timer("gemm_%dx%d" % (K, N), sample=12, flops=2.0 * M * N * K):
        rc = L.pafc_gemm_bf16_ph_ln(M, N, K, _lib.ptr(a), a.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(bias), _lib.ptr(residual),
                                    residual.stride(0) if residual is not None else 0, _lib.ptr(out), out.stride(0), float(alpha),
                                    _ACTS[act], 1 if csum is not None else 2, _lib.ptr(stats), _lib.ptr(csum),
                                    int(ln_c or (K if csum is not None else N)),
                                    float(eps), _ph_tile_m(M, N, min_tm=128), _lib.stream_of(a))
    _lib.check(rc, "pafc_gemm_bf16_ph_ln")
    return out


def split_planes(x: torch.Tensor, triple: bool = False) -> torch.Tensor:
    """fp32 (..., K) -> bf16 (..., 2K) = [hi | lo] with hi = bf16(x), lo = bf16(x - hi) (an activation for
    gemm_ph_ex(a_split=True)), or, triple, (..., 3K) = [hi | hi | lo] (its weight)."""
    _lib.require_gpu(x)
    if x.dtype != torch.float32 or x.stride(-1) != 1:
        raise _lib.PafcError("split_planes: contiguous fp32 input")
    K = x.shape[-1]
    rows = x.numel() // K
    out = torch.empty(x.shape[:-1] + ((3 if triple else 2) * K,), dtype=torch.bfloat16, device=x.device)
    L = _bind2()
    if not getattr(L, "_pafc_split_planes_bound", False):
        from ctypes import c_long
        _lib._sig(L.pafc_split_planes, c_int, c_long, c_int, c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_void_p)
        L._pafc_split_planes_bound = True
    _lib.check(L.pafc_split_planes(rows, K, _lib.ptr(x), K, _lib.ptr(out), out.shape[-1], K, int(triple), _lib.stream_of(x)),
               "pafc_split_planes")
    return out


def _ph_tile_m(M: int, N: int, batch: int = 1, min_tm: int = 64) -> int:
    """Rows per tile of the phase-pipelined GEMM for this problem (the count that needs the fewest rounds of one-tile-per-CU
    work, a round weighted by its rows plus a fixed per-tile part worth ~64 rows), as csrc/gemm_bf16.hip:ph_tile_m."""
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    nt = (N + 255) // 256
    best = None
    for tm in (256, 192, 128, 64):      # (64: a few thousand fp32 rows as split operands, profiles/r05_split_mid_rows.txt)
        if tm < min_tm:
            continue
        tiles = ((M + tm - 1) // tm) * nt * batch
        cost = ((tiles + cus - 1) // cus) * (tm + 64)
        if best is None or cost < best[0]:
            best = (cost, tm)
    return best[1]


def gemm_ph_ex(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: str = "none", alpha: float = 1.0,
               residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, a_split: bool = False,
               out_kind: str = "bf16", tile_m: int = 0, a_plane_block: int = 0) -> torch.Tensor:
    """The phase-pipelined GEMM with every operand form (include/pafc_encoder_ops.h: pafc_gemm_ph_ex).
    a: (M, K) bf16, or with a_split (M, 2K) planes [hi | lo] of an fp32 activation, then w: (N, 3K) = split_planes(W, triple=True);
    out_kind "bf16" | "f32" | "planes" ((M, 2N) bf16 = [hi | lo] of the fp32 result); residual bf16 (out bf16) or fp32 (out f32),
    may be `out`; bias bf16 for out_kind bf16, fp32 otherwise; act as gemm_bf16 (GLU: glu_interleave(w, 32) rows before splitting).
    a_plane_block: the planes of a split `a` alternate in blocks of that many columns ([hi PB | lo PB] ...; 0: [hi K | lo K])."""
    _lib.require_gpu(bias)
    for t in (a, w, residual, out):
        if t is not None and (not t.is_cuda or t.stride(-1) != 1):
            raise _lib.PafcError("gemm_ph_ex: GPU tensors with unit stride in the last dimension")
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or a.dim() != 2 or w.dim() != 2:
        raise _lib.PafcError("gemm_ph_ex: a and w are 2-d bf16 (planes of fp32 operands with a_split)")
    M = a.shape[0]
    N = w.shape[0]
    K = w.shape[1] // 3 if a_split else w.shape[1]
    if a.shape[1] != (2 * K if a_split else K) or (a_split and w.shape[1] != 3 * K):
        raise _lib.PafcError("gemm_ph_ex: a (M, K) x w (N, K); split: a (M, 2K) x w (N, 3K)")
    No = N // 2 if act == "glu" else N
    kinds = {"bf16": (0, torch.bfloat16, No), "f32": (1, torch.float32, No), "planes": (2, torch.bfloat16, 2 * No)}
    ok, odt, ocols = kinds[out_kind]
    if out is None:
        out = torch.empty((M, ocols), dtype=odt, device=a.device)
    if out.dtype != odt or tuple(out.shape) != (M, ocols):
        raise _lib.PafcError("gemm_ph_ex: out must be (M, N) in the output form's dtype ((M, 2N) bf16 for planes)")
    rk = 0
    if residual is not None:
        rk = 1 if residual.dtype == torch.bfloat16 else 2
        if tuple(residual.shape) != (M, N) or residual.dtype != (torch.bfloat16 if ok == 0 else torch.float32) or ok == 2:
            raise _lib.PafcError("gemm_ph_ex: residual (M, N) in the output dtype (bf16 / fp32), not with planes")
    if bias is not None and (bias.dtype != (torch.bfloat16 if ok == 0 else torch.float32) or bias.numel() != N):
        raise _lib.PafcError("gemm_ph_ex: bias (N) bf16 for a bf16 output, fp32 otherwise")
    L = _bind2()
    if not getattr(L, "_pafc_gemmex_bound", False):
        from ctypes import c_float, c_long
        P, I, G = c_void_p, c_int, c_long
        _lib._sig(L.pafc_gemm_ph_ex2, I, G, I, I, I, P, G, G, I, I, P, G, G, P, G, P, I, G, G, P, I, G, G, G, c_float, I, I, P)
        from ctypes import c_size_t
        _lib._sig(L.pafc_gemm_bf16_f32out_pb, I, G, I, I, P, G, I, I, P, G, P, P, G, P, I, G, G, c_float, I, P, c_size_t, P)
        _lib._sig(L.pafc_gemm_bf16_f32out_workspace_bytes, c_size_t, G, I, I, I)
        L._pafc_gemmex_bound = True
    from .profiling import op_timer
    pb_ok = not a_plane_block or (a_split and a_plane_block >= 64 and a_plane_block & (a_plane_block - 1) == 0 and K % a_plane_block == 0)
    if (M <= _SPLIT_SMALL_MAX_ROWS and M * N <= _SPLIT_SMALL_MAX_OUT and ok != 0 and act in ("none", "silu", "tanh", "relu")
            and pb_ok and not tile_m and K % 64 == 0):
        # few rows: the small tiles of csrc/gemm_bf16.hip (same operand forms, same epilogue order)
        nws = L.pafc_gemm_bf16_f32out_workspace_bytes(M, N, K, int(a_split))     # > 0: few rows x long K, K split over blocks
        ws = torch.empty(nws, dtype=torch.uint8, device=a.device) if nws else None
        with op_timer("gemm%ss_%dx%d" % ("3" if a_split else "", K, N), sample=12, flops=2.0 * M * N * K * (3 if a_split else 1)):
            rc = L.pafc_gemm_bf16_f32out_pb(M, N, K, _lib.ptr(a), a.stride(0), int(a_split), int(a_plane_block), _lib.ptr(w), w.stride(0),
                                            _lib.ptr(bias), _lib.ptr(residual), residual.stride(0) if residual is not None else 0,
                                            _lib.ptr(out), ok, out.stride(0), No if ok == 2 else 0, float(alpha), _ACTS[act], _lib.ptr(ws),
                                            nws, _lib.stream_of(a))
        _lib.check(rc, "pafc_gemm_bf16_f32out")
        return out
    with op_timer("gemm%s_%dx%d" % ("3" if a_split else "", K, N), sample=12, flops=2.0 * M * N * K * (3 if a_split else 1)):
        rc = L.pafc_gemm_ph_ex2(M, N, K, 1, _lib.ptr(a), a.stride(0), 0, int(a_split), int(a_plane_block), _lib.ptr(w), w.stride(0), 0,
                                _lib.ptr(bias), 0,
                               _lib.ptr(residual), rk, residual.stride(0) if residual is not None else 0, 0, _lib.ptr(out), ok,
                               out.stride(0), No if ok == 2 else 0, 0, float(alpha), _ACTS[act], int(tile_m or _ph_tile_m(M, N)),
                               _lib.stream_of(a))
    _lib.check(rc, "pafc_gemm_ph_ex")
    return out


_SPLIT_SMALL_MAX_ROWS = DISPATCH["split_small_max_rows"]
_SPLIT_SMALL_MAX_OUT = 1 << 22          # ... and at most this many outputs (rows x columns)


def wkv6_single_chunk(B: int, T: int, C: int, H: int) -> bool:
    """Does the library walk a (B, T, C) one-direction scan as ONE chunk (then the carried state may be updated in place)?"""
    return _lib.lib().pafc_wkv6_pick_chunk_len(B, T, C, H, 1) >= T


def _check_prev(prev: Optional[torch.Tensor], x: torch.Tensor):
    """prev: (B, 1, C) / (B, C) = the frame before each sequence's first one (streaming), same dtype as x, contiguous."""
    if prev is not None and (prev.dtype != x.dtype or prev.numel() != x.shape[0] * x.shape[2] or not prev.is_contiguous()
                             or prev.device != x.device):
        raise _lib.PafcError("tmix: prev must be a contiguous (B, C) tensor of x's dtype on x's device")


def tmix_shift_mix(x: torch.Tensor, maa_x0: torch.Tensor, maa_x1: Optional[torch.Tensor], reverse0: bool = False,
                   prev: Optional[torch.Tensor] = None):
    """(B, T, C) -> (ndir, B, T, C): x + (shift_d(x) - x) * maa_x_d.  prev: the frame before the chunk (streaming), or None."""
    _lib.require_gpu(x, maa_x0, maa_x1, prev)
    _check_prev(prev, x)
    B, T, C = x.shape
    ndir = 2 if maa_x1 is not None else 1
    out = torch.empty((ndir, B, T, C), dtype=x.dtype, device=x.device)
    rc = _bind2().pafc_tmix_shift_mix_prev(_lib.dtype_code(x.dtype), B, T, C, ndir, int(reverse0), _lib.ptr(x),
                                           _lib.ptr(maa_x0), _lib.ptr(maa_x1), _lib.ptr(prev), _lib.ptr(out), _lib.stream_of(x))
    _lib.check(rc, "pafc_tmix_shift_mix")
    return out


def tmix_mix4(x: torch.Tensor, m: torch.Tensor, maa: torch.Tensor, reverse0: bool = False):
    """x (B,T,C), m (ndir,4,B*T,C), maa (ndir,4,C) -> z (4,ndir,B*T,C)."""
    _lib.require_gpu(x, m, maa)
    B, T, C = x.shape
    ndir = m.shape[0]
    if m.dtype != x.dtype or maa.dtype != x.dtype:
        raise _lib.PafcError("tmix_mix4: x, m and maa must share one dtype")
    z = torch.empty((4, ndir, B * T, C), dtype=x.dtype, device=x.device)
    rc = _bind2().pafc_tmix_mix4(_lib.dtype_code(x.dtype), B, T, C, ndir, int(reverse0), _lib.ptr(x), _lib.ptr(m),
                                 _lib.ptr(maa), _lib.ptr(z), _lib.stream_of(x))
    _lib.check(rc, "pafc_tmix_mix4")
    return z


def _bind_tmix_bwd():
    L = _bind2()
    if not getattr(L, "_pafc_tmixbwd_bound", False):
        from ctypes import c_long, c_size_t
        P, I = c_void_p, c_int
        L.pafc_tmix_bwd_workspace_bytes.restype = c_size_t
        L.pafc_tmix_bwd_workspace_bytes.argtypes = [c_long, c_int]
        _lib._sig(L.pafc_tmix_shift_mix_bwd, I, I, I, I, I, I, P, P, P, P, P, P, c_size_t, P)
        _lib._sig(L.pafc_tmix_mix4_bwd, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, c_size_t, P)
        _lib._sig(L.pafc_tmix_mix4_bwd_rows, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, c_size_t, P)
        L._pafc_tmixbwd_bound = True
    return L


class _ShiftMixTrain(torch.autograd.Function):
    """xxx = x + (shift(x) - x) * maa_x (src/model.py:274-276) for one direction: forward tmix_shift_mix, backward one
    pass (pafc_tmix_shift_mix_bwd) instead of the framework's pad / sub / mul / add chain and its reduction."""

    @staticmethod
    def forward(ctx, x, maa_x, reverse):
        mx = maa_x.reshape(-1).to(x.dtype).contiguous()
        ctx.save_for_backward(x, mx)
        ctx.reverse, ctx.maa_shape, ctx.maa_dtype = reverse, maa_x.shape, maa_x.dtype
        return tmix_shift_mix(x, mx, None, reverse)[0]

    @staticmethod
    def backward(ctx, dxxx):
        x, mx = ctx.saved_tensors
        B, T, C = x.shape
        L = _bind_tmix_bwd()
        nbytes = L.pafc_tmix_bwd_workspace_bytes(B * T, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        dx = torch.empty_like(x)
        dmaa = torch.empty(C, dtype=torch.float32, device=x.device)
        dxxx = dxxx.contiguous()
        rc = L.pafc_tmix_shift_mix_bwd(_lib.dtype_code(x.dtype), B, T, C, int(ctx.reverse), _lib.ptr(x), _lib.ptr(mx),
                                       _lib.ptr(dxxx), _lib.ptr(dx), _lib.ptr(dmaa), _lib.ptr(ws), nbytes, _lib.stream_of(x))
        _lib.check(rc, "pafc_tmix_shift_mix_bwd")
        return dx, dmaa.to(ctx.maa_dtype).view(ctx.maa_shape), None


class _Mix4Train(torch.autograd.Function):
    """z_q = x + (shift(x) - x) * (maa_q + m_q), q = r, k, v, w (src/model.py:280-284) for one direction: forward
    tmix_mix4, backward one pass (pafc_tmix_mix4_bwd).  Returns the four maps as separate outputs so that autograd hands
    back four gradients instead of assembling one stacked tensor."""

    @staticmethod
    def forward(ctx, x, m, maa4, reverse):
        B, T, C = x.shape
        mm = m.reshape(1, 4, B * T, C).to(x.dtype).contiguous()     # under autocast the LoRA product is bf16, x may be fp32
        a4 = maa4.reshape(1, 4, C).to(x.dtype).contiguous()
        z = tmix_mix4(x, mm, a4, reverse)                           # (4, 1, B*T, C)
        ctx.save_for_backward(x, mm, a4)
        ctx.reverse, ctx.maa_shape, ctx.maa_dtype, ctx.m_shape, ctx.m_dtype = reverse, maa4.shape, maa4.dtype, m.shape, m.dtype
        return tuple(z[q, 0].view(B, T, C) for q in range(4))

    @staticmethod
    def backward(ctx, *dz):
        x, mm, a4 = ctx.saved_tensors
        B, T, C = x.shape
        dz = [(torch.zeros_like(x) if g is None else g.contiguous()) for g in dz]
        L = _bind_tmix_bwd()
        nbytes = L.pafc_tmix_bwd_workspace_bytes(B * T, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        dx = torch.empty_like(x)
        dm = torch.empty_like(mm)
        dmaa = torch.empty(4, C, dtype=torch.float32, device=x.device)
        P = _lib.ptr
        rc = L.pafc_tmix_mix4_bwd(_lib.dtype_code(x.dtype), B, T, C, int(ctx.reverse), P(x), P(mm), P(a4), P(dz[0]), P(dz[1]),
                                  P(dz[2]), P(dz[3]), P(dx), P(dm), P(dmaa), P(ws), nbytes, _lib.stream_of(x))
        _lib.check(rc, "pafc_tmix_mix4_bwd")
        return dx, dm.view(ctx.m_shape).to(ctx.m_dtype), dmaa.to(ctx.maa_dtype).view(ctx.maa_shape), None


_eye4 = {}      # device -> (4, 1, 4, 1) bf16 identity: the block-diagonal placement of the four LoRA-up matrices


class _LoraMix4Train(torch.autograd.Function):
    """m_q = t_q W2_q (the LoRA-up products, src/model.py:277-278: `torch.bmm` over four (B*T, 32) x (32, C) pairs) followed by the
    four lerps z_q = x + (shift(x) - x) (maa_q + m_q) (280-284), for one direction, every product on the hand-written kernels:
      forward   m = t (B*T, 128) against the four matrices laid out block-diagonally along K ((4, C, 128): block q of the K axis
                holds W2_q^T, zeros elsewhere -- 4 x the flops of a product that is 2 GFLOP) as ONE batched pafc_gemm_bf16, then
                pafc_tmix_mix4;
      backward  pafc_tmix_mix4_bwd_rows leaves dm as (B*T, 4, C); dt = ONE batched pafc_gemm_bf16 (dm's four column blocks against
                W2_q as they lie, into the four column blocks of dt); dW2 = the diagonal blocks of ONE pafc_gemm_tn_bf16
                t^T (128, B*T) x dm (B*T, 4 C).
    Round 5 ran the three products through the library (Cijk MT256x256x32 32 us, MT32x128x32 19 us, MT16x16x128 74 us per direction
    and layer) with a transposed copy of t in front of the last one."""

    @staticmethod
    def forward(ctx, x, t, w2, maa_r, maa_k, maa_v, maa_w, reverse):
        B, T, C = x.shape
        M, R = B * T, w2.shape[1]
        maas = (maa_r, maa_k, maa_v, maa_w)
        maa4 = _params_as_one(maas)                                               # (4, C): no launch once the four live in one buffer
        if maa4 is None:
            maa4 = torch.stack([m_.detach().reshape(C) for m_ in maas])
        tb = t.reshape(M, 4 * R)
        if tb.dtype != torch.bfloat16 or not tb.is_contiguous():
            tb = tb.to(torch.bfloat16).contiguous()
        w2b = _param_as(w2, torch.bfloat16).contiguous()                          # (4, R, C)
        eye = _eye4.get(x.device)
        if eye is None:
            eye = _eye4[x.device] = torch.eye(4, dtype=torch.bfloat16, device=x.device).view(4, 1, 4, 1)
        w2p = torch.empty((4, C, 4, R), dtype=torch.bfloat16, device=x.device)
        torch.mul(w2b.transpose(1, 2).unsqueeze(2), eye, out=w2p)                  # one launch
        w2p = w2p.view(4, C, 4 * R)
        m = gemm_bf16(tb.unsqueeze(0).expand(4, M, 4 * R), w2p)                    # (4, M, C)
        mm = m.view(1, 4, M, C) if x.dtype == torch.bfloat16 else m.view(1, 4, M, C).to(x.dtype)
        a4 = maa4.reshape(1, 4, C).to(x.dtype).contiguous()
        z = tmix_mix4(x, mm, a4, reverse)                                          # (4, 1, M, C)
        ctx.save_for_backward(x, tb, w2b, mm, a4)
        ctx.reverse, ctx.maa_shape, ctx.maa_dtype, ctx.t_shape, ctx.t_dtype, ctx.w2_dtype = (reverse, maa_r.shape, maa_r.dtype, t.shape,
                                                                                             t.dtype, w2.dtype)
        return tuple(z[q, 0].view(B, T, C) for q in range(4))

    @staticmethod
    def backward(ctx, *dz):
        x, tb, w2b, mm, a4 = ctx.saved_tensors
        B, T, C = x.shape
        M, R = B * T, w2b.shape[1]
        dz = [(torch.zeros_like(x) if g is None else g.contiguous()) for g in dz]
        L = _bind_tmix_bwd()
        nbytes = L.pafc_tmix_bwd_workspace_bytes(M, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        dx = torch.empty_like(x)
        dm = torch.empty((M, 4, C), dtype=x.dtype, device=x.device)
        dmaa = torch.empty(4, C, dtype=torch.float32, device=x.device)
        P = _lib.ptr
        rc = L.pafc_tmix_mix4_bwd_rows(_lib.dtype_code(x.dtype), B, T, C, int(ctx.reverse), P(x), P(mm), P(a4), P(dz[0]), P(dz[1]),
                                       P(dz[2]), P(dz[3]), P(dx), P(dm), P(dmaa), P(ws), nbytes, _lib.stream_of(x))
        _lib.check(rc, "pafc_tmix_mix4_bwd_rows")
        if dm.dtype != torch.bfloat16:
            dm = dm.to(torch.bfloat16)
        dt = dw2 = None
        if ctx.needs_input_grad[1]:
            dt = torch.empty((M, 4 * R), dtype=torch.bfloat16, device=x.device)
            gemm_bf16(dm.transpose(0, 1), w2b, out=dt.view(M, 4, R).transpose(0, 1))    # (4, M, C) x (4, R, C)^T -> (4, M, R)
            dt = dt.view(ctx.t_shape).to(ctx.t_dtype)
        if ctx.needs_input_grad[2]:
            od = ctx.w2_dtype if ctx.w2_dtype in (torch.float32, torch.bfloat16) else torch.float32
            full = gemm_tn(tb, dm.view(M, 4 * C), od)                                  # (4 R, 4 C): its diagonal blocks
            dw2 = torch.stack([full[q * R:(q + 1) * R, q * C:(q + 1) * C] for q in range(4)]).to(ctx.w2_dtype)
        dm4 = dmaa.to(ctx.maa_dtype)
        return (dx, dt, dw2, dm4[0].view(ctx.maa_shape), dm4[1].view(ctx.maa_shape), dm4[2].view(ctx.maa_shape),
                dm4[3].view(ctx.maa_shape), None)


def lora_mix4_train_eligible(x: torch.Tensor, t: torch.Tensor, w2: torch.Tensor) -> bool:
    """The LoRA-up + lerp group on own kernels: the element-wise group's conditions, bf16 LoRA activations, four (R, C) matrices with
    4 R a multiple of 64 (the K axis of the block-diagonal product) and C of 64 (that of the input-gradient product)."""
    if os.environ.get("PAFC_TRAIN_LORA_UP", "1") == "0":          # A/B runs: torch.bmm + mix4_train, as in round 5
        return False
    return (tmix_train_eligible(x) and train_gemms_own() and t.dtype == torch.bfloat16 and w2.dim() == 3 and w2.shape[0] == 4
            and (4 * w2.shape[1]) % 64 == 0 and w2.shape[2] == x.shape[-1] and x.shape[-1] % 64 == 0
            and w2.dtype in (torch.float32, torch.bfloat16) and t.numel() == x.shape[0] * x.shape[1] * 4 * w2.shape[1])


def lora_mix4_train(x: torch.Tensor, t: torch.Tensor, w2: torch.Tensor, maas, reverse: bool):
    """t: (B, T, 4 R) = tanh(xxx W1); w2: (4, R, C); maas: the four lerp coefficients (time_maa_r, _k, _v, _w), C elements each.
    -> z_r, z_k, z_v, z_w (B, T, C)."""
    return _LoraMix4Train.apply(x.contiguous(), t, w2, *maas, reverse)


def tmix_train_eligible(x: torch.Tensor) -> bool:
    return (x.is_cuda and torch.is_grad_enabled() and train_kernels_enabled() and x.dim() == 3
            and x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] % 8 == 0 and x.shape[-1] <= 1024)


def shift_mix_train(x: torch.Tensor, maa_x: torch.Tensor, reverse: bool) -> torch.Tensor:
    return _ShiftMixTrain.apply(x.contiguous(), maa_x, reverse)


def mix4_train(x: torch.Tensor, m: torch.Tensor, maa4: torch.Tensor, reverse: bool):
    """m: (4, B, T, C) LoRA outputs in the order r, k, v, w; maa4: (4, C) (or anything that reshapes to it)."""
    return _Mix4Train.apply(x.contiguous(), m, maa4, reverse)


def gemm_f32(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: str = "none", alpha: float = 1.0,
             residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Hand-written fp32 GEMM with a fused epilogue on the fp32 matrix cores (include/pafc_encoder_ops.h: pafc_gemm_f32; exact
    fp32 products, fp32 accumulation): act(alpha * a w^T + bias + residual).  a: (M, K) or (Z, M, K); w: (N, K) or (Z, N, K) in
    nn.Linear layout; bias (N) or (Z, N); residual / out (M, N) or (Z, M, N), `out` may be `residual`.  Rows and batch entries
    may be strided views (unit stride in the last dimension; K and the strides of a and w multiples of 4).  act: none / silu /
    tanh / relu.  No workspace, no plan objects: safe to issue from several streams at once."""
    _lib.require_gpu(bias)
    for t in (a, w, bias, residual, out):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1):
            raise _lib.PafcError("gemm_f32: fp32 GPU tensors with unit stride in the last dimension")
    L = _bind2()
    if not getattr(L, "_pafc_gemmf32_bound", False):
        from ctypes import c_float, c_long
        P, I, G = c_void_p, c_int, c_long
        _lib._sig(L.pafc_gemm_f32, I, G, I, I, I, P, G, G, P, G, G, P, G, P, G, G, P, G, G, c_float, I, P)
        L._pafc_gemmf32_bound = True
    batched = a.dim() == 3
    Z = a.shape[0] if batched else 1
    M, K = a.shape[-2], a.shape[-1]
    N = w.shape[-2]
    if w.shape[-1] != K or (batched and (w.dim() != 3 or w.shape[0] != Z)) or (not batched and (a.dim() != 2 or w.dim() != 2)):
        raise _lib.PafcError("gemm_f32: a (M, K) x w (N, K), or both with a leading batch")
    if act not in ("none", "silu", "tanh", "relu"):
        raise _lib.PafcError("gemm_f32: act is none / silu / tanh / relu")
    if out is None:
        out = torch.empty((Z, M, N) if batched else (M, N), dtype=a.dtype, device=a.device)
    for t in (residual, out):
        if t is not None and tuple(t.shape) != ((Z, M, N) if batched else (M, N)):
            raise _lib.PafcError("gemm_f32: residual / out must be (M, N) per batch entry")
    if bias is not None and bias.shape[-1] != N:
        raise _lib.PafcError("gemm_f32: bias must be (N) or (Z, N)")
    sb = bias.stride(0) if (bias is not None and bias.dim() == 2) else 0
    bs = lambda t: t.stride(0) if batched else 0
    from .profiling import op_timer
    with op_timer("gemmf32_%dx%d%s" % (K, N, "x%d" % Z if batched else ""), sample=12, flops=2.0 * Z * M * N * K):
        rc = L.pafc_gemm_f32(M, N, K, Z, _lib.ptr(a), a.stride(-2), bs(a), _lib.ptr(w), w.stride(-2), bs(w),
                             _lib.ptr(bias), sb, _lib.ptr(residual), residual.stride(-2) if residual is not None else 0,
                             bs(residual) if residual is not None else 0, _lib.ptr(out), out.stride(-2), bs(out),
                             float(alpha), _ACTS[act], _lib.stream_of(a))
    _lib.check(rc, "pafc_gemm_f32")
    return out


def gemm_f32_ok(a: torch.Tensor, w: torch.Tensor) -> bool:
    """Operand forms pafc_gemm_f32 takes: fp32, unit stride along K, K and the row / batch strides multiples of 4, 16-byte bases."""
    return (a.is_cuda and a.dtype == torch.float32 and w.dtype == torch.float32 and a.shape[-1] % 4 == 0
            and all(t.stride(-1) == 1 and all(st % 4 == 0 for st in t.stride()[:-1]) and t.data_ptr() % 16 == 0 for t in (a, w)))


def linear_bias_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: str = "silu",
                    alpha: float = 1.0, residual: Optional[torch.Tensor] = None, inplace: bool = False):
    """act(alpha * x @ weight.T + residual + bias) as ONE GEMM with a fused epilogue (act: 'silu' or 'none'), for the operands
    the tiled bf16 kernels do not take: fp32 (exact fp32 products on the fp32 matrix cores, pafc_gemm_f32) and bf16 shapes off the
    bf16 kernels' grid (K % 64 or N % 8: through the same kernel in fp32, one rounding at the end).  bias is added as given
    (not scaled by alpha).  inplace: write the result over ``residual``.

    Until round 6 this was a hipBLASLt plan per problem, falling back to the framework's F.linear where the library offered no
    workspace-free kernel.  Both are gone from the inference paths: the framework's fp32 GEMM stalls when two HIP streams issue
    it at once (DESIGN.md section 4 "the c2 stall"), and decode batches are issued two or three streams deep."""
    _lib.require_gpu(x, weight, bias, residual)
    N, K = weight.shape
    rows = x.numel() // K
    if x.shape[-1] != K or weight.dtype != x.dtype or (bias is not None and bias.dtype != x.dtype):
        raise _lib.PafcError("linear_bias_act: shape/dtype mismatch")
    if residual is not None and (residual.dtype != x.dtype or residual.numel() != rows * N):
        raise _lib.PafcError("linear_bias_act: residual must be (rows, N) in the activation dtype")
    if act not in ("silu", "none"):
        raise _lib.PafcError("linear_bias_act: act is 'silu' or 'none'")
    x2 = x.reshape(rows, K)
    r2 = residual.reshape(rows, N) if residual is not None else None
    if x.dtype == torch.float32 and gemm_f32_ok(x2, weight):
        out = gemm_f32(x2, weight, bias, act, alpha=alpha, residual=r2, out=r2 if (inplace and r2 is not None) else None)
        return residual if (inplace and residual is not None) else out.view(x.shape[:-1] + (N,))
    if x.dtype == torch.bfloat16 and K % 4 == 0:
        y = gemm_f32(x2.float(), weight.float(), None if bias is None else bias.float(), act, alpha=alpha,
                     residual=None if r2 is None else r2.float())
        if inplace and residual is not None:
            r2.copy_(y)
            return residual
        return y.to(torch.bfloat16).view(x.shape[:-1] + (N,))
    # K not a multiple of 4 (no layer of this package has one): the framework's operators, still on the GPU
    y = torch.nn.functional.linear(x, weight)
    if alpha != 1.0:
        y = y * alpha
    if bias is not None:
        y = y + bias
    if residual is not None:           # act(alpha * x W^T + residual + bias), as the fused kernels compute it
        y = y + residual.view(y.shape)
    if act == "silu":
        y = torch.nn.functional.silu(y)
    if residual is not None and inplace:
        residual.view(y.shape).copy_(y)
        return residual
    return y


# fp32 GEMMs of long inputs run on the bf16 matrix cores with split operands (csrc/gemm_ph.hip: three bf16 products per fp32
# product, fp32 accumulation, ~2^-16 relative); shorter ones take exact fp32 products on the fp32 matrix cores (gemm_f32).
_SPLIT_GEMM_MIN_ROWS = DISPATCH["split_gemm_min_rows"]
_split_weights = {}      # id(weight) -> (stamp, [hi | hi | lo] planes, weakref to the weight); bounded

# Derived copies of parameters (stacked / transposed / split / LayerNorm-folded weights of the inference plans) are keyed on
# (storage, Tensor._version).  Fused optimizers (`torch.optim.Adam(fused=True)`, `p.data` writes) update parameters WITHOUT
# touching `_version`, so the keys carry this process-wide epoch as well: it is bumped after every optimizer step of
# utils.train_utils.train_step and whenever an encoder switches between train() and eval() -- a CV pass between training
# steps therefore never multiplies stale weights.  Code that updates parameters behind torch's back calls it itself.
_param_epoch = 0


def param_epoch() -> int:
    return _param_epoch


def bump_param_epoch() -> None:
    global _param_epoch
    _param_epoch += 1


class DerivedFill:
    """Which stream issued the kernels that FILL a set of derived tensors (a plan's stacked / split / folded weights).  The fill
    is asynchronous: a forward pass on ANOTHER stream (utils.longform runs decode batches on side streams that only wait for the
    caller's stream) must wait for it before its kernels read the tensors.  `DerivedFill()` right after the fill records an event
    on the filling stream; `use()` at the top of every consumer makes the current stream wait for that event the first time it
    meets the fill (one dictionary look-up afterwards).  Under graph capture nothing is waited for: an eager pass on the same
    stream always precedes a capture (the graph caches capture a shape's SECOND sighting)."""
    __slots__ = ("event", "seen")

    def __init__(self, device=None):
        self.event, self.seen = None, set()
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            st = torch.cuda.current_stream(device)
            self.event = torch.cuda.Event()
            self.event.record(st)
            self.seen.add(st.cuda_stream)

    # an event belongs to the process and stream that recorded it: a copy (copy.deepcopy of a model for EMA / averaging,
    # pickling for spawn / torch.save) starts empty -- its tensors are copies that nobody is still filling
    def __deepcopy__(self, memo):
        return _empty_fill()

    def __reduce__(self):
        return (_empty_fill, ())

    def use(self, device=None) -> None:
        if self.event is None:
            return
        raw = _lib._raw_stream(device.index if device is not None and device.index is not None else torch.cuda.current_device()) \
            if _lib._raw_stream is not None else torch.cuda.current_stream(device).cuda_stream
        if raw in self.seen or torch.cuda.is_current_stream_capturing():
            return
        torch.cuda.current_stream(device).wait_event(self.event)
        self.seen.add(raw)



def _empty_fill() -> DerivedFill:
    f = DerivedFill.__new__(DerivedFill)
    f.event, f.seen = None, set()
    return f


def split_weight_cached(weight: torch.Tensor) -> torch.Tensor:
    """[hi | hi | lo] planes of a weight, cached per weight OBJECT: an entry is valid only for the very tensor it was made
    from (a weak reference is compared by identity -- a freed tensor's id and allocator block can both come back for a new
    tensor of the same shape, whose `_version` is 0 again) and for its stamp (storage, in-place version, shape)."""
    import weakref
    stamp = (weight.data_ptr(), weight._version, tuple(weight.shape), _param_epoch)
    ent = _split_weights.get(id(weight))
    if ent is None or ent[0] != stamp or ent[2]() is not weight:
        if len(_split_weights) > 64:
            _split_weights.clear()
        planes = split_planes(weight.detach().contiguous(), triple=True)
        ent = _split_weights[id(weight)] = (stamp, planes, weakref.ref(weight), DerivedFill(weight.device))
    else:
        ent[3].use(weight.device)            # filled on another stream: that fill first
    return ent[1]


def linear_fused(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: str = "none",
                 split_ok: bool = True) -> torch.Tensor:
    """act(x @ weight.T + bias) with the epilogue fused: bf16 operands on the hand-written GEMM (K % 64 == 0, N % 8 == 0),
    long fp32 inputs on the same kernel with split operands (unless split_ok is False: a pure-fp32 model's exact products),
    anything else through linear_bias_act (the hand-written fp32 GEMM)."""
    N, K = weight.shape
    rows = x.numel() // K
    if (x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.is_cuda and weight.is_contiguous()
            and skinny_ok(rows, N, K) and act != "glu"):         # a streaming chunk: the few-rows kernel
        return gemm_skinny(x.reshape(-1, K), weight, bias, act).view(x.shape[:-1] + (N,))
    if x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and K % 64 == 0 and N % 8 == 0:
        return gemm_bf16(x.reshape(-1, K), weight, bias, act).view(x.shape[:-1] + (N,))
    if (x.dtype == torch.float32 and weight.dtype == torch.float32 and x.is_cuda and K % 128 == 0 and N % 8 == 0 and N >= 256
            and rows >= _SPLIT_GEMM_MIN_ROWS and split_ok and act == "none" and not torch.is_grad_enabled()
            and (bias is None or bias.dtype == torch.float32)):
        planes = split_planes(x.reshape(rows, K) if x.is_contiguous() else x.reshape(rows, K).contiguous())
        return gemm_ph_ex(planes, split_weight_cached(weight), bias, a_split=True, out_kind="f32").view(x.shape[:-1] + (N,))
    return linear_bias_act(x, weight, bias, act)


def tmix_lora_mix4(x: torch.Tensor, t: torch.Tensor, w2t: torch.Tensor, maa: torch.Tensor, reverse0: bool = False,
                   prev: Optional[torch.Tensor] = None):
    """bf16: x (B,T,C), t (ndir,B*T,128) = tanh(xxx W1), w2t (ndir,4,C,32), maa (ndir,4,C) -> z (4,ndir,B*T,C).
    prev: the frame before the chunk (streaming), or None."""
    _lib.require_gpu(x, t, w2t, maa, prev)
    _check_prev(prev, x)
    B, T, C = x.shape
    ndir = t.shape[0]
    if x.dtype != torch.bfloat16 or t.shape != (ndir, B * T, 128) or w2t.shape != (ndir, 4, C, 32):
        raise _lib.PafcError("tmix_lora_mix4: bf16 only, t (ndir, B*T, 128), w2t (ndir, 4, C, 32)")
    z = torch.empty((4, ndir, B * T, C), dtype=x.dtype, device=x.device)
    rc = _bind2().pafc_tmix_lora_mix4_bf16_prev(B, T, C, ndir, int(reverse0), _lib.ptr(x), _lib.ptr(t), _lib.ptr(w2t),
                                                _lib.ptr(maa), _lib.ptr(prev), _lib.ptr(z), _lib.stream_of(x))
    _lib.check(rc, "pafc_tmix_lora_mix4_bf16")
    return z


# The one-pass LoRA kernels stage 128 KiB of weights into LDS per block: worth it from a couple of thousand rows on (30-minute
# file: 45 k rows, a c2 batch: ~16 k) and, as ONE launch instead of two or three, in the launch-bound streaming chunk step
# (`one_pass=True`, fused.layer_forward_carry); below it the small-GEMM path (DISPATCH table at the top of this file).
_LDS_RESIDENT_MIN_ROWS = DISPATCH["lds_resident_min_rows"]


def tmix_lora_down(x: torch.Tensor, maa_x: torch.Tensor, w1n: torch.Tensor, reverse0: bool = False,
                   prev: Optional[torch.Tensor] = None, one_pass: Optional[bool] = None):
    """bf16: t = tanh((x + (x_nb - x) * maa_x) w1n^T) in one pass.  x (B, T, C), maa_x (ndir, C), w1n (ndir, N, C) ->
    (ndir, B*T, N).  C = 512, N = 128 run the fused kernel (weights resident in LDS); other sizes take the shift/lerp pass and
    a GEMM with the same roundings."""
    _lib.require_gpu(x, maa_x, w1n, prev)
    _check_prev(prev, x)
    L = _bind2()
    B, T, C = x.shape
    ndir, N = w1n.shape[0], w1n.shape[1]
    if x.dtype != torch.bfloat16 or w1n.shape != (ndir, N, C) or maa_x.shape != (ndir, C):
        raise _lib.PafcError("tmix_lora_down: bf16 only, maa_x (ndir, C), w1n (ndir, N, C)")
    for a in (x, maa_x, w1n):
        if not a.is_contiguous() or a.dtype != torch.bfloat16:
            raise _lib.PafcError("tmix_lora_down: contiguous bf16 tensors")
    if C == 512 and N == 128 and (B * T >= _LDS_RESIDENT_MIN_ROWS if one_pass is None else one_pass):
        t = torch.empty((ndir, B * T, N), dtype=x.dtype, device=x.device)
        from .profiling import op_timer
        with op_timer("tmix_lora_down"):
            rc = L.pafc_tmix_lora_down_bf16_prev(B, T, C, N, ndir, int(reverse0), _lib.ptr(x), _lib.ptr(maa_x), _lib.ptr(w1n),
                                                 _lib.ptr(prev), _lib.ptr(t), _lib.stream_of(x))
        _lib.check(rc, "pafc_tmix_lora_down_bf16")
        return t
    xxx = tmix_shift_mix(x, maa_x[0], maa_x[1] if ndir == 2 else None, reverse0=reverse0, prev=prev)
    return gemm_bf16(xxx.view(ndir, B * T, C), w1n, act="tanh")


def decay_lora_one_pass(rows: int, C: int, H: int) -> bool:
    """Does decay_lora take its one-pass kernel (where a bias costs nothing) at this size?"""
    return C == 512 and H == 64 and rows >= _LDS_RESIDENT_MIN_ROWS


def decay_lora(zw: torch.Tensor, d1n: torch.Tensor, d2n: torch.Tensor, bias: Optional[torch.Tensor] = None,
               one_pass: Optional[bool] = None):
    """bf16: w = bf16(bf16(tanh(zw d1n^T)) d2n^T) [+ bias] in one pass.  zw (ndir, rows, C), d1n (ndir, H, C), d2n (ndir, C, H),
    bias (ndir, C) or None -> (ndir, rows, C).  C = 512, H = 64 run the fused kernel (weights resident in LDS); other sizes
    take two GEMMs with the same roundings."""
    _lib.require_gpu(zw, d1n, d2n, bias)
    L = _bind2()
    if not getattr(L, "_pafc_decay_bound", False):
        _lib._sig(L.pafc_decay_lora_bf16, c_int, ctypes.c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                  c_void_p, c_void_p)
        L._pafc_decay_bound = True
    ndir, rows, C = zw.shape
    H = d1n.shape[1]
    if zw.dtype != torch.bfloat16 or d1n.shape != (ndir, H, C) or d2n.shape != (ndir, C, H) or \
            (bias is not None and bias.numel() != ndir * C):
        raise _lib.PafcError("decay_lora: bf16 only, zw (ndir, rows, C), d1n (ndir, H, C), d2n (ndir, C, H), bias (ndir, C)")
    for a in (zw, d1n, d2n) + ((bias,) if bias is not None else ()):
        if not a.is_contiguous() or a.dtype != torch.bfloat16:
            raise _lib.PafcError("decay_lora: contiguous bf16 tensors")
    if C == 512 and H == 64 and (decay_lora_one_pass(rows, C, H) if one_pass is None else one_pass):
        w = torch.empty_like(zw)
        from .profiling import op_timer
        with op_timer("decay_lora"):
            rc = L.pafc_decay_lora_bf16(rows, C, H, ndir, _lib.ptr(zw), _lib.ptr(d1n), _lib.ptr(d2n), _lib.ptr(bias),
                                        _lib.ptr(w), _lib.stream_of(zw))
        _lib.check(rc, "pafc_decay_lora_bf16")
        return w
    w = gemm_bf16(gemm_bf16(zw, d1n, act="tanh"), d2n)
    return w if bias is None else w + bias.view(ndir, 1, C)


def conv3x3s2_nhwc(x: torch.Tensor, w_tap_co_ci: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = True):
    """x (B, T1, F1, Ci) bf16 NHWC, w (9, Co, Ci) -> (B, T2, F2, Co) = relu(conv3x3 stride 2 + bias)."""
    _lib.require_gpu(x, w_tap_co_ci, bias)
    L = _bind2()
    if not getattr(L, "_pafc_conv_bound", False):
        _lib._sig(L.pafc_conv3x3s2_nhwc_bf16, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                  c_void_p, c_int, c_void_p)
        L._pafc_conv_bound = True
    B, T1, F1, Ci = x.shape
    Co = w_tap_co_ci.shape[1]
    if x.dtype != torch.bfloat16 or w_tap_co_ci.shape != (9, Co, Ci) or w_tap_co_ci.dtype != x.dtype:
        raise _lib.PafcError("conv3x3s2_nhwc: bf16 NHWC input and a (9, Co, Ci) weight")
    out = torch.empty((B, (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1, Co), dtype=x.dtype, device=x.device)
    from .profiling import op_timer
    with op_timer("conv3x3s2", flops=2.0 * out.numel() * 9 * Ci):
        rc = L.pafc_conv3x3s2_nhwc_bf16(B, T1, F1, Ci, Co, _lib.ptr(x), _lib.ptr(w_tap_co_ci), _lib.ptr(bias),
                                        _lib.ptr(out), int(relu), _lib.stream_of(x))
    _lib.check(rc, "pafc_conv3x3s2_nhwc_bf16")
    return out


def conv3x3s2_nhwc_ph(x: torch.Tensor, w_tap_co_ci: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = True,
                      tile_m: int = 256):
    """The phase-pipelined implicit GEMM (csrc/gemm_ph.hip) by itself -- same arguments as conv3x3s2_nhwc, which picks it
    for the long-form shapes."""
    _lib.require_gpu(x, w_tap_co_ci, bias)
    L = _bind2()
    if not getattr(L, "_pafc_convph_bound", False):
        _lib._sig(L.pafc_conv3x3s2_nhwc_bf16_ph, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                  c_void_p, c_int, c_int, c_void_p)
        L._pafc_convph_bound = True
    B, T1, F1, Ci = x.shape
    Co = w_tap_co_ci.shape[1]
    if x.dtype != torch.bfloat16 or w_tap_co_ci.shape != (9, Co, Ci) or w_tap_co_ci.dtype != x.dtype:
        raise _lib.PafcError("conv3x3s2_nhwc_ph: bf16 NHWC input and a (9, Co, Ci) weight")
    out = torch.empty((B, (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1, Co), dtype=x.dtype, device=x.device)
    _lib.check(L.pafc_conv3x3s2_nhwc_bf16_ph(B, T1, F1, Ci, Co, _lib.ptr(x), _lib.ptr(w_tap_co_ci), _lib.ptr(bias), _lib.ptr(out),
                                             int(relu), int(tile_m), _lib.stream_of(x)), "pafc_conv3x3s2_nhwc_bf16_ph")
    return out


def ctc_greedy(scores: torch.Tensor, lens: Optional[torch.Tensor], blank_id: int = 0, want_frames: bool = False):
    """GPU-resident CTC greedy search (include/pafc_search.h): (B, T, V) scores -> (tokens (B, T) int32, ntok (B) int32
    [, first-frame index per token]).  Row b of ``tokens`` holds ``ntok[b]`` collapsed ids."""
    _lib.require_gpu(scores, lens)
    if scores.dim() != 3:
        raise _lib.PafcError("ctc_greedy wants (B, T, V) scores")
    L = _bind()
    B, T, V = scores.shape
    lens64 = None if lens is None else lens.to(torch.int64).contiguous()
    best = torch.empty(B, T, dtype=torch.int32, device=scores.device)
    tokens = torch.empty(B, T, dtype=torch.int32, device=scores.device)
    ntok = torch.empty(B, dtype=torch.int32, device=scores.device)
    frames = torch.empty(B, T, dtype=torch.int32, device=scores.device) if want_frames else None
    _lib.check(L.pafc_ctc_greedy(_lib.dtype_code(scores.dtype), B, T, V, _lib.ptr(scores), _lib.ptr(lens64), int(blank_id),
                                 _lib.ptr(best), _lib.ptr(tokens), _lib.ptr(ntok), _lib.ptr(frames),
                                 _lib.stream_of(scores)), "pafc_ctc_greedy")
    return (tokens, ntok, frames) if want_frames else (tokens, ntok)


_ACTS = {"none": 0, "silu": 1, "tanh": 2, "relu": 3, "glu": 4}


def gemm_bf16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: str = "none",
              alpha: float = 1.0, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """Hand-written bf16 GEMM with fused epilogue (include/pafc_encoder_ops.h: pafc_gemm_bf16).
    a: (M, K) or (Z, M, K); w: (N, K) or (Z, N, K) in nn.Linear layout; bias: (N) or (Z, N); residual / out:
    (M, N) or (Z, M, N), ``out`` may be ``residual``.  Rows may be strided views (unit stride in the last dim).
    act "glu": w / bias rows come in blocks of 128 = 64 value rows + the 64 gate rows of the same channels
    (``glu_interleave``); out is (M, N / 2)."""
    _lib.require_gpu(bias)
    for t in (a, w, residual, out):
        if t is not None and (not t.is_cuda or t.dtype != torch.bfloat16 or t.stride(-1) != 1):
            raise _lib.PafcError("gemm_bf16: bf16 GPU tensors with unit stride in the last dimension")
    L = _bind2()
    if not getattr(L, "_pafc_gemm2_bound", False):
        from ctypes import c_float, c_long
        P, I, G = c_void_p, c_int, c_long
        _lib._sig(L.pafc_gemm_bf16, I, G, I, I, I, P, G, G, P, G, G, P, G, P, G, G, P, G, G, c_float, I, P)
        L._pafc_gemm2_bound = True
    batched = a.dim() == 3
    Z = a.shape[0] if batched else 1
    M, K = a.shape[-2], a.shape[-1]
    N = w.shape[-2]
    if w.shape[-1] != K or (batched and (w.dim() != 3 or w.shape[0] != Z)) or (not batched and (a.dim() != 2 or w.dim() != 2)):
        raise _lib.PafcError("gemm_bf16: a (M, K) x w (N, K), or both with a leading batch")
    No = N // 2 if act == "glu" else N
    if out is None:
        out = torch.empty((Z, M, No) if batched else (M, No), dtype=a.dtype, device=a.device)
    for t in (residual, out):
        if t is not None and tuple(t.shape) != ((Z, M, No) if batched else (M, No)):
            raise _lib.PafcError("gemm_bf16: residual / out must be (M, N) per batch entry")
    if bias is not None and (bias.dtype != a.dtype or bias.shape[-1] != N):
        raise _lib.PafcError("gemm_bf16: bias must be (N) or (Z, N) bf16")
    sb = bias.stride(0) if (bias is not None and bias.dim() == 2) else 0
    bs = lambda t: t.stride(0) if batched else 0
    from .profiling import op_timer
    with op_timer("gemm_%dx%d%s" % (K, N, "x%d" % Z if batched else ""), sample=12, flops=2.0 * Z * M * N * K):
        rc = L.pafc_gemm_bf16(M, N, K, Z, _lib.ptr(a), a.stride(-2), bs(a), _lib.ptr(w), w.stride(-2), bs(w),
                              _lib.ptr(bias), sb, _lib.ptr(residual), residual.stride(-2) if residual is not None else 0,
                              bs(residual) if residual is not None else 0, _lib.ptr(out), out.stride(-2), bs(out),
                              float(alpha), _ACTS[act], _lib.stream_of(a))
    _lib.check(rc, "pafc_gemm_bf16")
    return out


SKINNY_MAX_ROWS = DISPATCH["skinny_max_rows"]      # (the table at the top of this file)


_chunk_step = threading.local()


@contextlib.contextmanager
def chunk_step():
    """Inside: the caller is a streaming chunk step (encoder.forward_chunk_carry) -- its few-row projections run on
    csrc/gemm_skinny.hip.  Outside (offline inputs that merely happen to be short) they keep the kernels the goldens of the
    offline path were recorded with."""
    prev = getattr(_chunk_step, "on", False)
    _chunk_step.on = True
    try:
        yield
    finally:
        _chunk_step.on = prev


def skinny_ok(rows: int, N: int, K: int, glu: bool = False) -> bool:
    """Shapes csrc/gemm_skinny.hip takes, asked from inside a streaming chunk step."""
    return (getattr(_chunk_step, "on", False) and 0 < rows <= SKINNY_MAX_ROWS and K % 32 == 0
            and N % (32 if glu else 16) == 0)


def decay_lora_skinny(x: torch.Tensor, d1n: torch.Tensor, d2n: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The decay LoRA for a handful of rows in one launch: bf16(bf16(tanh(x d1n^T)) d2n^T) [+ bias, rounded again] -- bit-identical
    to gemm_skinny(gemm_skinny(x, d1n, None, "tanh"), d2n, bias, round_first=True) (include/pafc_encoder_ops.h:
    pafc_decay_lora_skinny_bf16).  x (M, C) bf16 rows (unit column stride), d1n (64, C), d2n (C, 64), bias (C) -> (M, C)."""
    _lib.require_gpu(x, d1n, d2n, bias)
    M, C = x.shape
    H = d1n.shape[0]
    if x.dtype != torch.bfloat16 or x.stride(1) != 1 or d1n.shape != (H, C) or d2n.shape != (C, H) or \
            not (d1n.is_contiguous() and d2n.is_contiguous()) or (bias is not None and (bias.numel() != C or not bias.is_contiguous())):
        raise _lib.PafcError("decay_lora_skinny: x (M, C) bf16 rows, d1n (H, C), d2n (C, H), bias (C), contiguous weights")
    L = _bind2()
    if not getattr(L, "_pafc_dls_bound", False):
        _lib._sig(L.pafc_decay_lora_skinny_bf16, c_int, ctypes.c_long, c_int, c_int, c_void_p, ctypes.c_long, c_void_p, c_void_p, c_void_p,
                  c_void_p, ctypes.c_long, c_void_p)
        L._pafc_dls_bound = True
    out = torch.empty((M, C), dtype=x.dtype, device=x.device)
    rc = L.pafc_decay_lora_skinny_bf16(M, C, H, _lib.ptr(x), x.stride(0), _lib.ptr(d1n), _lib.ptr(d2n), _lib.ptr(bias), _lib.ptr(out), C,
                                       _lib.stream_of(x))
    _lib.check(rc, "pafc_decay_lora_skinny_bf16")
    return out


def gemm_skinny(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: str = "none", alpha: float = 1.0,
                residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, ln_stats: Optional[torch.Tensor] = None,
                ln_csum: Optional[torch.Tensor] = None, ln_eps: float = 1e-5, stats_out: Optional[torch.Tensor] = None,
                ln_self: bool = False, round_first: bool = False, mix_maa: Optional[torch.Tensor] = None,
                mix_prev: Optional[torch.Tensor] = None, mix_T: int = 0, norm_silu: Optional[tuple] = None):
    """The few-rows bf16 GEMM of the streaming chunk step (include/pafc_encoder_ops.h: pafc_gemm_skinny_bf16[_ex]), arguments as
    gemm_bf16 except act "glu": w is the module's own (2C, K) weight (value rows, then gate rows), out (M, C).
    ln_csum (N) fp32 with ln_stats (M, P, 2) fp32 or ln_self: the LayerNorm in front of the projection folded in (a = the
    UN-normalised rows, w / bias = the folded ones; statistics from a producer's partials or formed in the launch);
    stats_out (M, N_out / 16, 2) fp32 receives the partial row statistics of the result; round_first: bf16(alpha a w^T) + bias;
    mix_maa (K) [+ mix_prev (B, K)], mix_T: a = x of B sequences of mix_T rows, the operand is x + (x_prev - x) * maa;
    norm_silu = (gamma (K), beta (K), eps): the operand is silu(LayerNorm(a)), each rounded to bf16."""
    ng, nb, neps = norm_silu if norm_silu is not None else (None, None, 0.0)
    _lib.require_gpu(bias, ln_stats, ln_csum, stats_out, mix_maa, mix_prev, ng, nb)
    for t in (a, w, residual, out):
        if t is not None and (not t.is_cuda or t.dtype != torch.bfloat16 or t.stride(-1) != 1):
            raise _lib.PafcError("gemm_skinny: bf16 GPU tensors with unit stride in the last dimension")
    L = _bind2()
    if not getattr(L, "_pafc_skinny_bound", False):
        from ctypes import c_float, c_long
        P, I, G = c_void_p, c_int, c_long
        _lib._sig(L.pafc_gemm_skinny_bf16_ex, I, G, I, I, I, P, G, G, P, G, G, P, G, P, G, G, P, G, G, c_float, I, I, P, I, I, P,
                  c_float, P, P, P, I, P, P, c_float, P)
        L._pafc_skinny_bound = True
    batched = a.dim() == 3
    Z = a.shape[0] if batched else 1
    M, K = a.shape[-2], a.shape[-1]
    N = w.shape[-2]
    if w.shape[-1] != K or (batched and (w.dim() != 3 or w.shape[0] != Z)) or (not batched and (a.dim() != 2 or w.dim() != 2)):
        raise _lib.PafcError("gemm_skinny: a (M, K) x w (N, K), or both with a leading batch")
    No = N // 2 if act == "glu" else N
    shape = (Z, M, No) if batched else (M, No)
    if out is None:
        out = torch.empty(shape, dtype=a.dtype, device=a.device)
    for t in (residual, out):
        if t is not None and tuple(t.shape) != shape:
            raise _lib.PafcError("gemm_skinny: residual / out must be (M, N) per batch entry")
    if bias is not None and (bias.dtype != a.dtype or bias.shape[-1] != N):
        raise _lib.PafcError("gemm_skinny: bias must be (N) or (Z, N) bf16")
    parts = 0
    if ln_stats is not None or ln_self:
        if ln_csum is None or batched or ln_csum.dtype != torch.float32 or ln_csum.numel() != N or (ln_stats is not None and ln_self):
            raise _lib.PafcError("gemm_skinny: a folded LayerNorm wants ln_csum (N) fp32 and ln_stats or ln_self, unbatched")
        if ln_stats is not None:
            if ln_stats.dtype != torch.float32 or ln_stats.dim() != 3 or ln_stats.shape[0] != M or ln_stats.shape[2] != 2:
                raise _lib.PafcError("gemm_skinny: ln_stats (M, P, 2) fp32")
            parts = ln_stats.shape[1]
    elif ln_csum is not None:
        raise _lib.PafcError("gemm_skinny: ln_csum without ln_stats / ln_self")
    if stats_out is not None and (stats_out.dtype != torch.float32 or stats_out.numel() != Z * M * (No // 16) * 2):
        raise _lib.PafcError("gemm_skinny: stats_out must be fp32 (M, N_out / 16, 2) per batch entry")
    if mix_maa is not None:
        if (batched or mix_T <= 0 or M % mix_T or mix_maa.dtype != a.dtype or mix_maa.numel() != K or not a.is_contiguous()
                or (mix_prev is not None and (mix_prev.dtype != a.dtype or mix_prev.numel() != (M // mix_T) * K))):
            raise _lib.PafcError("gemm_skinny: mix wants contiguous a = (B * T, K), maa (K) and prev (B, K) of a's dtype")
    elif mix_prev is not None:
        raise _lib.PafcError("gemm_skinny: mix_prev without mix_maa")
    if ng is not None and (batched or ng.dtype != a.dtype or nb.dtype != a.dtype or ng.numel() != K or nb.numel() != K):
        raise _lib.PafcError("gemm_skinny: norm_silu wants gamma, beta (K) of a's dtype, unbatched")
    sb = bias.stride(0) if (bias is not None and bias.dim() == 2) else 0
    bs = lambda t: t.stride(0) if batched else 0
    rc = L.pafc_gemm_skinny_bf16_ex(M, N, K, Z, _lib.ptr(a), a.stride(-2), bs(a), _lib.ptr(w), w.stride(-2), bs(w), _lib.ptr(bias),
                                    sb, _lib.ptr(residual), residual.stride(-2) if residual is not None else 0,
                                    bs(residual) if residual is not None else 0, _lib.ptr(out), out.stride(-2), bs(out),
                                    float(alpha), _ACTS[act], int(round_first), _lib.ptr(ln_stats), parts, int(ln_self),
                                    _lib.ptr(ln_csum), float(ln_eps), _lib.ptr(stats_out), _lib.ptr(mix_maa), _lib.ptr(mix_prev),
                                    int(mix_T), _lib.ptr(ng), _lib.ptr(nb), float(neps), _lib.stream_of(a))
    _lib.check(rc, "pafc_gemm_skinny_bf16")
    return out


def gemm_bf16_ph(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: str = "none",
                 alpha: float = 1.0, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 tile_n: int = 256, tile_m: int = 256):
    """The phase-pipelined tile_m x tile_n kernel (csrc/gemm_ph.hip) by itself -- same arguments as gemm_bf16, which picks
    it for the long-form shapes; act "glu" wants glu_interleave(w, tile_n // 8)."""
    _lib.require_gpu(bias)
    for t in (a, w, residual, out):
        if t is not None and (not t.is_cuda or t.dtype != torch.bfloat16 or t.stride(-1) != 1):
            raise _lib.PafcError("gemm_bf16_ph: bf16 GPU tensors with unit stride in the last dimension")
    L = _bind2()
    if not getattr(L, "_pafc_gemm_ph_bound", False):
        from ctypes import c_float, c_long
        P, I, G = c_void_p, c_int, c_long
        _lib._sig(L.pafc_gemm_bf16_ph, I, G, I, I, I, P, G, G, P, G, G, P, G, P, G, G, P, G, G, c_float, I, I, I, P)
        L._pafc_gemm_ph_bound = True
    batched = a.dim() == 3
    Z = a.shape[0] if batched else 1
    M, K = a.shape[-2], a.shape[-1]
    N = w.shape[-2]
    if w.shape[-1] != K or (batched and (w.dim() != 3 or w.shape[0] != Z)) or (not batched and (a.dim() != 2 or w.dim() != 2)):
        raise _lib.PafcError("gemm_bf16_ph: a (M, K) x w (N, K), or both with a leading batch")
    No = N // 2 if act == "glu" else N
    if out is None:
        out = torch.empty((Z, M, No) if batched else (M, No), dtype=a.dtype, device=a.device)
    sb = bias.stride(0) if (bias is not None and bias.dim() == 2) else 0
    bs = lambda t: t.stride(0) if batched else 0
    rc = L.pafc_gemm_bf16_ph(M, N, K, Z, _lib.ptr(a), a.stride(-2), bs(a), _lib.ptr(w), w.stride(-2), bs(w),
                             _lib.ptr(bias), sb, _lib.ptr(residual), residual.stride(-2) if residual is not None else 0,
                             bs(residual) if residual is not None else 0, _lib.ptr(out), out.stride(-2), bs(out),
                             float(alpha), _ACTS[act], int(tile_n), int(tile_m), _lib.stream_of(a))
    _lib.check(rc, "pafc_gemm_bf16_ph")
    return out


def gemm_glu_half(M: int, N: int, K: int, batch: int = 1) -> int:
    """Row-block half size (64 or 32) pafc_gemm_bf16 wants for act "glu" on this problem (it depends on the kernel chosen)."""
    L = _bind2()
    if not getattr(L, "_pafc_gluhalf_bound", False):
        from ctypes import c_long
        _lib._sig(L.pafc_gemm_bf16_glu_half, c_int, c_long, c_int, c_int, c_int)
        L._pafc_gluhalf_bound = True
    return int(L.pafc_gemm_bf16_glu_half(M, N, K, batch))


def log_softmax_rows(x: torch.Tensor, inplace: bool = False) -> torch.Tensor:
    """log_softmax over the last dimension in one pass over HBM (include/pafc_search.h: pafc_log_softmax_rows)."""
    _lib.require_gpu(x)
    L = _bind()
    if not getattr(L, "_pafc_lsm_bound", False):
        from ctypes import c_long
        _lib._sig(L.pafc_log_softmax_rows, c_int, c_int, c_long, c_int, c_void_p, c_void_p, c_void_p)
        L._pafc_lsm_bound = True
    V = x.shape[-1]
    out = x if inplace else torch.empty_like(x)
    _lib.check(L.pafc_log_softmax_rows(_lib.dtype_code(x.dtype), x.numel() // V, V, _lib.ptr(x), _lib.ptr(out),
                                       _lib.stream_of(x)), "pafc_log_softmax_rows")
    return out


def glu_interleave(t: torch.Tensor, half: int = 64) -> torch.Tensor:
    """(2C, ...) parameter of a Linear followed by F.glu (values = rows [0, C), gates = rows [C, 2C)) -> the row order
    the GEMM kernels want for act "glu": blocks of `half` value rows followed by the `half` gate rows of the same channels
    (half = 64: pafc_gemm_bf16's 128 x 128 kernel; 32 / 16: the phase-pipelined kernel at tile_n 256 / 128, where a wave's
    columns are one block)."""
    C = t.shape[0] // 2
    if C % half:
        raise _lib.PafcError(f"glu_interleave: channels must be a multiple of {half}")
    v = t[:C].reshape(C // half, half, *t.shape[1:])
    g = t[C:].reshape(C // half, half, *t.shape[1:])
    return torch.cat([v, g], dim=1).reshape(t.shape).contiguous()


def conv3x3s2_c1_nhwc(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = True):
    """First subsampling convolution (1 input channel): x (B, T, F) bf16, weight (C, 1, 3, 3) -> (B, T1, F1, C)."""
    _lib.require_gpu(x, weight, bias)
    if x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16 or x.dim() != 3 or weight.shape[1:] != (1, 3, 3):
        raise _lib.PafcError("conv3x3s2_c1_nhwc: bf16 x (B, T, F) and weight (C, 1, 3, 3)")
    L = _bind()
    if not getattr(L, "_pafc_c1_bound", False):
        _lib._sig(L.pafc_conv3x3s2_c1_nhwc_bf16, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                  c_int, c_void_p)
        L._pafc_c1_bound = True
    B, T, Fd = x.shape
    C = weight.shape[0]
    out = torch.empty((B, (T - 3) // 2 + 1, (Fd - 3) // 2 + 1, C), dtype=x.dtype, device=x.device)
    _lib.check(L.pafc_conv3x3s2_c1_nhwc_bf16(B, T, Fd, C, _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out),
                                             int(relu), _lib.stream_of(x)), "pafc_conv3x3s2_c1_nhwc_bf16")
    return out


class _Conv1Train(torch.autograd.Function):
    """relu(Conv2d(1, C, 3, 2)(x)) in NHWC for the GPU training step (subsampling.py:201-226, conv[0:2]): forward = the
    inference kernel, backward = pafc_conv3x3s2_c1_wgrad_bf16 (the library runs this layer as im2col + one small GEMM per
    batch entry: 6 ms forward + backward at the c4 shape).  x: (B, T, F) bf16 features (no gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        wb = weight.to(torch.bfloat16)
        a = conv3x3s2_c1_nhwc(x, wb, None if bias is None else bias.to(torch.bfloat16), relu=True)
        ctx.save_for_backward(x, a)
        ctx.w_dtype, ctx.b_dtype, ctx.w_shape = weight.dtype, (None if bias is None else bias.dtype), weight.shape
        return a

    @staticmethod
    def backward(ctx, da):
        x, a = ctx.saved_tensors
        B, T, Fd = x.shape
        C = a.shape[-1]
        L = _bind()
        if not getattr(L, "_pafc_c1w_bound", False):
            from ctypes import c_size_t
            L.pafc_conv3x3s2_c1_wgrad_workspace_bytes.restype = c_size_t
            L.pafc_conv3x3s2_c1_wgrad_workspace_bytes.argtypes = [c_int, c_int, c_int]
            _lib._sig(L.pafc_conv3x3s2_c1_wgrad_bf16, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                      c_void_p, c_void_p, c_size_t, c_void_p)
            L._pafc_c1w_bound = True
        nbytes = L.pafc_conv3x3s2_c1_wgrad_workspace_bytes(B, T, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        out = torch.empty(10, C, dtype=torch.float32, device=x.device)
        da = da.contiguous()
        rc = L.pafc_conv3x3s2_c1_wgrad_bf16(B, T, Fd, C, _lib.ptr(x), _lib.ptr(a), _lib.ptr(da), _lib.ptr(out), _lib.ptr(ws),
                                            nbytes, _lib.stream_of(x))
        _lib.check(rc, "pafc_conv3x3s2_c1_wgrad_bf16")
        dw = out[:9].t().reshape(ctx.w_shape).to(ctx.w_dtype)
        db = out[9].to(ctx.b_dtype) if ctx.b_dtype is not None else None
        return None, dw, db


class _Conv2Train(torch.autograd.Function):
    """relu(Conv2d(C, C, 3, 2)(a)) NHWC -> NHWC for the GPU training step (subsampling.py conv[2:4]): forward = the
    hand-written implicit GEMM (1.0 PFLOP/s; the library's forward runs at a third of that), backward = the library's
    convolution backward on channels_last views of the same memory."""

    @staticmethod
    def forward(ctx, a, weight, bias):
        Co, Ci = weight.shape[0], weight.shape[1]
        wb = weight.to(torch.bfloat16)
        taps = wb.permute(2, 3, 0, 1).reshape(9, Co, Ci).contiguous()
        y = conv3x3s2_nhwc(a, taps, None if bias is None else bias.to(torch.bfloat16), relu=True)
        ctx.save_for_backward(a, y, wb)
        ctx.w_dtype, ctx.b_dtype = weight.dtype, (None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, y, wb = ctx.saved_tensors
        # relu's backward as the framework's own one-pass kernel (`dy * (y > 0)` was a compare, a 150 MB bool tensor and a
        # mixed-dtype multiply: 0.4 ms); NCHW view of NHWC memory (channels_last)
        g = torch.ops.aten.threshold_backward(dy.contiguous(), y, 0.0).permute(0, 3, 1, 2)
        gi, gw, gb = torch.ops.aten.convolution_backward(g, a.permute(0, 3, 1, 2), wb, [wb.shape[0]], [2, 2], [0, 0], [1, 1],
                                                         False, [0, 0], 1, [ctx.needs_input_grad[0], True,
                                                                            ctx.b_dtype is not None])
        da = gi.permute(0, 2, 3, 1).contiguous() if gi is not None else None
        return da, gw.to(ctx.w_dtype), (gb.to(ctx.b_dtype) if gb is not None else None)


def conv_sub_train(x: torch.Tensor, c1_weight, c1_bias, c2_weight, c2_bias) -> torch.Tensor:
    """The two subsampling convolutions + ReLUs of the bf16 training step, NHWC: x (B, T, F) -> (B, T', F', C) bf16."""
    a = _Conv1Train.apply(x.to(torch.bfloat16).contiguous(), c1_weight, c1_bias)
    return _Conv2Train.apply(a, c2_weight, c2_bias)


def ctc_prefix_beam(top_logp: torch.Tensor, top_idx: torch.Tensor, lens: Optional[torch.Tensor], beam: int,
                    blank_id: int = 0):
    """GPU-resident CTC prefix beam search (include/pafc_search.h).  top_logp (B, T, K) float32 / top_idx (B, T, K) =
    torch.topk of the CTC log-probs.  Returns (tokens (B, beam, T) int32, lengths (B, beam) int32 [-1 = unused],
    scores (B, beam) float64), best first."""
    _lib.require_gpu(top_logp, top_idx, lens)
    from ctypes import c_size_t
    L = _bind()
    if not getattr(L, "_pafc_beam_bound", False):
        P, I = c_void_p, c_int
        _lib._sig(L.pafc_ctc_prefix_beam_workspace_bytes, c_size_t, I, I, I)
        _lib._sig(L.pafc_ctc_prefix_beam_search, I, I, I, I, P, P, P, I, I, P, P, P, P, c_size_t, P)
        L._pafc_beam_bound = True
    if top_logp.dtype != torch.float32 or top_logp.shape != top_idx.shape or top_logp.dim() != 3:
        raise _lib.PafcError("ctc_prefix_beam: top_logp float32 (B, T, K) and top_idx of the same shape")
    B, T, K = top_logp.shape
    idx32 = top_idx.to(torch.int32).contiguous()
    lens64 = None if lens is None else lens.to(torch.int64).contiguous()
    dev = top_logp.device
    nws = L.pafc_ctc_prefix_beam_workspace_bytes(B, T, beam)
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
    tokens = torch.empty(B, beam, T, dtype=torch.int32, device=dev)
    lengths = torch.empty(B, beam, dtype=torch.int32, device=dev)
    scores = torch.empty(B, beam, dtype=torch.float64, device=dev)
    _lib.check(L.pafc_ctc_prefix_beam_search(B, T, K, _lib.ptr(top_logp), _lib.ptr(idx32), _lib.ptr(lens64), int(beam),
                                             int(blank_id), _lib.ptr(tokens), _lib.ptr(lengths), _lib.ptr(scores),
                                             _lib.ptr(ws), nws, _lib.stream_of(top_logp)), "pafc_ctc_prefix_beam_search")
    return tokens, lengths, scores



class RnntBeamState:
    """Device-side beams of the CTC-fused RNN-T prefix beam search (include/pafc_search.h: pafc_rnnt_beam_*)."""

    def __init__(self, B: int, T: int, beam: int, blank: int, device):
        from ctypes import c_size_t
        L = _bind()
        if not getattr(L, "_pafc_rnnt_bound", False):
            P, I = c_void_p, c_int
            _lib._sig(L.pafc_rnnt_beam_workspace_bytes, c_size_t, I, I, I)
            _lib._sig(L.pafc_rnnt_beam_init, I, I, I, I, I, P, c_size_t, P, P, P)
            _lib._sig(L.pafc_rnnt_beam_step, I, I, I, I, I, I, P, P, P, P, P, c_size_t, P, P, P)
            _lib._sig(L.pafc_rnnt_beam_finish, I, I, I, I, P, c_size_t, P, P, P, P)
            L._pafc_rnnt_bound = True
        self.L, self.B, self.T, self.beam, self.blank = L, B, T, beam, blank
        self.nws = L.pafc_rnnt_beam_workspace_bytes(B, T, beam)
        if self.nws == 0:
            raise _lib.PafcError("rnnt beam search: B, T, beam must be positive")
        self.ws = torch.empty(self.nws, dtype=torch.uint8, device=device)
        self.next_idx = torch.empty(B * beam, dtype=torch.int64, device=device)
        self.last_tok = torch.empty(B * beam, dtype=torch.int64, device=device)
        self.stream = _lib.stream_of(self.ws)
        _lib.check(L.pafc_rnnt_beam_init(B, T, beam, blank, _lib.ptr(self.ws), self.nws, _lib.ptr(self.next_idx),
                                         _lib.ptr(self.last_tok), self.stream), "pafc_rnnt_beam_init")

    def step(self, t: int, lens64: Optional[torch.Tensor], top_val: torch.Tensor, top_idx: torch.Tensor,
             t_dev: Optional[torch.Tensor] = None):
        """t_dev: device int64 scalar holding the frame index (graph replay); otherwise ``t``."""
        _lib.require_gpu(top_val, top_idx, lens64, t_dev)
        if top_val.dtype != torch.float32 or top_idx.dtype != torch.int64 or top_val.numel() != self.B * self.beam * self.beam:
            raise _lib.PafcError("rnnt beam step: top_val float32 / top_idx int64 of (B, beam, beam)")
        _lib.check(self.L.pafc_rnnt_beam_step(self.B, self.T, self.beam, self.blank, int(t), _lib.ptr(t_dev),
                                              _lib.ptr(lens64), _lib.ptr(top_val), _lib.ptr(top_idx), _lib.ptr(self.ws),
                                              self.nws, _lib.ptr(self.next_idx), _lib.ptr(self.last_tok),
                                              _lib.stream_of(top_val)), "pafc_rnnt_beam_step")

    def finish(self):
        dev = self.ws.device
        tokens = torch.empty(self.B, self.beam, self.T, dtype=torch.int32, device=dev)
        lengths = torch.empty(self.B, self.beam, dtype=torch.int32, device=dev)
        scores = torch.empty(self.B, self.beam, dtype=torch.float64, device=dev)
        _lib.check(self.L.pafc_rnnt_beam_finish(self.B, self.T, self.beam, _lib.ptr(self.ws), self.nws, _lib.ptr(tokens),
                                                _lib.ptr(lengths), _lib.ptr(scores), self.stream), "pafc_rnnt_beam_finish")
        return tokens, lengths, scores


def split_bf16(t: torch.Tensor):
    """fp32 tensor -> (hi, lo) bf16 planes with t ~= hi + lo (16 significant bits)."""
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def conv_sub_f32split(x: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], w2_hi: torch.Tensor,
                      w2_lo: torch.Tensor, b2: Optional[torch.Tensor]) -> torch.Tensor:
    """Both subsampling convolutions (+ ReLU) for fp32 activations on the bf16 matrix cores with split operands
    (include/pafc_encoder_ops.h: pafc_conv3x3s2_*_f32split).  x (B, T, F) fp32; w1 (C, 1, 3, 3) fp32; w2_hi / w2_lo
    (9, C, C) bf16 = split_bf16 of the second Conv2d weight in (tap, co, ci) order -> (B, T', F', C) fp32."""
    _lib.require_gpu(x, w1, b1, w2_hi, w2_lo, b2)
    if x.dtype != torch.float32 or w1.dtype != torch.float32 or w2_hi.dtype != torch.bfloat16 or x.dim() != 3:
        raise _lib.PafcError("conv_sub_f32split: fp32 x (B, T, F) / w1, bf16 split planes for w2")
    L = _bind()
    if not getattr(L, "_pafc_split_bound", False):
        P, I = c_void_p, c_int
        _lib._sig(L.pafc_conv3x3s2_c1_nhwc_f32split, I, I, I, I, I, P, P, P, P, P, I, P)
        _lib._sig(L.pafc_conv3x3s2_nhwc_f32split, I, I, I, I, I, I, P, P, P, P, P, P, I, P)
        L._pafc_split_bound = True
    B, T, Fd = x.shape
    C = w1.shape[0]
    T1, F1 = (T - 3) // 2 + 1, (Fd - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    hi = torch.empty((B, T1, F1, C), dtype=torch.bfloat16, device=x.device)
    lo = torch.empty_like(hi)
    st = _lib.stream_of(x)
    _lib.check(L.pafc_conv3x3s2_c1_nhwc_f32split(B, T, Fd, C, _lib.ptr(x), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(hi),
                                                 _lib.ptr(lo), 1, st), "pafc_conv3x3s2_c1_nhwc_f32split")
    out = torch.empty((B, T2, F2, C), dtype=torch.float32, device=x.device)
    _lib.check(L.pafc_conv3x3s2_nhwc_f32split(B, T1, F1, C, C, _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(w2_hi), _lib.ptr(w2_lo),
                                              _lib.ptr(b2), _lib.ptr(out), 1, st), "pafc_conv3x3s2_nhwc_f32split")
    return out


def conv_sub_f32split_planes(x: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], w2_3: torch.Tensor,
                             b2: Optional[torch.Tensor]) -> torch.Tensor:
    """Both subsampling convolutions (+ ReLU) of an fp32 model at long-form sizes: conv1 in fp32 arithmetic writing planes
    [hi C | lo C] per pixel, conv2 as the split-operand implicit GEMM of csrc/gemm_ph.hip (include/pafc_encoder_ops.h:
    pafc_conv3x3s2_nhwc_split_ph).  x (B, T, F) fp32; w1 (C, 1, 3, 3) fp32; w2_3 (9, C, 3 C) = split_planes(w2 taps, triple)
    -> (B, T', F', 2 C) bf16 planes [hi C | lo C] of the fp32 result, the A operand of gemm_ph_ex(a_split, a_plane_block=C)."""
    _lib.require_gpu(x, w1, b1, w2_3, b2)
    B, T, Fd = x.shape
    C = w1.shape[0]
    if x.dtype != torch.float32 or w1.dtype != torch.float32 or w2_3.dtype != torch.bfloat16 or tuple(w2_3.shape) != (9, C, 3 * C):
        raise _lib.PafcError("conv_sub_f32split_planes: fp32 x (B, T, F) / w1, w2 as (9, C, 3C) bf16 planes")
    L = _bind()
    if not getattr(L, "_pafc_splitph_bound", False):
        from ctypes import c_long
        P, I = c_void_p, c_int
        _lib._sig(L.pafc_conv3x3s2_c1_nhwc_f32split_ps, I, I, I, I, I, P, P, P, P, P, c_long, I, P)
        _lib._sig(L.pafc_conv3x3s2_nhwc_split_ph, I, I, I, I, I, I, P, P, P, P, I, I, P)
        L._pafc_splitph_bound = True
    T1, F1 = (T - 3) // 2 + 1, (Fd - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    y1 = torch.empty((B, T1, F1, 2 * C), dtype=torch.bfloat16, device=x.device)
    st = _lib.stream_of(x)
    lo = c_void_p(y1.data_ptr() + 2 * C)
    _lib.check(L.pafc_conv3x3s2_c1_nhwc_f32split_ps(B, T, Fd, C, _lib.ptr(x), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(y1), lo, 2 * C, 1, st),
               "pafc_conv3x3s2_c1_nhwc_f32split_ps")
    out = torch.empty((B, T2, F2, 2 * C), dtype=torch.bfloat16, device=x.device)
    from .profiling import op_timer
    with op_timer("conv3x3s2_split", flops=2.0 * B * T2 * F2 * C * 9 * C * 3):
        rc = L.pafc_conv3x3s2_nhwc_split_ph(B, T1, F1, C, C, _lib.ptr(y1), _lib.ptr(w2_3), _lib.ptr(b2), _lib.ptr(out), 1,
                                            _ph_tile_m(B * T2 * F2, C, min_tm=128), st)
    _lib.check(rc, "pafc_conv3x3s2_nhwc_split_ph")
    return out


def _bind_mamba():
    L = _bind()
    if not getattr(L, "_pafc_mamba_bound", False):
        from ctypes import c_float, c_long
        P, I, G = c_void_p, c_int, c_long
        _lib._sig(L.pafc_dwconv1d_cl_ex, I, I, I, I, I, I, I, I, P, G, P, P, P, I, P, P)
        _lib._sig(L.pafc_mamba2_prep, I, I, I, I, I, P, P, G, P, P, P, P, P, P, P, P, P)
        _lib._sig(L.pafc_mamba2_finish, I, I, I, I, I, P, P, P, P, G, P, G, P, P, P, c_float, I, P, P)
        from ctypes import c_size_t
        _lib._sig(L.pafc_mamba2_scan_workspace_bytes, c_size_t, I, I, I, I)
        _lib._sig(L.pafc_mamba2_scan, I, I, I, I, P, G, P, P, P, I, P, c_size_t, P)
        _lib._sig(L.pafc_mamba2_scan_dir, I, I, I, I, P, G, P, P, P, I, I, P, c_size_t, P)
        _lib._sig(L.pafc_mamba2_scan_skip_bf16, I, I, I, I, P, G, P, P, P, P, I, I, P, c_size_t, P)
        _lib._sig(L.pafc_mamba2_gate_norm, I, I, G, I, P, P, G, P, c_float, P, P)
        L._pafc_mamba_bound = True
    return L


def causal_conv_silu_cl(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], reverse: bool = False
                        ) -> torch.Tensor:
    """SiLU(causal depthwise conv1d) in channels-last layout: x (B, L, C) -- may be a column slice of a wider tensor --
    weight (C, 1, K) -> (B, L, C) contiguous (include/pafc_encoder_ops.h: pafc_dwconv1d_cl_ex, act 2).  reverse: the
    convolution that is causal in reversed time (what flip -> conv -> flip computes), i.e. the taps reversed and looking
    right: same kernel, left_pad 0."""
    _lib.require_gpu(weight, bias)
    if not x.is_cuda or x.dim() != 3 or x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
        raise _lib.PafcError("causal_conv_silu_cl: (B, L, C) GPU tensor, unit stride in C, batch stride L * row stride")
    B, Lq, C = x.shape
    K = weight.shape[-1]
    y = torch.empty((B, Lq, C), dtype=x.dtype, device=x.device)
    if reverse:
        weight = weight.flip(-1).contiguous()
    rc = _bind_mamba().pafc_dwconv1d_cl_ex(_lib.dtype_code(x.dtype), B, Lq, C, K, 0 if reverse else K - 1, Lq, _lib.ptr(x),
                                           x.stride(1), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(y), 2, None,
                                           _lib.stream_of(x))
    _lib.check(rc, "pafc_dwconv1d_cl_ex")
    return y


def mamba2_prep(xbc: torch.Tensor, dt_raw: torch.Tensor, dt_bias: torch.Tensor, A_log: torch.Tensor, d_inner: int):
    """xbc (B, L, d_inner + 256) contiguous, dt_raw (B, L, H) slice -> [r0, r1, k0, k1, v, w] fp32 (B, L, d_inner)."""
    _lib.require_gpu(xbc, dt_bias, A_log)
    B, Lq, _ = xbc.shape
    planes = [torch.empty((B, Lq, d_inner), dtype=torch.float32, device=xbc.device) for _ in range(6)]
    rc = _bind_mamba().pafc_mamba2_prep(_lib.dtype_code(xbc.dtype), B, Lq, d_inner, _lib.ptr(xbc), _lib.ptr(dt_raw),
                                        dt_raw.stride(1), _lib.ptr(dt_bias), _lib.ptr(A_log), *[_lib.ptr(t) for t in planes],
                                        _lib.stream_of(xbc))
    _lib.check(rc, "pafc_mamba2_prep")
    return planes


def mamba2_scan(xbc: torch.Tensor, dt: torch.Tensor, log_a: torch.Tensor, H: int, reverse: bool = False,
                D: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Mamba-2 selective scan on the dedicated SSD kernel: xbc (B, L, H*64 + 256) bf16 contiguous, dt / log_a (B, L, H)
    fp32 -> y (B, L, H*64) fp32 (include/pafc_encoder_ops.h: pafc_mamba2_scan_dir); with D (H) fp32 the scan returns
    bf16(y + D x) as mamba_ssm's does (pafc_mamba2_scan_skip_bf16)."""
    _lib.require_gpu(xbc, dt, log_a, D)
    if xbc.dtype != torch.bfloat16 or dt.dtype != torch.float32 or log_a.dtype != torch.float32:
        raise _lib.PafcError("mamba2_scan: bf16 xbc, fp32 dt / log_a")
    B, Lq, ldx = xbc.shape
    Lb = _bind_mamba()
    nws = Lb.pafc_mamba2_scan_workspace_bytes(B, Lq, H, 0)
    ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=xbc.device)
    if D is not None:
        if D.dtype != torch.float32 or D.shape != (H,):
            raise _lib.PafcError("mamba2_scan: D must be float32 (H)")
        y = torch.empty((B, Lq, H * 64), dtype=torch.bfloat16, device=xbc.device)
        rc = Lb.pafc_mamba2_scan_skip_bf16(B, Lq, H, _lib.ptr(xbc), ldx, _lib.ptr(dt), _lib.ptr(log_a), _lib.ptr(D),
                                           _lib.ptr(y), int(reverse), 0, _lib.ptr(ws) if nws else None, nws,
                                           _lib.stream_of(xbc))
        _lib.check(rc, "pafc_mamba2_scan_skip_bf16")
        return y
    y = torch.empty((B, Lq, H * 64), dtype=torch.float32, device=xbc.device)
    rc = Lb.pafc_mamba2_scan_dir(B, Lq, H, _lib.ptr(xbc), ldx, _lib.ptr(dt), _lib.ptr(log_a), _lib.ptr(y), int(reverse), 0,
                                 _lib.ptr(ws) if nws else None, nws, _lib.stream_of(xbc))
    _lib.check(rc, "pafc_mamba2_scan_dir")
    return y


def mamba2_gate_norm(y: torch.Tensor, z: torch.Tensor, norm_weight: torch.Tensor, eps: float) -> torch.Tensor:
    """RMSNorm(y * silu(z)) * norm_weight over the last axis; z may be a column slice of a wider tensor."""
    _lib.require_gpu(y, norm_weight)
    d = y.shape[-1]
    rows = y.numel() // d
    if z.dtype != y.dtype or norm_weight.dtype != y.dtype or z.stride(-1) != 1 or z.shape != y.shape or not z.is_cuda:
        raise _lib.PafcError("mamba2_gate_norm: y, z, norm_weight in one dtype; z shaped like y with unit channel stride")
    out = torch.empty_like(y)
    rc = _bind_mamba().pafc_mamba2_gate_norm(_lib.dtype_code(y.dtype), rows, d, _lib.ptr(y), _lib.ptr(z), z.stride(-2),
                                             _lib.ptr(norm_weight), float(eps), _lib.ptr(out), _lib.stream_of(y))
    _lib.check(rc, "pafc_mamba2_gate_norm")
    return out


def mamba2_finish(y0, y1, xbc, dt_raw, z, dt_bias, D, norm_weight, eps: float, d_inner: int, diag: bool = True
                  ) -> torch.Tensor:
    _lib.require_gpu(y0, y1, xbc, dt_bias, D, norm_weight)
    B, Lq, _ = xbc.shape
    out = torch.empty((B, Lq, d_inner), dtype=xbc.dtype, device=xbc.device)
    rc = _bind_mamba().pafc_mamba2_finish(_lib.dtype_code(xbc.dtype), B, Lq, d_inner, _lib.ptr(y0), _lib.ptr(y1), _lib.ptr(xbc),
                                          _lib.ptr(dt_raw), dt_raw.stride(1), _lib.ptr(z), z.stride(1), _lib.ptr(dt_bias),
                                          _lib.ptr(D), _lib.ptr(norm_weight), float(eps), int(diag), _lib.ptr(out),
                                          _lib.stream_of(xbc))
    _lib.check(rc, "pafc_mamba2_finish")
    return out
