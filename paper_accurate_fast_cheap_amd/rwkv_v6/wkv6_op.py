"""WKV-6 op: torch tensors -> C ABI (include/pafc_wkv6.h) -> gfx950 kernels.

Host-side counterpart of the reference's WKV_6 / WKV_6_FP32 autograd Functions
(wenet/rwkv_v6/src/model.py:108-214) and of its `torch.ops.wkv6.*` bindings
(cuda/wkv6_op.cpp:34-41): same argument meaning (r, k, v, w: (B, T, C)
contiguous; u: (H, N); y allocated here), same dtype rules (all operands one
dtype, bfloat16 or float32), errors raised instead of asserted.
"""
from typing import Optional, Tuple

import torch

from .. import _lib
from ..profiling import op_timer

HEAD_SIZE = 64


def _shape(r, u):
    B, T, C = r.shape
    H = u.shape[0]
    if u.shape != (H, C // H) or C % H:
        raise _lib.PafcError(f"u must be (H, C // H); got {tuple(u.shape)} for C={C}")
    return B, T, C, H


def _same(*ts):
    dt = ts[0].dtype
    for t in ts:
        if t.dtype != dt:
            raise _lib.PafcError("r, k, v, w, u must share one dtype (model.py:116-120)")
    return _lib.dtype_code(dt)


def _workspace(B, T, C, H, ndir, chunk_len, device):
    L = _lib.lib()
    nbytes = L.pafc_wkv6_fwd_workspace_bytes(B, T, C, H, ndir, chunk_len)
    if nbytes == 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def wkv6_forward(r, k, v, w, u, *, reverse: bool = False, s_in: Optional[torch.Tensor] = None,
                 want_state: bool = False, chunk_len: int = 0, s_out: Optional[torch.Tensor] = None):
    """y = WKV6(r, k, v, w, u); optionally from an initial state and returning the final one.

    chunk_len: 0 = library heuristic, T (or more) = serial schedule, else steps per chunk.
    s_out: where the final state goes (float32 (B, H, 64, 64); allocated here when want_state and None).  It may BE s_in
    -- a streaming step that updates its carried state in place -- when the sequence is walked as one chunk (each wave then
    reads its head's state before the first step and writes it after the last); refused otherwise.
    """
    _lib.require_gpu(r, k, v, w, u, s_in, s_out)
    B, T, C, H = _shape(r, u)
    code = _same(r, k, v, w, u)
    y = torch.empty_like(r)
    if s_out is not None:
        want_state = True
        if s_out.dtype != torch.float32 or s_out.shape != (B, H, HEAD_SIZE, HEAD_SIZE):
            raise _lib.PafcError("s_out must be float32 (B, H, 64, 64)")
        if s_in is not None and s_out.data_ptr() == s_in.data_ptr():
            picked = chunk_len if chunk_len > 0 else _lib.lib().pafc_wkv6_pick_chunk_len(B, T, C, H, 1)
            if picked < T:
                raise _lib.PafcError("wkv6_forward: s_out may alias s_in only when the sequence is one chunk")
    elif want_state:
        s_out = torch.empty(B, H, HEAD_SIZE, HEAD_SIZE, dtype=torch.float32, device=r.device)
    if s_in is not None and (s_in.dtype != torch.float32 or s_in.shape != (B, H, HEAD_SIZE, HEAD_SIZE)):
        raise _lib.PafcError("s_in must be float32 (B, H, 64, 64)")
    ws, nbytes = _workspace(B, T, C, H, 1, chunk_len, r.device)
    with op_timer("wkv6_fwd", B=B, T=T, C=C, elem_bytes=r.element_size(), ndir=1):
        rc = _lib.lib().pafc_wkv6_forward_state(code, B, T, C, H, _lib.ptr(r), _lib.ptr(k), _lib.ptr(v), _lib.ptr(w),
                                                _lib.ptr(u), _lib.ptr(y), _lib.ptr(s_in), _lib.ptr(s_out),
                                                int(reverse), chunk_len, _lib.ptr(ws), nbytes, _lib.stream_of(r))
    _lib.check(rc, "pafc_wkv6_forward_state")
    return (y, s_out) if want_state else y


def wkv6_forward_bidir(fwd: Tuple[torch.Tensor, ...], bwd: Tuple[torch.Tensor, ...], *, chunk_len: int = 0,
                       w_bias: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """Both directions of the bidirectional wrapper in one launch.

    fwd / bwd = (r, k, v, w, u) produced by the left-to-right / right-to-left parameter sets on the SAME,
    un-flipped time axis; returns (y_fwd, y_bwd), also un-flipped.
    """
    rf, kf, vf, wf, uf = fwd
    rb, kb, vb, wb, ub = bwd
    _lib.require_gpu(*fwd, *bwd)
    B, T, C, H = _shape(rf, uf)
    code = _same(*fwd, *bwd)
    if rb.shape != rf.shape:
        raise _lib.PafcError("both directions must have the same (B, T, C)")
    yf, yb = torch.empty_like(rf), torch.empty_like(rb)
    ws, nbytes = _workspace(B, T, C, H, 2, chunk_len, rf.device)
    P = _lib.ptr
    wbf, wbb = w_bias if w_bias is not None else (None, None)   # (H*N,) each: time_decay, added to w in-kernel
    _lib.require_gpu(wbf, wbb)
    with op_timer("wkv6_fwd_bidir", B=B, T=T, C=C, elem_bytes=rf.element_size(), ndir=2):
        rc = _lib.lib().pafc_wkv6_forward_bidir_wbias(code, B, T, C, H, P(rf), P(kf), P(vf), P(wf), P(uf), P(wbf), P(yf),
                                                      P(rb), P(kb), P(vb), P(wb), P(ub), P(wbb), P(yb),
                                                      chunk_len, P(ws), nbytes, _lib.stream_of(rf))
    _lib.check(rc, "pafc_wkv6_forward_bidir_wbias")
    return yf, yb


def wkv6_backward(r, k, v, w, u, gy, *, reverse: bool = False, chunk_len: int = 0, s_in=None, want_gs: bool = False):
    """(gr, gk, gv, gw, gu): what WKV_6.backward returns (model.py:135-152); gu already summed over the
    batch and shaped (H, N).  With s_in (float32 (B, H, N, N), the layout of wkv6_forward's state) the recurrence
    starts from it, and with want_gs the tuple ends with gs = dL/ds_in per batch entry (WKV_6STATE.backward,
    model.py:84-101, without its sum over the batch)."""
    _lib.require_gpu(r, k, v, w, u, gy, s_in)
    B, T, C, H = _shape(r, u)
    code = _same(r, k, v, w, u, gy)
    if s_in is not None and (s_in.dtype != torch.float32 or s_in.shape != (B, H, HEAD_SIZE, HEAD_SIZE)):
        raise _lib.PafcError("s_in must be float32 (B, H, 64, 64)")
    g4 = torch.empty((4,) + tuple(r.shape), dtype=r.dtype, device=r.device)     # one buffer: the r / k / v projections' backward reads
    gr, gk, gv, gw = g4[0], g4[1], g4[2], g4[3]                                 # g_r, g_k, g_v as one batched operand (no stacking copy)
    gu = torch.empty(B, C, dtype=r.dtype, device=r.device)
    gs = torch.empty(B, H, HEAD_SIZE, HEAD_SIZE, dtype=torch.float32, device=r.device) if want_gs else None
    L = _lib.lib()
    nbytes = L.pafc_wkv6_bwd_workspace_bytes(B, T, C, H, chunk_len)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=r.device) if nbytes else None
    P = _lib.ptr
    rc = L.pafc_wkv6_backward_state(code, B, T, C, H, P(r), P(k), P(v), P(w), P(u), P(s_in), P(gy), P(gr), P(gk), P(gv),
                                    P(gw), P(gu), P(gs), int(reverse), chunk_len, P(ws), nbytes, _lib.stream_of(r))
    _lib.check(rc, "pafc_wkv6_backward_state")
    # (model.py:151: torch.sum(gu, 0) on the bf16 per-batch sums -- ONE reduction kernel, fp32 accumulation, one rounding;
    # `gu.float()` in front and `.to()` behind it were two more launches per direction and layer for the same value)
    out = (gr, gk, gv, gw, torch.sum(gu, 0).view(H, C // H))
    return out + (gs,) if want_gs else out


class _WKV6State(torch.autograd.Function):
    """Autograd wrapper = the reference's WKV_6STATE (model.py:54-103): the recurrence from an initial state, with the
    gradient of that state (per batch entry; a state parameter broadcast over the batch gets their sum from autograd)."""

    @staticmethod
    def forward(ctx, r, k, v, w, u, s, reverse):
        ctx.save_for_backward(r, k, v, w, u, s)
        ctx.reverse = reverse
        return wkv6_forward(r, k, v, w, u, s_in=s, reverse=reverse)

    @staticmethod
    def backward(ctx, gy):
        r, k, v, w, u, s = ctx.saved_tensors
        gr, gk, gv, gw, gu, gs = wkv6_backward(r, k, v, w, u, gy.contiguous(), reverse=ctx.reverse, s_in=s, want_gs=True)
        return gr, gk, gv, gw, gu, gs, None


def wkv6_state(r, k, v, w, u, s, reverse: bool = False):
    """y = WKV6(...) started from state s (float32 (B, H, N, N)), differentiable in r, k, v, w, u and s."""
    u = u.contiguous()
    s = s.contiguous()
    if torch.is_grad_enabled() and any(t.requires_grad for t in (r, k, v, w, u, s)):
        return _WKV6State.apply(r, k, v, w, u, s, reverse)
    return wkv6_forward(r, k, v, w, u, s_in=s, reverse=reverse)


class _WKV6(torch.autograd.Function):
    """Autograd wrapper = the reference's WKV_6 / WKV_6_FP32 (model.py:108-214), plus the direction flag."""

    @staticmethod
    def forward(ctx, r, k, v, w, u, reverse):
        ctx.save_for_backward(r, k, v, w, u)
        ctx.reverse = reverse
        return wkv6_forward(r, k, v, w, u, reverse=reverse)

    @staticmethod
    def backward(ctx, gy):
        r, k, v, w, u = ctx.saved_tensors
        gr, gk, gv, gw, gu = wkv6_backward(r, k, v, w, u, gy.contiguous(), reverse=ctx.reverse)
        return gr, gk, gv, gw, gu, None


def wkv6(r, k, v, w, u, reverse: bool = False):
    """y = WKV6(...) with autograd when any operand needs a gradient."""
    u = u.contiguous()
    if torch.is_autocast_enabled() and len({t.dtype for t in (r, k, v, w, u)}) > 1:
        # under autocast the projections come out in the autocast dtype while the decay / bonus keep the parameter
        # dtype: the kernel wants one dtype (model.py:116-120) -- bf16 when that is the autocast dtype, else fp32
        dt = torch.bfloat16 if torch.get_autocast_dtype("cuda") == torch.bfloat16 else torch.float32
        r, k, v, w, u = (t.to(dt).contiguous() for t in (r, k, v, w, u))
    if torch.is_grad_enabled() and any(t.requires_grad for t in (r, k, v, w, u)):
        return _WKV6.apply(r, k, v, w, u, reverse)
    return wkv6_forward(r, k, v, w, u, reverse=reverse)
