"""ctypes binding for oracle/wkv6_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see the header of wkv6_oracle.c).  The product package
paper_accurate_fast_cheap_amd never does.

Tensors are torch CPU tensors, contiguous, dtype float32 or bfloat16, laid out
exactly like the reference op's arguments (wenet/rwkv_v6/cuda/wkv6_op.cpp:9-33):
r, k, v, w, y: (B, T, C); u: (H, N).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libwkv6_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (idempotent)."""
    src = os.path.join(_HERE, "wkv6_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _check(B, T, C, H, *ts):
    for t in ts:
        assert t.device.type == "cpu" and t.is_contiguous()
    assert C % H == 0


def _suffix(dtype):
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.bfloat16:
        return "bf16"
    raise TypeError(f"oracle supports float32/bfloat16, got {dtype}")


def forward(r, k, v, w, u, s_in=None, want_state=False, reverse=False):
    """y (and optionally the final state) of the WKV-6 recurrence.

    s_in / returned state: float32 (B, H, N, N) indexed [b, h, i(value), j(key)]
    as in wkv6state_cuda.cu:15,23-25.  reverse=True walks t = T-1..0 (what the
    bidirectional wrapper obtains by flipping its input and output,
    rwkv_wrapper_bidirectional.py:44-48).
    """
    B, T, C = r.shape
    H = u.shape[0]
    N = C // H
    _check(B, T, C, H, r, k, v, w, u)
    y = torch.empty_like(r)
    s_out = torch.empty(B, H, N, N, dtype=torch.float32) if want_state else None
    if s_in is not None:
        assert s_in.dtype == torch.float32 and s_in.is_contiguous() and s_in.shape == (B, H, N, N)
    fn = getattr(lib(), "wkv6_oracle_forward_" + _suffix(r.dtype))
    fn(B, T, C, H, _p(r), _p(k), _p(v), _p(w), _p(u), _p(y), _p(s_in), _p(s_out), int(reverse))
    return (y, s_out) if want_state else y


def backward(r, k, v, w, u, gy):
    """(gr, gk, gv, gw, gu) with gu already summed over B and viewed (H, N),
    as WKV_6.backward returns it (wenet/rwkv_v6/src/model.py:135-152)."""
    B, T, C = r.shape
    H = u.shape[0]
    _check(B, T, C, H, r, k, v, w, u, gy)
    gr, gk, gv, gw = (torch.empty_like(r) for _ in range(4))
    gu = torch.empty(B, C, dtype=r.dtype)
    fn = getattr(lib(), "wkv6_oracle_backward_" + _suffix(r.dtype))
    fn(B, T, C, H, _p(r), _p(k), _p(v), _p(w), _p(u), _p(gy), _p(gr), _p(gk), _p(gv), _p(gw), _p(gu))
    return gr, gk, gv, gw, torch.sum(gu, 0).view(H, C // H)


def forward_closed_form_f64(r, k, v, w, u):
    B, T, C = r.shape
    H = u.shape[0]
    args = [t.double().contiguous() for t in (r, k, v, w, u)]
    y = torch.empty(B, T, C, dtype=torch.float64)
    lib().wkv6_oracle_forward_closed_form_f64(B, T, C, H, *[_p(t) for t in args], _p(y))
    return y


def state_recurrence_f64(r, k, v, w, u, s):
    """The recurrence from an initial state as a differentiable float64 torch loop -- wkv6state_cuda.cu:6-65
    (kernel_forward: y_t[i] = sum_j r_t[j] (u[j] k_t[j] v_t[i] + S[i][j]);  S[i][j] <- S[i][j] exp(-exp(w_t[j])) +
    k_t[j] v_t[i]) -- so that autograd yields what kernel_backward_111/222 (:66-296) compute analytically: gr, gk, gv,
    gw, gu and gs.  r, k, v, w: (B, T, C); u: (H, N); s: (B, H, N, N) indexed [value i][key j].  Small T only."""
    import torch
    B, T, C = r.shape
    H, N = u.shape
    r, k, v, w = (t.double().view(B, T, H, N) for t in (r, k, v, w))
    u = u.double()
    S = s.double()
    ys = []
    for t in range(T):
        kv = v[:, t, :, :, None] * k[:, t, :, None, :]                      # (B, H, i, j)
        ys.append(((u[None, :, None, :] * kv + S) * r[:, t, :, None, :]).sum(-1))
        S = S * torch.exp(-torch.exp(w[:, t]))[:, :, None, :] + kv
    return torch.stack(ys, 1).reshape(B, T, C), S
