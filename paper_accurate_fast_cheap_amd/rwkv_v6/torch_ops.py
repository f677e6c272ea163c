"""`torch.ops.wkv6.*` -- the reference's native-op surface, backed by the C ABI.

The reference binds its CUDA kernels as `torch.ops.wkv6.{forward,backward,forward_fp32,backward_fp32}`
(wenet/rwkv_v6/cuda/wkv6_op.cpp:9-41: TORCH_LIBRARY(wkv6, m), every tensor passed by mutable reference, outputs
pre-allocated by the caller) and calls them from its autograd Functions exactly as

    torch.ops.wkv6.forward(B, T, C, H, r, k, v, w, u, y)                         (src/model.py:131-132, 186)
    torch.ops.wkv6.backward(B, T, C, H, r, k, v, w, u, gy, gr, gk, gv, gw, gu)   (src/model.py:147-150, 203-204)

`register()` defines the same four ops with the same argument lists, implemented for CUDA(=HIP) tensors by
`pafc_wkv6_forward_state` / `pafc_wkv6_backward_state` of libpafc_hip.so, so a reference checkout whose
`cpp_extension.load(...)` is skipped runs its own WKV_6 / WKV_6_FP32 unchanged on MI355X (INTEGRATION.md route C).
Differences from the reference binding, all in the caller's favour: launched on the tensors' current stream instead of
the default stream, dims / dtype / contiguity / device are checked and raise, no `T <= _T_` limit in backward.
The fp64 pair (wkv6_op.cpp:27-33, never called by the wrappers) is not provided.
"""
import torch

from .. import _lib

_SCHEMA_FWD = "(int B, int T, int C, int H, Tensor r, Tensor k, Tensor v, Tensor w, Tensor u, Tensor(a!) y) -> ()"
_SCHEMA_BWD = ("(int B, int T, int C, int H, Tensor r, Tensor k, Tensor v, Tensor w, Tensor u, Tensor gy, Tensor(a!) gr, "
               "Tensor(b!) gk, Tensor(c!) gv, Tensor(d!) gw, Tensor(e!) gu) -> ()")
_lib_handle = None


def _check(dtype, B, T, C, H, full, u, extra=()):
    for t in full:
        if t.dtype != dtype or tuple(t.shape) != (B, T, C) or not t.is_contiguous():
            raise _lib.PafcError(f"wkv6: operands must be contiguous (B, T, C) = ({B}, {T}, {C}) {dtype} tensors")
    if u.dtype != dtype or u.numel() != C or not u.is_contiguous():
        raise _lib.PafcError("wkv6: u must be a contiguous (H, N) tensor of the operands' dtype")
    for t, shape in extra:
        if t.dtype != dtype or tuple(t.shape) != shape or not t.is_contiguous():
            raise _lib.PafcError(f"wkv6: expected a contiguous {shape} {dtype} tensor")
    _lib.require_gpu(*full, u, *(t for t, _ in extra))


def _workspace(nbytes, device):
    return (torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes) if nbytes else (None, 0)


def _forward(dtype):
    def impl(B, T, C, H, r, k, v, w, u, y):
        _check(dtype, B, T, C, H, (r, k, v, w, y), u)
        L, P = _lib.lib(), _lib.ptr
        ws, n = _workspace(L.pafc_wkv6_fwd_workspace_bytes(B, T, C, H, 1, 0), r.device)
        _lib.check(L.pafc_wkv6_forward_state(_lib.dtype_code(dtype), B, T, C, H, P(r), P(k), P(v), P(w), P(u), P(y), None,
                                             None, 0, 0, P(ws), n, _lib.stream_of(r)), "pafc_wkv6_forward_state")
    return impl


def _backward(dtype):
    def impl(B, T, C, H, r, k, v, w, u, gy, gr, gk, gv, gw, gu):
        _check(dtype, B, T, C, H, (r, k, v, w, gy, gr, gk, gv, gw), u, extra=((gu, (B, C)),))
        L, P = _lib.lib(), _lib.ptr
        ws, n = _workspace(L.pafc_wkv6_bwd_workspace_bytes(B, T, C, H, 0), r.device)
        _lib.check(L.pafc_wkv6_backward_state(_lib.dtype_code(dtype), B, T, C, H, P(r), P(k), P(v), P(w), P(u), None, P(gy),
                                              P(gr), P(gk), P(gv), P(gw), P(gu), None, 0, 0, P(ws), n, _lib.stream_of(r)),
                   "pafc_wkv6_backward_state")
    return impl


def register() -> None:
    """Define the `wkv6` op namespace once per process (a second definition of the same namespace is an error in
    torch, e.g. after the reference's own extension was loaded: then that one stays)."""
    global _lib_handle
    if _lib_handle is not None:
        return
    try:
        lib = torch.library.Library("wkv6", "DEF")
    except RuntimeError as e:   # namespace already owned by another extension in this process
        raise _lib.PafcError(f"torch.ops.wkv6 is already defined in this process: {e}") from e
    for name, schema, fn in (("forward", _SCHEMA_FWD, _forward(torch.bfloat16)),
                             ("backward", _SCHEMA_BWD, _backward(torch.bfloat16)),
                             ("forward_fp32", _SCHEMA_FWD, _forward(torch.float32)),
                             ("backward_fp32", _SCHEMA_BWD, _backward(torch.float32))):
        lib.define(name + schema)
        lib.impl(name, fn, "CUDA")
    _lib_handle = lib
