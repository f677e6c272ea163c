"""ctypes binding of libpafc_hip.so -- the only way product code reaches a kernel.

There is no CPU fallback anywhere in this package: if the library is missing or
a tensor is not on the GPU the call raises.  (The CPU restatement lives under
oracle/ and is test infrastructure; nothing here imports it.)
"""
import ctypes
import os
from ctypes import c_int, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libpafc_hip.so")

PAFC_F32, PAFC_BF16 = 0, 1
_ERRORS = {
    -1: "null pointer", -2: "bad dims (B, T, C, H must be positive and C == H * 64)", -3: "head size must be 64",
    -4: "workspace too small", -5: "kernel launch failed", -6: "unsupported dtype", -7: "unsupported",
    -8: "r, k, v, w, y must be 16-byte aligned",
}


class PafcError(RuntimeError):
    pass


_lib = None


def _sig(fn, restype, *argtypes):
    fn.restype = restype
    fn.argtypes = list(argtypes)


def lib():
    """Load (once) and return the C-ABI library; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    variant = os.environ.get("PAFC_SO_PATH")     # A/B measurements: a variant of the library built by csrc.build --extra / --out
    if variant:
        if not os.path.exists(variant):
            raise PafcError(f"PAFC_SO_PATH={variant} does not exist")
        _lib = _bind(ctypes.CDLL(variant))
        return _lib
    if not os.path.exists(SO_PATH):
        raise PafcError(
            f"{SO_PATH} is missing: the HIP extension has not been built "
            "(python -m paper_accurate_fast_cheap_amd.csrc.build). There is no CPU fallback.")
    from .csrc import build as _build
    if _build.built_key() != _build.source_key():
        raise PafcError(
            f"{SO_PATH} was not built from the sources in this tree (libpafc_hip.so.json: {_build.built_key()}, sources: "
            f"{_build.source_key()}): rebuild with python -m paper_accurate_fast_cheap_amd.csrc.build")
    _lib = _bind(ctypes.CDLL(SO_PATH))
    return _lib


def _bind(L):
    P, I, Z = c_void_p, c_int, c_size_t
    _sig(L.pafc_abi_version, I)
    _sig(L.pafc_wkv6_pick_chunk_len, I, I, I, I, I, I)
    _sig(L.pafc_wkv6_fwd_workspace_bytes, Z, I, I, I, I, I, I)
    _sig(L.pafc_wkv6_forward_bf16, I, I, I, I, I, P, P, P, P, P, P, I, P, Z, P)
    _sig(L.pafc_wkv6_forward_f32, I, I, I, I, I, P, P, P, P, P, P, I, P, Z, P)
    _sig(L.pafc_wkv6_forward_state, I, I, I, I, I, I, P, P, P, P, P, P, P, P, I, I, P, Z, P)
    _sig(L.pafc_wkv6_forward_bidir, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, I, P, Z, P)
    _sig(L.pafc_wkv6_forward_bidir_wbias, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, P, Z, P)
    _sig(L.pafc_wkv6_bwd_workspace_bytes, Z, I, I, I, I, I)
    _sig(L.pafc_wkv6_backward, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, I, I, P, Z, P)
    _sig(L.pafc_wkv6_backward_state, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, P, Z, P)
    _sig(L.pafc_ctc_greedy, I, I, I, I, I, P, P, I, P, P, P, P, P)
    return L


def check(code: int, what: str):
    if code != 0:
        raise PafcError(f"{what}: {_ERRORS.get(code, code)}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return PAFC_F32
    if dt == torch.bfloat16:
        return PAFC_BF16
    raise PafcError(f"kernels take float32 or bfloat16, got {dt}")


# Host cost matters: a decode batch of short utterances is ~300 launches whose GPU time is below what Python needs to issue
# them (5.8 ms per encoder pass before, 4.2 ms after this and the cached plan stamps: profiles/r04o_host_issue_time.txt).
# torch.cuda.current_stream() builds a Stream object per call (~4 us); the raw-stream query is a plain C call.  Pointers and the
# stream stay c_void_p OBJECTS: ctypes passes those pointer-sized whatever a function's declared argtypes say.
def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t: torch.Tensor) -> c_void_p:
    """torch's CURRENT stream on the tensor's device as a raw hipStream_t: the stream the kernels launch on."""
    if _raw_stream is not None:
        idx = t.device.index
        return c_void_p(_raw_stream(idx if idx is not None else torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise PafcError("this op runs on the MI355X only (tensor is on %s); there is no CPU fallback" % t.device)
        if not t.is_contiguous():
            raise PafcError("kernel operands must be contiguous")
