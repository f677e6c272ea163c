"""Import the reference's Python (read-only, /root/reference) in THIS container.

TEST INFRASTRUCTURE ONLY -- used by tests/golden/make_goldens.py to capture
golden vectors and by tests that validate the oracle when /root/reference is
present.  Never imported by the product package, never needed on the GPU box
(the goldens are committed as data; the reference cannot travel).

Recipe = SURVEY.md section 8(c):
  1. `wenet` becomes a namespace stub whose __path__ is the reference tree
     (wenet/__init__.py imports wenet.cli.model -> torchaudio, absent here).
  2. Modules that are absent from the image or from the release get empty
     stand-ins: nvtx, torchaudio, whisper, wenet.rwkv_v7 (class_utils.py:36
     imports it; the directory was never released), wenet.transformer.decoder
     (swallowed by .gitignore:44).  None of them is on the hot path.
  3. torch 2.10 no longer re-exports typing names from torch.nn.modules.conv
     (wenet/squeezeformer/conv2d.py:17 wants them).
  4. torch.utils.cpp_extension.load -> no-op (model.py:105 would call nvcc).
  5. torch.ops.wkv6.{forward,forward_fp32,backward,backward_fp32} are defined
     with the reference's schema (cuda/wkv6_op.cpp:34-41) and implemented by
     oracle/wkv6_oracle.c -- the reference has no CPU implementation of its own.
"""
import os
import sys
import types
import typing

import torch

REFERENCE_ROOT = "/root/reference"
_installed = False


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "wenet"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present; goldens are the fallback")
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True

    wenet = types.ModuleType("wenet")
    wenet.__path__ = [os.path.join(REFERENCE_ROOT, "wenet")]
    sys.modules["wenet"] = wenet

    class _Annot:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def __call__(self, f):
            return f

    _stub("nvtx", annotate=_Annot)
    _stub("torchaudio")
    _stub("whisper")
    _stub("whisper.tokenizer", LANGUAGES={})
    _stub("wenet.rwkv_v7")

    class _Absent(torch.nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("stand-in for a module the reference release does not contain")

    _stub("wenet.rwkv_v7.rwkv_v7_wrapper_v6", RWKV_TmixWrapper=_Absent)
    _stub("wenet.transformer.decoder", TransformerDecoder=_Absent, BiTransformerDecoder=_Absent,
          LanguageSpecificTransformerDecoder=_Absent, LanguageSpecificBiTransformerDecoder=_Absent)

    import torch.nn.modules.conv as _conv
    for n in ("Union", "Optional", "List", "Tuple"):
        if not hasattr(_conv, n):
            setattr(_conv, n, getattr(typing, n))

    import torch.utils.cpp_extension as _ext
    _ext.load = lambda *a, **k: None

    _define_wkv6_cpu_op()
    _installed = True


_wkv6_lib = None


def _define_wkv6_cpu_op():
    global _wkv6_lib
    from oracle import wkv6_oracle as O

    lib = torch.library.Library("wkv6", "DEF")
    fwd = "(int B, int T, int C, int H, Tensor r, Tensor k, Tensor v, Tensor w, Tensor u, Tensor(a!) y) -> ()"
    bwd = ("(int B, int T, int C, int H, Tensor r, Tensor k, Tensor v, Tensor w, Tensor u, Tensor gy, "
           "Tensor(a!) gr, Tensor(b!) gk, Tensor(c!) gv, Tensor(d!) gw, Tensor(e!) gu) -> ()")
    for name in ("forward", "forward_fp32"):
        lib.define(name + fwd)
    for name in ("backward", "backward_fp32"):
        lib.define(name + bwd)

    def _fwd(B, T, C, H, r, k, v, w, u, y):
        y.copy_(O.forward(r, k, v, w, u.contiguous()))

    def _bwd(B, T, C, H, r, k, v, w, u, gy, gr, gk, gv, gw, gu):
        # gu is the (B, C) per-batch partial in the reference ABI; the oracle
        # wrapper returns the B-summed (H, N) form, so call the C symbol directly.
        import ctypes
        fn = getattr(O.lib(), "wkv6_oracle_backward_" + ("f32" if r.dtype == torch.float32 else "bf16"))
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        fn(B, T, C, H, p(r), p(k), p(v), p(w), p(u.contiguous()), p(gy), p(gr), p(gk), p(gv), p(gw), p(gu))

    for name in ("forward", "forward_fp32"):
        lib.impl(name, _fwd, "CPU")
    for name in ("backward", "backward_fp32"):
        lib.impl(name, _bwd, "CPU")
    _wkv6_lib = lib  # keep alive
