"""CPU-only checks of the boundary: the library builds for gfx950 here (hipcc cross-compiles), loads, and
exports every symbol include/*.h declares.  No compute call is made."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so_path():
    from paper_accurate_fast_cheap_amd.csrc import build
    if not os.path.exists("/opt/rocm/bin/hipcc") and not os.path.exists(build.OUT):
        pytest.skip("no hipcc and no prebuilt library")
    return build.build() if os.path.exists("/opt/rocm/bin/hipcc") else build.OUT


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(pafc_[a-z0-9_]+)\s*\(", text))
    return names


def test_every_declared_symbol_is_exported(so_path):
    lib = ctypes.CDLL(so_path)
    declared = _declared_symbols()
    assert len(declared) >= 9
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing


def test_abi_version_and_host_side_helpers(so_path):
    lib = ctypes.CDLL(so_path)
    assert lib.pafc_abi_version() >= 1
    lib.pafc_wkv6_fwd_workspace_bytes.restype = ctypes.c_size_t
    # pure host arithmetic: serial schedule needs no scratch, the chunked one needs (N*N+N) floats per chunk
    assert lib.pafc_wkv6_fwd_workspace_bytes(1, 1000, 512, 8, 2, 1000) == 0
    assert lib.pafc_wkv6_fwd_workspace_bytes(1, 1000, 512, 8, 2, 250) == 4 * 2 * 8 * 4 * (64 * 64 + 64)
    L = lib.pafc_wkv6_pick_chunk_len(1, 44998, 512, 8, 2)
    assert 64 <= L < 44998 and L % 8 == 0
    assert lib.pafc_wkv6_pick_chunk_len(512, 400, 512, 8, 2) == 400  # enough sequences: serial


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "paper_accurate_fast_cheap_amd")
    for path in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True) + \
            glob.glob(os.path.join(pkg, "csrc", "*")):
        if os.path.isfile(path) and not path.endswith(".so"):
            src = open(path, errors="ignore").read()
            assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace("oracle/ ", ""), path


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from paper_accurate_fast_cheap_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PafcError, match="no CPU fallback"):
        _lib.lib()


def test_argument_validation_returns_error_codes_without_a_gpu(so_path):
    """Every entry point validates pointers and dimensions before it touches the device and reports a negative
    PAFC_ERR_* (the reference's op asserts or faults instead): callable here, on a box without a GPU."""
    L = ctypes.CDLL(so_path)
    P, I, G, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
    NULL = P(0)
    ERR_NULL, ERR_DIMS, ERR_UNSUP = -1, -2, -7
    one = P(16)      # a non-null, 16-byte aligned address that is never dereferenced (validation fails first)
    L.pafc_gemm_bf16.argtypes = [G, I, I, I, P, G, G, P, G, G, P, G, P, G, G, P, G, G, F, I, P]
    assert L.pafc_gemm_bf16(8, 8, 64, 1, NULL, 64, 0, one, 64, 0, NULL, 0, NULL, 0, 0, one, 8, 0, 1.0, 0, NULL) == ERR_NULL
    assert L.pafc_gemm_bf16(0, 8, 64, 1, one, 64, 0, one, 64, 0, NULL, 0, NULL, 0, 0, one, 8, 0, 1.0, 0, NULL) == ERR_DIMS
    assert L.pafc_gemm_bf16(8, 8, 48, 1, one, 48, 0, one, 48, 0, NULL, 0, NULL, 0, 0, one, 8, 0, 1.0, 0, NULL) == ERR_UNSUP
    assert L.pafc_gemm_bf16(8, 8, 64, 1, one, 64, 0, one, 64, 0, NULL, 0, NULL, 0, 0, one, 8, 0, 1.0, 9, NULL) == ERR_UNSUP
    L.pafc_conv3x3s2_nhwc_bf16.argtypes = [I, I, I, I, I, P, P, P, P, I, P]
    assert L.pafc_conv3x3s2_nhwc_bf16(1, 9, 9, 64, 128, NULL, one, NULL, one, 1, NULL) == ERR_NULL
    assert L.pafc_conv3x3s2_nhwc_bf16(1, 2, 9, 64, 128, one, one, NULL, one, 1, NULL) == ERR_DIMS
    assert L.pafc_conv3x3s2_nhwc_bf16(1, 9, 9, 60, 128, one, one, NULL, one, 1, NULL) == ERR_DIMS
    L.pafc_ctc_greedy.argtypes = [I, I, I, I, P, P, I, P, P, P, P, P]
    assert L.pafc_ctc_greedy(1, 2, 4, 5, NULL, NULL, 0, one, one, one, NULL, NULL) == ERR_NULL
    assert L.pafc_ctc_greedy(1, 2, 4, 5, one, NULL, 7, one, one, one, NULL, NULL) == ERR_DIMS     # blank >= V
    assert L.pafc_ctc_greedy(5, 2, 4, 5, one, NULL, 0, one, one, one, NULL, NULL) == -6           # dtype
    L.pafc_ctc_prefix_beam_search.argtypes = [I, I, I, P, P, P, I, I, P, P, P, P, ctypes.c_size_t, P]
    assert L.pafc_ctc_prefix_beam_search(1, 4, 17, one, one, NULL, 8, 0, one, one, one, one, 1 << 20, NULL) == ERR_UNSUP
    assert L.pafc_ctc_prefix_beam_search(1, 4, 8, one, one, NULL, 8, 0, one, one, one, one, 8, NULL) == -4  # workspace
    L.pafc_mamba2_scan.argtypes = [I, I, I, P, G, P, P, P, I, P, ctypes.c_size_t, P]
    assert L.pafc_mamba2_scan(1, 64, 4, NULL, 512, one, one, one, 0, NULL, 0, NULL) == ERR_NULL
    assert L.pafc_mamba2_scan(1, 64, 4, one, 100, one, one, one, 0, NULL, 0, NULL) == ERR_DIMS     # row too short
    L.pafc_mamba2_scan_workspace_bytes.restype = ctypes.c_size_t
    assert L.pafc_mamba2_scan_workspace_bytes(1, 64, 4, 0) == 0                                     # short: one chunk
    assert L.pafc_mamba2_scan_workspace_bytes(1, 44998, 16, 0) > 0
    L.pafc_rnnt_beam_workspace_bytes.restype = ctypes.c_size_t
    assert L.pafc_rnnt_beam_workspace_bytes(8, 250, 8) > 0 and L.pafc_rnnt_beam_workspace_bytes(0, 250, 8) == 0
    L.pafc_log_softmax_rows.argtypes = [I, G, I, P, P, P]
    assert L.pafc_log_softmax_rows(1, 0, 5, one, one, NULL) == ERR_DIMS
    # the fp32 GEMM: argument checks answer before anything is launched
    L.pafc_gemm_f32.argtypes = [G, I, I, I, P, G, G, P, G, G, P, G, P, G, G, P, G, G, F, I, P]
    assert L.pafc_gemm_f32(8, 8, 8, 1, NULL, 8, 0, one, 8, 0, NULL, 0, NULL, 0, 0, one, 8, 0, 1.0, 0, NULL) == ERR_NULL
    assert L.pafc_gemm_f32(0, 8, 8, 1, one, 8, 0, one, 8, 0, NULL, 0, NULL, 0, 0, one, 8, 0, 1.0, 0, NULL) == ERR_DIMS
    assert L.pafc_gemm_f32(8, 8, 6, 1, one, 8, 0, one, 8, 0, NULL, 0, NULL, 0, 0, one, 8, 0, 1.0, 0, NULL) == -7    # K % 4
    assert L.pafc_gemm_f32(8, 8, 8, 1, P(8), 8, 0, one, 8, 0, NULL, 0, NULL, 0, 0, one, 8, 0, 1.0, 0, NULL) == -8   # alignment
    assert L.pafc_gemm_f32(8, 8, 8, 1, one, 8, 0, one, 8, 0, NULL, 0, NULL, 0, 0, one, 8, 0, 1.0, 4, NULL) == -7    # no GLU here
    # the CTC loss kernels: workspace query and argument checks
    Z = ctypes.c_size_t
    L.pafc_ctc_loss_workspace_bytes.restype = Z
    L.pafc_ctc_loss_workspace_bytes.argtypes = [I, I, I]
    assert L.pafc_ctc_loss_workspace_bytes(32, 499, 160) == (32 * 499 * (1 + 2 * 321) + 32) * 4
    assert L.pafc_ctc_loss_workspace_bytes(0, 499, 160) == 0
    L.pafc_ctc_loss_forward.argtypes = [I, I, I, I, P, G, P, P, I, P, I, I, P, P, Z, P]
    assert L.pafc_ctc_loss_forward(1, 2, 8, 16, NULL, 16, one, one, 4, one, 4, 0, one, one, 1 << 20, NULL) == ERR_NULL
    assert L.pafc_ctc_loss_forward(1, 2, 8, 16, one, 8, one, one, 4, one, 4, 0, one, one, 1 << 20, NULL) == ERR_DIMS     # ldl < V
    assert L.pafc_ctc_loss_forward(1, 2, 8, 16, one, 16, one, one, 4, one, 4, 0, one, one, 16, NULL) == -4             # workspace
    L.pafc_ctc_loss_backward.argtypes = [I, I, I, I, P, G, P, P, I, P, I, I, P, P, F, P, G, P, Z, P]
    assert L.pafc_ctc_loss_backward(1, 2, 8, 16, one, 16, one, one, 4, one, 4, 0, one, NULL, 1.0, one, 64, one, 1 << 20, NULL) == ERR_NULL
    assert L.pafc_ctc_loss_backward(1, 2, 8, 16, one, 16, one, one, 4, one, 4, 0, one, one, 1.0, one, 8, one, 1 << 20, NULL) == ERR_DIMS   # ldg < V
    L.pafc_wkv6_forward_bf16.argtypes = [I, I, I, I, P, P, P, P, P, P, I, P, ctypes.c_size_t, P]
    assert L.pafc_wkv6_forward_bf16(1, 8, 128, 2, one, one, one, one, one, NULL, 0, NULL, 0, NULL) == ERR_NULL
    assert L.pafc_wkv6_forward_bf16(1, 8, 100, 2, one, one, one, one, one, one, 0, NULL, 0, NULL) == -3    # head size
    assert L.pafc_wkv6_forward_bf16(1, 8, 128, 2, P(8), one, one, one, one, one, 0, NULL, 0, NULL) == -8   # alignment


def test_torch_ops_wkv6_schema_matches_the_reference_binding(so_path):
    """torch.ops.wkv6.{forward,backward,forward_fp32,backward_fp32} exist with the reference's argument lists
    (wenet/rwkv_v6/cuda/wkv6_op.cpp:9-41) and have no CPU implementation: a host tensor fails loudly."""
    import torch
    from paper_accurate_fast_cheap_amd.rwkv_v6 import torch_ops
    torch_ops.register()
    torch_ops.register()          # idempotent
    fwd_args = ["B", "T", "C", "H", "r", "k", "v", "w", "u", "y"]
    bwd_args = ["B", "T", "C", "H", "r", "k", "v", "w", "u", "gy", "gr", "gk", "gv", "gw", "gu"]
    for name, args in (("forward", fwd_args), ("forward_fp32", fwd_args), ("backward", bwd_args), ("backward_fp32", bwd_args)):
        schema = getattr(torch.ops.wkv6, name).default._schema
        assert [a.name for a in schema.arguments] == args, name
        assert len(schema.returns) == 0
        written = [a.name for a in schema.arguments if a.alias_info is not None and a.alias_info.is_write]
        assert written == (["y"] if "forward" in name else ["gr", "gk", "gv", "gw", "gu"])
    t = torch.zeros(1, 4, 64)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.wkv6.forward_fp32(1, 4, 64, 1, t, t, t, t, torch.zeros(1, 64), torch.empty(1, 4, 64))


def test_ctypes_signatures_have_the_arity_of_the_header_prototypes():
    """Every `_lib._sig(L.pafc_x, restype, *argtypes)` in the package declares exactly as many arguments as the prototype of
    pafc_x in include/*.h has (a short list lets ctypes pass the tail by its default rules: a Python int as a 32-bit C int --
    a truncated pointer or stream handle)."""
    import glob
    import re
    hdr = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "include", "*.h")))
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|size_t|void|long)\s+(pafc_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    src = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "paper_accurate_fast_cheap_amd", "**", "*.py"), recursive=True))
    seen = 0
    for m in re.finditer(r"_sig\(\s*\w+\.(pafc_\w+)\s*,(.*?)\)\n", src, flags=re.S):
        name, rest = m.group(1), m.group(2)
        n = len([a for a in rest.replace("\n", " ").split(",") if a.strip()]) - 1          # minus the return type
        assert name in protos, f"{name}: bound but not declared in include/*.h"
        assert protos[name] == n, f"{name}: header has {protos[name]} arguments, the binding declares {n}"
        seen += 1
    assert seen >= 40
