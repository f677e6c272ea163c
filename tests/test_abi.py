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
