"""Static checks on the compiled phase-pipelined GEMM (csrc/gemm_ph.hip), no GPU needed: hipcc cross-compiles to gfx950
assembly here.  The kernel's counted `s_waitcnt vmcnt(N)` waits are exact only if every vector-memory instruction between
them is one the source wrote: a register spill (scratch_load / scratch_store) or a compiler-chosen global_ / flat_ access in
the wrong place would shift the counts and turn a wait into a race that passes most runs."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "paper_accurate_fast_cheap_amd", "csrc", "gemm_ph.hip")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("asm") / "gemm_ph.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-I",
                           os.path.join(ROOT, "include"), "-I", os.path.dirname(SRC), "-S", "--cuda-device-only", SRC, "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return out.read_text()


def _kernels(asm):
    parts = re.split(r"\n(?=_ZN4pafc[^\n]*gemm_ph_kernel[^\n]*:\s)", asm)
    return {p.split(":")[0]: p.split("s_endpgm")[0] for p in parts[1:]}


def test_gemm_ph_vector_memory_instructions_are_only_the_ones_written(asm):
    ks = _kernels(asm)
    assert len(ks) >= 18                                     # every instantiation the library dispatches to
    for name, body in ks.items():
        ops = re.findall(r"^\s+((?:buffer|global|flat|scratch)_\w+)[^\n]*?(\blds\b)?\s*$", body, flags=re.M)
        kinds = {(op, bool(l)) for op, l in ops}
        assert all(op.startswith("buffer_") for op, _ in kinds), (name, sorted(kinds))       # descriptors only, no spills
        assert ("buffer_load_dwordx4", True) in kinds and ("buffer_store_dwordx4", False) in kinds
        plain_loads = sum(1 for op, l in ops if op == "buffer_load_dwordx4" and not l)
        m = re.search(r"gemm_ph_kernelILb[01]ELi\d+ELi(\d)ELi\d+ELb[01]ELi(\d)E", name)
        res, lnf = int(m.group(1)), int(m.group(2))
        # the residual (16 / 32 loads) and nothing else (bias, csum and row statistics arrive by LDS-DMA)
        assert plain_loads == {0: 0, 1: 16, 2: 32}[res], (name, plain_loads)
        assert (("buffer_store_dwordx2", False) in kinds) == (lnf == 2), name


def test_gemm_ph_has_no_register_spills(asm):
    spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", asm)
    flat = re.findall(r"gemm_ph_kernel\w+\.uses_flat_scratch, (\d+)", asm)
    assert spills and all(int(v) == 0 for v in spills)
    assert flat and all(int(v) == 0 for v in flat)          # (a reserved but unused stack slot is harmless; an access is not)
    assert "scratch_" not in asm
