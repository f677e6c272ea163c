"""Build libpafc_hip.so (the C-ABI library of include/*.h) for gfx950 with hipcc.

    python -m paper_accurate_fast_cheap_amd.csrc.build [--force]

hipcc cross-compiles without a GPU.  The .so is written next to the package
(paper_accurate_fast_cheap_amd/libpafc_hip.so): in-tree, git-ignored, shipped to
the GPU box with the snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libpafc_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include"), "-I", HERE]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip"))) + sorted(glob.glob(os.path.join(HERE, "*.cpp")))


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "*.inc")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", OUT + ".tmp"] + sources() + ["-L/opt/rocm/lib", "-lhipblaslt", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
