"""Build libpafc_hip.so (the C-ABI library of include/*.h) for gfx950 with hipcc.

    python -m paper_accurate_fast_cheap_amd.csrc.build [--force] [-v]

hipcc cross-compiles without a GPU.  Every source becomes one object, keyed by the sha256 of its text, of every header /
.inc it could include and of the flags: an object is rebuilt exactly when that key changes (content, not mtime), objects
compile in parallel, and the link writes `libpafc_hip.so` next to the package (in-tree, git-ignored, shipped to the GPU box
with the snapshot) together with `libpafc_hip.so.json`: the key of the sources it was linked from.  `_lib.lib()` compares
that key with the sources in the tree and refuses a stale library.
"""
import glob
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libpafc_hip.so")
MANIFEST = OUT + ".json"
OBJ = os.path.join(HERE, "_obj")
ARCH = "gfx950"
_CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]   # these enter the keys
CFLAGS = _CFLAGS + ["-I", os.path.join(ROOT, "include"), "-I", HERE]      # (the include paths depend on where the tree lies)
LDFLAGS = ["-shared", "-fPIC", f"--offload-arch={ARCH}"]     # no library beyond the HIP runtime: every kernel is in csrc/


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip"))) + sorted(glob.glob(os.path.join(HERE, "*.cpp")))


def _headers():
    return sorted(glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "*.inc"))
                  + glob.glob(os.path.join(ROOT, "include", "*.h")))


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def source_key() -> str:
    """Key of everything the library is built from (sources, headers, flags)."""
    return _sha(sources() + _headers(), " ".join(_CFLAGS + LDFLAGS))


def built_key():
    try:
        with open(MANIFEST) as f:
            return json.load(f).get("source_key")
    except (OSError, ValueError):
        return None


def stale() -> bool:
    return not os.path.exists(OUT) or built_key() != source_key()


def build(force: bool = False, verbose: bool = False, extra_flags=None, out: str = None) -> str:
    """Build the library.  `extra_flags` / `out` build a VARIANT from the same sources with additional compiler flags into another
    file (A/B measurements of a flag: `_lib` loads it when PAFC_SO_PATH names it); the tree's own library is left alone."""
    variant = bool(extra_flags) or out is not None
    if variant:
        return _build_variant(list(extra_flags or []), out or (OUT + ".variant"), verbose)
    if not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdr_key = _sha(_headers(), " ".join(_CFLAGS))
    jobs, objs = [], []
    for src in sources():
        key = _sha([src], hdr_key)
        obj = os.path.join(OBJ, f"{os.path.basename(src)}.{key}.o")
        objs.append(obj)
        if force or not os.path.exists(obj):
            cmd = [hipcc] + CFLAGS + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + ["-c", src, "-o", obj + ".tmp"]
            jobs.append((cmd, obj))

    def run(job):
        cmd, obj = job
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(obj + ".tmp", obj)

    workers = max(1, min(len(jobs), int(os.environ.get("PAFC_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))))
    if jobs:
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(run, jobs))
    subprocess.check_call([hipcc] + objs + LDFLAGS + ["-o", OUT + ".tmp"])
    os.replace(OUT + ".tmp", OUT)
    with open(MANIFEST, "w") as f:
        json.dump({"source_key": source_key(), "arch": ARCH, "objects": [os.path.basename(o) for o in objs],
                   "compiled_now": len(jobs)}, f, indent=1)
    keep = set(objs)                                   # objects of older source versions
    for old in glob.glob(os.path.join(OBJ, "*.o")):
        if old not in keep:
            os.remove(old)
    return OUT


def _build_variant(extra, out, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    vdir = os.path.join(OBJ, "variant_" + hashlib.sha256(" ".join(extra).encode()).hexdigest()[:8])
    os.makedirs(vdir, exist_ok=True)
    hdr_key = _sha(_headers(), " ".join(_CFLAGS + extra))
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(vdir, f"{os.path.basename(src)}.{_sha([src], hdr_key)}.o")
        objs.append(obj)
        if not os.path.exists(obj):
            jobs.append(([hipcc] + CFLAGS + extra + ["-c", src, "-o", obj + ".tmp"], obj))

    def run(job):
        cmd, obj = job
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(obj + ".tmp", obj)

    if jobs:
        with ThreadPoolExecutor(max(1, min(len(jobs), int(os.environ.get("PAFC_BUILD_JOBS", "8"))))) as ex:
            list(ex.map(run, jobs))
    subprocess.check_call([hipcc] + objs + LDFLAGS + ["-o", out])
    return out


if __name__ == "__main__":
    # python -m ...csrc.build [--force] [-v] [--extra "<flags>" --out <file>]
    extra = sys.argv[sys.argv.index("--extra") + 1].split() if "--extra" in sys.argv else None
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, extra_flags=extra, out=out))
