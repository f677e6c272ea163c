"""Timing experiments on the matrix-core scan: build variants of the kernel with parts of the work removed (the
results are WRONG on purpose) to see what the block time is made of.  Not part of the product or the tests.

  python tools/exp_wkv_variants.py build      # here (hipcc cross-compiles): build_exp/libpafc_<variant>.so
  python tools/exp_wkv_variants.py run        # on the GPU box: time pass A / pass C per variant
"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "paper_accurate_fast_cheap_amd", "csrc")
OUT = os.path.join(ROOT, "build_exp")

P2 = """                    X[0] = mfma4(k, r, X[0]);
                    X[1] = mfma4(k * G1, r * E1, X[1]);
                    X[2] = mfma4(k * G2, r * E2, X[2]);
                    X[3] = mfma4(k * G3, r * E3, X[3]);
"""
P2_ONE = """                    X[0] = mfma4(k + k * G1 + k * G2 + k * G3, r + r * E1 + r * E2 + r * E3, X[0]);
"""
P1 = """                    for (int in = 0; in < 4; ++in) Y[in] = mfma4(rt, S[jm][in][g], Y[in]);
"""
P1_ONE = """                    for (int in = 0; in < 1; ++in) Y[in] = mfma4(rt, S[jm][0][g] + S[jm][1][g] + S[jm][2][g] + S[jm][3][g], Y[in]);
"""
DEC_A = "                const float n1 = LaneOps::xor1(d);"
DEC_B = "                const float kt = kk[g] * G4;"
DEC_CHEAP = """                const float Dtot = d, G1 = d, G2 = d, G3 = d, G4 = d, n1 = d, n2 = d, n3 = d, n4 = d;
"""
EDEC_A = "                    const float E1 = b0 ? n1 : 1.f;"
EDEC_B = "                    const float r = rr[g], k = kk[g];"
EDEC_CHEAP = """                    const float E1 = d, E2 = d, E3 = d, E4 = d;
"""
EXP = "const float d = __expf(-__expf(ww[g]));"
EXP_NONE = "const float d = ww[g];"
SDEC = "#pragma unroll\n                for (int in = 0; in < 4; ++in) S[jm][in][g] *= Dtot;\n"


def cut(s, a, b, new):
    i, j = s.index(a), s.index(b)
    return s[:i] + new + s[j:]


VARIANTS = {
    "base": lambda s: s,
    "p2one": lambda s: s.replace(P2, P2_ONE),
    "p1one": lambda s: s.replace(P1, P1_ONE),
    "p1p2one": lambda s: s.replace(P2, P2_ONE).replace(P1, P1_ONE),
    "cheapdec": lambda s: cut(cut(s, DEC_A, DEC_B, DEC_CHEAP), EDEC_A, EDEC_B, EDEC_CHEAP),
    "noexp": lambda s: s.replace(EXP, EXP_NONE),
    "nosdec": lambda s: s.replace(SDEC, ""),
    "allcheap": lambda s: cut(cut(s.replace(P2, P2_ONE).replace(P1, P1_ONE).replace(EXP, EXP_NONE).replace(SDEC, ""),
                                  DEC_A, DEC_B, DEC_CHEAP), EDEC_A, EDEC_B, EDEC_CHEAP),
}


def build():
    for name, fn in VARIANTS.items():
        d = os.path.join(OUT, name)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(os.path.join(d, "include"))
        for f in os.listdir(SRC):
            if f.endswith((".hip", ".inc", ".h")):
                shutil.copy(os.path.join(SRC, f), d)
        src = open(os.path.join(SRC, "wkv6_mfma.inc")).read()
        new = fn(src)
        assert name == "base" or new != src, name
        open(os.path.join(d, "wkv6_mfma.inc"), "w").write(new)
        txt = open(os.path.join(d, "wkv6.hip")).read().replace('"../../include/', '"' + os.path.join(ROOT, "include") + "/")
        open(os.path.join(d, "wkv6.hip"), "w").write(txt)
        so = os.path.join(OUT, f"libpafc_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                               os.path.join(d, "wkv6.hip"), "-o", so])
        print("built", so)


def run():
    import ctypes, torch
    B, T, C, H = 1, 44998, 512, 8
    dev = "cuda"
    mk = lambda: [torch.randn(B, T, C, device=dev).mul_(0.5).bfloat16() for _ in range(3)] + \
        [(torch.randn(B, T, C, device=dev) - 3).bfloat16(), (torch.randn(H, 64, device=dev) * 0.3).bfloat16(),
         torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)]
    f, b = mk(), mk()
    P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    for name in VARIANTS:
        L = ctypes.CDLL(os.path.join(OUT, f"libpafc_{name}.so"))
        L.pafc_wkv6_fwd_workspace_bytes.restype = ctypes.c_size_t
        L.pafc_wkv6_fwd_workspace_bytes.argtypes = [ctypes.c_int] * 6
        nb = L.pafc_wkv6_fwd_workspace_bytes(B, T, C, H, 2, 0)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        fn = L.pafc_wkv6_forward_bidir
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 12 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        call = lambda: fn(1, B, T, C, H, *[P(x) for x in f], *[P(x) for x in b], 0, P(ws), nb, st)
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:10s} {e0.elapsed_time(e1) * 100:.1f} us per op", flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
