"""The CPU oracle (C restatement of the CUDA kernels) under AddressSanitizer + UBSan: the checker itself must not read
or write out of bounds.  (GPU sanitizers are not available on the pool; the HIP kernels are covered by their parity
tests across ragged shapes instead.)  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
void wkv6_oracle_forward_f32(int, int, int, int, const float*, const float*, const float*, const float*, const float*,
                             float*, const float*, float*, int);
void wkv6_oracle_forward_bf16(int, int, int, int, const uint16_t*, const uint16_t*, const uint16_t*, const uint16_t*,
                              const uint16_t*, uint16_t*, const float*, float*, int);
void wkv6_oracle_backward_f32(int, int, int, int, const float*, const float*, const float*, const float*, const float*,
                              const float*, float*, float*, float*, float*, float*);
void wkv6_oracle_forward_closed_form_f64(int, int, int, int, const double*, const double*, const double*, const double*,
                                         const double*, double*);
static float rnd(unsigned *s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xffff) / 65536.0f - 0.5f; }
int main(void) {
    const int shapes[4][3] = {{1, 1, 64}, {2, 7, 128}, {1, 33, 64}, {3, 2, 192}};   /* B, T, C (N = 64) */
    unsigned seed = 7;
    for (int si = 0; si < 4; ++si) {
        const int B = shapes[si][0], T = shapes[si][1], C = shapes[si][2], H = C / 64;
        const size_t n = (size_t)B * T * C;
        float *r = malloc(n * 4), *k = malloc(n * 4), *v = malloc(n * 4), *w = malloc(n * 4), *y = malloc(n * 4);
        float *gy = malloc(n * 4), *gr = malloc(n * 4), *gk = malloc(n * 4), *gv = malloc(n * 4), *gw = malloc(n * 4);
        float *u = malloc(C * 4), *gu = malloc((size_t)B * C * 4);
        float *s0 = malloc((size_t)B * H * 64 * 64 * 4), *s1 = malloc((size_t)B * H * 64 * 64 * 4);
        uint16_t *hb = malloc(n * 2), *yb = malloc(n * 2), *ub = malloc(C * 2);
        double *rd = malloc(n * 8), *yd = malloc(n * 8), *ud = malloc(C * 8);
        for (size_t i = 0; i < n; ++i) {
            r[i] = rnd(&seed); k[i] = rnd(&seed); v[i] = rnd(&seed); w[i] = rnd(&seed) - 2.f; gy[i] = rnd(&seed);
            hb[i] = 0x3e00 + (i & 0xff); rd[i] = r[i];
        }
        for (int i = 0; i < C; ++i) { u[i] = rnd(&seed); ub[i] = 0x3d80; ud[i] = u[i]; }
        for (size_t i = 0; i < (size_t)B * H * 64 * 64; ++i) s0[i] = rnd(&seed);
        wkv6_oracle_forward_f32(B, T, C, H, r, k, v, w, u, y, NULL, NULL, 0);
        wkv6_oracle_forward_f32(B, T, C, H, r, k, v, w, u, y, s0, s1, 1);
        wkv6_oracle_forward_bf16(B, T, C, H, hb, hb, hb, hb, ub, yb, s0, s1, 0);
        wkv6_oracle_backward_f32(B, T, C, H, r, k, v, w, u, gy, gr, gk, gv, gw, gu);
        wkv6_oracle_forward_closed_form_f64(B, T, C, H, rd, rd, rd, rd, ud, yd);
        free(r); free(k); free(v); free(w); free(y); free(gy); free(gr); free(gk); free(gv); free(gw); free(u); free(gu);
        free(s0); free(s1); free(hb); free(yb); free(ub); free(rd); free(yd); free(ud);
    }
    puts("oracle sanitizer run ok");
    return 0;
}
'''


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    drv = tmp_path / "driver.c"
    drv.write_text(DRIVER)
    exe = tmp_path / "oracle_asan"
    cmd = ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-ffp-contract=off", str(drv), os.path.join(ROOT, "oracle", "wkv6_oracle.c"), "-lm", "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr.lower() and "cannot find" in build.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "oracle sanitizer run ok" in run.stdout
    assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
