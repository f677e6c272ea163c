// Diagnostic build of csrc/gemm_ph.hip with in-kernel shader-clock stamps (PH_STAMPS): where a tile's cycles go --
// prologue fill, each K-step, re-join, residual DMA, accumulator -> LDS, LDS -> global.  Stamps go to a buffer no other
// code reads; in the shipped library no stamp executes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I paper_accurate_fast_cheap_amd/csrc tools/micro/gemm_ph_stamps.cpp -o tools/micro/bin/gemm_ph_stamps
//   tools/micro/bin/gemm_ph_stamps M N K mode(0 SiLU / 1 bf16 residual / 2 plain / 3 GLU / 4 SiLU with folded LayerNorm / 5 residual + row statistics /
//                                         split operands with shared fragments (round 5, the headline's kernels; K = the logical K):
//                                         6 planes in -> SiLU -> planes out (w_1) / 7 planes in -> fp32 out + fp32 residual (w_2, pointwise_conv2) /
//                                         8 planes in -> GLU -> fp32 out (pointwise_conv1)) tile_m
// Round 3 stamps: 0 tile start (its prologue units are already in flight, issued ahead of the previous tile's epilogue) |
// 1 first operands landed | 2.. K-steps | 50 re-join | 51 residual folded into the accumulators (RES 1) | 52 next tile set up
// and its prologue issued | 53 epilogue arithmetic + stores issued.
#define PH_STAMPS 1
#include "../../paper_accurate_fast_cheap_amd/csrc/gemm_ph.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void fill_rand(unsigned short *p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        const float f = ((int)(h & 0xffff) - 32768) / 32768.0f;
        unsigned u; memcpy(&u, &f, 4);
        p[i] = (unsigned short)(u >> 16);
    }
}

int main(int argc, char **argv) {
    const long M = argc > 1 ? atol(argv[1]) : 44998;
    const int N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512, res = argc > 4 ? atoi(argv[4]) : 0;
    const int tn = 256, tm = argc > 5 ? atoi(argv[5]) : 256;
    const int mode = res;
    const bool spl = mode >= 6;
    unsigned short *A, *W, *O, *R, *B;
    // (split operands: A = planes [hi K | lo K], W = [hi | hi | lo]; outputs, residual and bias fp32: twice the bytes -- the
    //  buffers are sized for that, their contents are arbitrary finite bf16 / fp32 patterns)
    hipMalloc(&A, M * K * 2 * (spl ? 2 : 1)); hipMalloc(&W, (size_t)N * K * 2 * (spl ? 3 : 1)); hipMalloc(&O, M * N * 4); hipMalloc(&R, M * N * 4);
    hipMalloc(&B, N * 4);
    fill_rand<<<2048, 256>>>(A, M * K * (spl ? 2 : 1), 1); fill_rand<<<512, 256>>>(W, (size_t)N * K * (spl ? 3 : 1), 2);
    hipMemset(R, 0, M * N * 4); hipMemset(B, 0, N * 4);
    if (!spl) { fill_rand<<<2048, 256>>>(R, M * N, 3); fill_rand<<<8, 256>>>(B, N, 4); }
    const long mt = (M + tm - 1) / tm, nt = (N + tn - 1) / tn, nblk = mt * nt;
    unsigned long long *st;
    hipMalloc(&st, nblk * 2 * 64 * 8);
    hipMemset(st, 0, nblk * 2 * 64 * 8);
    pafc::PhParams p{};
    p.A = A; p.W = W; p.bias = B; p.res = (mode == 1 || mode == 5 || mode == 7) ? R : nullptr; p.out = O; p.nk1 = INT_MAX / 4; p.pb_shift = 31;
    float *stats, *csum;
    hipMalloc(&stats, M * 64); hipMalloc(&csum, (size_t)N * 4);
    hipMemset(stats, 0, M * 64); hipMemset(csum, 0, (size_t)N * 4);
    p.ln_stats = stats; p.ln_csum = csum; p.ln_eps = 1e-5f; p.ln_inv_c = 1.f / 512;
    p.M = M; p.N = N; p.K = K; p.lda = K; p.ldw = K; p.ldo = N; p.ldr = N; p.alpha = 1.f;
    if (spl) {     // as pafc_gemm_ph_ex2 sets the shared-fragment form up
        p.K = 3 * K; p.lda = 2 * K; p.ldw = 3 * K; p.nk1 = K / 64; p.pb_bytes = (long)K * 2; p.pb_shift = 30;
        p.nsteps = K / 32; p.a_lo = K * 2; p.w_lo = 2 * K * 2; p.w_step = 64;
        p.ldo = mode == 6 ? 2 * N : mode == 8 ? N / 2 : N; p.lo_off = mode == 6 ? N : 0;
    }
    p.mtiles = (int)mt; p.ntiles = (int)nt; p.tm = tm; p.stamps = st; p.batch = 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        int rc;
        rc = mode == 1 ? pafc::launch_ph<false, 0, 1, 0>(p, 1, 0) : mode == 2 ? pafc::launch_ph<false, 0, 0, 0>(p, 1, 0)
             : mode == 3 ? pafc::launch_ph<true, 0, 0, 0>(p, 1, 0) : mode == 4 ? pafc::launch_ph<false, 1, 0, 0, false, 1>(p, 1, 0)
             : mode == 5 ? pafc::launch_ph<false, 0, 1, 0, false, 2>(p, 1, 0)
             : mode == 6 ? pafc::launch_ph<false, 1, 0, 2, false, 0, true>(p, 1, 0) : mode == 7 ? pafc::launch_ph<false, 0, 2, 1, false, 0, true>(p, 1, 0)
             : mode == 8 ? pafc::launch_ph<true, 0, 0, 1, false, 0, true>(p, 1, 0) : pafc::launch_ph<false, 1, 0, 0>(p, 1, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("launch failed %d\n", rc); return 1; }
    }
    std::vector<unsigned long long> h(nblk * 2 * 64);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    const int nk = spl ? K / 32 : K / 64;      // K-steps per tile (split operands: 32 columns of both planes per step, 24 MFMAs per phase)
    printf("M %ld N %d K %d mode %d tile %d x %d: %ld tiles, %.1f us with stamps (%.0f TF/s%s)\n", M, N, K, mode, tm, tn, nblk, ms * 1e3,
           2.0 * M * N * K * (spl ? 3 : 1) / ms / 1e9, spl ? " executed, 3 MFMAs per fp32 product" : "");
    for (int g = 0; g < 2; ++g) {
        // medians over the blocks of: prologue, mean K-step, first / last K-step, re-join, residual DMA, acc -> LDS, store, total
        std::vector<double> pro, ks, k0, kl, rj, rd, wr, so, tot;
        for (long b = 0; b < nblk; ++b) {
            const unsigned long long *s = &h[(b * 2 + g) * 64];
            if (!s[53]) continue;
            pro.push_back((double)(s[1] - s[0]));
            ks.push_back((double)(s[1 + nk] - s[1]) / nk);
            k0.push_back((double)(s[2] - s[1]));
            kl.push_back((double)(s[1 + nk] - s[nk]));
            rj.push_back((double)(s[50] - s[1 + nk]));
            rd.push_back(0.0);
            wr.push_back(0.0);
            so.push_back((double)(s[53] - s[50]));
            tot.push_back((double)(s[53] - s[0]));
        }
        auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
        printf("  wave %d (median cycles over %zu tiles): tile-top barrier wait %.0f | K-step mean %.0f (first %.0f, last %.0f; ideal %d) x %d | re-join %.0f | "
               "(%.0f %.0f) epilogue %.0f | tile total %.0f\n",
               g * 4, tot.size(), med(pro), med(ks), med(k0), med(kl), spl ? 1536 : (tn == 256 ? 2048 : 1024), nk, med(rj), med(rd), med(wr), med(so), med(tot));
    }
    return 0;
}
