// Diagnostic build of csrc/wkv6.hip: (1) the bidirectional scan timed launch by launch with events (A/B of compile-time variants:
// build once per -D flag), (2) with -DPAFC_WKV6_STAMPS in-kernel shader-clock stamps of one block of pass C.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I paper_accurate_fast_cheap_amd/csrc [-DPAFC_WKV6_STAMPS] [-DPAFC_SPLIT_SUB] \
//         tools/micro/wkv6_stamps.cpp -o tools/micro/bin/wkv6_stamps
//   tools/micro/bin/wkv6_stamps [B T [with_bias]]
#include "../../paper_accurate_fast_cheap_amd/csrc/wkv6.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static unsigned short bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float gauss() {
    const float a = (rand() + 1.0f) / ((float)RAND_MAX + 2.0f), b = (rand() + 1.0f) / ((float)RAND_MAX + 2.0f);
    return sqrtf(-2.f * logf(a)) * cosf(6.2831853f * b);
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1, T = argc > 2 ? atoi(argv[2]) : 44998, with_bias = argc > 3 ? atoi(argv[3]) : 0;
    const int C = 512, H = 8;
    const size_t n = (size_t)B * T * C;
    srand(11);
    std::vector<unsigned short> h(n);
    unsigned short *dev[2][6];   // r k v w y u
    for (int d = 0; d < 2; ++d) {
        for (int t = 0; t < 4; ++t) {
            for (size_t i = 0; i < n; ++i) h[i] = bf16_rne(t == 3 ? gauss() - 3.f : 0.5f * gauss());
            hipMalloc(&dev[d][t], n * 2);
            hipMemcpy(dev[d][t], h.data(), n * 2, hipMemcpyHostToDevice);
        }
        hipMalloc(&dev[d][4], n * 2);
        for (int i = 0; i < C; ++i) h[i] = bf16_rne(0.3f * gauss());
        hipMalloc(&dev[d][5], C * 2);
        hipMemcpy(dev[d][5], h.data(), C * 2, hipMemcpyHostToDevice);
    }
    unsigned short *wb = nullptr;
    if (with_bias) { hipMalloc(&wb, C * 2); hipMemset(wb, 0, C * 2); }
    const size_t wsb = pafc_wkv6_fwd_workspace_bytes(B, T, C, H, 2, 0);
    void *ws = nullptr;
    const size_t wsb2 = 4 * wsb + 16;           // room for the head-group experiment's shorter chunks
    hipMalloc(&ws, wsb2);
#ifdef PAFC_WKV6_STAMPS
    unsigned long long *st;
    const size_t nst = (size_t)2 * B * H * 4096 * 12;
    hipMalloc(&st, nst * 8);
    hipMemset(st, 0, nst * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(pafc::g_wkv_stamps), &st, sizeof(st));
#endif
    // experiment: the heads in `groups` launches of H / groups heads each (same row stride; a group's k, v, w read by its pass A
    // may still be in the Infinity Cache when its pass C reads them again)
    const int groups = argc > 4 ? atoi(argv[4]) : 1;
    if (groups > 1) {
        hipEvent_t g0, g1; hipEventCreate(&g0); hipEventCreate(&g1);
        const int Hg = H / groups;
        float gbest = 1e9f, gsum = 0;
        for (int rep = 0; rep < 23; ++rep) {
            hipEventRecord(g0, 0);
            for (int g = 0; g < groups; ++g) {
                const size_t o = (size_t)g * Hg * 64;
                pafc::DirArgs da[2] = {{dev[0][0] + o, dev[0][1] + o, dev[0][2] + o, dev[0][3] + o, dev[0][5] + o, dev[0][4] + o, nullptr, nullptr, 0, nullptr},
                                       {dev[1][0] + o, dev[1][1] + o, dev[1][2] + o, dev[1][3] + o, dev[1][5] + o, dev[1][4] + o, nullptr, nullptr, 1, nullptr}};
                pafc::FwdParams p{};
                p.d[0] = da[0]; p.d[1] = da[1];
                p.B = B; p.T = T; p.C = C; p.H = Hg;
                int L = pafc::pick_chunk_len(B, T, Hg, 2);
                L = (L + 15) / 16 * 16; if (L >= T) L = T;
                p.L = L; p.NC = (T + L - 1) / L;
                p.ws_state = (float *)ws;
                p.ws_decay = p.ws_state + (size_t)2 * B * Hg * p.NC * (64 * 64);
                if (pafc::ws_bytes(B, T, Hg, 2, L) > wsb2) { printf("workspace\n"); return 1; }
                pafc::launch_fwd<pafc::bf16_t>(p, 2, false, 0);
            }
            hipEventRecord(g1, 0); hipEventSynchronize(g1);
            float ms; hipEventElapsedTime(&ms, g0, g1);
            if (rep >= 3) { gsum += ms; if (ms < gbest) gbest = ms; }
        }
        printf("%d head groups: %.1f us mean, %.1f us best of 20\n", groups, gsum / 20 * 1e3, gbest * 1e3);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    const int reps = 20;
    for (int rep = 0; rep < reps + 3; ++rep) {
        hipEventRecord(e0, 0);
        const int rc = pafc_wkv6_forward_bidir_wbias(PAFC_BF16, B, T, C, H, dev[0][0], dev[0][1], dev[0][2], dev[0][3], dev[0][5], wb, dev[0][4],
                                                     dev[1][0], dev[1][1], dev[1][2], dev[1][3], dev[1][5], wb, dev[1][4], 0, ws, wsb, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (rc != 0) { printf("rc %d\n", rc); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    printf("B %d T %d bias %d: bidirectional scan %.1f us mean, %.1f us best of %d\n", B, T, with_bias, sum / reps * 1e3, best * 1e3, reps);
    // checksum of y (variants must agree)
    hipMemcpy(h.data(), dev[0][4], n * 2, hipMemcpyDeviceToHost);
    double cs = 0; for (size_t i = 0; i < n; ++i) { unsigned u = (unsigned)h[i] << 16; float f; memcpy(&f, &u, 4); cs += fabs(f); }
    unsigned long long hx = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { hx ^= h[i]; hx *= 1099511628211ull; }
    printf("sum |y_fwd| %.6f  fnv %016llx\n", cs, hx);
#ifdef PAFC_WKV6_STAMPS
    std::vector<unsigned long long> hs(nst);
    hipMemcpy(hs.data(), st, nst * 8, hipMemcpyDeviceToHost);
    double acc[12] = {0}; long cnt = 0;
    for (size_t w = 0; w < nst / 12; ++w) {
        const unsigned long long *s = &hs[w * 12];
        if (s[0] == 0 || s[11] <= s[0]) continue;
        for (int i = 1; i < 12; ++i) acc[i] += (double)(s[i] - s[i - 1]);
        ++cnt;
    }
    const char *name[12] = {"", "wait for operands + y flush", "LDS reads of the inputs", "prefetch issue (+ ragged masks)", "per-channel chains + k~ split",
                            "operands 1, 2 -> LDS", "read 1, 2; operands 3, r~ -> LDS; level MFMAs 1, 2", "read 3, r~; level MFMA 3",
                            "inter-block term + state split + decay", "A assembly + intra-block term", "state update", "barrier, y -> LDS, barrier"};
    double tot = 0; for (int i = 1; i < 12; ++i) tot += acc[i] / (cnt ? cnt : 1);
    printf("stamps: %ld waves, block 10 of their chunk, %.0f clocks per block (s_memtime units)\n", cnt, tot);
    for (int i = 1; i < 12; ++i) printf("  %2d %-58s %8.0f  %5.1f %%\n", i, name[i], acc[i] / (cnt ? cnt : 1), 100.0 * acc[i] / (cnt ? cnt : 1) / tot);
#endif
    return 0;
}
