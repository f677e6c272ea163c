// Micro-benchmark: what does a CU's store path sustain for the GEMM epilogue's access pattern, and does the shape of one wave
// instruction's footprint matter?  One 512-thread block per CU walks 256 x 256 bf16 output tiles of an (M x N) matrix exactly
// as gemm_ph's epilogue does (wave (wr, wc) owns 128 rows x 64 columns = 128 B per row) and only stores.
//   pattern 0: one instruction = 16 rows x 64 B   (lane & 15 = row, lane >> 4 = 16-byte piece; the MFMA C layout as it falls)
//   pattern 1: one instruction = 8 rows x 128 B   (full 128-byte lines: lane & 7 = row, lane >> 3 = piece)
//   pattern 2: one instruction = 4 rows x 256 B   (two waves' slices; only as a yardstick)
//   pattern 3: one instruction = 1 KiB contiguous (yardstick)
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_patterns.hip -o tools/micro/bin/store_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int AUX = 0>
__global__ __launch_bounds__(512) void k(unsigned char *out, long M, int N, int mtiles, int ntiles, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
    const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
    const u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    const long total = (long)mtiles * ntiles;
    for (long tt = blockIdx.x; tt < total * reps; tt += gridDim.x) {
        const long t = tt % total;
        const long m0 = (t / ntiles) * 256, n0 = (t % ntiles) * 256;
        const long base = ((m0 + wr * 128) * N + n0 + wc * 64) * 2;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            long off;
            if (PAT == 0) { const int g = i >> 1, nj = i & 1; off = (long)(g * 16 + (lane & 15)) * N * 2 + nj * 64 + (lane >> 4) * 16; }
            else if (PAT == 1) { off = (long)(i * 8 + (lane & 7)) * N * 2 + (lane >> 3) * 16; }
            else if (PAT == 2) { off = (long)(i * 8 + (wc & 1) * 4 + (lane & 3)) * N * 2 + (lane >> 2) * 16 - (wc & 1) * 128; }
            else { off = (long)(i * 8 + wc * 2 + (lane >> 5)) * N * 2 + (lane & 31) * 16 - wc * 128; }
            __builtin_amdgcn_raw_buffer_store_b128(v, R, (unsigned)(base + off), 0, AUX);
        }
    }
}

template <int PAT, int AUX = 0>
void run(const char *name, unsigned char *out, long M, int N, int reps = 1) {
    const int mt = (int)(M / 256), nt = N / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PAT, AUX>), dim3(256), dim3(512), 0, 0, out, M, N, mt, nt, reps);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<PAT, AUX>), dim3(256), dim3(512), 0, 0, out, M, N, mt, nt, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)mt * nt * 256 * 256 * 2 * reps;
    printf("%-34s M %6ld N %5d: %8.1f us  %7.1f GB/s  (%.1f B / clk / CU at 2.1 GHz, tiles per CU %.2f)\n", name, M, N, best * 1e3,
           bytes / best / 1e6, bytes / (best * 1e-3) / (mt * nt < 256 ? mt * nt : 256) / 2.1e9, (double)mt * nt / 256);
}

int main() {
    unsigned char *out;
    const long M = 45056;    // 176 row tiles
    hipMalloc(&out, (size_t)M * 2048 * 2 + (1 << 20));
    for (int N : {512, 2048}) {
        run<0>("16 rows x 64 B (as the epilogue)", out, M, N);
        run<1>("8 rows x 128 B (full lines)", out, M, N);
        run<2>("4 rows x 256 B", out, M, N);
        run<3>("2 rows x 512 B", out, M, N);
    }
    // an L2-resident footprint: the CU-side path alone
    for (int N : {512}) {
        run<0>("L2-resident 16 rows x 64 B", out, 4096, N, 64);
        run<1>("L2-resident 8 rows x 128 B", out, 4096, N, 64);
        run<3>("L2-resident 2 rows x 512 B", out, 4096, N, 64);
    }
    // cache-policy bits of the store (aux: 1 = sc0, 2 = nt, 16 = sc1), L2-resident and HBM-bound
    run<0, 1>("L2-resident 16x64 aux sc0", out, 4096, 512, 64);
    run<0, 2>("L2-resident 16x64 aux nt", out, 4096, 512, 64);
    run<0, 3>("L2-resident 16x64 aux sc0 nt", out, 4096, 512, 64);
    run<0, 16>("L2-resident 16x64 aux sc1", out, 4096, 512, 64);
    run<0, 17>("L2-resident 16x64 aux sc0 sc1", out, 4096, 512, 64);
    run<0, 2>("16x64 aux nt", out, M, 2048);
    run<0, 17>("16x64 aux sc0 sc1", out, M, 2048);
    hipFree(out);
    return 0;
}
