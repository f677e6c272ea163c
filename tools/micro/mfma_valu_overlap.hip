// Micro-benchmark: do fp32 MFMA and VALU instructions of one SIMD overlap (same wave / different waves)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o build_exp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF> __device__ __forceinline__ f32x4 mm(float a, float b, f32x4 c) {
    if constexpr (BF) {
        bf16x8 x, y;
        for (int i = 0; i < 8; ++i) { x[i] = (__bf16)a; y[i] = (__bf16)b; }
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
}
template <int MODE, bool BF>   // 0: MFMA only, 1: VALU only, 2: both interleaved in one wave, 3: even waves MFMA / odd waves VALU
__global__ void k(float *out, int iters, float a, float b) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + threadIdx.x + i;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 8) & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 8) & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_m && do_v) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = mm<BF>(a, b, acc[u]);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], b, a);   // 8 independent VALU per MFMA
            }
        } else if (do_m) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = mm<BF>(a, b, acc[u]);
        } else if (do_v) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], b, a);
        }
    }
    float s = 0;
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, bool BF>
void run(const char *name, int waves_per_block, float *out) {
    const int iters = 20000, blocks = 256 * 4 / (waves_per_block >= 4 ? 1 : 1);   // 4 blocks per CU -> 1 block per SIMD when 1 wave
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // one block per CU (256 CUs), waves_per_block waves: waves spread over the 4 SIMDs of the CU
    hipLaunchKernelGGL((k<MODE, BF>), dim3(256), dim3(64 * waves_per_block), 0, 0, out, 10, 1.0f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, BF>), dim3(256), dim3(64 * waves_per_block), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per iteration: 4 MFMA (and/or 32 VALU) per wave
    printf("%-32s waves/CU %2d: %.3f ms  -> %.1f ns per iteration (4 MFMA = 128 cyc, 32 VALU = 128 cyc issue)\n", name,
           waves_per_block, ms, ms * 1e6 / iters);
    (void)blocks;
}

int main() {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    for (int w : {4, 8, 16}) {
        run<0, false>("f32 mfma only", w, out);
        run<1, false>("valu only", w, out);
        run<2, false>("f32 mfma + valu, same wave", w, out);
        if (w >= 8) run<3, false>("f32 mfma waves + valu waves", w, out);
        run<0, true>("bf16 mfma only", w, out);
        run<2, true>("bf16 mfma + valu, same wave", w, out);
        if (w >= 8) run<3, true>("bf16 mfma waves + valu waves", w, out);
    }
    return 0;
}
