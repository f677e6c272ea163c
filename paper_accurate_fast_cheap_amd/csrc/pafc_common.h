// Shared device helpers for the gfx950 kernels (wave = 64 lanes, everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pafc {

constexpr int kWave = 64;

typedef uint16_t bf16_t;  // raw bits; arithmetic is always float

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

// round-to-nearest-even through the hardware conversion (v_cvt_pk_bf16_f32: one instruction per two values; a
// hand-rolled integer rounding costs ~5 VALU ops per value and was the top cost of the element-wise kernels).
// Same mapping as c10::BFloat16 / the CPU oracle for every non-NaN input; NaN stays a (quiet) NaN.
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    return (uint32_t)__builtin_bit_cast(unsigned short, static_cast<__bf16>(f));
}

__device__ __forceinline__ float round_bf16(float f) { return static_cast<float>(static_cast<__bf16>(f)); }

template <typename ET> struct Elem;
template <> struct Elem<float> {
    static constexpr int kPerLane = 4;  // elements in one 16-byte lane access
    __device__ static __forceinline__ void unpack(const uint4 &q, float *f) {
        f[0] = __uint_as_float(q.x); f[1] = __uint_as_float(q.y);
        f[2] = __uint_as_float(q.z); f[3] = __uint_as_float(q.w);
    }
    __device__ static __forceinline__ float load(const float *p) { return *p; }
    __device__ static __forceinline__ void store(float *p, float v) { *p = v; }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kPerLane = 8;
    __device__ static __forceinline__ void unpack(const uint4 &q, float *f) {
        f[0] = bf16_bits_to_f32(q.x & 0xffffu); f[1] = __uint_as_float(q.x & 0xffff0000u);
        f[2] = bf16_bits_to_f32(q.y & 0xffffu); f[3] = __uint_as_float(q.y & 0xffff0000u);
        f[4] = bf16_bits_to_f32(q.z & 0xffffu); f[5] = __uint_as_float(q.z & 0xffff0000u);
        f[6] = bf16_bits_to_f32(q.w & 0xffffu); f[7] = __uint_as_float(q.w & 0xffff0000u);
    }
    __device__ static __forceinline__ float load(const bf16_t *p) { return bf16_bits_to_f32(*p); }
    __device__ static __forceinline__ void store(bf16_t *p, float v) { *p = (bf16_t)f32_to_bf16_bits(v); }
    __device__ static __forceinline__ float round(float v) { return round_bf16(v); }
};

// Compute units of the current device, asked once per device and process (host side; 256 if the query fails).
inline int device_cus() {
    static int cached[64];                 // 0 = not asked yet; a racing first call writes the same value twice
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int c = cached[dev];
    if (c <= 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cached[dev] = c;
    }
    return c;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    return x;
}

}  // namespace pafc
