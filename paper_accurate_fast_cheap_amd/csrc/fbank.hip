// Kaldi-compatible log-mel filterbank on gfx950 (C ABI: include/pafc_fbank.h).
//
// Replaces torchaudio.compliance.kaldi.fbank as the reference calls it (wenet/dataset/processor.py:363-369,
// wenet/bin/encoder-rtf.py:575-583): 25 ms / 10 ms frames at 16 kHz (400 / 160 samples), snip_edges, optional
// dither, DC removal, pre-emphasis 0.97, povey window, 512-point power spectrum, triangular mel filters, log.
//
// One 256-thread block = 64 frames.  (1) each wave conditions 16 frames (mean, pre-emphasis, window) straight
// from the waveform -- frames overlap by 60 %, so the block touches 10.5 k samples once -- into an LDS tile
// A[64][400] (row stride 401 words: conflict-free column reads).  (2) the 512-point real DFT of all 64 frames is
// one fp32 GEMM A[64x400] x W[400x514] on the matrix cores with v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains;
// the zero-padded tail 400..511 contributes nothing, so K = 400).  W interleaves cos / -sin per bin, which puts
// the real and imaginary part of a bin on adjacent lanes of the accumulator.  (3) |X|^2 = re^2 + im^2 by one
// lane exchange, written to LDS P[64][257].  (4) 80 mel sums over each filter's support, log(max(., eps)),
// staged and stored as 64 contiguous rows.  fp32 throughout (bf16 operands would put ~0.4 % noise on the power
// spectrum).  The DFT is 206 k MAC per frame = 74 GFLOP for 30 minutes of audio.
#include "pafc_common.h"
#include "../../include/pafc_fbank.h"

namespace pafc {
namespace {

constexpr int WIN = 400, SHIFT = 160, NBIN = 257;
constexpr int NTILE = 17, NCOL = NTILE * 32;  // 544 >= 2 * 257
constexpr int FPB = 64;                       // frames per block
constexpr int LDA = WIN + 1;                  // LDS row stride (words)
constexpr int LDP = NBIN;                     // 257: odd, conflict-free
constexpr int MAXMEL = 128;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FbankParams {
    const float *wave;      // (S)
    long S;
    int m;                  // frames
    const float *window;    // (400)
    const float *tw;        // (400, 544): col 2k = cos(2 pi k n / 512), col 2k+1 = -sin(.), zero padded
    const float *melw;      // (nmel, 257)
    const int *mel_lo;      // (nmel) first bin with non-zero weight
    const int *mel_hi;      // (nmel) one past the last
    int nmel;
    const float *noise;     // (m, 400) or null
    float dither;
    float preemph;
    float *out;             // (m, nmel)
};

__global__ __launch_bounds__(256) void fbank_kernel(const FbankParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *A = lds;                            // [64][401]
    float *P = lds;                            // [64][257], aliases A after the GEMM
    float *O = lds + FPB * LDP + 64;           // [64][nmel] staging, behind P
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m0 = blockIdx.x * FPB;

    // ---- (1) frame conditioning ----------------------------------------------------------------------
    for (int ff = 0; ff < 16; ++ff) {
        const int f = wv * 16 + ff;
        const int fr = m0 + f;
        float x[7], xp[7];
        float sum = 0.f;
        const bool live = fr < p.m;
        const float *src = p.wave + (long)fr * SHIFT;
        const float *nz = p.noise ? p.noise + (long)fr * WIN : nullptr;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int n = lane + 64 * q;
            x[q] = 0.f; xp[q] = 0.f;
            if (live && n < WIN) {
                x[q] = src[n];
                const int np = n > 0 ? n - 1 : 0;   // replicate-padded predecessor
                xp[q] = src[np];
                if (nz) { x[q] += nz[n] * p.dither; xp[q] += nz[np] * p.dither; }
                sum += x[q];
            }
        }
        const float mean = wave_sum(sum) * (1.f / WIN);
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int n = lane + 64 * q;
            if (n < WIN) {
                const float y = (x[q] - mean) - p.preemph * (xp[q] - mean);
                A[f * LDA + n] = live ? y * p.window[n] : 0.f;
            }
        }
    }
    __syncthreads();

    // ---- (2) DFT as an fp32 MFMA GEMM: [64 x 400] x [400 x 544] -----------------------------------
    f32x16 acc[2][5];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ci = 0; ci < 5; ++ci)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][ci][r] = 0.f;
    const int lr = lane & 31, lk = lane >> 5;
    const float *twl = p.tw + lr;
#pragma unroll 2
    for (int k0 = 0; k0 < WIN; k0 += 2) {
        const int kk = k0 + lk;
        const float a0 = A[lr * LDA + kk];
        const float a1 = A[(32 + lr) * LDA + kk];
        const float *twr = twl + (long)kk * NCOL;
#pragma unroll
        for (int ci = 0; ci < 5; ++ci) {
            const int ct = wv + 4 * ci;          // wave-uniform
            if (ct < NTILE) {
                const float b = twr[ct * 32];
                acc[0][ci] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][ci], 0, 0, 0);
                acc[1][ci] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][ci], 0, 0, 0);
            }
        }
    }
    __syncthreads();   // every wave is done with A before P overwrites it

    // ---- (3) power spectrum: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ---
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ci = 0; ci < 5; ++ci) {
            const int ct = wv + 4 * ci;
            if (ct < NTILE) {
                const int bin = (ct * 32 + lr) >> 1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float sq = acc[mt][ci][r] * acc[mt][ci][r];
                    sq += __shfl_xor(sq, 1, 64);     // re^2 + im^2 (adjacent columns = adjacent lanes)
                    const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (!(lr & 1) && bin < NBIN) P[row * LDP + bin] = sq;
                }
            }
        }
    __syncthreads();

    // ---- (4) mel energies + log: lane = frame, wave = a quarter of the filters ------------------------
    const int per = (p.nmel + 3) / 4;
    const int ldo = p.nmel | 1;                  // odd row stride: conflict-free lane-per-frame writes
    const float eps = 1.1920928955078125e-07f;   // float32 machine epsilon, as torchaudio's floor
    for (int bi = 0; bi < per; ++bi) {
        const int b = wv * per + bi;             // wave-uniform
        if (b < p.nmel) {
            const int lo = p.mel_lo[b], hi = p.mel_hi[b];
            const float *wrow = p.melw + (long)b * NBIN;
            float e = 0.f;
            for (int k = lo; k < hi; ++k) e = fmaf(P[lane * LDP + k], wrow[k], e);
            O[lane * ldo + b] = logf(fmaxf(e, eps));
        }
    }
    __syncthreads();
    const int nvalid = min(FPB, p.m - m0);
    float *dst = p.out + (long)m0 * p.nmel;
    for (int i = tid; i < nvalid * p.nmel; i += 256) dst[i] = O[(i / p.nmel) * ldo + (i % p.nmel)];
}

}  // namespace
}  // namespace pafc

extern "C" {

long pafc_fbank_num_frames(long num_samples) {
    return num_samples < pafc::WIN ? 0 : 1 + (num_samples - pafc::WIN) / pafc::SHIFT;
}

int pafc_fbank_tables_cols(void) { return pafc::NCOL; }

int pafc_fbank_f32(const float *wave, long num_samples, const float *window, const float *dft_table,
                   const float *mel_weights, const int *mel_lo, const int *mel_hi, int num_mel_bins,
                   const float *noise, float dither, float preemph, float *out, pafc_stream_t stream) {
    if (!wave || !window || !dft_table || !mel_weights || !mel_lo || !mel_hi || !out) return PAFC_ERR_NULL_POINTER;
    if (num_mel_bins <= 0 || num_mel_bins > pafc::MAXMEL) return PAFC_ERR_BAD_DIMS;
    const long m = pafc_fbank_num_frames(num_samples);
    if (m <= 0 || m > 0x7fffffffL) return PAFC_ERR_BAD_DIMS;
    pafc::FbankParams p{wave, num_samples, (int)m, window, dft_table, mel_weights, mel_lo, mel_hi, num_mel_bins,
                        noise, dither, preemph, out};
    const size_t a_bytes = sizeof(float) * pafc::FPB * pafc::LDA;
    const size_t po_bytes = sizeof(float) * (pafc::FPB * pafc::LDP + 64 + pafc::FPB * (num_mel_bins | 1));
    const size_t lds = a_bytes > po_bytes ? a_bytes : po_bytes;
    // > 64 KiB of dynamic LDS needs the attribute; it is per device and idempotent, so set it on every call
    if (hipFuncSetAttribute((const void *)pafc::fbank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
        return PAFC_ERR_LAUNCH;
    const unsigned blocks = (unsigned)((m + pafc::FPB - 1) / pafc::FPB);
    hipLaunchKernelGGL(pafc::fbank_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // extern "C"
