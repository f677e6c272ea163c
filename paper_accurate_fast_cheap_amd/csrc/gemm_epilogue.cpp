// Linear + bias (+ SiLU) (+ residual) as ONE library GEMM with a fused epilogue (C ABI: include/pafc_encoder_ops.h).
//
// Replaces `activation(w_1(x))` of PositionwiseFeedForward.forward (wenet/transformer/positionwise_feed_forward.py:47-55,
// activation swish = SiLU): in the reference (and through torch) that is a GEMM+bias kernel followed by an
// element-wise SiLU kernel over the (rows, 2048) hidden tensor -- its largest per-layer activation, read and
// written once more.  hipBLASLt's SWISH_BIAS epilogue applies bias and x*sigmoid(x) to the fp32 accumulator before
// the single bf16 rounding.  This is a plain library GEMM (no hand-written tiling here), so it lives in a .cpp.
//
// With `residual` the same call computes  out = residual + alpha * x.W^T + bias  (beta = 1, C = residual): the
// `x = residual + ff_scale * ff(x)` / `x = residual + branch(x)` adds of ConformerEncoderLayer.forward
// (wenet/transformer/encoder_layer.py:201-259) happen on the fp32 accumulator, and the pre-norm pass that follows
// reads one tensor instead of two.
//
// The hipBLASLt handle and the heuristic's algorithm choice per (rows, N, K) are cached per process and device: they
// are library objects, not state of the computation (results never depend on them).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

#include "../../include/pafc_encoder_ops.h"

namespace {

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, d = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool ok = false;
};

std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;                                   // per device
std::map<std::tuple<int, int, long, int, int, int>, Plan> g_plans;   // (device, dtype, rows, N, K, act | bias | residual)

constexpr size_t kMaxWorkspace = 64u << 20;

}  // namespace

extern "C" {

size_t pafc_linear_act_workspace_bytes(void) { return kMaxWorkspace; }

int pafc_linear_bias_act(int dtype, long rows, int N, int K, const void *x, const void *weight, const void *bias,
                         void *out, int act, float alpha, const void *residual, void *workspace,
                         size_t workspace_bytes, pafc_stream_t stream) {
    if (!x || !weight || !out) return PAFC_ERR_NULL_POINTER;
    if (rows <= 0 || N <= 0 || K <= 0) return PAFC_ERR_BAD_DIMS;
    if (dtype != PAFC_BF16 && dtype != PAFC_F32) return PAFC_ERR_DTYPE;
    if (act != 0 && act != 1) return PAFC_ERR_UNSUPPORTED;   // 0: identity, 1: SiLU
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return PAFC_ERR_LAUNCH;
    const hipDataType dt = dtype == PAFC_BF16 ? HIP_R_16BF : HIP_R_32F;

    std::lock_guard<std::mutex> lock(g_mu);
    hipblasLtHandle_t &handle = g_handles[dev];
    if (!handle && hipblasLtCreate(&handle) != HIPBLAS_STATUS_SUCCESS) return PAFC_ERR_LAUNCH;
    Plan &p = g_plans[std::make_tuple(dev, dtype, rows, N, K, act | (bias ? 2 : 0) | (residual ? 4 : 0))];
    if (!p.ok) {
        // row-major out (rows, N) = x (rows, K) . W(N, K)^T   <=>   column-major D (N, rows) = W_cm(K, N)^T . x_cm(K, rows)
        if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return PAFC_ERR_LAUNCH;
        const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
        hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
        hipblasLtEpilogue_t epi = bias ? (act ? HIPBLASLT_EPILOGUE_SWISH_BIAS_EXT : HIPBLASLT_EPILOGUE_BIAS)
                                       : (act ? HIPBLASLT_EPILOGUE_SWISH_EXT : HIPBLASLT_EPILOGUE_DEFAULT);
        hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi));
        if (bias) {
            const int32_t bt = (int32_t)dt;
            hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
        }
        if (act) {
            const float one = 1.f;   // Swish(x, 1) = SiLU
            hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE_ACT_ARG0_EXT, &one, sizeof(one));
        }
        hipblasLtMatrixLayoutCreate(&p.a, dt, K, N, K);       // W as column-major (K, N), ld K
        hipblasLtMatrixLayoutCreate(&p.b, dt, K, rows, K);    // x as column-major (K, rows)
        hipblasLtMatrixLayoutCreate(&p.d, dt, N, rows, N);    // out as column-major (N, rows)
        hipblasLtMatmulPreference_t pref;
        hipblasLtMatmulPreferenceCreate(&pref);
        const uint64_t maxws = kMaxWorkspace;
        hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &maxws, sizeof(maxws));
        // the bias pointer takes part in the heuristic's validity check
        if (bias) hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
        hipblasLtMatmulHeuristicResult_t res[1];
        int found = 0;
        const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(handle, p.desc, p.a, p.b, p.d, p.d, pref, 1, res, &found);
        hipblasLtMatmulPreferenceDestroy(pref);
        if (st != HIPBLAS_STATUS_SUCCESS || found < 1) return PAFC_ERR_UNSUPPORTED;
        p.algo = res[0].algo;
        p.ws = res[0].workspaceSize;
        p.ok = true;
    }
    if (p.ws > 0 && (!workspace || workspace_bytes < p.ws)) return PAFC_ERR_WORKSPACE;
    if (bias) hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
    const float beta = residual ? 1.f : 0.f;
    const void *c = residual ? residual : out;
    const hipblasStatus_t st = hipblasLtMatmul(handle, p.desc, &alpha, weight, p.a, x, p.b, &beta, c, p.d, out, p.d,
                                               &p.algo, workspace, p.ws, (hipStream_t)stream);
    return st == HIPBLAS_STATUS_SUCCESS ? PAFC_OK : PAFC_ERR_LAUNCH;
}

}  // extern "C"
