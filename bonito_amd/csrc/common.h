// Shared device/host helpers for the gfx950 (MI355X, CDNA4) basecalling kernels.
// Everything here is written for wave64 + MFMA directly; there is no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bh {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

// Activation ids shared by the host ABI (include/bonito_hip.h: BH_ACT_*) and the kernels.
enum Act : int { ACT_NONE = 0, ACT_SWISH = 1, ACT_TANH = 2, ACT_RELU = 3 };

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

// sigmoid / tanh in fp32. exp-based forms; |err| ~1e-7 relative, far below fp16 output resolution.
// v_rcp_f32 (1 ulp) instead of the IEEE divide sequence: ~10x fewer VALU ops in the LSTM gate math.
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); saturates cleanly for |x| large (exp -> inf or 0).
    // Near zero that form cancels, so use the odd Taylor series there (rel. err < 3e-8 for |x|<1/8).
    float e = __expf(2.0f * x);
    float big = 1.0f - 2.0f * rcpf_(e + 1.0f);
    float x2 = x * x;
    float small = x * (1.0f + x2 * (-0.33333334f + x2 * 0.13333334f));
    return fabsf(x) < 0.125f ? small : big;
}
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }

template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
    if constexpr (ACT == ACT_SWISH) return swishf_(x);
    else if constexpr (ACT == ACT_TANH) return tanhf_(x);
    else if constexpr (ACT == ACT_RELU) return fmaxf(x, 0.0f);
    else return x;
}
__device__ __forceinline__ float apply_act_rt(float x, int act) {
    switch (act) {
        case ACT_SWISH: return swishf_(x);
        case ACT_TANH: return tanhf_(x);
        case ACT_RELU: return fmaxf(x, 0.0f);
        default: return x;
    }
}

// One 16x16x32 f16 MFMA: D[16x16] += A[16x32] * B[32x16].
// lane l holds A[row=l&15][k=(l>>4)*8..+8], B[k=(l>>4)*8..+8][col=l&15],
// D[row=(l>>4)*4+i][col=l&15] in acc[i].
__device__ __forceinline__ float4_t mfma16(half8_t a, half8_t b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

}  // namespace bh

// Host-side error plumbing for the C ABI (thread-local message, int status).
void bh_set_error(const char* fmt, ...);
// hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, bytes), issued only when (fn, current device) has not been raised to `bytes` yet
// (it used to run in front of every launch of the large-LDS kernels: host overhead on the hot path; engine.cpp)
hipError_t bh_max_lds(const void* fn, int bytes);
// compute units of the current device, cached per device (256 when the query fails)
int bh_cu_count();
#define BH_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            bh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                     \
            return -1;                                                                  \
        }                                                                               \
    } while (0)
#define BH_REQUIRE(cond, ...)             \
    do {                                  \
        if (!(cond)) {                    \
            bh_set_error(__VA_ARGS__);    \
            return -2;                    \
        }                                 \
    } while (0)
