// 1-D convolution front-end for gfx950 (replaces cuDNN behind bonito.nn.Convolution,
// /root/reference bonito/nn.py:222-241, with BatchNorm already folded as nn.py:447-454 does).
//
// Activations between convolutions are kept CHANNEL-MINOR ([N][L][C] fp16). With that layout the
// im2col row of output position t -- all (k, c) taps -- is ONE contiguous run of K*Cin halves
// starting at input position t*stride - pad, so the convolution is an implicit GEMM whose B
// fragments are plain 16-byte LDS reads with no gather:
//     out[t][f] = sum_{kk < K*Cin} Wp[f][kk] * in_flat[(t*stride - pad)*Cin + kk]
// Wp is the conv weight re-packed [Cout][k*Cin + c] and zero-padded to a multiple of 32.
//
//  * bh_k_conv_first : Cin == 1 (raw signal, [N][L] fp16) on the VALU, writes channel-minor.
//  * bh_k_conv_igemm : Cin % 8 == 0 on MFMA 16x16x32 f16; W is the A operand so a lane owns 4
//    consecutive output features of one position (8-byte packed stores, lane-local epilogue).
//    The output may be written NTC ([N][T][C]) or TNC ([T][N][C], what the LSTM stack consumes;
//    this folds nn.Permute([2,0,1]), nn.py:331-338, into the store).
#include "common.h"
#include "kernels.h"
#include <cstring>

namespace bh {

// ---------------------------------------------------------------------------------------------
struct ConvFirstArgs {
    const half_t* sig;  // [N][Lin]
    const float* w;     // [Cout][K]
    const float* bias;  // [Cout]
    half_t* out;
    int N, Lin, Lout, Cout, K, stride, pad, act;
    float clamp_lo, clamp_hi;
    long os_n, os_t;
    int vec8;
};

__global__ __launch_bounds__(256) void conv_first_kernel(ConvFirstArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = (float*)smem;           // [Cout][K]
    float* bl = wl + p.Cout * p.K;      // [Cout]
    float* sl = bl + p.Cout;            // signal span of this workgroup, zero padded
    const int n = blockIdx.y;
    const int tb = blockIdx.x * 256;
    const int span = 255 * p.stride + p.K;
    const half_t* s = p.sig + (long)n * p.Lin;
    for (int i = threadIdx.x; i < p.Cout * p.K; i += 256) wl[i] = p.w[i];
    for (int i = threadIdx.x; i < p.Cout; i += 256) bl[i] = p.bias ? p.bias[i] : 0.0f;
    for (int i = threadIdx.x; i < span; i += 256) {
        int pos = tb * p.stride - p.pad + i;
        sl[i] = (pos >= 0 && pos < p.Lin) ? (float)s[pos] : 0.0f;
    }
    __syncthreads();
    const int t = tb + threadIdx.x;
    if (t >= p.Lout) return;
    const float* x = sl + threadIdx.x * p.stride;
    half_t* dst = p.out + (long)n * p.os_n + (long)t * p.os_t;
    auto channel = [&](int c) {
        const float* wr = wl + c * p.K;
        float a = bl[c];
        for (int k = 0; k < p.K; ++k) a = fmaf(wr[k], x[k], a);
        a = apply_act_rt(a, p.act);
        return (half_t)fminf(fmaxf(a, p.clamp_lo), p.clamp_hi);
    };
    int c0 = 0;
    if (p.vec8) {   // Cout and both output strides are multiples of 8: 16-byte packed stores
        for (; c0 + 8 <= p.Cout; c0 += 8) {
            half8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = channel(c0 + j);
            *(half8_t*)(dst + c0) = o;
        }
    }
    for (; c0 < p.Cout; ++c0) dst[c0] = channel(c0);
}

// ---------------------------------------------------------------------------------------------
struct ConvArgs {
    const half_t* in;   // [N][Lin][Cin]
    const half_t* wpk;  // [Cout16][Kp]
    const float* bias;  // [Cout]
    half_t* out;
    int N, Lin, Lout, Cin, Cout, K, stride, pad, act;
    int Kp;             // padded K*Cin (multiple of 32)
    float clamp_lo, clamp_hi;
    long os_n, os_t;
};

// FS (round 4, layers with a multiple of 64 output channels): the four waves split the FEATURE tiles and each covers all 4 * NTT
// position tiles of the workgroup, instead of splitting the positions and each walking all feature tiles: a weight fragment is then
// fetched once per workgroup (not once per wave) and feeds 4 * NTT MFMAs instead of NTT. Same accumulation order per output.
template <int NTT, bool FS = false>  // position tiles (of 16) per wave and feature tile
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* xin = (half_t*)smem;
    constexpr int PW = NTT * 16;
    constexpr int PB = 4 * PW;  // positions per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * PB;

    // ---- stage the contiguous input span (zero outside [0, Lin)) -----------------------------
    const int span_pos = (PB - 1) * p.stride + p.K;
    const int span_halves = span_pos * p.Cin + 32 + 8;  // tail read by the zero-padded K columns
    const int p_start = t0 * p.stride - p.pad;
    const half_t* src = p.in + (long)n * p.Lin * p.Cin;
    for (int e = tid * 8; e < span_halves; e += 256 * 8) {
        int pos = p_start + e / p.Cin;
        uint4_t v = {0, 0, 0, 0};
        if (pos >= 0 && pos < p.Lin && e < span_pos * p.Cin)
            v = *(const uint4_t*)(src + (long)pos * p.Cin + (e % p.Cin));
        *(uint4_t*)(xin + e) = v;
    }
    __syncthreads();

    const int nks = p.Kp >> 5;
    const int nft = (p.Cout + 15) >> 4;
    constexpr int NT = FS ? 4 * NTT : NTT;           // position tiles this wave covers
    const int pbase = FS ? 0 : wave * PW;
    int boff[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
        boff[tt] = (pbase + tt * 16 + r) * p.stride * p.Cin + kg * 8;

    for (int ft = FS ? wave : 0; ft < nft; ft += FS ? 4 : 1) {
        float4_t acc[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[tt] = float4_t{0.f, 0.f, 0.f, 0.f};
        const half_t* wrow = p.wpk + (long)(ft * 16 + r) * p.Kp + kg * 8;
        // four k-steps per trip: their weight fragments (global, L2-resident) are requested together, so a trip waits for one round
        // trip instead of four (one k-step per trip left 1-2 MFMAs per exposed load at 16 / 32 positions per wave)
        int ks = 0;
        for (; FS && ks + 4 <= nks; ks += 4) {       // (FS instances only: the 16-channel layers of the LSTM models have 3 k-steps and measured 5-10 % slower with this loop in front of theirs)
            half8_t a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = *(const half8_t*)(wrow + (ks + u) * 32);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    half8_t b = *(const half8_t*)(xin + boff[tt] + (ks + u) * 32);
                    acc[tt] = mfma16(a[u], b, acc[tt]);
                }
        }
        for (; ks < nks; ++ks) {
            half8_t a = *(const half8_t*)(wrow + ks * 32);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                half8_t b = *(const half8_t*)(xin + boff[tt] + ks * 32);
                acc[tt] = mfma16(a, b, acc[tt]);
            }
        }
        const int f = ft * 16 + kg * 4;
        if (f < p.Cout) {
            float bv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[g] = (p.bias && f + g < p.Cout) ? p.bias[f + g] : 0.0f;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                int t = t0 + pbase + tt * 16 + r;
                if (t >= p.Lout) continue;
                half4_t o;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float x = apply_act_rt(acc[tt][g] + bv[g], p.act);
                    x = fminf(fmaxf(x, p.clamp_lo), p.clamp_hi);
                    o[g] = (half_t)x;
                }
                half_t* dst = p.out + (long)n * p.os_n + (long)t * p.os_t + f;
                if (f + 4 <= p.Cout) *(half4_t*)dst = o;
                else
                    for (int g = 0; g < 4; ++g)
                        if (f + g < p.Cout) dst[g] = o[g];
            }
        }
    }
}

// Weight-stationary variant for the layer that feeds the recurrent stack (conv3: 16 -> 384 channels in hac-sized models,
// 16 -> 96 in fast-sized ones; 19 taps): the WAVES waves of a workgroup split the feature tiles (FPW each) and keep their weight fragments in registers for the whole block of 256
// output positions; every wave walks the 16 position tiles, reading each tile's NKS input fragments from LDS once and
// using them for FPW MFMAs each. There is no global load inside the loop, so nothing ever waits on `vmcnt` and the output
// stores stream out behind the arithmetic (in conv_igemm_kernel the wait for the next tile's weights also drains the
// previous tile's stores: loads and stores share one in-order counter on gfx950). Same accumulation order as
// conv_igemm_kernel (k-steps ascending into one accumulator), so the two kernels give identical bytes.
template <int FPW, int NKS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv_ws_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* xin = (half_t*)smem;
    constexpr int PB = 256;                    // positions per workgroup = 16 tiles of 16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * PB;

    // this wave's weight fragments: FPW feature tiles x NKS k-steps (requested first, they land during the staging)
    half8_t afr[FPW][NKS];
#pragma unroll
    for (int f = 0; f < FPW; ++f) {
        const half_t* wrow = p.wpk + (long)((wave * FPW + f) * 16 + r) * p.Kp + kg * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) afr[f][ks] = *(const half8_t*)(wrow + ks * 32);
    }
    float4_t bv[FPW];
#pragma unroll
    for (int f = 0; f < FPW; ++f) {
        bv[f] = float4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv[f] = *(const float4_t*)(p.bias + (wave * FPW + f) * 16 + kg * 4);
    }

    // ---- stage the contiguous input span (zero outside [0, Lin)), as conv_igemm_kernel ----------
    const int span_pos = (PB - 1) * p.stride + p.K;
    const int span_halves = span_pos * p.Cin + 32 + 8;
    const int p_start = t0 * p.stride - p.pad;
    const half_t* src = p.in + (long)n * p.Lin * p.Cin;
    for (int e = tid * 8; e < span_halves; e += 64 * WAVES * 8) {
        int pos = p_start + e / p.Cin;
        uint4_t v = {0, 0, 0, 0};
        if (pos >= 0 && pos < p.Lin && e < span_pos * p.Cin)
            v = *(const uint4_t*)(src + (long)pos * p.Cin + (e % p.Cin));
        *(uint4_t*)(xin + e) = v;
    }
    __syncthreads();

    const int RS = p.stride * p.Cin;
    for (int pt = 0; pt < PB / 16; ++pt) {
        const int t = t0 + pt * 16 + r;
        if (t0 + pt * 16 >= p.Lout) break;
        const half_t* xrow = xin + (pt * 16 + r) * RS + kg * 8;
        half8_t b[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) b[ks] = *(const half8_t*)(xrow + ks * 32);
        float4_t acc[FPW];
#pragma unroll
        for (int f = 0; f < FPW; ++f) acc[f] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int f = 0; f < FPW; ++f) acc[f] = mfma16(afr[f][ks], b[ks], acc[f]);
#pragma unroll
        for (int f = 0; f < FPW; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[f][g] += bv[f][g];
        switch (p.act) {
            case ACT_SWISH:
#pragma unroll
                for (int f = 0; f < FPW; ++f)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[f][g] = swishf_(acc[f][g]);
                break;
            case ACT_TANH:
#pragma unroll
                for (int f = 0; f < FPW; ++f)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[f][g] = tanhf_(acc[f][g]);
                break;
            case ACT_RELU:
#pragma unroll
                for (int f = 0; f < FPW; ++f)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[f][g] = fmaxf(acc[f][g], 0.0f);
                break;
            default: break;
        }
        if (t < p.Lout) {
            half_t* drow = p.out + (long)n * p.os_n + (long)t * p.os_t + kg * 4;
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
                half4_t o;
#pragma unroll
                for (int g = 0; g < 4; ++g) o[g] = (half_t)fminf(fmaxf(acc[f][g], p.clamp_lo), p.clamp_hi);
                *(half4_t*)(drow + (wave * FPW + f) * 16) = o;
            }
        }
    }
}

}  // namespace bh

int bh_k_conv_first(const void* signal, const float* w, const float* bias, void* out, int N, int Lin,
                    int Lout, int Cout, int K, int stride, int pad, int act, float clamp_lo,
                    float clamp_hi, long os_n, long os_t, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(K >= 1 && Cout >= 1, "conv_first: bad shape (K=%d Cout=%d)", K, Cout);
    const int vec8 = (Cout % 8 == 0 && os_t % 8 == 0 && os_n % 8 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    ConvFirstArgs a{(const half_t*)signal, w, bias, (half_t*)out, N, Lin, Lout, Cout, K, stride, pad,
                    act, clamp_lo, clamp_hi, os_n, os_t, vec8};
    size_t lds = (size_t)(Cout * K + Cout + 255 * stride + K) * sizeof(float);
    hipLaunchKernelGGL(conv_first_kernel, dim3((Lout + 255) / 256, N), dim3(256), lds, stream, a);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

int g_conv_ws = 1;      // bh_set_option("conv_ws", 0): always the generic implicit-GEMM kernel (A/B, regression tests)
int g_conv_fs = 1;      // bh_set_option("conv_fs", 0): never the feature-split instance of conv_igemm_kernel (A/B, tests)
int g_conv_lds_kb = 64; // bh_set_option("conv_lds_kb", v): LDS a workgroup of conv_igemm_kernel may take for its input span; the positions per
                        // workgroup follow. Measured on the v5 sup model (256 x 12000, conv class per batch): position-split instances
                        // 3.85 ms at 64 KiB, 3.25 at 80, 3.68 at 104, 4.28 at 150 (one workgroup per CU); with four k-steps of weight
                        // fragments per trip 2.99 at 80; feature-split instances (FS) 2.4-2.6 anywhere from 24 to 64 KiB, 2.69 at 80
int bh_k_conv_set_option(const char* name, int value) {
    if (name && !strcmp(name, "conv_ws")) { g_conv_ws = value; return 0; }
    if (name && !strcmp(name, "conv_fs")) { g_conv_fs = value; return 0; }
    if (name && !strcmp(name, "conv_lds_kb")) { g_conv_lds_kb = value > 0 ? value : 64; return 0; }
    return 1;
}

int bh_k_conv_igemm(const void* in, const void* wpk, const float* bias, void* out, int N, int Lin,
                    int Lout, int Cin, int Cout, int K, int stride, int pad, int act, float clamp_lo,
                    float clamp_hi, long os_n, long os_t, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(Cin % 8 == 0 && Cout % 4 == 0, "conv_igemm: need Cin%%8==0, Cout%%4==0 (Cin=%d Cout=%d)", Cin, Cout);
    BH_REQUIRE(os_t % 4 == 0 && os_n % 4 == 0, "conv_igemm: output strides must be multiples of 4");
    ConvArgs a{(const half_t*)in, (const half_t*)wpk, bias, (half_t*)out, N, Lin, Lout, Cin, Cout, K,
               stride, pad, act, ((K * Cin + 31) / 32) * 32, clamp_lo, clamp_hi, os_n, os_t};
    auto lds_for = [&](int pw) { return (size_t)(((4 * pw - 1) * stride + K) * Cin + 40) * 2 + 16; };
    // wide output layer with the k-step count of the bonito conv3 (19 taps x 16 channels): weight-stationary kernel
    if (g_conv_ws && (Cout == 384 || Cout == 96) && a.Kp == 320 && lds_for(64) <= 64 * 1024) {
        const dim3 wgrid((Lout + 255) / 256, N);
        if (Cout == 384) hipLaunchKernelGGL((conv_ws_kernel<3, 10, 8>), wgrid, dim3(512), lds_for(64), stream, a);
        else hipLaunchKernelGGL((conv_ws_kernel<1, 10, 6>), wgrid, dim3(384), lds_for(64), stream, a);
        BH_CHECK_HIP(hipGetLastError());
        return 0;
    }
    int pw = 64;
    while (pw > 16 && lds_for(pw) > (size_t)g_conv_lds_kb * 1024) pw >>= 1;
    size_t lds = lds_for(pw);
    BH_REQUIRE(lds <= 160 * 1024, "conv_igemm: input span does not fit LDS (%zu bytes)", lds);
    dim3 grid((Lout + 4 * pw - 1) / (4 * pw), N);
    if (lds > 64 * 1024) {
        const void* fn = pw == 64 ? (const void*)conv_igemm_kernel<4>
                         : pw == 32 ? (const void*)conv_igemm_kernel<2> : (const void*)conv_igemm_kernel<1>;
        BH_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const bool fs = g_conv_fs && Cout % 64 == 0;
    if (fs) {
        if (lds > 64 * 1024) {
            const void* fn = pw == 64 ? (const void*)conv_igemm_kernel<4, true>
                             : pw == 32 ? (const void*)conv_igemm_kernel<2, true> : (const void*)conv_igemm_kernel<1, true>;
            BH_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (pw == 64) hipLaunchKernelGGL((conv_igemm_kernel<4, true>), grid, dim3(256), lds, stream, a);
        else if (pw == 32) hipLaunchKernelGGL((conv_igemm_kernel<2, true>), grid, dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((conv_igemm_kernel<1, true>), grid, dim3(256), lds, stream, a);
    } else if (pw == 64) hipLaunchKernelGGL(conv_igemm_kernel<4>, grid, dim3(256), lds, stream, a);
    else if (pw == 32) hipLaunchKernelGGL(conv_igemm_kernel<2>, grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(conv_igemm_kernel<1>, grid, dim3(256), lds, stream, a);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}
