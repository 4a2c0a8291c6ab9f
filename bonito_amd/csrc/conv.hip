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
        if (p.K == 5 && p.act == ACT_SWISH) {       // every bonito model: taps unrolled, the activation switch outside the channel loop
            const float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], x4 = x[4];
            for (; c0 + 8 <= p.Cout; c0 += 8) {
                half8_t o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* wr = wl + (c0 + j) * 5;
                    float a = bl[c0 + j];
                    a = fmaf(wr[0], x0, a);
                    a = fmaf(wr[1], x1, a);
                    a = fmaf(wr[2], x2, a);
                    a = fmaf(wr[3], x3, a);
                    a = fmaf(wr[4], x4, a);
                    o[j] = (half_t)fminf(fmaxf(swishf_(a), p.clamp_lo), p.clamp_hi);
                }
                *(half8_t*)(dst + c0) = o;
            }
        }
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
                float xv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) xv[g] = acc[tt][g] + bv[g];
                if (p.act == ACT_SWISH) {                   // (the activation switch once per four outputs, not once per output)
#pragma unroll
                    for (int g = 0; g < 4; ++g) xv[g] = swishf_(xv[g]);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) xv[g] = apply_act_rt(xv[g], p.act);
                }
                half4_t o;
#pragma unroll
                for (int g = 0; g < 4; ++g) o[g] = (half_t)fminf(fmaxf(xv[g], p.clamp_lo), p.clamp_hi);
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

// ---------------------------------------------------------------------------------------------
// Fused front-end (round 4): conv1 (1 -> 16 channels, VALU) -> conv2 (16 -> 16, MFMA) -> conv3 (the weight-stationary
// kernel above) in ONE kernel. The 16-channel intermediates never leave the CU: a workgroup of conv3 needs conv2's outputs for its span
// of (PB - 1) * stride + K positions, which it now COMPUTES into the LDS buffer conv_ws_kernel used to fill from global memory - in
// chunks of 256 positions: conv1 of the chunk (+ K2 - 1 halo rows) from the staged signal into a small LDS buffer, a barrier, conv2 of
// the chunk by MFMA from there into the span buffer, a barrier. The halo is recomputed (1.6 % of conv1 / conv2), 0.33 GB of writes and
// 0.33 GB of reads per hac batch are gone, and two launches. Every output is computed by the operations of the three separate kernels
// in their order (conv1: bias + fmaf over the taps; conv2: the same three k-steps into one accumulator, bias added behind them;
// positions outside a layer's output are the ZEROS of the next layer's padding, not evaluations on a padded input): identical bytes
// (tests/test_gpu_ops.py::test_fused_conv_front_end_equals_three_kernels, "conv_fuse" 0 restores the three kernels).
struct ConvFront3Args {
    const half_t* sig;     // [N][L0]
    const float* w1;       // [16][K1]
    const float* b1;       // [16]
    const half_t* w2pk;    // [16][Kp2 = 96]
    const float* b2;       // [16]
    int L0, L1, L2;        // lengths: signal, conv1 output, conv2 output
    int K1, pad1, act1;
    int K2, pad2, act2;
    float lo1, hi1, lo2, hi2;
    ConvArgs c3;           // conv3 exactly as conv_ws_kernel takes it (c3.in unused, c3.Lin = L2)
};

template <int FPW, int NKS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv_front3_kernel(ConvFront3Args q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvArgs& p = q.c3;
    constexpr int PB = 256, CH = 256, C16 = 16, NT = 64 * WAVES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * PB;
    const int span_pos = (PB - 1) * p.stride + p.K;                  // conv2 positions conv3 reads
    const int span_halves = span_pos * C16 + 32 + 8;
    half_t* xin = (half_t*)smem;                                      // conv3's input span [span_pos][16] (+ tail)
    half_t* a1 = xin + ((span_halves + 7) & ~7);                      // conv1 outputs of one chunk [CH + K2 - 1 (+ 6 rows read by the zero-padded k columns)][16]
    const int a1_rows = CH + q.K2 - 1 + 6;
    float* sl = (float*)(a1 + a1_rows * C16);                         // signal span, zero padded
    float* wl = sl + span_pos + q.K2 - 1 + q.K1 - 1 + 8;              // conv1 weights [16][K1] and bias [16]
    float* bl = wl + C16 * q.K1;

    // conv2's three weight fragments and bias (conv3's 120 registers of fragments are fetched behind the front phase: held across it they
    // pushed the kernel to the register limit)
    half8_t a2[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) a2[ks] = *(const half8_t*)(q.w2pk + (long)r * 96 + kg * 8 + ks * 32);
    float b2v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) b2v[g] = q.b2 ? q.b2[kg * 4 + g] : 0.0f;

    // ---- signal span and conv1's weights -> LDS ----------------------------------------------------------------------------------
    const int p3_start = t0 * p.stride - p.pad;                       // conv2 position of span row 0
    const int s_start = p3_start - q.pad2 - q.pad1;                   // signal position of sl[0]
    const int s_len = span_pos + q.K2 - 1 + q.K1 - 1;
    const half_t* sg = q.sig + (long)n * q.L0;
    for (int i = tid; i < s_len; i += NT) {
        const int pos = s_start + i;
        sl[i] = (pos >= 0 && pos < q.L0) ? (float)sg[pos] : 0.0f;
    }
    for (int i = tid; i < C16 * q.K1; i += NT) wl[i] = q.w1[i];
    if (tid < C16) bl[tid] = q.b1 ? q.b1[tid] : 0.0f;
    for (int e = span_pos * C16 + tid; e < span_halves; e += NT) xin[e] = (half_t)0.0f;     // tail read by conv3's zero-padded k columns
    for (int e = (CH + q.K2 - 1) * C16 + tid; e < a1_rows * C16; e += NT) a1[e] = (half_t)0.0f;   // ... and by conv2's
    __syncthreads();
    const int c0t = (tid & 1) * 8;
    float w1r[8][5], w1b[8];
    if (q.K1 == 5) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            w1b[c] = bl[c0t + c];
#pragma unroll
            for (int k = 0; k < 5; ++k) w1r[c][k] = wl[(c0t + c) * 5 + k];
        }
    }

    // ---- conv1 -> conv2 -> span buffer, 256 positions of conv2 at a time ---------------------------------------------------------
    for (int q0 = 0; q0 < span_pos; q0 += CH) {
        // conv1 rows of this chunk: a1 row i = conv1 position p3_start + q0 - pad2 + i, i < CH + K2 - 1
        const int rows1 = min(CH, span_pos - q0) + q.K2 - 1;
        // work item: eight channels (c0 = 0 or 8: fixed per thread, NT is even) of one position. K1 == 5 (every bonito model): the
        // thread's 40 weights and 8 biases live in registers and the taps are unrolled - with the generic loop below (two LDS reads
        // in front of every dependent fmaf, eight waves per CU to hide them) this phase took longer than conv3 itself
        if (q.K1 == 5) {
            for (int w = tid; w < 2 * rows1; w += NT) {
                const int i = w >> 1;
                const int u = p3_start + q0 - q.pad2 + i;
                const float* x = sl + (q0 + i);
                const float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], x4 = x[4];
                float av[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float a = w1b[c];
                    a = fmaf(w1r[c][0], x0, a);
                    a = fmaf(w1r[c][1], x1, a);
                    a = fmaf(w1r[c][2], x2, a);
                    a = fmaf(w1r[c][3], x3, a);
                    a = fmaf(w1r[c][4], x4, a);
                    av[c] = a;
                }
                if (q.act1 == ACT_SWISH) {                  // (the switch of apply_act_rt once per item instead of once per channel)
#pragma unroll
                    for (int c = 0; c < 8; ++c) av[c] = swishf_(av[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) av[c] = apply_act_rt(av[c], q.act1);
                }
                const bool inside = u >= 0 && u < q.L1;
                half8_t o;
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = inside ? (half_t)fminf(fmaxf(av[c], q.lo1), q.hi1) : (half_t)0.0f;
                *(half8_t*)(a1 + i * C16 + c0t) = o;
            }
        } else
        for (int w = tid; w < 2 * rows1; w += NT) {                   // generic tap count
            const int i = w >> 1, c0 = (w & 1) * 8;
            const int u = p3_start + q0 - q.pad2 + i;                 // conv1 output position
            const float* x = sl + (q0 + i);                          // its first tap: signal position u - pad1 = s_start + q0 + i
            half8_t o;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float* wr = wl + (c0 + c) * q.K1;
                float a = bl[c0 + c];
                for (int k = 0; k < q.K1; ++k) a = fmaf(wr[k], x[k], a);
                a = apply_act_rt(a, q.act1);
                const half_t hv = (half_t)fminf(fmaxf(a, q.lo1), q.hi1);
                o[c] = (u >= 0 && u < q.L1) ? hv : (half_t)0.0f;      // outside conv1's output: conv2's zero padding
            }
            *(half8_t*)(a1 + i * C16 + c0) = o;
        }
        __syncthreads();
        // conv2 of the chunk: position tiles wave, wave + WAVES, ... ; K = K2 * 16 halves of a row run, padded to 96 with zero weights
        const int tiles = (min(CH, span_pos - q0) + 15) >> 4;
        for (int pt = wave; pt < tiles; pt += WAVES) {
            const half_t* xrow = a1 + (pt * 16 + r) * C16 + kg * 8;
            float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) acc = mfma16(a2[ks], *(const half8_t*)(xrow + ks * 32), acc);
            const int j = q0 + pt * 16 + r;                           // span row = conv2 position p3_start + j
            const int v = p3_start + j;
            float xv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) xv[g] = acc[g] + b2v[g];
            if (q.act2 == ACT_SWISH) {
#pragma unroll
                for (int g = 0; g < 4; ++g) xv[g] = swishf_(xv[g]);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) xv[g] = apply_act_rt(xv[g], q.act2);
            }
            const bool inside = v >= 0 && v < q.L2;                                  // outside conv2's output: conv3's zero padding
            half4_t o;
#pragma unroll
            for (int g = 0; g < 4; ++g) o[g] = inside ? (half_t)fminf(fmaxf(xv[g], q.lo2), q.hi2) : (half_t)0.0f;
            if (j < span_pos) *(half4_t*)(xin + j * C16 + kg * 4) = o;
        }
        __syncthreads();
    }

    // ---- conv3: conv_ws_kernel's loop on the span buffer ---------------------------------------------------------------------------
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
    const int RS = p.stride * C16;
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
#ifdef BH_CONV_EXPT_NOSTORE       // timing experiment (wrong results): the kernel without conv3's output stores
                if (o[0] == (half_t)12345.0f)
#endif
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
int g_conv_fuse = 1;    // bh_set_option("conv_fuse", 0): the three separate kernels instead of conv_front3_kernel (A/B, regression tests)
int g_conv_fs = 1;      // bh_set_option("conv_fs", 0): never the feature-split instance of conv_igemm_kernel (A/B, tests)
int g_conv_lds_kb = 64; // bh_set_option("conv_lds_kb", v): LDS a workgroup of conv_igemm_kernel may take for its input span; the positions per
                        // workgroup follow. Measured on the v5 sup model (256 x 12000, conv class per batch): position-split instances
                        // 3.85 ms at 64 KiB, 3.25 at 80, 3.68 at 104, 4.28 at 150 (one workgroup per CU); with four k-steps of weight
                        // fragments per trip 2.99 at 80; feature-split instances (FS) 2.4-2.6 anywhere from 24 to 64 KiB, 2.69 at 80
int bh_k_conv_set_option(const char* name, int value) {
    if (name && !strcmp(name, "conv_ws")) { g_conv_ws = value; return 0; }
    if (name && !strcmp(name, "conv_fs")) { g_conv_fs = value; return 0; }
    if (name && !strcmp(name, "conv_fuse")) { g_conv_fuse = value; return 0; }
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
        BH_CHECK_HIP(bh_max_lds(fn, (int)lds));
    }
    const bool fs = g_conv_fs && Cout % 64 == 0;
    if (fs) {
        if (lds > 64 * 1024) {
            const void* fn = pw == 64 ? (const void*)conv_igemm_kernel<4, true>
                             : pw == 32 ? (const void*)conv_igemm_kernel<2, true> : (const void*)conv_igemm_kernel<1, true>;
            BH_CHECK_HIP(bh_max_lds(fn, (int)lds));
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


// Can the three convolutions at the head of an LSTM model run as conv_front3_kernel? conv1: 1 -> <= 16 channels (16 with padding),
// stride 1, K1 <= 8; conv2: 16 -> 16 (padded), stride 1, K2 * 16 <= 96; conv3: what conv_ws_kernel serves (16 -> 384 / 96 channels, Kp = 320).
int bh_k_conv_front3_ok(int c1_eff, int K1, int s1, int c2_in_eff, int c2_eff, int K2, int s2, int c3_in_eff, int c3_out, int K3, int s3) {
    if (!g_conv_fuse || !g_conv_ws) return 0;
    if (c1_eff != 16 || c2_in_eff != 16 || c2_eff != 16 || c3_in_eff != 16) return 0;
    if (s1 != 1 || s2 != 1 || K1 < 1 || K1 > 8 || K2 < 1 || K2 * 16 > 96) return 0;
    if (!(c3_out == 384 || c3_out == 96) || ((K3 * 16 + 31) / 32) * 32 != 320) return 0;
    // 96 channels (the fast models): correct (tests run it with "conv_fuse" 2) but not the default - those models keep three batches in
    // flight whose recurrent kernels share the CUs with the convolutions, and the 65 KiB, 384-thread fused workgroups cost that
    // pipeline more than the 0.04 ms of convolution time they save (bench step 2.43 -> 2.55-2.65 ms)
    if (c3_out == 96 && g_conv_fuse < 2) return 0;
    const size_t span = (size_t)255 * s3 + K3;
    const size_t lds = ((span * 16 + 40 + 7) & ~(size_t)7) * 2 + (size_t)(256 + K2 - 1 + 6) * 16 * 2 + (span + K2 + K1 + 6) * 4 + (size_t)(16 * K1 + 16) * 4;
    return lds <= 80 * 1024 ? 1 : 0;
}

int bh_k_conv_front3(const void* signal, int N, int L0, const float* w1, const float* b1, int K1, int pad1, int act1, float lo1, float hi1,
                     const void* w2pk, const float* b2, int K2, int pad2, int act2, float lo2, float hi2, const void* w3pk,
                     const float* b3, int Cout3, int K3, int stride3, int pad3, int act3, float lo3, float hi3, void* out, long os_n,
                     long os_t, hipStream_t stream) {
    using namespace bh;
    const int L1 = L0 + 2 * pad1 - K1 + 1, L2 = L1 + 2 * pad2 - K2 + 1, L3 = (L2 + 2 * pad3 - K3) / stride3 + 1;
    BH_REQUIRE(L1 > 0 && L2 > 0 && L3 > 0, "conv_front3: chunk of %d samples is too short", L0);
    BH_REQUIRE(os_t % 4 == 0 && os_n % 4 == 0, "conv_front3: output strides must be multiples of 4");
    ConvFront3Args a{(const half_t*)signal, w1, b1, (const half_t*)w2pk, b2, L0, L1, L2, K1, pad1, act1, K2, pad2, act2, lo1, hi1, lo2, hi2,
                     ConvArgs{nullptr, (const half_t*)w3pk, b3, (half_t*)out, N, L2, L3, 16, Cout3, K3, stride3, pad3, act3, 320, lo3, hi3,
                              os_n, os_t}};
    const size_t span = (size_t)255 * stride3 + K3;
    const size_t lds = ((span * 16 + 40 + 7) & ~(size_t)7) * 2 + (size_t)(256 + K2 - 1 + 6) * 16 * 2 + (span + K2 + K1 + 6) * 4 + (size_t)(16 * K1 + 16) * 4;
    const dim3 grid((L3 + 255) / 256, N);
    if (Cout3 == 384) {
        if (lds > 64 * 1024) BH_CHECK_HIP(bh_max_lds((const void*)conv_front3_kernel<3, 10, 8>, (int)lds));
        hipLaunchKernelGGL((conv_front3_kernel<3, 10, 8>), grid, dim3(512), lds, stream, a);
    } else {
        if (lds > 64 * 1024) BH_CHECK_HIP(bh_max_lds((const void*)conv_front3_kernel<1, 10, 6>, (int)lds));
        hipLaunchKernelGGL((conv_front3_kernel<1, 10, 6>), grid, dim3(384), lds, stream, a);
    }
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}
