// CTC-CRF Viterbi decode on gfx950 -- integer-exact best path / move table.
// Restates the Max-semiring path of CTC_CRF (/root/reference bonito/crf/model.py:30-42 idx table,
// :98-103 viterbi, :105-108 path_to_str), which the reference delegates to koi's cupy kernels
// (crf/model.py:9-10,52,61,67).
//
//   states j in [0, S), S = 4^state_len (k-mer, oldest base most significant)
//   Ms[t][j][0]   = stay in j            (5S layout: scores[t][5j];   4S/koi layout: blank_score)
//   Ms[t][j][1+r] = move into j from idx[j][1+r] = r*S/4 + j/4   (5S: scores[t][5j+1+r]; 4S: scores[t][4j+r])
//   alpha_0 = 0;  alpha_{t+1}[j] = max_k Ms[t][j][k] + alpha_t[idx[j][k]]   (ties: lowest k)
//   final state = argmax_j alpha_T[j] (ties: lowest j); traceback gives k_t, j_t:
//   move_t = (k_t != 0), path_t = move_t ? 1 + (j_t & 3) : 0.
// fp32 left-fold of fp16 inputs with exactly one addend per step => bit-identical to the CPU oracle.
//
// Round 1 kernel (`crf_viterbi_kernel`; kept for the reference's [T][N][5S] layout, for S < 64 and for unaligned strides): one workgroup
// per chunk, one thread per state; alpha ping-pongs in LDS (one barrier per step); score loads are register-prefetched U steps ahead so
// the serial alpha chain never waits on HBM; 3-bit back-pointers go to a [N][T][S] byte workspace and are chased back in LDS-staged blocks.
//
// Round 5 kernel (`crf_viterbi_quad_kernel`, koi layout, S >= 64): the lane <-> state mapping of the BS-2 forward scan (beam.hip). Thread i
// of a chunk owns the FOUR states 4i .. 4i+3: they share their four predecessors r S/4 + i, and the 16 transition scores they need are
// the 32 contiguous bytes row[16 i .. 16 i + 15] - every byte a lane loads is used, the four stay terms come from the lane's own registers,
// and a step is two LDS reads of two floats (stride S/4: ds_read2st64_b32 at 256 states), sixteen adds and compares, one 16-byte LDS
// write. 256 states = ONE WAVE per chunk: no barrier at all (a wave's LDS operations complete in order), four chunks per workgroup;
// 64 states = four chunks per wave; 1024 states = four waves per chunk, one barrier per step. Back-pointers: the four 3-bit values of a
// thread as one 16-bit word, [N][T][S/4] (half the bytes of round 1's). Same arithmetic, same tie rules -> the same bytes as the round-1
// kernel and the CPU oracle (tests/test_gpu_decode.py). Per call on MI355X (tools/decode_bench.py ... viterbi): 2048 x 1667 x 1024 (hac) 3.50 -> 1.62 ms
// (7.9 GB of scores and back-pointers = 4.9 TB/s), 2048 x 1667 x 256 (fast) 2.70 -> 0.98, 512 x 2000 x 4096 (sup) 4.48 -> 2.18.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace bh {

struct VitArgs {
    const half_t* scores;
    int N, T, S;
    float blank;
    long s_n, s_t;
    uint8_t* bp;     // [N][T][S]
    int8_t* moves;   // [N][T]
    int8_t* path;    // [N][T]
    float* best;     // [N] best path score (may be null)
};

constexpr int VU = 8;  // prefetch depth (time steps)
int g_viterbi_quad = 1;       // "viterbi_quad": 1 = four states per thread (round 5), 0 = the round-1 kernel everywhere

template <bool L5S>
__global__ void crf_viterbi_kernel(VitArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = p.S;
    int& s_state = *(int*)smem;               // all LDS lives in the dynamic region (16-byte aligned)
    float* al = (float*)(smem + 16);          // [2][S]
    uint8_t* stage = (uint8_t*)(al + 2 * S);  // traceback staging, TB*S bytes + 2*TB result bytes
    const int n = blockIdx.x;
    const int j = threadIdx.x;
    const bool active = j < S;
    const int q = S >> 2;
    const half_t* sc = p.scores + (long)n * p.s_n;
    uint8_t* bp = p.bp + (long)n * p.T * S;

    if (active) al[j] = 0.0f;
    __syncthreads();

    float cur[VU][5], nxt[VU][5];
    auto load = [&](float (&dst)[VU][5], int t0) {
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            int t = t0 + u;
            if (active && t < p.T) {
                const half_t* s = sc + (long)t * p.s_t;
                if constexpr (L5S) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) dst[u][k] = (float)s[j * 5 + k];
                } else {
                    half4_t v = *(const half4_t*)(s + j * 4);
                    dst[u][0] = p.blank;
#pragma unroll
                    for (int k = 0; k < 4; ++k) dst[u][1 + k] = (float)v[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 5; ++k) dst[u][k] = 0.0f;
            }
        }
    };

    int cb = 0;
    load(cur, 0);
    for (int t0 = 0; t0 < p.T; t0 += VU) {
        load(nxt, t0 + VU);
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            int t = t0 + u;
            if (t < p.T) {   // uniform across the workgroup
                const float* a = al + cb * S;
                if (active) {
                    float best = a[j] + cur[u][0];
                    int k = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float cand = a[r * q + (j >> 2)] + cur[u][1 + r];
                        if (cand > best) { best = cand; k = 1 + r; }
                    }
                    al[(cb ^ 1) * S + j] = best;
                    bp[(long)t * S + j] = (uint8_t)k;
                }
                cb ^= 1;
                __syncthreads();
            }
        }
#pragma unroll
        for (int u = 0; u < VU; ++u)
#pragma unroll
            for (int k = 0; k < 5; ++k) cur[u][k] = nxt[u][k];
    }

    // ---- final state: argmax_j alpha_T[j], lowest j on ties (serial scan by one thread per 64) ----
    if (j == 0) {
        const float* a = al + cb * S;
        float best = a[0];
        int bj = 0;
        for (int i = 1; i < S; ++i)
            if (a[i] > best) { best = a[i]; bj = i; }
        s_state = bj;
        if (p.best) p.best[n] = best;
    }
    __threadfence();   // back-pointer stores of this workgroup -> visible to its own later loads
    __syncthreads();

    // ---- traceback in LDS-staged blocks of TB steps ----
    const int TB = max(1, min(512, (32 * 1024) / S));
    int8_t* res_m = (int8_t*)(stage + TB * S);
    int8_t* res_p = res_m + TB;
    int8_t* mo = p.moves + (long)n * p.T;
    int8_t* pa = p.path + (long)n * p.T;
    for (int thi = p.T; thi > 0; thi -= TB) {
        const int tlo = max(0, thi - TB);
        const int nb = (thi - tlo) * S;
        const uint8_t* src = bp + (long)tlo * S;
        for (int i = threadIdx.x * 4; i < nb; i += blockDim.x * 4)   // S % 4 == 0
            *(unsigned*)(stage + i) = *(const unsigned*)(src + i);
        __syncthreads();
        if (j == 0) {
            int st = s_state;
            for (int t = thi - 1; t >= tlo; --t) {
                int k = stage[(t - tlo) * S + st];
                res_m[t - tlo] = (int8_t)(k != 0);
                res_p[t - tlo] = (int8_t)(k != 0 ? 1 + (st & 3) : 0);
                if (k != 0) st = (k - 1) * q + (st >> 2);
            }
            s_state = st;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < thi - tlo; i += blockDim.x) {
            mo[tlo + i] = res_m[i];
            pa[tlo + i] = res_p[i];
        }
        __syncthreads();
    }
}

template <int SL>
struct VitQuadGeo {
    static constexpr int S = 1 << (2 * SL), Q = S / 4;              // Q threads per chunk, four states each
    static constexpr int CPB = Q >= 256 ? 1 : 256 / Q;              // chunks per workgroup of 256 threads
    static constexpr bool BARRIER = Q > 64;                         // a chunk spans several waves
    static constexpr int TB = 64;                                   // traceback block (time steps)
    static constexpr int ALPHA = (BARRIER ? 2 : 1) * S * 4;         // bytes per chunk
    static constexpr int STAGE = TB * Q * 2;                        // back-pointer block, bytes per chunk
    static constexpr int PER_CHUNK = ALPHA + STAGE + 2 * TB + 16;   // + the block's moves / path + final state
    static constexpr int LDS = CPB * PER_CHUNK;
};

constexpr int VQU = 4;  // prefetch depth (time steps)

template <int SL>
__global__ __launch_bounds__(256) void crf_viterbi_quad_kernel(VitArgs p) {
    using G = VitQuadGeo<SL>;
    constexpr int S = G::S, Q = G::Q;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int c = threadIdx.x / Q, i = threadIdx.x % Q;
    const int n_raw = blockIdx.x * G::CPB + c;
    const bool valid = n_raw < p.N;                                 // a ragged last workgroup computes chunk N-1 again and stores nothing
    const int n = valid ? n_raw : p.N - 1;
    char* base = smem + c * G::PER_CHUNK;
    float* al = (float*)base;                                       // [1 or 2][S]
    unsigned short* stage = (unsigned short*)(base + G::ALPHA);     // [TB][Q]
    int8_t* res_m = (int8_t*)(base + G::ALPHA + G::STAGE);
    int8_t* res_p = res_m + G::TB;
    int* s_state = (int*)(res_p + G::TB);
    const half_t* sc = p.scores + (long)n * p.s_n + 16 * i;
    unsigned short* bp = (unsigned short*)p.bp + (long)n * p.T * Q;

    float own[4] = {0.f, 0.f, 0.f, 0.f};
    *(float4_t*)(al + 4 * i) = float4_t{0.f, 0.f, 0.f, 0.f};
    if (G::BARRIER) __syncthreads();

    uint4_t cur[VQU][2], nxt[VQU][2];
    auto load = [&](uint4_t (&dst)[VQU][2], int t0) {
#pragma unroll
        for (int u = 0; u < VQU; ++u) {
            const int t = min(t0 + u, p.T - 1);                     // (rows beyond the end are never used)
            const uint4_t* s = (const uint4_t*)(sc + (long)t * p.s_t);
            dst[u][0] = s[0];
            dst[u][1] = s[1];
        }
    };
    int cb = 0;
    load(cur, 0);
    for (int t0 = 0; t0 < p.T; t0 += VQU) {
        load(nxt, t0 + VQU);
#pragma unroll
        for (int u = 0; u < VQU; ++u) {
            const int t = t0 + u;
            if (t < p.T) {                                          // uniform across the workgroup
                const float* a = al + (G::BARRIER ? cb * S : 0);
                float pa[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) pa[r] = a[r * Q + i];
                const half8_t h0 = __builtin_bit_cast(half8_t, cur[u][0]), h1 = __builtin_bit_cast(half8_t, cur[u][1]);
                unsigned kk = 0;
                float nw[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    float best = own[x] + p.blank;
                    unsigned k = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = 4 * x + r;
                        const float cand = pa[r] + (float)(e < 8 ? h0[e] : h1[e - 8]);
                        if (cand > best) { best = cand; k = 1 + r; }
                    }
                    nw[x] = best;
                    kk |= k << (3 * x);
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) own[x] = nw[x];
                // (one wave per chunk: every lane's predecessor reads above were issued before this write; LDS operations of a wave complete in order)
                *(float4_t*)(al + (G::BARRIER ? (cb ^ 1) * S : 0) + 4 * i) = float4_t{nw[0], nw[1], nw[2], nw[3]};
                if (valid) bp[(long)t * Q + i] = (unsigned short)kk;
                if (G::BARRIER) { cb ^= 1; __syncthreads(); }
            }
        }
#pragma unroll
        for (int u = 0; u < VQU; ++u) { cur[u][0] = nxt[u][0]; cur[u][1] = nxt[u][1]; }
    }

    // ---- final state: argmax_j alpha_T[j], lowest j on ties (one thread per chunk scans the S values in LDS) ----
    __syncthreads();
    if (i == 0) {
        const float* a = al + (G::BARRIER ? cb * S : 0);
        float best = a[0];
        int bj = 0;
        for (int j = 1; j < S; ++j)
            if (a[j] > best) { best = a[j]; bj = j; }
        *s_state = bj;
        if (p.best && valid) p.best[n] = best;
    }
    __threadfence();   // back-pointer stores of this workgroup -> visible to its own later loads
    __syncthreads();

    // ---- traceback in LDS-staged blocks of TB steps: the chunk's threads copy, its first thread chases ----
    int8_t* mo = p.moves + (long)n * p.T;
    int8_t* pa_out = p.path + (long)n * p.T;
    for (int thi = p.T; thi > 0; thi -= G::TB) {
        const int tlo = max(0, thi - G::TB);
        const int n16 = (thi - tlo) * Q / 8;                        // 16-byte pieces (Q % 8 == 0)
        const uint4_t* src = (const uint4_t*)(bp + (long)tlo * Q);
        for (int e = i; e < n16; e += Q) ((uint4_t*)stage)[e] = src[e];
        __syncthreads();
        if (i == 0) {
            int st = *s_state;
            for (int t = thi - 1; t >= tlo; --t) {
                const int k = (stage[(t - tlo) * Q + (st >> 2)] >> (3 * (st & 3))) & 7;
                res_m[t - tlo] = (int8_t)(k != 0);
                res_p[t - tlo] = (int8_t)(k != 0 ? 1 + (st & 3) : 0);
                if (k != 0) st = (k - 1) * Q + (st >> 2);
            }
            *s_state = st;
        }
        __syncthreads();
        if (valid)
            for (int e = i; e < thi - tlo; e += Q) {
                mo[tlo + e] = res_m[e];
                pa_out[tlo + e] = res_p[e];
            }
        __syncthreads();
    }
}

}  // namespace bh

int bh_k_crf_viterbi(const void* scores, int N, int T, int state_len, int layout_5s, float blank_score,
                     long s_n, long s_t, void* bp_ws, float* alpha_ws, int8_t* moves, int8_t* path,
                     float* best_score, hipStream_t stream) {
    using namespace bh;
    (void)alpha_ws;
    BH_REQUIRE(state_len >= 1 && state_len <= 5, "viterbi: state_len must be in 1..5 (got %d)", state_len);
    BH_REQUIRE(N > 0 && T > 0, "viterbi: empty problem N=%d T=%d", N, T);
    int S = 1;
    for (int i = 0; i < state_len; ++i) S *= 4;
    BH_REQUIRE(layout_5s || (s_t % 4 == 0 && s_n % 4 == 0), "viterbi: 4S layout needs strides %% 4 == 0");
    VitArgs a{(const half_t*)scores, N, T, S, blank_score, s_n, s_t, (uint8_t*)bp_ws, moves, path, best_score};
    // koi layout, 64 states and more, rows 16-byte aligned: four states per thread (round 5); "viterbi_quad" 0 = the round-1 kernel
    if (!layout_5s && state_len >= 3 && g_viterbi_quad && s_t % 8 == 0 && s_n % 8 == 0 && ((uintptr_t)scores & 15) == 0) {
#define BH_VIT_QUAD(SL)                                                                                                        \
    {                                                                                                                          \
        using G = VitQuadGeo<SL>;                                                                                              \
        BH_CHECK_HIP(bh_max_lds((const void*)crf_viterbi_quad_kernel<SL>, G::LDS));                                            \
        hipLaunchKernelGGL(crf_viterbi_quad_kernel<SL>, dim3((N + G::CPB - 1) / G::CPB), dim3(256), G::LDS, stream, a);        \
    }
        if (state_len == 3) BH_VIT_QUAD(3) else if (state_len == 4) BH_VIT_QUAD(4) else BH_VIT_QUAD(5)
#undef BH_VIT_QUAD
        BH_CHECK_HIP(hipGetLastError());
        return 0;
    }
    int threads = S < 64 ? 64 : S;
    int TB = (32 * 1024) / S; if (TB > 512) TB = 512; if (TB < 1) TB = 1;
    size_t lds = 16 + (size_t)2 * S * sizeof(float) + (size_t)TB * S + 2 * TB + 16;
    if (layout_5s) hipLaunchKernelGGL(crf_viterbi_kernel<true>, dim3(N), dim3(threads), lds, stream, a);
    else hipLaunchKernelGGL(crf_viterbi_kernel<false>, dim3(N), dim3(threads), lds, stream, a);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// CTC_CRF.reverse_complement (/root/reference bonito/crf/model.py:84-96) as one gather kernel: the score
// tensor is permuted so that decoding it yields the reverse-complement strand.  A transition into state
// j = (d1..dk) that dropped base r is the (k+1)-mer (r, d1..dk); on the other strand it becomes the
// transition into j' = comp(d_{k-1}, .., d1, r) dropping r' = comp(dk), at time T-1-t.  Stay scores move
// to the reverse-complemented state.  Works on the koi layout [N][T][4S] and on [T][N][5S].
namespace bh {

struct RcArgs {
    const half_t* in;
    half_t* out;
    int N, T, S, k, five;
    long s_n, s_t;       // element strides of both tensors
};

__device__ __forceinline__ int rc_digits(int j, int k) {   // reverse the k base-4 digits and complement them
    int r = 0;
    for (int i = 0; i < k; ++i) { r = (r << 2) | (3 - (j & 3)); j >>= 2; }
    return r;
}

__global__ void crf_revcomp_kernel(RcArgs p) {
    const int W = p.five ? 5 * p.S : 4 * p.S;
    const long total = (long)p.N * p.T * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c2 = (int)(i % W);
        const long nt = i / W;
        const int t2 = (int)(nt % p.T), n = (int)(nt / p.T);
        const int per = p.five ? 5 : 4;
        const int j2 = c2 / per, m = c2 % per;
        int src_c;
        if (p.five && m == 0) {
            src_c = rc_digits(j2, p.k) * 5;                      // stay score of the rev-comp state
        } else {
            const int r2 = p.five ? m - 1 : m;
            // (k+1)-mer on this strand: (r2, digits of j2) ; source (k+1)-mer = its reverse complement
            const int kmer1 = r2 * p.S + j2;                     // r2 is the most significant digit
            const int src = rc_digits(kmer1, p.k + 1);           // = (r, d1..dk) of the source
            const int r = src / p.S, j = src % p.S;
            src_c = p.five ? j * 5 + 1 + r : j * 4 + r;
        }
        p.out[(long)n * p.s_n + (long)t2 * p.s_t + c2] = p.in[(long)n * p.s_n + (long)(p.T - 1 - t2) * p.s_t + src_c];
    }
}

}  // namespace bh

int bh_k_crf_revcomp(const void* in, void* out, int N, int T, int state_len, int layout_5s, long s_n, long s_t,
                     hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(state_len >= 1 && state_len <= 5 && N > 0 && T > 0, "revcomp: bad shape");
    BH_REQUIRE(in != out, "revcomp: in-place operation is not supported");
    int S = 1;
    for (int i = 0; i < state_len; ++i) S *= 4;
    RcArgs a{(const half_t*)in, (half_t*)out, N, T, S, state_len, layout_5s, s_n, s_t};
    const long total = (long)N * T * (layout_5s ? 5 : 4) * S;
    int blocks = (int)std::min<long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(crf_revcomp_kernel, dim3(blocks), dim3(256), 0, stream, a);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}
