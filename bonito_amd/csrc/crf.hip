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
// One workgroup per chunk, one thread per state; alpha ping-pongs in LDS (one barrier per step);
// score loads are register-prefetched U steps ahead so the serial alpha chain never waits on HBM;
// 3-bit back-pointers go to a [N][T][S] byte workspace and are chased back in LDS-staged blocks.
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
