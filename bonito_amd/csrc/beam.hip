// Beam-search CRF decode for gfx950: the MI355X replacement for koi.decode.beam_search
// (call site /root/reference bonito/crf/basecall.py:36-40; koi itself is not in the reference tree).
// Algorithm "BS-1" is defined in DESIGN.md and restated on the CPU in oracle/crf_oracle.c; this file
// implements exactly that definition:
//
//   K1 bs2_backward_kernel      the guide b[N][T+1][S] in the LINEAR domain (deterministic exp, round 5: "BS-2" below)
//   K2 scan wave of K3 / bs2_forward_post_kernel   class posteriors P[N][T][4] (linear domain)
//   K3 beam_kernel              beam of <=32 (state, sequence-hash) elements; back-pointers bp[N][T][32]
//   K4 beam_finalize_kernel     traceback, sequence / moves / q-string
//
// All Log-semiring arithmetic that decides the decoded sequence (K1, K3) uses lse2(a,b) = max + table
// interpolation with plain IEEE fp32 ops (include/bh_lse_table.h, shared with the oracle), so sequences
// and move tables are bit-identical to the CPU oracle; only the q-scores (K2, expf) are tolerance-level.
//
// Mapping: K1/K2 one workgroup per chunk, one thread per k-mer state, alpha/beta ping-pong in LDS with
// one barrier per step, score rows prefetched in registers; the scans are HBM-bound streaming reads of
// the score tensor (2*C/stride bytes per signal sample each). K3 one wave per chunk, wave-synchronous,
// score / guide rows staged through LDS in blocks of 8 steps, top-W selection by a 32-step radix select
// on ballots (no sort), slots assigned by prefix popcount.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include <cstring>
#include "../../include/bh_lse_table.h"

namespace bh {

__device__ const float g_lse_tab[BH_LSE_TABLE_SIZE] = {BH_LSE_TABLE_VALUES};

// deterministic log(exp(a)+exp(b)); `tab` may point to LDS or global memory
__device__ __forceinline__ float lse2_tab(float a, float b, const float* tab) {
    float m = fmaxf(a, b);
    float d = fabsf(a - b);
    // (m == -inf needs no test of its own: then both are -inf and d is NaN, or one is and d is +inf - neither is < RANGE)
    if (!(d < BH_LSE_RANGE)) return m;
    float x = d * BH_LSE_SCALE;
    int i = (int)x;
    float f = x - (float)i;
    float t0 = tab[i];
    float sp = __fmaf_rn(f, tab[i + 1] - t0, t0);
    return m + sp;
}

// lse2_tab without control flow (same arithmetic): safe for garbage `b` as long as the result is discarded
__device__ __forceinline__ float lse2_tab_nb(float a, float b, const float* tab) {
    const float m = fmaxf(a, b);
    const float d = fabsf(a - b);
    const bool plain = !(d < BH_LSE_RANGE);          // covers m == -inf (d is NaN or +inf then)
    const float x = d * BH_LSE_SCALE;
    // d < RANGE: x < RANGE * SCALE = TABLE_SIZE - 2, the clamp changes nothing. Otherwise (d >= RANGE, +inf or NaN) the index only
    // has to stay inside the table: fminf returns the bound for NaN, d is never negative
    const int i = (int)fminf(x, (float)(BH_LSE_TABLE_SIZE - 2));
    const float f = x - (float)i;
    const float t0 = tab[i];
    const float sp = __fmaf_rn(f, tab[i + 1] - t0, t0);
    return plain ? m : m + sp;
}

// The scans call the branch-free form (round 4): with the early return every one of the 16-20 LSEs of a lane and step sat in its own
// exec-masked branch (s_and_saveexec / s_cbranch_execz / s_or, an lgkmcnt(0) in front of each), which also kept the four independent
// state chains of a lane from overlapping their table reads. Same operations on the same values: bit-identical rows.
#ifndef BH_LSE_BRANCHY
#define lse2_scan lse2_tab_nb
#else
#define lse2_scan lse2_tab
#endif

// wave-wide max without LDS traffic: DPP butterflies inside each row of 16 lanes, then row broadcasts;
// the total ends up in lane 63 (classic GCN reduction, valid on the gfx9 family incl. gfx950).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                                CTRL, ROW_MASK, 0xF, false));
}

// ------------------------------------------------------------------------------------------------
struct ScanArgs {
    const half_t* scores;  // [N][T][4S]
    int N, T, S, state_len;
    float blank;
    float* beta;           // [N][T+1][S]
    double* Bcum;          // [N][T+1]
    double* logZ;          // [N]
    float* P;              // [N][T][4]   (forward only)
    int nt;                // stream scores / guide with the non-temporal cache policy (they are read once; keeps the L2 for the
                           // recurrent kernels' exchange buffers when the decoder runs beside the next batch's encoder)
    int cpb;               // backward scan: chunks per workgroup (backward_geometry)
};

constexpr int SU = 4;  // score prefetch depth

__global__ void crf_backward_kernel(ScanArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Several chunks per workgroup (`p.cpb`, threads = cpb * max(S, 64)) share the 16 KiB lse table: at 256 states a chunk then takes
    // 16.4 / 2 + 6 = 14 KiB of LDS instead of 22.5, eight chunks fit a CU instead of six (the 32-wave limit), and the 2048 chunks of a
    // bench call run in ONE round on 256 CUs instead of a full one and a third-full one. The chunks of a workgroup run the same number
    // of steps, so they share the per-step barrier.
    const int S = p.S, T = p.T;
    const int tpc = S < 64 ? 64 : S;           // threads per chunk
    const int slot = threadIdx.x / tpc, s = threadIdx.x - slot * tpc;
    const int n = blockIdx.x * p.cpb + slot;
    float* tab = (float*)smem;                 // BH_LSE_TABLE_SIZE
    float* buf = (float*)(smem + ((BH_LSE_TABLE_SIZE * 4 + 15) & ~15) + (size_t)slot * 24 * S);   // [2][S], 16-byte aligned (vector reads below)
    half_t* rows = (half_t*)(buf + 2 * S);     // [2][4S] score rows, double buffered
    const bool active = s < S && n < p.N;
    for (int i = threadIdx.x; i < BH_LSE_TABLE_SIZE; i += blockDim.x) tab[i] = g_lse_tab[i];
    if (active) buf[s] = 0.0f;
    const int lead = s >> (2 * (p.state_len - 1));
    const bool vec = S >= 16;                  // (S = 4: a row is 16 halves in all, the scalar gathers stay)
    const int sm = (s & ((S >> 2) - 1));       // successors of s are the states sm*4 + x
    const int nn = n < p.N ? n : p.N - 1;      // (a slot beyond the batch only keeps the barriers company)
    const half_t* sc = p.scores + (long)nn * T * 4 * S + s * 4;   // this thread stages halves [4s, 4s+4) of a row
    float* bn = p.beta + (long)nn * (T + 1) * S;
    double* Bn = p.Bcum + (long)nn * (T + 1);
    if (active) bn[(long)T * S + s] = 0.0f;
    if (s == 0 && n < p.N) Bn[T] = 0.0;

    // register prefetch ring: pre[u] holds this thread's 8 bytes of row (thi - 1 - u)
    half4_t cur[SU], nxt[SU];
    auto load = [&](half4_t (&dst)[SU], int thi) {
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            int t = thi - 1 - u;
            if (active && t >= 0) dst[u] = p.nt ? __builtin_nontemporal_load((const half4_t*)(sc + (long)t * 4 * S))
                                                : *(const half4_t*)(sc + (long)t * 4 * S);
        }
    };
    load(cur, T);
    if (active && T > 0) *(half4_t*)(rows + ((T - 1) & 1) * 4 * S + s * 4) = cur[0];
    __syncthreads();

    double cum = 0.0;
    int cb = 0;
    for (int thi = T; thi > 0; thi -= SU) {
        load(nxt, thi - SU);
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int t = thi - 1 - u;
            if (t >= 0) {
                const float* prev = buf + cb * S;
                const half_t* row = rows + (t & 1) * 4 * S;
                const float ref = prev[0];
                if (active) {
                    float acc = p.blank + (prev[s] - ref);
                    if (vec) {
                        // The four successors of state s are the states sm*4 .. sm*4+3: their guide values are 16 contiguous bytes
                        // and their 16 transition scores 32 contiguous bytes of the row. Read as vectors and pick half `lead` of
                        // each group of four in registers - the per-element gathers (prev[sm*4+x]: stride 16 bytes across lanes,
                        // row[(sm*4+x)*4+lead]: 2-byte reads at stride 32 bytes) were 4- and 8-way bank conflicts: the kernel spent
                        // 89 % of its CU cycles in LDS conflict cycles (SQ_LDS_BANK_CONFLICT, profiles/r03a_sq_counters.txt).
                        // Same values, same order of operations: the same bits.
                        const float4_t pv = *(const float4_t*)(prev + sm * 4);
                        const uint4_t r0 = *(const uint4_t*)(row + sm * 16), r1 = *(const uint4_t*)(row + sm * 16 + 8);
                        const bool hi = (lead & 2) != 0;
                        const unsigned sh16 = (lead & 1) * 16;
                        const unsigned w[4] = {hi ? r0.y : r0.x, hi ? r0.w : r0.z, hi ? r1.y : r1.x, hi ? r1.w : r1.z};
                        const float pvx[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const half_t m = __builtin_bit_cast(half_t, (unsigned short)(w[x] >> sh16));
                            acc = lse2_scan(acc, (float)m + (pvx[x] - ref), tab);
                        }
                    } else {
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const int s2 = sm * 4 + x;
                            acc = lse2_scan(acc, (float)row[s2 * 4 + lead] + (prev[s2] - ref), tab);
                        }
                    }
                    buf[(cb ^ 1) * S + s] = acc;
                    // stage the next (earlier) row for the following step
                    if (t > 0) {
                        const half4_t v = (u + 1 < SU) ? cur[(u + 1) % SU] : nxt[0];
                        *(half4_t*)(rows + ((t - 1) & 1) * 4 * S + s * 4) = v;
                    }
                }
                if (s == 0 && n < p.N) { cum += (double)ref; Bn[t] = cum; }
                cb ^= 1;
                __syncthreads();
                if (active) {
                    const float* now = buf + cb * S;
                    if (p.nt) __builtin_nontemporal_store(now[s] - now[0], bn + (long)t * S + s);
                    else bn[(long)t * S + s] = now[s] - now[0];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) cur[u] = nxt[u];
    }
    // logZ = B_0 + raw_0[0] + LSE_s beta~_0[s]   (alpha_0 = 0); once per chunk, serial in double
    if (s == 0 && n < p.N) {
        const float* now = buf + cb * S;
        double m = -INFINITY, sum = 0.0;
        for (int i = 0; i < S; ++i) m = fmax(m, (double)(now[i] - now[0]));
        for (int i = 0; i < S; ++i) sum += exp((double)(now[i] - now[0]) - m);
        p.logZ[n] = cum + (double)now[0] + m + log(sum);
    }
}

// chunks per workgroup / threads / LDS bytes of crf_backward_kernel for S states
static void backward_geometry(int S, int N, int& cpb, int& threads, size_t& lds) {
    const int tpc = S < 64 ? 64 : S;
    cpb = tpc >= 1024 ? 1 : tpc >= 256 ? 2 : 4;          // workgroups of at most 512 threads (1024 at 1024 states)
    if (cpb > N) cpb = N;
    threads = cpb * tpc;
    lds = (size_t)((BH_LSE_TABLE_SIZE * 4 + 15) & ~15) + (size_t)cpb * 24 * S + 64;
}

// ------------------------------------------------------------------------------------------------
// Posterior decoding = SeqdistModel.decode_batch (/root/reference bonito/crf/model.py:196-199):
//     post = posteriors(scores) + 1e-8 ; path = viterbi(log post)
// i.e. the Max-semiring best path over the LOG EDGE POSTERIORS
//     log p[t][j][k] = alpha_t[idx[j][k]] + Ms[t][j][k] + beta_{t+1}[j] - logZ .
// One forward pass carries both recurrences: the log-semiring alpha (table lse2, as K2) and, on the fly,
// v_{t+1}[j] = max_k log(p[t][j][k] + 1e-8) + v_t[idx[j][k]] with 3-bit back-pointers; then the same
// LDS-staged traceback as the plain Viterbi kernel. Needs beta~/B/logZ from crf_backward_kernel.
struct PostVitArgs {
    ScanArgs sc;
    uint8_t* bp;      // [N][T][S]
    int8_t* moves;    // [N][T]
    int8_t* path;     // [N][T]
};

__global__ void crf_posterior_viterbi_kernel(PostVitArgs pa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ScanArgs& p = pa.sc;
    const int S = p.S, T = p.T, q = S >> 2;
    float* tab = (float*)smem;
    float* buf = tab + BH_LSE_TABLE_SIZE + 2;   // [2][S] alpha raw
    float* vb = buf + 2 * S;                    // [2][S] max-semiring scores
    int* s_state = (int*)(vb + 2 * S);          // [4]
    uint8_t* stage = (uint8_t*)(s_state + 4);   // traceback staging
    const int n = blockIdx.x, j = threadIdx.x;
    const bool active = j < S;
    for (int i = threadIdx.x; i < BH_LSE_TABLE_SIZE; i += blockDim.x) tab[i] = g_lse_tab[i];
    if (active) { buf[j] = 0.0f; vb[j] = 0.0f; }
    const half_t* sc = p.scores + (long)n * T * 4 * S + j * 4;
    const float* bn = p.beta + (long)n * (T + 1) * S;
    const double* Bn = p.Bcum + (long)n * (T + 1);
    const double lz = p.logZ[n];
    uint8_t* bp = pa.bp + (long)n * T * S;
    __syncthreads();

    double A = 0.0;
    int cb = 0;
    for (int t = 0; t < T; ++t) {
        const float* prev = buf + cb * S;
        const float* vprev = vb + cb * S;
        const float ref = prev[0];
        if (active) {
            const half4_t m4 = *(const half4_t*)(sc + (long)t * 4 * S);
            const float b1 = bn[(long)(t + 1) * S + j];
            // log edge posterior = alpha_t[src] + Ms + beta_{t+1}[j] - logZ, alpha_t = prev + D_t, beta_{t+1} = b1 + B_t
            const double off = A + (double)b1 + Bn[t] - lz;
            float acc = p.blank + (prev[j] - ref);
            float e0 = (float)((double)prev[j] + (double)p.blank + off);
            float best = __logf(__expf(e0) + 1e-8f) + vprev[j];
            int bk = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int src = r * q + (j >> 2);
                const float ms = (float)m4[r];
                acc = lse2_scan(acc, ms + (prev[src] - ref), tab);
                const float e = (float)((double)prev[src] + (double)ms + off);
                const float cand = __logf(__expf(e) + 1e-8f) + vprev[src];
                if (cand > best) { best = cand; bk = 1 + r; }
            }
            buf[(cb ^ 1) * S + j] = acc;
            vb[(cb ^ 1) * S + j] = best;
            bp[(long)t * S + j] = (uint8_t)bk;
        }
        A += (double)ref;
        cb ^= 1;
        __syncthreads();
    }
    if (j == 0) {
        const float* a = vb + cb * S;
        float best = a[0];
        int bj = 0;
        for (int i = 1; i < S; ++i)
            if (a[i] > best) { best = a[i]; bj = i; }
        s_state[0] = bj;
    }
    __threadfence();
    __syncthreads();
    const int TB = max(1, min(512, (32 * 1024) / S));
    int8_t* res_m = (int8_t*)(stage + TB * S);
    int8_t* res_p = res_m + TB;
    int8_t* mo = pa.moves + (long)n * T;
    int8_t* pth = pa.path + (long)n * T;
    for (int thi = T; thi > 0; thi -= TB) {
        const int tlo = max(0, thi - TB);
        const int nbytes = (thi - tlo) * S;
        const uint8_t* src = bp + (long)tlo * S;
        for (int i = threadIdx.x * 4; i < nbytes; i += blockDim.x * 4) *(unsigned*)(stage + i) = *(const unsigned*)(src + i);
        __syncthreads();
        if (j == 0) {
            int st = s_state[0];
            for (int t = thi - 1; t >= tlo; --t) {
                const int k = stage[(t - tlo) * S + st];
                res_m[t - tlo] = (int8_t)(k != 0);
                res_p[t - tlo] = (int8_t)(k != 0 ? 1 + (st & 3) : 0);
                if (k != 0) st = (k - 1) * q + (st >> 2);
            }
            s_state[0] = st;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < thi - tlo; i += blockDim.x) { mo[tlo + i] = res_m[i]; pth[tlo + i] = res_p[i]; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
struct BeamArgs {
    const half_t* scores;  // [N][T][4S]
    const float* beta;     // [N][T+1][S] the LINEAR guide b of bs2_backward_kernel
    int N, T, S, state_len, W;
    float blank, cut;      // cut = log(beam_cut)
    uint8_t* bp;           // [N][T][32]  parent | move << 5 | base << 6
    int* final_slot;       // [N]
    long long* dbg;        // optional [N][8] per-section cycle counters (BH_BEAM_DEBUG)
    float inv_bin;         // 256 / cut: selection histogram bins per unit of key; 0 puts every key into one bin, which
                           // turns the selection into the plain radix search (bh_set_option("beam_select", 1))
    float* P;              // [N][T][4] class posteriors, written by the fused scan wave (FUSE instantiations)
    int nt;                // non-temporal staging of scores / guide
};

// Steps per staged block of the beam kernel; two blocks are resident (the next one streams in under the current one). Round 2 went from four
// to two (12 KiB less LDS per chunk at 256 states, decode 4.35 -> 3.65 ms per hac batch); round 5: ONE at 256 states and above - a step of
// a chunk is ~5 k cycles, enough to hide the DMA of the next row, and at 14.1 KiB per chunk eight chunks share a CU (g_beam_cpw): 12.97 ->
// 8.72 ms per 2048-chunk call. Below 256 states it stays two: there the four beam waves of a workgroup and their shared scan wave meet at
// one barrier per block, and a barrier per step costs the fast-sized models' pipeline 12 % (2.35 -> 2.65 ms per batch, measured).
// -DBH_BTB=n forces a depth (experiments).
template <int STATE_LEN>
__host__ __device__ constexpr int beam_btb() {
#ifdef BH_BTB
    return BH_BTB;
#else
    return STATE_LEN >= 4 ? 1 : 2;
#endif
}
constexpr int MAXW = 32;

__device__ __forceinline__ unsigned bs_hash0(int s) { return ((unsigned)s + 1u) * 2654435761u; }
__device__ __forceinline__ unsigned bs_mix(unsigned h, int x) {
    h = (h ^ ((unsigned)x + 1u)) * 16777619u;
    return h ^ (h >> 15);
}
__device__ __forceinline__ unsigned bs_ukey(float k) {
    unsigned u = __float_as_uint(k);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// Select the `want` largest ukeys among 3 per lane (0 = dead); ties at the threshold by ascending
// candidate index c = lane + 64*i. Returns per-candidate selected flags and slot numbers
// (slots in candidate-index order). MSB-first radix search on ballots, two bits per round, with an
// early exit as soon as some prefix isolates exactly `want` keys.
__device__ __forceinline__ int count_ge(const unsigned (&uk)[3], unsigned trial) {
    return popc64(__ballot(uk[0] >= trial)) + popc64(__ballot(uk[1] >= trial)) + popc64(__ballot(uk[2] >= trial));
}
__device__ __forceinline__ int radix_select(const unsigned (&uk)[3], int want, int lane, bool (&sel)[3], int (&slot)[3],
                                            unsigned ulo, unsigned uhi) {
    unsigned long long selm[3];
    int n_alive = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { selm[i] = __ballot(uk[i] != 0u); n_alive += popc64(selm[i]); }
    if (n_alive > want) {
        // every live key lies in [ulo, uhi]: their common leading bits are already known
        const int common = __clz((int)(ulo ^ uhi)) & ~1;            // even number of shared leading bits (32 if equal)
        unsigned prefix = common >= 32 ? uhi : (uhi & ~(0xffffffffu >> common));
        bool exact = false;
        for (int bit = 30 - common; bit >= 0; bit -= 2) {
            const unsigned t1 = prefix | (1u << bit), t2 = prefix | (2u << bit), t3 = prefix | (3u << bit);
            const int c1 = count_ge(uk, t1), c2 = count_ge(uk, t2), c3 = count_ge(uk, t3);
            int cnt = -1;
            if (c3 >= want) { prefix = t3; cnt = c3; }
            else if (c2 >= want) { prefix = t2; cnt = c2; }
            else if (c1 >= want) { prefix = t1; cnt = c1; }
            if (cnt == want) { exact = true; break; }
        }
        // prefix is now either an exact separator (exact) or the want-th largest key value
        unsigned long long gt[3], eq[3];
        int n_gt = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            gt[i] = exact ? __ballot(uk[i] >= prefix) : __ballot(uk[i] > prefix);
            eq[i] = exact ? 0ull : __ballot(uk[i] == prefix);
            n_gt += popc64(gt[i]);
        }
        const int need = want - n_gt;    // ties to take, in index order
        int tie_before = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int my_rank = tie_before + popc64(eq[i] & lanemask_lt(lane));
            const bool tie_ok = ((eq[i] >> lane) & 1ull) && my_rank < need;
            const bool s_ = ((gt[i] >> lane) & 1ull) || tie_ok;
            tie_before += popc64(eq[i]);
            selm[i] = __ballot(s_);
        }
    }
    int before = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        sel[i] = (selm[i] >> lane) & 1ull;
        slot[i] = before + popc64(selm[i] & lanemask_lt(lane));
        before += popc64(selm[i]);
    }
    return before;
}

__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_f<0xB1, 0xF>(v));     // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f<0x4E, 0xF>(v));     // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f<0x124, 0xF>(v));    // row_ror:4
    v = fmaxf(v, dpp_f<0x128, 0xF>(v));    // row_ror:8   -> every lane holds its row's max
    v = fmaxf(v, dpp_f<0x142, 0xA>(v));    // row_bcast15 into rows 1 and 3
    v = fmaxf(v, dpp_f<0x143, 0xC>(v));    // row_bcast31 into rows 2 and 3 -> lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_max_pos(float v) { return wave_max_f32(v); }

template <int CTRL>
__device__ __forceinline__ int dpp_i0(int v) {       // DPP move; lanes whose source falls outside the row read 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16 (row_shr), then the three row totals
__device__ __forceinline__ int wave_scan_add(int v, int lane) {
    v += dpp_i0<0x111>(v);     // row_shr:1
    v += dpp_i0<0x112>(v);     // row_shr:2
    v += dpp_i0<0x114>(v);     // row_shr:4
    v += dpp_i0<0x118>(v);     // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    return v + (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, dpp_u<0xB1, 0xF>(v));
    v = max(v, dpp_u<0x4E, 0xF>(v));
    v = max(v, dpp_u<0x124, 0xF>(v));
    v = max(v, dpp_u<0x128, 0xF>(v));
    v = max(v, dpp_u<0x142, 0xA>(v));
    v = max(v, dpp_u<0x143, 0xC>(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Same contract as radix_select (the `want` largest keys, ties by ascending candidate index), for keys known to lie
// in [thr, thr + 256 / inv_bin]: one 256-bin histogram pass (LDS atomics, four bins per lane, a wave prefix sum) finds
// the bin that holds the want-th key; the few candidates in that bin are ranked exactly by repeated wave maxima. A
// crowded boundary bin (many equal or near-equal keys) falls back to the radix search on that bin alone.
// One wave per workgroup: LDS operations complete in issue order, no barrier is needed between the phases.
constexpr int HBINS = 256;
__device__ __forceinline__ int hist_select(const float (&key)[3], const unsigned (&uk)[3], int want, int lane, float thr,
                                           float inv_bin, int* hist, bool (&sel)[3], int (&slot)[3], unsigned ulo, unsigned uhi) {
    unsigned long long selm[3];
    int n_alive = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { selm[i] = __ballot(uk[i] != 0u); n_alive += popc64(selm[i]); }
    if (n_alive > want) {
        int bin[3];                                     // (the histogram arrives zeroed and is handed back zeroed)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bin[i] = min(HBINS - 1, (int)((key[i] - thr) * inv_bin));
            if (uk[i] != 0u) atomicAdd(&hist[HBINS - 1 - bin[i]], 1);
        }
        const uint4_t h4 = *(const uint4_t*)(hist + 4 * lane);    // lane l: bins 255-4l, 254-4l, 253-4l, 252-4l
        *(uint4_t*)(hist + 4 * lane) = uint4_t{0u, 0u, 0u, 0u};
        const int c0 = (int)h4.x, c1 = c0 + (int)h4.y, c2 = c1 + (int)h4.z, c3 = c2 + (int)h4.w;
        const int inc = wave_scan_add(c3, lane);        // candidates in bins >= 252 - 4l
        const unsigned long long ge = __ballot(inc >= want);
        const int ls = __ffsll((long long)ge) - 1;      // exists: inc[63] = n_alive > want
        const int e = __builtin_amdgcn_readlane(inc - c3, ls);     // candidates in the bins above lane ls's four
        const int g0 = __builtin_amdgcn_readlane((int)h4.x, ls), g1 = __builtin_amdgcn_readlane((int)h4.y, ls);
        const int g2 = __builtin_amdgcn_readlane((int)h4.z, ls), g3 = __builtin_amdgcn_readlane((int)h4.w, ls);
        int k, n_hi, m;
        if (e + g0 >= want) { k = 0; n_hi = e; m = g0; }
        else if (e + g0 + g1 >= want) { k = 1; n_hi = e + g0; m = g1; }
        else if (e + g0 + g1 + g2 >= want) { k = 2; n_hi = e + g0 + g1; m = g2; }
        else { k = 3; n_hi = e + g0 + g1 + g2; m = g3; }
        const int need = want - n_hi;                   // 1 <= need <= m
        const int bstar = HBINS - 1 - (4 * ls + k);
        bool hi[3], bnd[3], selb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            hi[i] = uk[i] != 0u && bin[i] > bstar;
            bnd[i] = uk[i] != 0u && bin[i] == bstar;
            selb[i] = bnd[i];
        }
        if (m > need) {
            if (m <= 8) {
#pragma unroll
                for (int i = 0; i < 3; ++i) selb[i] = false;
                int taken = 0;
                while (taken < need) {
                    unsigned mine = 0u;
#pragma unroll
                    for (int i = 0; i < 3; ++i) mine = max(mine, (bnd[i] && !selb[i]) ? uk[i] : 0u);
                    const unsigned top = wave_max_u32(mine);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const unsigned long long mm = __ballot(bnd[i] && !selb[i] && uk[i] == top);
                        const int room = need - taken;
                        const bool take = ((mm >> lane) & 1ull) && popc64(mm & lanemask_lt(lane)) < room;
                        selb[i] = selb[i] || take;
                        taken += min(popc64(mm), room);
                    }
                }
            } else {
                unsigned ukb[3];
                int dummy[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) ukb[i] = bnd[i] ? uk[i] : 0u;
                radix_select(ukb, need, lane, selb, dummy, ulo, uhi);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) selm[i] = __ballot(hi[i] || selb[i]);
    }
    int before = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        sel[i] = (selm[i] >> lane) & 1ull;
        slot[i] = before + popc64(selm[i] & lanemask_lt(lane));
        before += popc64(selm[i]);
    }
    return before;
}

// Stay elements keyed by sequence hash: NBK buckets of BKE entries (hash, step tag << 15 | state << 5 | slot), filled
// with one LDS atomic per element at the head of the step (by the lane that owns the element's stay); a move candidate reads its whole bucket with one 16-byte load
// and compares in registers - one LDS round trip and no divergent probe loop. Entries of earlier steps carry another
// tag and never match, so nothing is cleared but the bucket fill counters. A third element in a bucket (a few percent
// of the steps) goes to an overflow list that every lookup then scans. Ties between equal (hash, state) stays resolve to the lowest slot, like
// the oracle's first-match scan.
constexpr int NBK = 256;
constexpr int BKE = 2;      // entries per bucket
constexpr unsigned TAG_MASK = 0x1ffffu;     // 17 tag bits above 10 state bits and 5 slot bits

struct BeamTable {
    int* cnt;          // [NBK] elements hashed to the bucket this step (may exceed 4)
    uint2_t* ent;      // [NBK][BKE]
    int* ov_cnt;       // [1] (+3 pad)
    uint2_t* ov;       // [MAXW]
};

// slot (0..31) of the lowest stay element matching (hash, want = tag | state << 5), or >= 32
__device__ __forceinline__ unsigned bucket_match(const uint4_t& e0, unsigned hash, unsigned want) {
    unsigned d = 32u;
    d = min(d, e0.x == hash ? e0.y - want : 32u);
    d = min(d, e0.z == hash ? e0.w - want : 32u);
    return d;
}

// One 1 KiB global -> LDS DMA: lane l moves 16 bytes from g (per lane) to lds + 16 l (lds is wave-uniform).
__device__ __forceinline__ void dma16(const char* g, char* lds) {
    const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory");
}
__device__ __forceinline__ void dma16_nt(const char* g, char* lds) {      // same, non-temporal (streamed-once data)
    const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(l), "v"(g) : "memory");
}

// =================================================================================================================================
// BS-2 (round 5): the guide and the posterior scan in the LINEAR domain.
//
// The table-lse2 scans of rounds 1-4 were chains: four dependent lse2 per state and step, each with two dependent LDS table reads at
// random addresses (SQ counters of round 5: the backward scan waited 51 % of its wave cycles, its LDS bank-conflict cycles were 2.6 x
// the cycles its LDS instructions were active). In the linear domain a step is five multiply-adds per state on exponentials that do not
// depend on the recurrence, no table, no chain. What the search needs - guide rows that are bit-identical on the GPU and in the CPU
// oracle - is kept by evaluating the exponential (and the logarithm the beam wave takes of the guide values it ranks with) with plain
// IEEE operations only: polynomials by fmaf and integer arithmetic on the exponent bits (include/bh_bs2.h, oracle/crf_oracle.c
// oracle_bs2_exp / oracle_bs2_log / oracle_bs2_backward: the same operations in the same order). Packed fp32 instructions
// (v_pk_fma_f32, two IEEE fmas per lane) carry the polynomial.
//
// State <-> lane mapping of the backward scan: lane m owns the four states {m + lead * S/4}, i.e. the four k-mers that differ in their
// LEADING base. Their successors are the same four states 4m .. 4m+3, and the 16 transition scores they need are the 32 contiguous
// bytes row[16m .. 16m+15] - every byte a lane reads is used, reads are linear across the wave (no bank conflicts), and a half needs no
// extraction beyond the conversion. (With one state per thread, as before, a thread picked one half out of every eight bytes.)
// =================================================================================================================================
#include "../../include/bh_bs2.h"
typedef float float2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2_t fma2(float2_t a, float2_t b, float2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float2_t splat2(float v) { return float2_t{v, v}; }

// e^x for two values: oracle_bs2_exp, operation for operation (v_med3_f32, v_pk_fma_f32, v_pk_add_f32, v_lshl_add_u32)
__device__ __forceinline__ float2_t bs2_exp2(float2_t x) {
    x.x = __builtin_amdgcn_fmed3f(x.x, -BH_BS2_XMAX, BH_BS2_XMAX);
    x.y = __builtin_amdgcn_fmed3f(x.y, -BH_BS2_XMAX, BH_BS2_XMAX);
    const float2_t t = fma2(x, splat2(BH_BS2_LOG2E), splat2(BH_BS2_MAGIC));
    const float2_t n = t - splat2(BH_BS2_MAGIC);
    const float2_t f = fma2(x, splat2(BH_BS2_LOG2E), -n);
    float2_t p = splat2(BH_BS2_E5);
    p = fma2(p, f, splat2(BH_BS2_E4));
    p = fma2(p, f, splat2(BH_BS2_E3));
    p = fma2(p, f, splat2(BH_BS2_E2));
    p = fma2(p, f, splat2(BH_BS2_E1));
    p = fma2(p, f, splat2(BH_BS2_E0));
    // 2^n by adding n to the exponent field: n sits in the low mantissa bits of t, and the low nine bits of the magic constant are zero,
    // so (bits(t) << 23) IS n << 23 (mod 2^32)
    float2_t r;
    r.x = __uint_as_float(__float_as_uint(p.x) + (__float_as_uint(t.x) << 23));
    r.y = __uint_as_float(__float_as_uint(p.y) + (__float_as_uint(t.y) << 23));
    return r;
}
__device__ __forceinline__ float bs2_exp1(float x) { return bs2_exp2(float2_t{x, x}).x; }

// ln v of a positive normal fp32: oracle_bs2_log, operation for operation (eleven instructions)
__device__ __forceinline__ float bs2_log(float v) {
    const unsigned u = __float_as_uint(v);
    const float ef = (float)(int)(u >> 23);
    const float z = __uint_as_float((u & 0x007fffffu) | 0x3f800000u) - 1.0f;
    float q = BH_BS2_L5;
    q = __fmaf_rn(q, z, BH_BS2_L4);
    q = __fmaf_rn(q, z, BH_BS2_L3);
    q = __fmaf_rn(q, z, BH_BS2_L2);
    q = __fmaf_rn(q, z, BH_BS2_L1);
    q = __fmaf_rn(q, z, BH_BS2_L0);
    return __fmaf_rn(ef, BH_BS2_LN2, __fmaf_rn(z, q, BH_BS2_LOGC));
}

struct Bs2Args {
    const half_t* scores;  // [N][T][4S]
    float* b;              // [N][T+1][S] the linear-domain guide, every row scaled by a power of two (maximum in [1, 2))
    int N, T;
    float blank;
    int nt;
};

constexpr int BS2_R = 4;       // score rows per staged block of the backward scan (two blocks resident)

// max / sum over the GL lanes of a lane group (GL = 64: the wave, 16: a DPP row, 4: a quad, 1: the lane itself); every lane of the group
// ends up with the result
template <int GL>
__device__ __forceinline__ float group_max(float v) {
    if constexpr (GL >= 4) {
        v = fmaxf(v, dpp_f<0xB1, 0xF>(v));     // quad_perm [1,0,3,2]
        v = fmaxf(v, dpp_f<0x4E, 0xF>(v));     // quad_perm [2,3,0,1]
    }
    if constexpr (GL >= 16) {
        v = fmaxf(v, dpp_f<0x124, 0xF>(v));    // row_ror:4
        v = fmaxf(v, dpp_f<0x128, 0xF>(v));    // row_ror:8
    }
    if constexpr (GL == 64) {
        v = fmaxf(v, dpp_f<0x142, 0xA>(v));    // row_bcast15 into rows 1 and 3
        v = fmaxf(v, dpp_f<0x143, 0xC>(v));    // row_bcast31 into rows 2 and 3 -> lane 63 holds the total
        v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    }
    return v;
}
template <int GL>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (GL >= 4) {
        v += dpp_f<0xB1, 0xF>(v);
        v += dpp_f<0x4E, 0xF>(v);
    }
    if constexpr (GL >= 16) {
        v += dpp_f<0x124, 0xF>(v);
        v += dpp_f<0x128, 0xF>(v);
    }
    if constexpr (GL == 64) {
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 15));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 47));
        const float r4 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
        v = (r1 + r2) + (r3 + r4);
    }
    return v;
}

// Geometry of the backward scan. A chunk needs Q = S/4 lanes. Up to 256 states that is at most one wave, and for the small state spaces
// SEVERAL CHUNKS SHARE A WAVE (16 / 4 / 1 lanes each at 64 / 16 / 4 states): the instruction stream of a step then serves 4 / 16 / 64
// chunks - with one chunk per wave a 64-state scan cost as many instructions as a 256-state one. At 1024 states a chunk is four waves
// and one barrier per step. A "unit" is what runs in lockstep: the chunks of one wave, or the waves of one chunk.
template <int STATE_LEN>
struct Bs2Geo {
    static constexpr int S = 1 << (2 * STATE_LEN), Q = S / 4;
    static constexpr int GL = Q < 64 ? Q : 64;           // lanes per chunk inside a wave
    static constexpr int CPWV = 64 / GL;                 // chunks per wave
    static constexpr int WPC = Q <= 64 ? 1 : Q / 64;     // waves per chunk
    static constexpr int UT = WPC * 64;                  // threads per unit
    static constexpr int UPB = CPWV > 1 ? 1 : 256 / UT;  // units per workgroup: 256 threads, but ONE wave where a wave already holds several chunks
                                                         // (fast-sized models run three batch lanes: small workgroups find room beside the encoder)
    static constexpr int CPB = UPB * CPWV;               // chunks per workgroup
    static constexpr int ROW = 4 * S * 2;                // bytes of a score row
    static constexpr int PP = CPWV * S / 2;              // 16-byte pieces of one row of every chunk of a unit (128, or 512 at 1024 states)
    static constexpr int UNIT_LDS = 2 * BS2_R * CPWV * ROW + CPWV * 2 * S * 4 + 64;      // staged rows, b ping-pong per chunk, wave maxima
};

template <int STATE_LEN>
__global__ __launch_bounds__(256) void bs2_backward_kernel(Bs2Args p) {
    using G = Bs2Geo<STATE_LEN>;
    constexpr int S = G::S, Q = G::Q, WPC = G::WPC, GL = G::GL, CPWV = G::CPWV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = p.T;
    const int unit = threadIdx.x / G::UT, tu = threadIdx.x - unit * G::UT;
    const int lane = threadIdx.x & 63, wv = tu >> 6;
    const int c = WPC == 1 ? lane / GL : 0;               // chunk inside the wave
    const int m = WPC == 1 ? lane - c * GL : tu;          // the lane owns the states m + lead * Q
    const int n0 = (blockIdx.x * G::UPB + unit) * CPWV;   // first chunk of the unit
    const int n = n0 + c;
    const bool live = n < p.N;
    const int nn = live ? n : p.N - 1;                    // (a slot beyond the batch repeats the last chunk's reads and stores nothing)
    char* mine = smem + (size_t)unit * G::UNIT_LDS;
    half_t* rows = (half_t*)mine;                         // [2][BS2_R][CPWV][4S]
    float* pb = (float*)(mine + 2 * BS2_R * CPWV * G::ROW) + (size_t)c * 2 * S;      // this chunk's [2][S]
    float* mxs = (float*)(mine + 2 * BS2_R * CPWV * G::ROW) + (size_t)CPWV * 2 * S;  // [2][4] per-wave maxima (WPC > 1)
    float* bn = p.b + (long)nn * (T + 1) * S;
    const float eb = bs2_exp1(p.blank);

    // step u = 0 .. T-1 handles row t = T-1-u; block k = steps k*R .. k*R+R-1 in buffer k & 1, slot r = u - k*R.
    // One row of all chunks of the unit is PP 16-byte pieces, laid out [chunk][piece] = in piece order: lane l of a DMA moves piece
    // q = i0 + 64 wv + l, which belongs to chunk q / (S/2).
    auto stage = [&](int k) {
        const int u0 = k * BS2_R, nr = min(BS2_R, T - u0);
        for (int r = 0; r < nr; ++r) {
            char* dst = (char*)(rows + (size_t)((k & 1) * BS2_R + r) * CPWV * 4 * S);
            const long trow = (long)(T - 1 - (u0 + r)) * 4 * S;
#pragma unroll
            for (int i0 = 0; i0 < G::PP; i0 += G::UT) {
                const int q = i0 + wv * 64 + lane;
                const int qc = q / (S / 2), qp = q - qc * (S / 2);
                const int qn = min(n0 + qc, p.N - 1);
                const char* src = (const char*)(p.scores + (long)qn * T * 4 * S + trow) + (long)qp * 16;
                if (p.nt) dma16_nt(src, dst + (size_t)(i0 + wv * 64) * 16);
                else dma16(src, dst + (size_t)(i0 + wv * 64) * 16);
            }
        }
    };
    // b_T = 1 (its maximum is 1: scale 2^0)
#pragma unroll
    for (int l = 0; l < 4; ++l) pb[m + l * Q] = 1.0f;
    if (WPC > 1 && tu < 4) mxs[tu] = 1.0f;
    if (WPC == 1 && live) {
#pragma unroll
        for (int l = 0; l < 4; ++l) bn[(long)T * S + m + l * Q] = 1.0f;
    }
    if (T > 0) stage(0);
    int cur = 0;
    const int nblk = (T + BS2_R - 1) / BS2_R;
    for (int k = 0; k < nblk; ++k) {
        const int u0 = k * BS2_R, nr = min(BS2_R, T - u0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // block k has landed
        if (WPC > 1) __syncthreads();                               // ... for every wave of the chunk, and block k-1 is consumed
        if (k + 1 < nblk) stage(k + 1);
        for (int r = 0; r < nr; ++r) {
            const int t = T - 1 - (u0 + r);
            const float* prev = pb + cur * S;
            const half_t* row = rows + ((size_t)((k & 1) * BS2_R + r) * CPWV + c) * 4 * S;
            const uint4_t w0 = *(const uint4_t*)(row + 16 * m);
            const uint4_t w1 = *(const uint4_t*)(row + 16 * m + 8);
            float4_t succ = *(const float4_t*)(prev + 4 * m);
            float own[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) own[l] = prev[m + l * Q];
            if (WPC > 1) {
                // the previous step left its values unscaled: its row maximum is known only behind the barrier
                const float4_t mv = *(const float4_t*)(mxs + cur * 4);
                const float mx = fmaxf(fmaxf(mv.x, mv.y), fmaxf(mv.z, mv.w));
                const unsigned eb23 = ((__float_as_uint(mx) >> 23) - 127u) << 23;
#pragma unroll
                for (int l = 0; l < 4; ++l) own[l] = __uint_as_float(__float_as_uint(own[l]) - eb23);
#pragma unroll
                for (int x = 0; x < 4; ++x) succ[x] = __uint_as_float(__float_as_uint(succ[x]) - eb23);
                if (live) {                                         // ... which makes them row t+1 of the guide
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        if (p.nt) __builtin_nontemporal_store(own[l], bn + (long)(t + 1) * S + m + l * Q);
                        else bn[(long)(t + 1) * S + m + l * Q] = own[l];
                    }
                }
            }
            // the lane's 16 scores: word j of (w0, w1) holds successor x = j / 2, leads 2 (j & 1) and 2 (j & 1) + 1
            const unsigned w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            float2_t acc01, acc23;                                  // raw[lead 0, 1], raw[lead 2, 3]
            acc01 = fma2(splat2(eb), float2_t{own[0], own[1]}, splat2(BH_BS2_TINY));
            acc23 = fma2(splat2(eb), float2_t{own[2], own[3]}, splat2(BH_BS2_TINY));
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const half2_t h01 = __builtin_bit_cast(half2_t, w[2 * x]), h23 = __builtin_bit_cast(half2_t, w[2 * x + 1]);
                const float2_t e01 = bs2_exp2(float2_t{(float)h01.x, (float)h01.y});
                const float2_t e23 = bs2_exp2(float2_t{(float)h23.x, (float)h23.y});
                acc01 = fma2(e01, splat2(succ[x]), acc01);
                acc23 = fma2(e23, splat2(succ[x]), acc23);
            }
            float raw[4] = {acc01.x, acc01.y, acc23.x, acc23.y};
            const float mx = group_max<GL>(fmaxf(fmaxf(raw[0], raw[1]), fmaxf(raw[2], raw[3])));
            float* nextb = pb + (cur ^ 1) * S;
            if (WPC > 1) {
                if (lane == 0) mxs[(cur ^ 1) * 4 + wv] = mx;
#pragma unroll
                for (int l = 0; l < 4; ++l) nextb[m + l * Q] = raw[l];
                __syncthreads();
            } else {
                const unsigned eb23 = ((__float_as_uint(mx) >> 23) - 127u) << 23;
#pragma unroll
                for (int l = 0; l < 4; ++l) raw[l] = __uint_as_float(__float_as_uint(raw[l]) - eb23);
#pragma unroll
                for (int l = 0; l < 4; ++l) nextb[m + l * Q] = raw[l];
                if (live) {
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        if (p.nt) __builtin_nontemporal_store(raw[l], bn + (long)t * S + m + l * Q);
                        else bn[(long)t * S + m + l * Q] = raw[l];
                    }
                }
            }
            cur ^= 1;
        }
    }
    if (WPC > 1 && live) {          // row 0: scale and store what the last step left
        const float* prev = pb + cur * S;
        const float4_t mv = *(const float4_t*)(mxs + cur * 4);
        const float mx = fmaxf(fmaxf(mv.x, mv.y), fmaxf(mv.z, mv.w));
        const unsigned eb23 = ((__float_as_uint(mx) >> 23) - 127u) << 23;
#pragma unroll
        for (int l = 0; l < 4; ++l) bn[m + l * Q] = __uint_as_float(__float_as_uint(prev[m + l * Q]) - eb23);
    }
}

// STATE_LEN is a template parameter so that every LDS region sits at a constant offset (immediate DS offsets, no address
// arithmetic or scalar registers spent on them); DBG compiles the per-section cycle counters in.
// One wave per chunk, CPW chunks (waves) per workgroup: the waves share nothing but the 16 KiB lse table - each has its own
// staging buffers, beam and hash table, and they never synchronise after the table is loaded. (Sharing the table is what
// keeps the LDS footprint per chunk low enough for several decode kernels and a recurrent layer to be co-resident.)
template <int STATE_LEN>
__host__ __device__ constexpr int beam_wave_lds() {
    constexpr int S = 1 << (2 * STATE_LEN), BTB = beam_btb<STATE_LEN>();
    return 2 * (BTB * 4 * S * 2 + BTB * S * 4) + NBK * BKE * 8 + MAXW * (16 + 8 + 8) + (NBK + 4) * 4 + BTB * MAXW;
}
constexpr int BEAM_TAB_LDS = (BH_LSE_TABLE_SIZE + 2) * 4;

// LDS of one fused scan wave: alpha~ ping-pong [2][S]
template <int STATE_LEN>
__host__ __device__ constexpr int scan_wave_lds() { return 2 * (1 << (2 * STATE_LEN)) * 4 + 64; }
// wave-wide sum; the total ends up in lane 63 and is broadcast from there
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_f<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x124, 0xF>(v);     // row_ror:4
    v += dpp_f<0x128, 0xF>(v);     // row_ror:8   -> every lane holds its row's sum
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 15));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 47));
    const float r4 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    return (r1 + r2) + (r3 + r4);
}

// One step of the forward / posterior scan in the LINEAR domain (BS-2) for the lane's four states j = 4m .. 4m+3 (classes 0 .. 3):
//     a'[j] = e^blank a[j] + sum_r e^{score[j][r]} a[r S/4 + m]            (the four states share their four predecessors)
// and the class products pv[j] = a'[j] b_{t+1}[j] (b = the guide row, linear) are formed by the caller from the RESCALED a' (with every
// score at the cap a row grows by 2^62: products of unscaled values would leave the fp32 range). Nothing here feeds the search - the class posteriors are a
// tolerance-level output (q-scores within 1e-3 of an fp64 scan) - so the exponentials are the hardware's (v_exp_f32).
template <int STATE_LEN>
__device__ __forceinline__ void scan_lin_lane(float ebl, const half_t* row, const float* ap, int m, float (&anew)[4]) {
    constexpr int S = 1 << (2 * STATE_LEN), Q = S / 4;
    const uint4_t w0 = *(const uint4_t*)(row + 16 * m), w1 = *(const uint4_t*)(row + 16 * m + 8);
    const float4_t own = *(const float4_t*)(ap + 4 * m);
    float pr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pr[r] = ap[r * Q + m];
    const unsigned w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // scores above BH_BS2_XMAX are capped like the guide's (v_pk_min_f16: the exponential must not overflow; far below -XMAX it
        // underflows to zero, which is what the oracle's clamp to -XMAX amounts to at fp32 resolution)
        const half2_t cap = {(half_t)BH_BS2_XMAX, (half_t)BH_BS2_XMAX};
        const half2_t h01 = __builtin_elementwise_min(__builtin_bit_cast(half2_t, w[2 * k]), cap);
        const half2_t h23 = __builtin_elementwise_min(__builtin_bit_cast(half2_t, w[2 * k + 1]), cap);
        float a = __fmaf_rn(ebl, own[k], BH_BS2_TINY);      // (the floor keeps every value a normal number: the rescaling below is exact)
        a = __fmaf_rn(__expf((float)h01.x), pr[0], a);
        a = __fmaf_rn(__expf((float)h01.y), pr[1], a);
        a = __fmaf_rn(__expf((float)h23.x), pr[2], a);
        a = __fmaf_rn(__expf((float)h23.y), pr[3], a);
        anew[k] = a;
    }
}

// The forward / posterior scan as ONE wave beside the beam wave of the same chunk (FUSE, <= 256 states): it reads the score rows and the
// guide rows from the LDS blocks the beam wave stages anyway, so the score tensor and the guide are read from HBM once for both. Lane m
// < S/4 owns the states 4m .. 4m+3; alpha lives in a private LDS ping-pong (one wave: LDS operations complete in issue order, no
// barrier inside a step), rescaled every step by the power of two of its maximum. Below 256 states ONE scan wave serves all chunks of the
// workgroup (16 / 4 / 1 lanes each, group_max / group_sum over the lanes of a chunk): a 64-state step costs the instructions of a
// 256-state one, so four chunks share them.
template <int STATE_LEN>
__device__ __forceinline__ void scan_step(float ebl, const half_t* row, const float* bnext, float* ap, float* an, int m, bool act, float* Pt) {
    constexpr int S = 1 << (2 * STATE_LEN), Q = S / 4, GL = Q < 64 ? Q : 64;
    float anew[4];
    scan_lin_lane<STATE_LEN>(ebl, row, ap, m, anew);
    const float4_t bnext4 = *(const float4_t*)(bnext + 4 * m);
    const float mx = group_max<GL>(fmaxf(fmaxf(anew[0], anew[1]), fmaxf(anew[2], anew[3])));
    const unsigned e23 = ((__float_as_uint(mx) >> 23) - 127u) << 23;
    float4_t o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = __uint_as_float(__float_as_uint(anew[k]) - e23);
    if (act) *(float4_t*)(an + 4 * m) = o;
    float cls[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cls[k] = group_sum<GL>(o[k] * bnext4[k]);
    const float inv = 1.0f / ((cls[0] + cls[1]) + (cls[2] + cls[3]));
    if (GL == 1) {
        if (act) *(float4_t*)Pt = float4_t{cls[0] * inv, cls[1] * inv, cls[2] * inv, cls[3] * inv};
    } else if (act && m < 4) {
        Pt[m] = (m == 0 ? cls[0] : m == 1 ? cls[1] : m == 2 ? cls[2] : cls[3]) * inv;
    }
}

// The same scan as a kernel of its own: 1024 states (four waves per chunk, one barrier per step; "beam_fork" runs it beside the beam
// kernel on a helper stream) and the "beam_fuse" 0 arrangement of the smaller state spaces. Score rows and guide rows arrive by LDS-DMA
// in blocks of BS2_FR steps, two blocks resident.
struct Bs2FwdArgs {
    const half_t* scores;  // [N][T][4S]
    const float* b;        // [N][T+1][S] linear guide
    float* P;              // [N][T][4]
    int N, T;
    float blank;
    int nt;
};
constexpr int BS2_FR = 2;      // (one row per block - 33 KiB per 1024-state chunk, two beam waves and two scan workgroups per CU - measured slower: 8.67 vs 8.37 ms per 512 x 2000 x 4096)
template <int STATE_LEN>
struct Bs2FwdGeo {
    static constexpr int S = 1 << (2 * STATE_LEN), Q = S / 4;
    static constexpr int TPC = Q < 64 ? 64 : Q, WPC = TPC / 64, CPB = 256 / TPC;
    static constexpr int ROW = 4 * S * 2, GROW = S * 4;
    static constexpr int CHUNK_LDS = 2 * BS2_FR * (ROW + GROW) + 2 * S * 4 + 2 * 4 * 4 + 2 * 4 * 4 * 4;     // rows, alpha ping-pong, maxima, class sums
};

template <int STATE_LEN>
__global__ __launch_bounds__(256) void bs2_forward_post_kernel(Bs2FwdArgs p) {
    using G = Bs2FwdGeo<STATE_LEN>;
    constexpr int S = G::S, Q = G::Q, WPC = G::WPC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = p.T;
    const int slot = threadIdx.x / G::TPC, m_raw = threadIdx.x - slot * G::TPC;
    const int lane = threadIdx.x & 63, wv = m_raw >> 6;
    const int n = blockIdx.x * G::CPB + slot;
    const bool live = n < p.N;
    const int nn = live ? n : p.N - 1;
    const bool act = m_raw < Q;
    const int m = act ? m_raw : Q - 1;
    char* mine = smem + (size_t)slot * G::CHUNK_LDS;
    half_t* rows = (half_t*)mine;                                   // [2][FR][4S]
    float* grows = (float*)(mine + 2 * BS2_FR * G::ROW);            // [2][FR][S]
    float* al = grows + 2 * BS2_FR * S;                             // [2][S]
    float* mxs = al + 2 * S;                                        // [2][4]
    float* part = mxs + 8;                                          // [2][4 waves][4 classes]
    const half_t* sc = p.scores + (long)nn * T * 4 * S;
    const float* bn = p.b + (long)nn * (T + 1) * S;
    float* Pn = p.P + (long)nn * T * 4;
    const float ebl = __expf(__builtin_amdgcn_fmed3f(p.blank, -BH_BS2_XMAX, BH_BS2_XMAX));
    auto dma_row = [&](const char* src, char* dst, int n16) {
        for (int i0 = 0; i0 < n16; i0 += G::TPC) {
            const int piece = i0 + wv * 64 + lane;
            if (i0 + wv * 64 < n16 && piece < n16) {
                if (p.nt) dma16_nt(src + (long)piece * 16, dst + (size_t)(i0 + wv * 64) * 16);
                else dma16(src + (long)piece * 16, dst + (size_t)(i0 + wv * 64) * 16);
            }
        }
    };
    auto stage = [&](int k) {
        const int t0 = k * BS2_FR, nr = min(BS2_FR, T - t0);
        for (int r = 0; r < nr; ++r) {
            dma_row((const char*)(sc + (long)(t0 + r) * 4 * S), (char*)(rows + (size_t)((k & 1) * BS2_FR + r) * 4 * S), S / 2);
            dma_row((const char*)(bn + (long)(t0 + r + 1) * S), (char*)(grows + (size_t)((k & 1) * BS2_FR + r) * S), S / 4);
        }
    };
    if (act) *(float4_t*)(al + 4 * m) = float4_t{1.0f, 1.0f, 1.0f, 1.0f};
    if (m_raw < 4) mxs[m_raw] = 1.0f;
    if (T > 0) stage(0);
    int cur = 0;
    const int nblk = (T + BS2_FR - 1) / BS2_FR;
    for (int k = 0; k < nblk; ++k) {
        const int t0 = k * BS2_FR, nr = min(BS2_FR, T - t0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (WPC > 1) __syncthreads();
        if (k + 1 < nblk) stage(k + 1);
        for (int r = 0; r < nr; ++r) {
            const int t = t0 + r;
            const half_t* row = rows + (size_t)((k & 1) * BS2_FR + r) * 4 * S;
            const float4_t bnext4 = *(const float4_t*)(grows + (size_t)((k & 1) * BS2_FR + r) * S + 4 * m);
            float anew[4];
            scan_lin_lane<STATE_LEN>(ebl, row, al + cur * S, m, anew);
            float mx = act ? fmaxf(fmaxf(anew[0], anew[1]), fmaxf(anew[2], anew[3])) : 0.0f;
            mx = wave_max_f32(mx);
            // rescale: by this row's own maximum where one wave holds the chunk, by the maximum of the row this step READ where four do
            // (this row's is known only behind the barrier; rows then stay within [g, 2g), g = the growth of a step <= 2^62)
            unsigned sc23;
            if (WPC > 1) {
                const float4_t mv = *(const float4_t*)(mxs + cur * 4);
                sc23 = ((__float_as_uint(fmaxf(fmaxf(mv.x, mv.y), fmaxf(mv.z, mv.w))) >> 23) - 127u) << 23;
            } else {
                sc23 = ((__float_as_uint(mx) >> 23) - 127u) << 23;
            }
            float4_t o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = __uint_as_float(__float_as_uint(anew[c]) - sc23);
            float cls[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) cls[c] = wave_sum_f32(act ? o[c] * bnext4[c] : 0.0f);
            float* an = al + (cur ^ 1) * S;
            if (WPC > 1) {
                // per-wave maxima (in the scale of the stored row) and class sums meet behind the barrier
                if (act) *(float4_t*)(an + 4 * m) = o;
                if (lane == 0) {
                    mxs[(cur ^ 1) * 4 + wv] = __uint_as_float(__float_as_uint(mx) - sc23);
                    *(float4_t*)(part + ((cur ^ 1) * 4 + wv) * 4) = float4_t{cls[0], cls[1], cls[2], cls[3]};
                }
                __syncthreads();
                if (m_raw < 4 && live) {
                    const float* pp = part + (cur ^ 1) * 16;
                    float c4[4], tot = 0.0f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { c4[c] = (pp[c] + pp[4 + c]) + (pp[8 + c] + pp[12 + c]); tot += c4[c]; }
                    Pn[(long)t * 4 + m_raw] = (m_raw == 0 ? c4[0] : m_raw == 1 ? c4[1] : m_raw == 2 ? c4[2] : c4[3]) / tot;
                }
            } else {
                if (act) *(float4_t*)(an + 4 * m) = o;
                const float inv = 1.0f / ((cls[0] + cls[1]) + (cls[2] + cls[3]));
                if (lane < 4 && live) Pn[(long)t * 4 + lane] = (lane == 0 ? cls[0] : lane == 1 ? cls[1] : lane == 2 ? cls[2] : cls[3]) * inv;
            }
            cur ^= 1;
        }
    }
}

// scan waves of a fused workgroup: one per chunk at 256 states, one for all chunks below
template <int STATE_LEN, int CPW, bool FUSE>
__host__ __device__ constexpr int beam_scan_waves() { return !FUSE ? 0 : (1 << (2 * STATE_LEN)) / 4 >= 64 ? CPW : 1; }

template <int STATE_LEN, int CPW, bool DBG, bool FUSE = false>
__global__ __launch_bounds__(64 * (CPW + beam_scan_waves<STATE_LEN, CPW, FUSE>())) __attribute__((amdgpu_waves_per_eu(1, 8))) void beam_kernel(BeamArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int S = 1 << (2 * STATE_LEN);
    constexpr int sh = 2 * (STATE_LEN - 1);
    constexpr int BTB = beam_btb<STATE_LEN>();
    constexpr int QS = S / 4, GLS = QS < 64 ? QS : 64;       // scan wave: lanes per chunk
    constexpr int NTHR = 64 * (CPW + beam_scan_waves<STATE_LEN, CPW, FUSE>());
    static_assert(!FUSE || QS >= 64 || CPW * QS <= 64, "one scan wave must hold the chunks of the workgroup");
    static_assert(NBK == HBINS, "the selection histogram reuses the bucket fill counters");
    static_assert(beam_wave_lds<STATE_LEN>() % 16 == 0 && BEAM_TAB_LDS % 16 == 0, "LDS regions must stay 16-byte aligned");
    const int T = p.T, W = p.W;
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool scan_role = FUSE && wave_all >= CPW;
    // chunk slot inside the workgroup: the wave's own for a beam wave (and for a scan wave at 256 states), per lane group in the shared
    // scan wave of the smaller state spaces
    const int slot_raw = !scan_role ? wave_all : QS >= 64 ? wave_all - CPW : lane / GLS;
    const bool slot_ok = slot_raw < CPW && blockIdx.x * CPW + slot_raw < p.N;
    const int wave = slot_ok ? slot_raw : 0;
    const int n = blockIdx.x * CPW + wave;
    // LDS carve
    float* tab = (float*)smem;                           // lse table (shared by the waves)
    for (int i = threadIdx.x; i < BH_LSE_TABLE_SIZE; i += NTHR) tab[i] = g_lse_tab[i];
    __syncthreads();
    if (!slot_ok && (!scan_role || QS >= 64)) return;        // (a shared scan wave stays: its other lane groups have chunks)
    char* mine = smem + BEAM_TAB_LDS + wave * beam_wave_lds<STATE_LEN>();
    half_t* st_sc = (half_t*)mine;                       // [2][BTB][4S]
    float* st_b = (float*)(st_sc + 2 * BTB * 4 * S);     // [2][BTB][S]
    constexpr int SCAN_LDS = scan_wave_lds<STATE_LEN>();
    char* scan_mem = smem + BEAM_TAB_LDS + CPW * beam_wave_lds<STATE_LEN>() + wave * SCAN_LDS;
    if (scan_role) {
        // ---- forward / posterior scan (linear domain) over the blocks the beam wave stages ---------------------------------
        float* al = (float*)scan_mem;
        const int sm = QS >= 64 ? lane : lane - (lane / GLS) * GLS;          // the lane owns the states 4 sm .. 4 sm + 3 of its chunk
        if (slot_ok) *(float4_t*)(al + 4 * sm) = float4_t{1.0f, 1.0f, 1.0f, 1.0f};       // alpha_0 = 1
        const float ebl = __expf(__builtin_amdgcn_fmed3f(p.blank, -BH_BS2_XMAX, BH_BS2_XMAX));
        float* Pn = p.P + (long)n * T * 4;
        int cb = 0;
        for (int tb0 = 0, blk = 0; tb0 < T; tb0 += BTB, ++blk) {
            const int nsteps = min(BTB, T - tb0);
            __syncthreads();                                                   // block `blk` has landed (beam wave waited for its DMA)
            const half_t* blk_sc = st_sc + (blk & 1) * BTB * 4 * S;
            const float* blk_b = st_b + (blk & 1) * BTB * S;
            for (int u = 0; u < nsteps; ++u) {
                scan_step<STATE_LEN>(ebl, blk_sc + u * 4 * S, blk_b + u * S, al + cb * S, al + (cb ^ 1) * S, sm, slot_ok, Pn + (long)(tb0 + u) * 4);
                cb ^= 1;
            }
        }
        return;
    }
    BeamTable tb;
    tb.ent = (uint2_t*)(st_b + 2 * BTB * S);             // [NBK][BKE], 16-byte aligned
    uint4_t* b_elem = (uint4_t*)(tb.ent + NBK * BKE);    // [32] beam element: state, hash, score bits, -
    uint2_t* m_pair = (uint2_t*)(b_elem + MAXW);         // [32] merged-in move: score bits, info (or -1)
    tb.ov = m_pair + MAXW;                               // [32]
    tb.cnt = (int*)(tb.ov + MAXW);                       // [NBK]; doubles as the selection histogram (zero between uses)
    tb.ov_cnt = tb.cnt + NBK;                            // [4]
    int* hist = tb.cnt;
    uint8_t* st_bp = (uint8_t*)(tb.ov_cnt + 4);          // [BTB][32]

    const half_t* sc = p.scores + (long)n * T * 4 * S;
    const float* bn = p.beta + (long)n * (T + 1) * S;
    uint8_t* bpn = p.bp + (long)n * T * MAXW;

    long long dsec[6] = {0, 0, 0, 0, 0, 0};
    // The lane's three candidate slots (time-invariant; round 5): slot 0 = the STAY of element `lane` (lanes < 32), slots 1 and 2 = the
    // MOVES 4 e + x = lane and 64 + lane (elements lane >> 2 and 16 + (lane >> 2), base x = lane & 3). Candidate order - ties, slots of the
    // new beam - is (slot, lane): the stays by element, then the moves by (element, base), as in oracle_beam_search. A merge then costs
    // ONE lse2 per lane (slot 0) and the hash lookup TWO slots; with the candidates interleaved as c = 5 e + j (rounds 1-4) every slot of
    // every lane ran both.
    const int ce[3] = {lane & (MAXW - 1), lane >> 2, 16 + (lane >> 2)};
    const int mx = lane & 3;
    const bool stay_lane = lane < MAXW;

    // ---- init: top-W states by beta~_0 (ties: lower state), slots in state order ------------------
    int nb;
#pragma unroll
    for (int i = 0; i < NBK * BKE / 64; ++i) tb.ent[lane + 64 * i] = uint2_t{0u, 0xffffffffu};     // matches no tag
    if (lane < MAXW) tb.ov[lane] = uint2_t{0u, 0xffffffffu};
    *(uint4_t*)(tb.cnt + 4 * lane) = uint4_t{0u, 0u, 0u, 0u};
    if (lane < 4) tb.ov_cnt[lane] = 0;
    if (lane < MAXW) m_pair[lane] = uint2_t{0u, 0xffffffffu};
    if (lane < MAXW) b_elem[lane] = uint4_t{0u, 0u, 0u, 0u};
    {
        const int per = (S + 63) / 64;
        unsigned prefix = 0u;
        const int want = W < S ? W : S;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned trial = prefix | (1u << bit);
            int cnt = 0;
            for (int i = 0; i < per; ++i) {
                const int s = i * 64 + lane;
                const unsigned u = s < S ? bs_ukey(bn[s]) : 0u;
                cnt += popc64(__ballot(u >= trial));
            }
            if (cnt >= want) prefix = trial;
        }
        int n_gt = 0;
        for (int i = 0; i < per; ++i) {
            const int s = i * 64 + lane;
            const unsigned u = s < S ? bs_ukey(bn[s]) : 0u;
            n_gt += popc64(__ballot(u > prefix));
        }
        const int need = want - n_gt;
        int tie_before = 0, before = 0;
        for (int i = 0; i < per; ++i) {
            const int s = i * 64 + lane;
            const unsigned u = s < S ? bs_ukey(bn[s]) : 0u;
            const unsigned long long eq = __ballot(s < S && u == prefix);
            const bool tie_ok = ((eq >> lane) & 1ull) && (tie_before + popc64(eq & lanemask_lt(lane))) < need;
            const bool take = (s < S && u > prefix) || tie_ok;
            const unsigned long long tm = __ballot(take);
            if (take) {          // step tag 0
                const int slot = before + popc64(tm & lanemask_lt(lane));
                b_elem[slot] = uint4_t{(unsigned)s, bs_hash0(s), 0u /* score 0.0f */, 0u};      // (the hash table is built at the head of every step)
            }
            tie_before += popc64(eq);
            before += popc64(tm);
        }
        nb = before;
    }

    // Staging: score rows tb0..tb0+nsteps-1 and guide rows tb0+1..tb0+nsteps of a block go global -> LDS by DMA, 1 KiB per
    // instruction, into the buffer the previous block is not using; the block after the current one is requested before
    // the current one is processed, so its latency hides under four beam steps. The DMA is issued from inline assembly:
    // the compiler then does not know of an outstanding LDS write and does not drain the memory queue before every LDS read.
    auto dma_rows = [&](const char* src, char* dst, int n16) {
        for (int i0 = 0; i0 < n16; i0 += 64)
            if (i0 + lane < n16) { if (p.nt) dma16_nt(src + (long)(i0 + lane) * 16, dst + i0 * 16); else dma16(src + (long)(i0 + lane) * 16, dst + i0 * 16); }
    };
    auto stage = [&](int tb0, int which) {
        const int nsteps = min(BTB, T - tb0);
        dma_rows((const char*)(sc + (long)tb0 * 4 * S), (char*)(st_sc + which * BTB * 4 * S), nsteps * S / 2);     // 4S halves per step
        dma_rows((const char*)(bn + (long)(tb0 + 1) * S), (char*)(st_b + which * BTB * S), nsteps * S / 4);         // S floats per step
    };
    if (T > 0) stage(0, 0);
    for (int tb0 = 0, blk = 0; tb0 < T; tb0 += BTB, ++blk) {
        const int nsteps = min(BTB, T - tb0);
        {
            long long ts0 = 0;
            if (DBG) ts0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this block has landed
            // unfused: wave-private buffers, no barrier. Fused: one workgroup barrier per block publishes the landed block to
            // the scan wave and proves that it is done with the block before last, whose buffer the next DMA overwrites
            if (FUSE) __syncthreads();
            if (DBG) dsec[5] += __builtin_readcyclecounter() - ts0;
            if (tb0 + BTB < T) stage(tb0 + BTB, (blk + 1) & 1);  // (everything read from that buffer was consumed a block ago)
        }
        const half_t* blk_sc = st_sc + (blk & 1) * BTB * 4 * S;
        const float* blk_b = st_b + (blk & 1) * BTB * S;
        for (int u = 0; u < nsteps; ++u) {
            const half_t* row = blk_sc + u * 4 * S;
            const float* b1 = blk_b + u * S;
            const unsigned tag = ((unsigned)(tb0 + u) & TAG_MASK) << 15;            // of the table built for this step
            long long tc0 = 0;
            if (DBG) tc0 = __builtin_readcyclecounter();
            // ---- (b) candidates; a move that spells the same sequence as a stay is folded into it ---
            // Every LDS level is issued for all slots before it is consumed, and nothing is conditional on loaded data:
            // level 1 = parent element, level 2 = transition score + guide + hash bucket (moves), guide (stay).
            // A candidate that does not exist (beyond the beam) or was folded into a stay carries score -inf.
            float cs[3];
            unsigned ch[3];
            int cst[3], cinfo[3];
            uint4_t el[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) el[i] = b_elem[ce[i]];
            float mv[3], bg[3];
            uint4_t ent0[3];
            {   // slot 0: the stay
                const int es = (int)(el[0].x & (unsigned)(S - 1));      // (stale slots beyond the beam stay in range)
                cst[0] = es; ch[0] = el[0].y; cinfo[0] = ce[0];
                bg[0] = b1[es];
                // ... which is what the moves look up: the beam's elements enter this step's hash table here, ONE LDS atomic per element
                // (round 5; rounds 2-4 inserted the selected candidates at the end of the previous step from three divergent blocks with
                // an atomic each). One wave: the bucket reads below are issued behind these writes and LDS operations complete in order.
                if (stay_lane && ce[0] < nb) {
                    const int bk = (int)(ch[0] & (NBK - 1));
                    const int pos = atomicAdd(&tb.cnt[bk], 1);
                    const uint2_t e{ch[0], tag | ((unsigned)es << 5) | (unsigned)lane};
                    if (pos < BKE) tb.ent[bk * BKE + pos] = e;
                    else tb.ov[atomicAdd(tb.ov_cnt, 1)] = e;
                }
            }
#pragma unroll
            for (int i = 1; i < 3; ++i) {
                const int es = (int)(el[i].x & (unsigned)(S - 1));
                const int s2 = ((es << 2) | mx) & (S - 1);
                cst[i] = s2;
                ch[i] = bs_mix(el[i].y, mx);
                cinfo[i] = ce[i] | (1 << 5) | (mx << 6);
                mv[i] = (float)row[s2 * 4 + (es >> sh)];
                bg[i] = b1[s2];
                ent0[i] = *(const uint4_t*)(tb.ent + (int)(ch[i] & (NBK - 1)) * BKE);
            }
            const int n_ov = tb.ov_cnt[0];
            cs[0] = (stay_lane && ce[0] < nb) ? __uint_as_float(el[0].z) + p.blank : -INFINITY;
            unsigned dhit[3];
#pragma unroll
            for (int i = 1; i < 3; ++i) {
                cs[i] = ce[i] < nb ? __uint_as_float(el[i].z) + mv[i] : -INFINITY;
                dhit[i] = bucket_match(ent0[i], ch[i], tag | ((unsigned)cst[i] << 5));
            }
            if (n_ov > 0) {                              // rare: some bucket held more than two elements
                for (int k = 0; k < n_ov; ++k) {
                    const uint2_t e = tb.ov[k];
#pragma unroll
                    for (int i = 1; i < 3; ++i)
                        dhit[i] = min(dhit[i], e.x == ch[i] ? e.y - (tag | ((unsigned)cst[i] << 5)) : 32u);
                }
            }
#pragma unroll
            for (int i = 1; i < 3; ++i)
                if (ce[i] < nb && dhit[i] < 32u) {
                    m_pair[dhit[i]] = uint2_t{__float_as_uint(cs[i]), (unsigned)cinfo[i]};
                    cs[i] = -INFINITY;
                }
            // the lookups of this step are done (one wave: LDS operations complete in issue order): reset the fill counters
            *(uint4_t*)(tb.cnt + 4 * lane) = uint4_t{0u, 0u, 0u, 0u};
            if (lane < 4) tb.ov_cnt[lane] = 0;
            if (DBG) { const long long t1 = __builtin_readcyclecounter(); dsec[0] += t1 - tc0; tc0 = t1; }
            {   // (no barriers inside a step: one wave per workgroup, and LDS operations complete in issue order)
                const uint2_t mp = m_pair[ce[0]];
                const bool merged = stay_lane && ce[0] < nb && (int)mp.y >= 0;
                const float ms = __uint_as_float(mp.x);
                const float lse = lse2_tab_nb(cs[0], ms, tab);
                if (merged && ms > cs[0]) cinfo[0] = (int)mp.y;
                if (merged) cs[0] = lse;
            }
            // ---- (c) keys, cut ---------------------------------------------------------------------
            float key[3];
            float best = -INFINITY;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                key[i] = cs[i] + bs2_log(bg[i]);         // -inf for absent / folded candidates (the guide value is always a positive normal)
                best = fmaxf(best, key[i]);
            }
            best = wave_max_f32(best);
            const float thr = best - p.cut;
            unsigned uk[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) uk[i] = key[i] >= thr ? bs_ukey(key[i]) : 0u;
            if (DBG) { const long long t1 = __builtin_readcyclecounter(); dsec[1] += t1 - tc0; tc0 = t1; }
            // ---- (d) top-W, slots in candidate order -----------------------------------------------
            bool sel[3];
            int slot[3];
            const int nnew = hist_select(key, uk, W, lane, thr, p.inv_bin, hist, sel, slot, bs_ukey(thr), bs_ukey(best));
            if (DBG) { const long long t1 = __builtin_readcyclecounter(); dsec[2] += t1 - tc0; tc0 = t1; }
            // best selected candidate (max key, lowest index) gives the renormalisation shift
            const unsigned ubest = bs_ukey(best);
            float shift = 0.0f;
            {
                int found = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const unsigned long long mm = __ballot(sel[i] && uk[i] == ubest);
                    if (!found && mm) {
                        const int src = __ffsll((long long)mm) - 1;
                        shift = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cs[i]), src));
                        found = 1;
                    }
                }
            }
            // the new beam (its elements enter the hash table at the head of the next step)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (sel[i]) {
                    b_elem[slot[i]] = uint4_t{(unsigned)cst[i], ch[i], __float_as_uint(cs[i] - shift), 0u};
                    st_bp[u * MAXW + slot[i]] = (uint8_t)cinfo[i];
                }
            if (lane < MAXW) m_pair[lane] = uint2_t{0u, 0xffffffffu};
            nb = nnew;
            if (DBG) { const long long t1 = __builtin_readcyclecounter(); dsec[3] += t1 - tc0; dsec[4] += nb; }
        }
        // ---- flush back-pointers of this block ------------------------------------------------------
        for (int i = lane; i < nsteps * MAXW / 4; i += 64)
            ((unsigned*)(bpn + (long)tb0 * MAXW))[i] = ((const unsigned*)st_bp)[i];
    }
    // ---- best final element: max score (beta~_T = 0), ties -> lower slot ---------------------------
    {
        const unsigned u = lane < nb ? bs_ukey(__uint_as_float(b_elem[lane].z)) : 0u;
        unsigned m = u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
        const unsigned long long mm = __ballot(u == m && lane < nb);
        if (lane == 0) p.final_slot[n] = __ffsll((long long)mm) - 1;
    }
    if (DBG && lane == 0)
        for (int i = 0; i < 6; ++i) p.dbg[(long)n * 8 + i] = dsec[i];
}

struct FinArgs {
    const uint8_t* bp;     // [N][T][32]
    const int* final_slot; // [N]
    const float* P;        // [N][T][4]
    int N, T;
    float q_scale, q_offset;
    int8_t* sequence;
    int8_t* qstring;
    int8_t* moves;
    float* qfloat;         // optional [N][T]
};

constexpr int FTB = 256;   // back-pointer rows staged per block (8 KiB)

// One wave per chunk. The pointer chase is serial (lane 0) but runs out of LDS-staged blocks; the
// q-score pass is parallel over emitting steps.
__global__ __launch_bounds__(64) void beam_finalize_kernel(FinArgs p) {
    __shared__ __attribute__((aligned(16))) uint8_t st[FTB * MAXW];
    __shared__ int8_t res_m[FTB], res_s[FTB];
    __shared__ int s_r;
    const int n = blockIdx.x, lane = threadIdx.x;
    const int T = p.T;
    const uint8_t* bp = p.bp + (long)n * T * MAXW;
    const float* P = p.P + (long)n * T * 4;
    int8_t* sq = p.sequence + (long)n * T;
    int8_t* qs = p.qstring + (long)n * T;
    int8_t* mv = p.moves + (long)n * T;
    float* qf = p.qfloat ? p.qfloat + (long)n * T : nullptr;
    if (lane == 0) s_r = p.final_slot[n];
    for (int thi = T; thi > 0; thi -= FTB) {
        const int tlo = max(0, thi - FTB);
        const int nbytes = (thi - tlo) * MAXW;
        const uint4_t* src = (const uint4_t*)(bp + (long)tlo * MAXW);
        {   // all loads of the block first, then the LDS writes (one load + wait + write per iteration cost 8 round trips per block)
            constexpr int PER = FTB * MAXW / 16 / 64;
            uint4_t tmp[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (lane + 64 * k < nbytes / 16) tmp[k] = src[lane + 64 * k];
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (lane + 64 * k < nbytes / 16) ((uint4_t*)st)[lane + 64 * k] = tmp[k];
        }
        __syncthreads();
        if (lane == 0) {
            int r = s_r;
            for (int t = thi - 1; t >= tlo; --t) {
                const int info = st[(t - tlo) * MAXW + r];
                const int is_move = (info >> 5) & 1;
                res_m[t - tlo] = (int8_t)is_move;
                res_s[t - tlo] = is_move ? (int8_t)((0x54474341u >> (8 * (info >> 6))) & 0xffu) : (int8_t)0;      // "ACGT"[x] without a load in the chase
                r = info & 31;
            }
            s_r = r;
        }
        __syncthreads();
        for (int i = lane; i < thi - tlo; i += 64) {
            mv[tlo + i] = res_m[i];
            sq[tlo + i] = res_s[i];
            qs[tlo + i] = 0;
            if (qf) qf[tlo + i] = 0.0f;
        }
        __syncthreads();
    }
    __threadfence();
    __syncthreads();
    // q-scores: each lane takes emitting steps t = lane, lane+64, ...
    for (int t = lane; t < T; t += 64) {
        if (!mv[t]) continue;
        int t2 = t + 1;
        while (t2 < T && !mv[t2]) ++t2;
        const int c = sq[t];
        const int x = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
        double err = 0.0;
        for (int u = t; u < t2; ++u) {
            const float4_t pv = *(const float4_t*)(P + (long)u * 4);
#pragma unroll
            for (int y = 0; y < 4; ++y)
                if (y != x) err += (double)pv[y];
        }
        err /= (double)(t2 - t);
        if (err < 1e-10) err = 1e-10;
        float qv = (float)(-10.0 * log10(err)) * p.q_scale + p.q_offset;
        qv = fminf(fmaxf(qv, 1.0f), 50.0f);
        if (qf) qf[t] = qv;
        qs[t] = (int8_t)(33 + (int)floorf(qv + 0.5f));
    }
}

}  // namespace bh

static int g_decode_nt = 0;     // bh_set_option("decode_nt", v)

size_t bh_k_beam_workspace(int N, int T, int state_len) {
    size_t S = 1;
    for (int i = 0; i < state_len; ++i) S *= 4;
    size_t b = 0;
    b += (size_t)N * (T + 1) * S * sizeof(float) + 256;   // beta~
    b += (size_t)N * (T + 1) * sizeof(double) + 256;      // Bcum
    b += (size_t)N * sizeof(double) + 256;                // logZ
    b += (size_t)N * T * 4 * sizeof(float) + 256;         // P
    b += (size_t)N * T * 32 + 256;                        // bp
    b += (size_t)N * sizeof(int) + 256;                   // final slot
    b += (size_t)N * 8 * sizeof(long long) + 256;         // debug counters
    return b;
}

// Log-semiring partition function (CTC_CRF.logZ, bonito/crf/model.py:47-52) on koi-layout scores: the
// backward scan of the beam decoder already produces it. workspace: bh_k_beam_workspace bytes.
int bh_k_crf_logz(const void* scores, int N, int T, int state_len, float blank, void* workspace, double* logz_out,
                  hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(state_len >= 1 && state_len <= 5 && N > 0 && T > 0, "crf_logz: bad shape");
    int S = 1;
    for (int i = 0; i < state_len; ++i) S *= 4;
    auto align = [](size_t x) { return (x + 255) / 256 * 256; };
    char* w = (char*)workspace;
    float* beta = (float*)w;   w += align((size_t)N * (T + 1) * S * sizeof(float));
    double* Bcum = (double*)w;
    ScanArgs sa{(const half_t*)scores, N, T, S, state_len, blank, beta, Bcum, logz_out, nullptr, g_decode_nt, 1};
    {
        int b_threads = 0;
        size_t b_lds = 0;
        backward_geometry(S, N, sa.cpb, b_threads, b_lds);
        if (b_lds > 64 * 1024) BH_CHECK_HIP(bh_max_lds((const void*)crf_backward_kernel, (int)b_lds));
        hipLaunchKernelGGL(crf_backward_kernel, dim3((N + sa.cpb - 1) / sa.cpb), dim3(b_threads), b_lds, stream, sa);
        sa.cpb = 1;
    }
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

size_t bh_k_posterior_viterbi_workspace(int N, int T, int state_len) {
    size_t S = 1;
    for (int i = 0; i < state_len; ++i) S *= 4;
    return bh_k_beam_workspace(N, T, state_len) + (size_t)N * T * S + 256;
}

int bh_k_posterior_viterbi(const void* scores, int N, int T, int state_len, float blank, void* workspace, int8_t* moves,
                           int8_t* path, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(state_len >= 1 && state_len <= 5 && N > 0 && T > 0, "posterior_viterbi: bad shape");
    int S = 1;
    for (int i = 0; i < state_len; ++i) S *= 4;
    auto align = [](size_t x) { return (x + 255) / 256 * 256; };
    char* w = (char*)workspace;
    float* beta = (float*)w;   w += align((size_t)N * (T + 1) * S * sizeof(float));
    double* Bcum = (double*)w; w += align((size_t)N * (T + 1) * sizeof(double));
    double* logZ = (double*)w;
    uint8_t* bp = (uint8_t*)workspace + bh_k_beam_workspace(N, T, state_len);
    ScanArgs sa{(const half_t*)scores, N, T, S, state_len, blank, beta, Bcum, logZ, nullptr, g_decode_nt, 1};
    const int threads = S < 64 ? 64 : S;
    {
        int b_threads = 0;
        size_t b_lds = 0;
        backward_geometry(S, N, sa.cpb, b_threads, b_lds);
        if (b_lds > 64 * 1024) BH_CHECK_HIP(bh_max_lds((const void*)crf_backward_kernel, (int)b_lds));
        hipLaunchKernelGGL(crf_backward_kernel, dim3((N + sa.cpb - 1) / sa.cpb), dim3(b_threads), b_lds, stream, sa);
        sa.cpb = 1;
    }
    PostVitArgs pa{sa, bp, moves, path};
    int TB = (32 * 1024) / S; if (TB > 512) TB = 512; if (TB < 1) TB = 1;
    const size_t lds_pv = (size_t)(BH_LSE_TABLE_SIZE + 2 + 4 * S + 4) * sizeof(float) + (size_t)TB * S + 2 * TB + 32;
    hipLaunchKernelGGL(crf_posterior_viterbi_kernel, dim3(N), dim3(threads), lds_pv, stream, pa);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

namespace {
// One helper stream + fork/join events per (device, host thread): bh_beam_search is re-entrant per thread, and a decode
// worker thread drives one device.
struct SideStream { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
int g_beam_select = 0;     // 0 histogram top-W selection, 1 radix search (A/B and regression tests)
int g_beam_fork = -1;      // -1 auto (fork for small state spaces), 0 never, 1 always; bh_set_option("beam_fork", v)
int g_beam_cpw = 0;        // chunks per workgroup of the fused beam kernel at 256 states ("beam_cpw": 0 = automatic, 1, 2, 4). The decode
                           // kernels are chains of dependent LDS round trips, ballots and scalar branches - ~5.5 k cycles per time step
                           // of a chunk whatever else runs - so what counts is how many chunks a CU works on AT ONCE, and whether the
                           // call fits ONE round of resident workgroups. The 16 KiB lse table is shared by the chunks of a workgroup
                           // and a chunk's own LDS is 14.1 KiB (one-step staging blocks, BTB = 1): one chunk per workgroup = five
                           // chunks per CU, two = six, four = eight (two workgroups of 72.5 KiB). Round 5, MI355X, 2048 x 1667 steps
                           // at 256 states, decode stage alone: 12.97 ms (BTB 2, one per workgroup: four per CU, two rounds) ->
                           // 12.00 (BTB 1, five per CU) -> 8.72 ms (BTB 1, four per workgroup: eight per CU, one round). Round 3 had
                           // tried two and three per workgroup with BTB 2 (3.71 / 3.56 / 3.63 ms per batch) and read the flat result
                           // as "instruction bound": those geometries still needed two rounds. Automatic = the smallest of 1 / 2 / 4
                           // that lets the call's chunks be resident together.
int g_beam_fuse = -1;      // forward / posterior scan as a second wave of the beam kernel's workgroups: -1 auto (<= 256 states: one
                           // scan wave keeps up with the beam wave; at 1024 states its 16 states per lane make the beam wave wait:
                           // sup-LSTM 256 x 3334 decode 18 -> 36 ms), 0 never (own kernel), 1 always
SideStream* side_stream(int S) {
    // Measured (MI355X, 512 x 1667 steps): forking shortens the decode stage 8.3 -> 6.7 ms (S=64) / 11.3 -> 9.3 ms (S=256).
    // Where the decoder is the pipeline bottleneck (fast-sized models) that is a net win (10.0 -> 9.4 ms per step); next to
    // the latency-bound LSTM of a hac-sized model the denser decode burst costs the encoder more than it saves (24.4-25.0 ->
    // 25.3-25.5 ms per step), so auto mode forks for S <= 64 - and for 1024 states, where the scan is not a wave of the beam
    // kernel and that kernel is one wave per chunk on an otherwise idle CU (round 4, 256-chunk batches: decode 9.9 -> 7.5 ms of
    // the transformer sup model with its step unchanged at 67.7 ms, 17.5 -> 13.5 ms of the LSTM sup model, step 112.7 -> 111.0).
    if (g_beam_fork == 0 || (g_beam_fork < 0 && S > 64 && S < 1024)) return nullptr;
    thread_local SideStream per_dev[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream& s = per_dev[dev];
    if (!s.stream) {
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) { s.stream = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    return &s;
}
}  // namespace

int bh_k_beam_search(const void* scores, int N, int T, int state_len, int beam_width, float beam_cut,
                     float blank, float q_scale, float q_offset, void* workspace, int8_t* sequence,
                     int8_t* qstring, int8_t* moves, float* qfloat, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(state_len >= 1 && state_len <= 5, "beam_search: state_len must be in 1..5 (got %d)", state_len);
    BH_REQUIRE(N > 0 && T > 0, "beam_search: empty problem N=%d T=%d", N, T);
    BH_REQUIRE(beam_width >= 1 && beam_width <= 32, "beam_search: beam_width must be in 1..32 (got %d)", beam_width);
    BH_REQUIRE(beam_cut >= 1.0f, "beam_search: beam_cut must be >= 1");
    BH_REQUIRE(T < (1 << 17), "beam_search: at most 131071 steps per chunk (got %d)", T);
    int S = 1;
    for (int i = 0; i < state_len; ++i) S *= 4;
    auto align = [](size_t x) { return (x + 255) / 256 * 256; };
    char* w = (char*)workspace;
    float* beta = (float*)w;   w += align((size_t)N * (T + 1) * S * sizeof(float));
    w += align((size_t)N * (T + 1) * sizeof(double));      // (B_t and logZ of the Log-semiring scans: bh_crf_logz / posterior Viterbi share
    w += align((size_t)N * sizeof(double));                //  this workspace layout; the beam search no longer needs them)
    float* P = (float*)w;      w += align((size_t)N * T * 4 * sizeof(float));
    uint8_t* bp = (uint8_t*)w; w += align((size_t)N * T * 32);
    int* fin = (int*)w;         w += align((size_t)N * sizeof(int));
    long long* dbg = getenv("BH_BEAM_DEBUG") ? (long long*)w : nullptr;

    const bool fuse = g_beam_fuse != 0 && S <= 256;      // (the scan wave owns four states per lane: up to 256 states)
    // ---- guide: linear-domain backward scan (BS-2) -------------------------------------------------------------------------------
    {
        Bs2Args a2{(const half_t*)scores, beta, N, T, blank, g_decode_nt};
        auto launch_bwd = [&](auto kern, int cpb, size_t lds, int threads) -> int {
            if (lds > 64 * 1024) BH_CHECK_HIP(bh_max_lds((const void*)kern, (int)lds));
            hipLaunchKernelGGL(kern, dim3((N + cpb - 1) / cpb), dim3(threads), lds, stream, a2);
            return 0;
        };
        int rc = -2;
        switch (state_len) {
            case 1: rc = launch_bwd(bs2_backward_kernel<1>, Bs2Geo<1>::CPB, (size_t)Bs2Geo<1>::UPB * Bs2Geo<1>::UNIT_LDS, Bs2Geo<1>::UPB * Bs2Geo<1>::UT); break;
            case 2: rc = launch_bwd(bs2_backward_kernel<2>, Bs2Geo<2>::CPB, (size_t)Bs2Geo<2>::UPB * Bs2Geo<2>::UNIT_LDS, Bs2Geo<2>::UPB * Bs2Geo<2>::UT); break;
            case 3: rc = launch_bwd(bs2_backward_kernel<3>, Bs2Geo<3>::CPB, (size_t)Bs2Geo<3>::UPB * Bs2Geo<3>::UNIT_LDS, Bs2Geo<3>::UPB * Bs2Geo<3>::UT); break;
            case 4: rc = launch_bwd(bs2_backward_kernel<4>, Bs2Geo<4>::CPB, (size_t)Bs2Geo<4>::UPB * Bs2Geo<4>::UNIT_LDS, Bs2Geo<4>::UPB * Bs2Geo<4>::UT); break;
            case 5: rc = launch_bwd(bs2_backward_kernel<5>, Bs2Geo<5>::CPB, (size_t)Bs2Geo<5>::UPB * Bs2Geo<5>::UNIT_LDS, Bs2Geo<5>::UPB * Bs2Geo<5>::UT); break;
        }
        if (rc) return rc;
    }
    // The forward / posterior scan and the beam kernel both depend only on the backward scan. Default for <= 256 states: the scan runs as
    // a second wave inside the beam kernel's workgroups (FUSE), sharing the staged score / guide blocks. Otherwise ("beam_fuse" 0, 1024
    // states) it is a kernel of its own, and with "beam_fork" it runs BESIDE the beam kernel on a per-device helper stream, forked from
    // and joined back into the caller's stream with events.
    SideStream* side = fuse ? nullptr : side_stream(S);
    const bool fork = side != nullptr;
    if (fork) {
        BH_CHECK_HIP(hipEventRecord(side->fork, stream));
        BH_CHECK_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
    }
    if (!fuse) {
        Bs2FwdArgs f2{(const half_t*)scores, beta, P, N, T, blank, g_decode_nt};
        hipStream_t fs = fork ? side->stream : stream;
        auto launch_fwd = [&](auto kern, int cpb, size_t chunk_lds) -> int {
            const size_t lds = (size_t)cpb * chunk_lds;
            if (lds > 64 * 1024) BH_CHECK_HIP(bh_max_lds((const void*)kern, (int)lds));
            hipLaunchKernelGGL(kern, dim3((N + cpb - 1) / cpb), dim3(256), lds, fs, f2);
            return 0;
        };
        int rc = -2;
        switch (state_len) {
            case 1: rc = launch_fwd(bs2_forward_post_kernel<1>, Bs2FwdGeo<1>::CPB, Bs2FwdGeo<1>::CHUNK_LDS); break;
            case 2: rc = launch_fwd(bs2_forward_post_kernel<2>, Bs2FwdGeo<2>::CPB, Bs2FwdGeo<2>::CHUNK_LDS); break;
            case 3: rc = launch_fwd(bs2_forward_post_kernel<3>, Bs2FwdGeo<3>::CPB, Bs2FwdGeo<3>::CHUNK_LDS); break;
            case 4: rc = launch_fwd(bs2_forward_post_kernel<4>, Bs2FwdGeo<4>::CPB, Bs2FwdGeo<4>::CHUNK_LDS); break;
            case 5: rc = launch_fwd(bs2_forward_post_kernel<5>, Bs2FwdGeo<5>::CPB, Bs2FwdGeo<5>::CHUNK_LDS); break;
        }
        if (rc) return rc;
    }
    if (fork) BH_CHECK_HIP(hipEventRecord(side->join, side->stream));
    BeamArgs ba{(const half_t*)scores, beta, N, T, S, state_len, beam_width, blank, logf(beam_cut), bp, fin, dbg,
                g_beam_select ? 0.0f : 256.0f / fmaxf(logf(beam_cut), 1e-6f), P, g_decode_nt};
    // Chunks (waves) per workgroup, measured on MI355X next to the encoder of the same model: four for the narrow state
    // spaces (fast-sized models, three lanes: 1.20e9 -> 1.26e9 samples/s); one for 256 states - two waves per workgroup
    // there cost the hac pipeline 6 % (the 78 KiB workgroups find room beside the recurrent layer's workgroups later).
    auto launch_beam = [&](auto kern, int cpw, size_t wave_lds, size_t scan_lds = 0) -> int {
        const size_t lds_beam = (size_t)BEAM_TAB_LDS + cpw * (wave_lds + scan_lds);
        if (lds_beam > 64 * 1024)
            BH_CHECK_HIP(bh_max_lds((const void*)kern, (int)lds_beam));
        hipLaunchKernelGGL(kern, dim3((N + cpw - 1) / cpw), dim3(64 * (cpw + (!scan_lds ? 0 : S >= 256 ? cpw : 1))), lds_beam, stream, ba);
        return 0;
    };
    int lrc = -2;
    int cpw4 = g_beam_cpw;           // chunks per workgroup at 256 states
    if (fuse && state_len == 4 && cpw4 <= 0) {
        const long cus = bh_cu_count();
        cpw4 = (long)N <= 5 * cus ? 1 : (long)N <= 6 * cus ? 2 : 4;
    }
    if (false) {
    } else if (fuse && state_len == 4 && !dbg && cpw4 == 4) {
        lrc = launch_beam(beam_kernel<4, 4, false, true>, 4, beam_wave_lds<4>(), scan_wave_lds<4>());
    } else if (fuse && state_len == 4 && !dbg && cpw4 == 2) {
        lrc = launch_beam(beam_kernel<4, 2, false, true>, 2, beam_wave_lds<4>(), scan_wave_lds<4>());
    } else if (fuse) {
        switch (state_len * 2 + (dbg ? 1 : 0)) {
            case 2: lrc = launch_beam(beam_kernel<1, 4, false, true>, 4, beam_wave_lds<1>(), scan_wave_lds<1>()); break;
            case 3: lrc = launch_beam(beam_kernel<1, 4, true, true>, 4, beam_wave_lds<1>(), scan_wave_lds<1>()); break;
            case 4: lrc = launch_beam(beam_kernel<2, 4, false, true>, 4, beam_wave_lds<2>(), scan_wave_lds<2>()); break;
            case 5: lrc = launch_beam(beam_kernel<2, 4, true, true>, 4, beam_wave_lds<2>(), scan_wave_lds<2>()); break;
            case 6: lrc = launch_beam(beam_kernel<3, 4, false, true>, 4, beam_wave_lds<3>(), scan_wave_lds<3>()); break;
            case 7: lrc = launch_beam(beam_kernel<3, 4, true, true>, 4, beam_wave_lds<3>(), scan_wave_lds<3>()); break;
            case 8: lrc = launch_beam(beam_kernel<4, 1, false, true>, 1, beam_wave_lds<4>(), scan_wave_lds<4>()); break;
            case 9: lrc = launch_beam(beam_kernel<4, 1, true, true>, 1, beam_wave_lds<4>(), scan_wave_lds<4>()); break;
        }
    } else
    switch (state_len * 2 + (dbg ? 1 : 0)) {
        case 2: lrc = launch_beam(beam_kernel<1, 4, false>, 4, beam_wave_lds<1>()); break;
        case 3: lrc = launch_beam(beam_kernel<1, 4, true>, 4, beam_wave_lds<1>()); break;
        case 4: lrc = launch_beam(beam_kernel<2, 4, false>, 4, beam_wave_lds<2>()); break;
        case 5: lrc = launch_beam(beam_kernel<2, 4, true>, 4, beam_wave_lds<2>()); break;
        case 6: lrc = launch_beam(beam_kernel<3, 4, false>, 4, beam_wave_lds<3>()); break;
        case 7: lrc = launch_beam(beam_kernel<3, 4, true>, 4, beam_wave_lds<3>()); break;
        case 8: lrc = launch_beam(beam_kernel<4, 1, false>, 1, beam_wave_lds<4>()); break;
        case 9: lrc = launch_beam(beam_kernel<4, 1, true>, 1, beam_wave_lds<4>()); break;
        case 10: lrc = launch_beam(beam_kernel<5, 1, false>, 1, beam_wave_lds<5>()); break;
        case 11: lrc = launch_beam(beam_kernel<5, 1, true>, 1, beam_wave_lds<5>()); break;
    }
    if (lrc) return lrc;
    if (fork) BH_CHECK_HIP(hipStreamWaitEvent(stream, side->join, 0));
    FinArgs fa{bp, fin, P, N, T, q_scale, q_offset, sequence, qstring, moves, qfloat};
    hipLaunchKernelGGL(beam_finalize_kernel, dim3(N), dim3(64), 0, stream, fa);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

int bh_k_decode_set_option(const char* name, int value) {
    if (name && !strcmp(name, "beam_fork")) { g_beam_fork = value; return 0; }
    if (name && !strcmp(name, "beam_select")) { g_beam_select = value; return 0; }
    if (name && !strcmp(name, "beam_fuse")) { g_beam_fuse = value; return 0; }
    if (name && !strcmp(name, "beam_cpw")) { g_beam_cpw = value; return 0; }
    if (name && !strcmp(name, "decode_nt")) { g_decode_nt = value; return 0; }
    if (name && !strcmp(name, "viterbi_quad")) { bh::g_viterbi_quad = value; return 0; }
    return 1;     // not a decoder option
}
