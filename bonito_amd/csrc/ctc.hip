// Kernels specific to the QuartzNet CTC models (legacy dna_r9.4.1 `bonito.ctc`, /root/reference
// bonito/ctc/model.py): depthwise time-channel-separable convolution (TCSConv1d.depthwise, :99-103),
// the 1x1 decoder convolution + log_softmax (Decoder, :195-207), and the CTC decoders that replace the
// fast_ctc_decode Rust crate (:39-46).
#include "common.h"
#include "kernels.h"

namespace bh {

// ---- depthwise conv, channel-minor: out[n][t][c] = sum_k w[c][k] * in[n][t*stride + k - pad][c] ----
struct DwArgs {
    const half_t* in;
    const float* w;   // [C][K]
    half_t* out;
    int N, Lin, Lout, C, K, stride, pad;
};

constexpr int DW_T = 64;   // output steps per workgroup
constexpr int DW_C = 64;   // channels per workgroup

__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int span = (DW_T - 1) * p.stride + p.K;
    half_t* xs = (half_t*)smem;                         // [span][DW_C]
    float* ws = (float*)(xs + (size_t)span * DW_C);     // [K][DW_C]  (tap-major: lanes read consecutive channels)
    const int n = blockIdx.z, c0 = blockIdx.y * DW_C, t0 = blockIdx.x * DW_T;
    const int tid = threadIdx.x;
    const int p0 = t0 * p.stride - p.pad;
    const half_t* src = p.in + (long)n * p.Lin * p.C;
    for (int i = tid; i < span * (DW_C / 8); i += 256) {
        const int r = i / (DW_C / 8), cc = (i % (DW_C / 8)) * 8;
        const int pos = p0 + r;
        uint4_t v = {0, 0, 0, 0};
        if (pos >= 0 && pos < p.Lin && c0 + cc < p.C) v = *(const uint4_t*)(src + (long)pos * p.C + c0 + cc);
        *(uint4_t*)(xs + r * DW_C + cc) = v;
    }
    for (int i = tid; i < p.K * DW_C; i += 256) {
        const int k = i / DW_C, c = i % DW_C;
        ws[i] = (c0 + c < p.C) ? p.w[(long)(c0 + c) * p.K + k] : 0.0f;
    }
    __syncthreads();
    // thread -> (8 channels, 2 output steps)
    const int cc = (tid & 7) * 8;
    for (int tt = tid >> 3; tt < DW_T; tt += 32) {
        const int t = t0 + tt;
        if (t >= p.Lout || c0 + cc >= p.C) continue;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const half_t* xr = xs + tt * p.stride * DW_C + cc;
        for (int k = 0; k < p.K; ++k) {
            const half8_t xv = *(const half8_t*)(xr + k * DW_C);
            const float* wk = ws + k * DW_C + cc;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(wk[e], (float)xv[e], acc[e]);
        }
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
        *(half8_t*)(p.out + ((long)n * p.Lout + t) * p.C + c0 + cc) = o;
    }
}

// ---- CTC head: logits = x W^T + b (classes <= 8), log_softmax over classes -> fp16 [M][classes] ----
struct HeadArgs {
    const half_t* in;   // [M][F]
    const float* w;     // [classes][F]
    const float* b;     // [classes]
    half_t* out;        // [M][classes]
    long M;
    int F, classes;
};

__global__ __launch_bounds__(256) void ctc_head_kernel(HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ws = (float*)smem;   // [classes][F]
    for (int i = threadIdx.x; i < p.classes * p.F; i += 256) ws[i] = p.w[i];
    __syncthreads();
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= p.M) return;
    float logit[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) logit[c] = (c < p.classes) ? p.b[c] : -INFINITY;
    const half_t* x = p.in + m * p.F;
    for (int f = 0; f < p.F; f += 8) {
        const half8_t xv = *(const half8_t*)(x + f);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < p.classes) {
                const float* wc = ws + c * p.F + f;
#pragma unroll
                for (int e = 0; e < 8; ++e) logit[c] = fmaf(wc[e], (float)xv[e], logit[c]);
            }
    }
    float mx = logit[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) mx = fmaxf(mx, logit[c]);
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) sum += (c < p.classes) ? __expf(logit[c] - mx) : 0.0f;
    const float lse = mx + __logf(sum);
    for (int c = 0; c < p.classes; ++c) p.out[m * p.classes + c] = (half_t)(logit[c] - lse);
}

}  // namespace bh

int bh_k_dwconv(const void* in, const float* w, void* out, int N, int Lin, int Lout, int C, int K, int stride,
                int pad, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(C % 8 == 0 && K >= 1 && stride >= 1, "dwconv: need C%%8==0 (C=%d K=%d)", C, K);
    DwArgs a{(const half_t*)in, w, (half_t*)out, N, Lin, Lout, C, K, stride, pad};
    const size_t lds = (size_t)((DW_T - 1) * stride + K) * DW_C * 2 + (size_t)K * DW_C * 4;
    BH_REQUIRE(lds <= 160 * 1024, "dwconv: kernel %d x stride %d does not fit LDS", K, stride);
    if (lds > 64 * 1024)
        BH_CHECK_HIP(bh_max_lds((const void*)dwconv_kernel, (int)lds));
    dim3 grid((Lout + DW_T - 1) / DW_T, (C + DW_C - 1) / DW_C, N);
    hipLaunchKernelGGL(dwconv_kernel, grid, dim3(256), lds, stream, a);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

int bh_k_ctc_head(const void* in, const float* w, const float* bias, void* out, long M, int features, int classes,
                  hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(classes >= 1 && classes <= 8 && features % 8 == 0 && features > 0,
               "ctc_head: need classes<=8 and features%%8==0 (classes=%d features=%d)", classes, features);
    HeadArgs a{(const half_t*)in, w, bias, (half_t*)out, M, features, classes};
    const size_t lds = (size_t)classes * features * 4;
    hipLaunchKernelGGL(ctc_head_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), lds, stream, a);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Greedy CTC decode of R stitched reads in one launch (replaces fast_ctc_decode.viterbi_search,
// /root/reference bonito/ctc/model.py:39-42): per step argmax label (ties: lowest label), collapse
// repeats, drop blank (label 0). For every emitted base: path = first step of its run, quality from the
// MEAN probability of the label over the run:  q = -10 log10(max(1 - p, 1e-4)) * scale + bias,
// char = 33 + round(q)  (bonito/util.py:105-111 `phred`).  One workgroup per read.
namespace bh {

struct CtcArgs {
    const float* logp;     // concatenated [sum T_r][C] log-probabilities
    const long* offs;      // [R+1] step offsets
    int R, C;
    float qscale, qbias;
    int8_t* seq;           // [sum T_r] label index (1..C-1) per emitted base, compacted per read at offs[r]
    int8_t* qual;          // [sum T_r] phred char
    int* path;             // [sum T_r] step index within the read
    int* count;            // [R] emitted bases
};

__device__ __forceinline__ int ctc_argmax(const float* row, int C) {
    int best = 0;
    float bv = row[0];
    for (int c = 1; c < C; ++c)
        if (row[c] > bv) { bv = row[c]; best = c; }
    return best;
}

__global__ __launch_bounds__(256) void ctc_greedy_kernel(CtcArgs p) {
    __shared__ int scan[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const long o0 = p.offs[r];
    const int T = (int)(p.offs[r + 1] - o0);
    const float* lp = p.logp + o0 * p.C;
    const int per = (T + 255) / 256;
    const int t0 = tid * per, t1 = min(T, t0 + per);
    // pass 1: count emissions in my segment (a run start: label != 0 and label != previous step's label)
    int cnt = 0;
    {
        int prev = (t0 > 0 && t0 < T) ? ctc_argmax(lp + (long)(t0 - 1) * p.C, p.C) : 0;
        for (int t = t0; t < t1; ++t) {
            const int lab = ctc_argmax(lp + (long)t * p.C, p.C);
            cnt += (lab != 0 && lab != prev);
            prev = lab;
        }
    }
    scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {     // inclusive Hillis-Steele scan
        int v = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int pos = scan[tid] - cnt;
    if (tid == 255) p.count[r] = scan[255];
    // pass 2: emit
    int prev = (t0 > 0 && t0 < T) ? ctc_argmax(lp + (long)(t0 - 1) * p.C, p.C) : 0;
    for (int t = t0; t < t1; ++t) {
        const int lab = ctc_argmax(lp + (long)t * p.C, p.C);
        if (lab != 0 && lab != prev) {
            double sum = 0.0;
            int n = 0;
            for (int u = t; u < T; ++u) {            // the run may extend past my segment
                const float* row = lp + (long)u * p.C;
                if (ctc_argmax(row, p.C) != lab) break;
                sum += (double)__expf(row[lab]);
                ++n;
            }
            const float prob = (float)(sum / (double)n);
            const float e = fmaxf(1.0f - prob, 1e-4f);
            const float q = -10.0f * log10f(e) * p.qscale + p.qbias;
            p.seq[o0 + pos] = (int8_t)lab;
            p.qual[o0 + pos] = (int8_t)((int)rintf(q) + 33);
            p.path[o0 + pos] = t;
            ++pos;
        }
        prev = lab;
    }
}

}  // namespace bh

int bh_k_ctc_greedy(const float* logp, const long* offs, int R, int C, float qscale, float qbias, int8_t* seq,
                    int8_t* qual, int* path, int* count, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(R > 0 && C >= 2 && C <= 8, "ctc_greedy: need R > 0 and 2 <= classes <= 8 (R=%d C=%d)", R, C);
    CtcArgs a{logp, offs, R, C, qscale, qbias, seq, qual, path, count};
    hipLaunchKernelGGL(ctc_greedy_kernel, dim3(R), dim3(256), 0, stream, a);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// CTC prefix beam search "PB-1" (replaces fast_ctc_decode.beam_search, /root/reference bonito/ctc/model.py:44;
// definition and CPU restatement: oracle/crf_oracle.c::oracle_ctc_prefix_beam). The search over one read is a
// short serial recurrence over T steps with a beam of <= 16 prefixes x <= 8 labels, so the parallelism is
// ACROSS reads: one lane per read, R reads per launch. Log-space arithmetic uses the shared lse2 table, so
// results are bit-identical to the oracle.
#include "../../include/bh_lse_table.h"
namespace bh {

__device__ const float g_lse_tab_ctc[BH_LSE_TABLE_SIZE] = {BH_LSE_TABLE_VALUES};

__device__ __forceinline__ float lse2_g(float a, float b) {
    float m = fmaxf(a, b);
    float d = fabsf(a - b);
    if (!(d < BH_LSE_RANGE) || m == -INFINITY) return m;
    float x = d * BH_LSE_SCALE;
    int i = (int)x;
    float f = x - (float)i;
    float t0 = g_lse_tab_ctc[i];
    return m + __fmaf_rn(f, g_lse_tab_ctc[i + 1] - t0, t0);
}

constexpr int PB_MAXB = 16, PB_MAXC = 8;

struct PbArgs {
    const float* logp;     // concatenated [sum T_r][C]
    const long* offs;      // [R+1]
    int R, C, B;
    float lthr;            // log(threshold)
    int* nodes;            // workspace: per read (1 + T_r*B) nodes x (3 + C-1) ints, at node offset offs[r]*B + r
    int8_t* labels;        // [sum T_r] compacted per read at offs[r]
    int* path;             // [sum T_r]
    int* count;            // [R]
};

// Round 6: the per-step candidate and beam arrays are indexed at run time, so hipcc put them in SCRATCH (private memory behind the L2:
// ~1 us per dependent access) and a 1334-step read took 0.68 s. They now live in LDS, element i of lane l at [i * L + l] (lanes that
// walk the same i hit distinct banks); L = lanes per workgroup is chosen by the launcher so that L * 24 B (C + 1) fits 64 KiB.
__global__ __launch_bounds__(64) void ctc_prefix_beam_kernel(PbArgs p, int L) {
    extern __shared__ __align__(8) unsigned char pb_smem[];
    const int lane = threadIdx.x;
    const int r = blockIdx.x * L + lane;
    if (lane >= L || r >= p.R) return;
    const long o0 = p.offs[r];
    const int T = (int)(p.offs[r + 1] - o0);
    const int C = p.C, B = p.B, NI = 3 + C - 1, NC = B * C;
    long* c_key_ = (long*)pb_smem;                               // [NC][L]
    float* c_pb_ = (float*)(c_key_ + (size_t)NC * L);            // [NC][L]
    float* c_pnb_ = c_pb_ + (size_t)NC * L;
    float* sc_ = c_pnb_ + (size_t)NC * L;
    int* used_ = (int*)(sc_ + (size_t)NC * L);
    int* b_node_ = used_ + (size_t)NC * L;                       // [B][L] each from here on
    float* b_pb_ = (float*)(b_node_ + (size_t)B * L);
    float* b_pnb_ = b_pb_ + (size_t)B * L;
    int* s_node_ = (int*)(b_pnb_ + (size_t)B * L);
    float* s_pb_ = (float*)(s_node_ + (size_t)B * L);
    float* s_pnb_ = s_pb_ + (size_t)B * L;
#define c_key(i) c_key_[(i) * L + lane]
#define c_pb(i) c_pb_[(i) * L + lane]
#define c_pnb(i) c_pnb_[(i) * L + lane]
#define sc(i) sc_[(i) * L + lane]
#define used(i) used_[(i) * L + lane]
#define b_node(i) b_node_[(i) * L + lane]
#define b_pb(i) b_pb_[(i) * L + lane]
#define b_pnb(i) b_pnb_[(i) * L + lane]
#define s_node(i) s_node_[(i) * L + lane]
#define s_pb(i) s_pb_[(i) * L + lane]
#define s_pnb(i) s_pnb_[(i) * L + lane]
    int* nd = p.nodes + ((long)o0 * B + r) * NI;          // node m: [parent, label, tstep, child_1..child_{C-1}]
    const float* lp = p.logp + o0 * C;
    nd[0] = -1; nd[1] = 0; nd[2] = -1;
    for (int c = 1; c < C; ++c) nd[2 + c] = -1;
    int n_nodes = 1, nb = 1;
    b_node(0) = 0; b_pb(0) = 0.0f; b_pnb(0) = -INFINITY;
    for (int t = 0; t < T; ++t) {
        const float* row = lp + (long)t * C;
        int nc = 0;
        auto find = [&](long key) {
            for (int q = 0; q < nc; ++q)
                if (c_key(q) == key) return q;
            c_key(nc) = key; c_pb(nc) = -INFINITY; c_pnb(nc) = -INFINITY;
            return nc++;
        };
        for (int e = 0; e < nb; ++e) {
            const int n = b_node(e);
            const float tot = lse2_g(b_pb(e), b_pnb(e));
            int idx = find((long)n);
            c_pb(idx) = lse2_g(c_pb(idx), tot + row[0]);
            const int lab_n = nd[(long)n * NI + 1];
            for (int c = 1; c < C; ++c) {
                if (!(row[c] >= p.lthr)) continue;
                float contrib;
                if (n != 0 && lab_n == c) {
                    idx = find((long)n);
                    c_pnb(idx) = lse2_g(c_pnb(idx), b_pnb(e) + row[c]);
                    contrib = b_pb(e) + row[c];
                } else {
                    contrib = tot + row[c];
                }
                const int ch = nd[(long)n * NI + 2 + c];
                const long key = ch >= 0 ? (long)ch : -((long)n * 8 + c) - 1;
                idx = find(key);
                c_pnb(idx) = lse2_g(c_pnb(idx), contrib);
            }
        }
        for (int i = 0; i < nc; ++i) { sc(i) = lse2_g(c_pb(i), c_pnb(i)); used(i) = 0; }
        int nn = 0;
        for (int k = 0; k < B && k < nc; ++k) {
            int bi = -1;
            float best = -INFINITY;
            for (int i = 0; i < nc; ++i) {
                const float v = sc(i);
                if (!used(i) && v > -INFINITY && (bi < 0 || v > best)) { bi = i; best = v; }
            }
            if (bi < 0) break;
            used(bi) = 1;
            int node;
            const long kb = c_key(bi);
            if (kb >= 0) node = (int)kb;
            else {
                const long pk = -(kb + 1);
                const int pn = (int)(pk / 8), c = (int)(pk % 8);
                node = n_nodes++;
                int* m = nd + (long)node * NI;
                m[0] = pn; m[1] = c; m[2] = t;
                for (int cc = 1; cc < C; ++cc) m[2 + cc] = -1;
                nd[(long)pn * NI + 2 + c] = node;
            }
            s_node(nn) = node; s_pb(nn) = c_pb(bi); s_pnb(nn) = c_pnb(bi); ++nn;
        }
        const float shift = nn ? lse2_g(s_pb(0), s_pnb(0)) : 0.0f;
        nb = nn;
        for (int i = 0; i < nn; ++i) { b_node(i) = s_node(i); b_pb(i) = s_pb(i) - shift; b_pnb(i) = s_pnb(i) - shift; }
    }
    const int n = nb ? b_node(0) : 0;
    int len = 0;
    for (int m = n; m > 0; m = nd[(long)m * NI]) ++len;
    p.count[r] = len;
    int i = len - 1;
    for (int m = n; m > 0; m = nd[(long)m * NI], --i) {
        p.labels[o0 + i] = (int8_t)nd[(long)m * NI + 1];
        p.path[o0 + i] = nd[(long)m * NI + 2];
    }
#undef c_key
#undef c_pb
#undef c_pnb
#undef sc
#undef used
#undef b_node
#undef b_pb
#undef b_pnb
#undef s_node
#undef s_pb
#undef s_pnb
}

}  // namespace bh

size_t bh_k_ctc_beam_workspace(long total_steps, int R, int C, int beam_size) {
    return (size_t)(total_steps * beam_size + R) * (3 + C - 1) * sizeof(int) + 256;
}

int bh_k_ctc_prefix_beam(const float* logp, const long* offs, int R, int C, int beam_size, float threshold, void* workspace,
                         int8_t* labels, int* path, int* count, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(R > 0 && C >= 2 && C <= PB_MAXC, "ctc_beam: need R > 0 and 2 <= classes <= 8 (R=%d C=%d)", R, C);
    BH_REQUIRE(beam_size >= 1 && beam_size <= PB_MAXB, "ctc_beam: beam_size must be in 1..16 (got %d)", beam_size);
    BH_REQUIRE(threshold >= 0.0f && threshold < 1.0f, "ctc_beam: threshold must be in [0, 1)");
    PbArgs a{logp, offs, R, C, beam_size, threshold > 0.0f ? logf(threshold) : -INFINITY, (int*)workspace, labels, path, count};
    // bytes of LDS per lane: five candidate arrays of B * C entries (one of them 8 bytes wide) + six beam arrays of B entries
    const size_t per_lane = (size_t)beam_size * C * 24 + (size_t)beam_size * 24;
    int L = 64;
    while (L > 1 && L * per_lane > 60 * 1024) L >>= 1;
    hipLaunchKernelGGL(ctc_prefix_beam_kernel, dim3((R + L - 1) / L), dim3(64), L * per_lane, stream, a, L);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}
