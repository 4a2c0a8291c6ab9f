/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the CTC-CRF arithmetic on the reference's hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's shared object
 * (oracle/liboracle.so); the product path (bonito_amd/, libbonito_hip.so) never does.
 *
 * What it restates (paths relative to /root/reference):
 *   - CTC_CRF.idx            bonito/crf/model.py:37-42   idx[j][0] = j, idx[j][1+r] = r*S/4 + j/4
 *   - CTC_CRF.logZ           bonito/crf/model.py:47-52   alpha_0 = 0, alpha_{t+1}[j] = (+)_k Ms[t][j][k] (x) alpha_t[idx[j][k]]
 *   - CTC_CRF.viterbi        bonito/crf/model.py:98-103  Max-semiring dlogZ/dMs one-hot -> move = (a%5)!=0, base = 1+(a//5)%4
 *   - LinearCRFEncoder blank expansion  bonito/nn.py:291-297  (4S "koi" layout + scalar blank <-> 5S layout)
 * The semiring kernels themselves (koi.ctc.logZ_cu_sparse etc., ont-koi==0.5.4, requirements.txt:19) are a
 * third-party dependency absent from /root/reference: their published algorithm (sparse forward scan over
 * idx) is what is restated.  PARITY UNPINNED by the reference's own tests (it has none on this path,
 * test/test_cli.py, test/test_download.py); pinned here against exhaustive path enumeration
 * (tests/test_oracle_crf.py) and a torch autograd restatement of crf/model.py:98-103.
 *
 * Tie-breaking (defined here and mirrored by the HIP kernel): lowest k first, then lowest j.
 * Arithmetic: fp16 inputs widened to fp32, one fp32 add per step -> bit-reproducible.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float h2f(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal */
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static int ipow4(int n) { int s = 1; while (n-- > 0) s *= 4; return s; }

/* Ms[t][j][k] accessor for both layouts; sc points at scores of chunk n, step t. */
static inline float ms(const uint16_t* sc, int layout_5s, float blank, int j, int k) {
    if (layout_5s) return h2f(sc[j * 5 + k]);
    return k == 0 ? blank : h2f(sc[j * 4 + (k - 1)]);
}

/* Viterbi best path.  scores: fp16 bits, element strides (s_n, s_t).  moves/path: [N][T] int8.
 * best: [N] fp32 or NULL.  Returns 0, or -1 on allocation failure. */
int oracle_crf_viterbi(const uint16_t* scores, int N, int T, int state_len, int layout_5s, float blank,
                       long s_n, long s_t, int8_t* moves, int8_t* path, float* best) {
    const int S = ipow4(state_len), q = S / 4;
    float* a0 = (float*)malloc(sizeof(float) * S);
    float* a1 = (float*)malloc(sizeof(float) * S);
    uint8_t* bp = (uint8_t*)malloc((size_t)T * S);
    if (!a0 || !a1 || !bp) { free(a0); free(a1); free(bp); return -1; }
    for (int n = 0; n < N; ++n) {
        for (int j = 0; j < S; ++j) a0[j] = 0.0f;
        for (int t = 0; t < T; ++t) {
            const uint16_t* sc = scores + (long)n * s_n + (long)t * s_t;
            for (int j = 0; j < S; ++j) {
                float b = a0[j] + ms(sc, layout_5s, blank, j, 0);
                int bk = 0;
                for (int r = 0; r < 4; ++r) {
                    float c = a0[r * q + (j >> 2)] + ms(sc, layout_5s, blank, j, 1 + r);
                    if (c > b) { b = c; bk = 1 + r; }
                }
                a1[j] = b;
                bp[(size_t)t * S + j] = (uint8_t)bk;
            }
            float* tmp = a0; a0 = a1; a1 = tmp;
        }
        int st = 0;
        float bs = a0[0];
        for (int j = 1; j < S; ++j) if (a0[j] > bs) { bs = a0[j]; st = j; }
        if (best) best[n] = bs;
        for (int t = T - 1; t >= 0; --t) {
            int k = bp[(size_t)t * S + st];
            moves[(long)n * T + t] = (int8_t)(k != 0);
            path[(long)n * T + t] = (int8_t)(k != 0 ? 1 + (st & 3) : 0);
            if (k != 0) st = (k - 1) * q + (st >> 2);
        }
    }
    free(a0); free(a1); free(bp);
    return 0;
}

static inline float lse2_libm(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    float m = a > b ? a : b;
    return m + logf(expf(a - m) + expf(b - m));
}

/* Log-semiring partition function per chunk (CTC_CRF.logZ, crf/model.py:47-52).  out: [N] fp32. */
int oracle_crf_logz(const uint16_t* scores, int N, int T, int state_len, int layout_5s, float blank,
                    long s_n, long s_t, float* out) {
    const int S = ipow4(state_len), q = S / 4;
    double* a0 = (double*)malloc(sizeof(double) * S);
    double* a1 = (double*)malloc(sizeof(double) * S);
    if (!a0 || !a1) { free(a0); free(a1); return -1; }
    for (int n = 0; n < N; ++n) {
        for (int j = 0; j < S; ++j) a0[j] = 0.0;
        for (int t = 0; t < T; ++t) {
            const uint16_t* sc = scores + (long)n * s_n + (long)t * s_t;
            for (int j = 0; j < S; ++j) {
                double v[5], m;
                v[0] = a0[j] + ms(sc, layout_5s, blank, j, 0);
                m = v[0];
                for (int r = 0; r < 4; ++r) {
                    v[1 + r] = a0[r * q + (j >> 2)] + ms(sc, layout_5s, blank, j, 1 + r);
                    if (v[1 + r] > m) m = v[1 + r];
                }
                double s = 0.0;
                for (int k = 0; k < 5; ++k) s += exp(v[k] - m);
                a1[j] = m + log(s);
            }
            double* tmp = a0; a0 = a1; a1 = tmp;
        }
        double m = a0[0];
        for (int j = 1; j < S; ++j) if (a0[j] > m) m = a0[j];
        double s = 0.0;
        for (int j = 0; j < S; ++j) s += exp(a0[j] - m);
        out[n] = (float)(m + log(s));
    }
    free(a0); free(a1);
    return 0;
}

/* =================================================================================================
 * Beam-search decode ("BS-2"; round 5, BS-1 = the same search with a Log-semiring table-lse2 guide).  koi.decode.beam_search
 * (ont-koi==0.5.4, requirements.txt:19; call site bonito/crf/basecall.py:36-40) is a closed third-party dependency: its exact
 * merging / q-score rules are NOT available in /root/reference, so the algorithm below is OUR definition (DESIGN.md "Beam search
 * BS-2"), PARITY UNPINNED against koi; the HIP kernels (bonito_amd/csrc/beam.hip) implement exactly this and are tested
 * against it bit for bit (sequence, moves) / within 1e-3 (q-scores).
 *
 * Scores: fp16 koi layout [N][T][4S]; entry s'*4 + r = move INTO state s' having dropped base r; stay
 * (blank) score is the scalar `blank`.
 *
 *  1. backward guide in the LINEAR domain (oracle_bs2_backward): E = bs2_exp(score) - a deterministic fp32 exponential (polynomial +
 *     exponent bits, include/bh_bs2.h) - and
 *        b_T[s] = 1
 *        raw_t[s] = fma(E_3, b[s'_3], fma(E_2, b[s'_2], fma(E_1, b[s'_1], fma(E_0, b[s'_0], fma(e^blank, b[s], 2^-60))))),  b = b_{t+1}
 *        s'_x = ((s << 2) | x) & (S-1),  E_x = bs2_exp(score[t][s'_x*4 + (s >> 2(k-1))])
 *        b_t[s] = raw_t[s] * 2^-e,  e = exponent of max_s raw_t[s]      (an exact scaling: the row's maximum lands in [1, 2))
 *     stored for t = 0..T ([T+1][S] fp32). Only ratios inside a row matter to the search, so no offsets are kept.
 *  2. class posteriors P_u[x] = sum_{s & 3 == x} alpha_u[s] beta_u[s] / sum_s alpha_u[s] beta_u[s], u = 1..T: here in fp64 with libm
 *     (oracle_crf_posteriors_f64 - the TRUE posteriors of the model; the HIP kernels run a linear-domain fp32 scan and are held to
 *     1e-3 on the q-scores derived from them).
 *  3. beam search over (state, sequence hash): stay / 4 moves per element (candidate order: the stays by element, then the moves by
 *     element and base), merge a move into the stay that spells the same sequence (table lse2), rank by score + bs2_log(b_{t+1}[state]), cut at best - log(beam_cut),
 *     keep `beam_width` (ties: lower candidate index), slots in candidate-index order.
 *  4. traceback from the best final element; per emitted base x with dwell u = t+1..t':
 *        err = mean_u sum_{y != x} P_u[y];  q = -10 log10(max(err, 1e-10)) * scale + offset, clamped to [1, 50]
 *        qstring = 33 + floor(q + 0.5) at the emitting step.
 *
 *  oracle_crf_backward / oracle_crf_forward_post below are the Log-semiring (table lse2) scans of round 1-4: they still define
 *  bh_crf_logz and the posterior-Viterbi decoder (D1), whose kernels are unchanged.
 * ================================================================================================= */
#include "../include/bh_lse_table.h"
static const float LSE_TAB[BH_LSE_TABLE_SIZE] = {BH_LSE_TABLE_VALUES};

float oracle_lse2(float a, float b) {
    float m = a > b ? a : b;
    if (m == -INFINITY) return m;
    float d = fabsf(a - b);
    if (!(d < BH_LSE_RANGE)) return m;
    float x = d * BH_LSE_SCALE;
    int i = (int)x;
    float f = x - (float)i;
    float t0 = LSE_TAB[i];
    float sp = fmaf(f, LSE_TAB[i + 1] - t0, t0);
    return m + sp;
}

/* beta~ [N][T+1][S] fp32, Bcum [N][T+1] double (B_t), logZ [N] double */
int oracle_crf_backward(const uint16_t* scores, int N, int T, int state_len, float blank, float* beta,
                        double* Bcum, double* logZ) {
    const int S = ipow4(state_len), sh = 2 * (state_len - 1);
    float* raw = (float*)malloc(sizeof(float) * S);
    float* nxt = (float*)malloc(sizeof(float) * S);
    if (!raw || !nxt) { free(raw); free(nxt); return -1; }
    for (int n = 0; n < N; ++n) {
        float* bn = beta + (size_t)n * (T + 1) * S;
        double* Bn = Bcum + (size_t)n * (T + 1);
        for (int s = 0; s < S; ++s) { raw[s] = 0.0f; bn[(size_t)T * S + s] = 0.0f; }
        Bn[T] = 0.0;
        double cum = 0.0;   /* sum of raw_u[0] for u > t */
        for (int t = T - 1; t >= 0; --t) {
            const uint16_t* sc = scores + ((size_t)n * T + t) * 4 * S;
            const float ref = raw[0];
            cum += (double)ref;                 /* raw_{t+1}[0] joins B_t */
            Bn[t] = cum;
            for (int s = 0; s < S; ++s) {
                float acc = blank + (raw[s] - ref);
                const int lead = s >> sh;
                for (int x = 0; x < 4; ++x) {
                    int s2 = ((s << 2) | x) & (S - 1);
                    acc = oracle_lse2(acc, h2f(sc[s2 * 4 + lead]) + (raw[s2] - ref));
                }
                nxt[s] = acc;
            }
            float* tmp = raw; raw = nxt; nxt = tmp;
            for (int s = 0; s < S; ++s) bn[(size_t)t * S + s] = raw[s] - raw[0];
        }
        /* logZ = B_0 + raw_0[0] + LSE_s beta~_0[s]  (alpha_0 = 0) */
        double m = -INFINITY, sum = 0.0;
        for (int s = 0; s < S; ++s) if (bn[s] > m) m = bn[s];
        for (int s = 0; s < S; ++s) sum += exp((double)bn[s] - m);
        logZ[n] = cum + (double)raw[0] + m + log(sum);
    }
    free(raw); free(nxt);
    return 0;
}

/* class posteriors P [N][T][4] fp32 (boundary u = t+1 stored at index t) */
int oracle_crf_forward_post(const uint16_t* scores, int N, int T, int state_len, float blank, const float* beta,
                            const double* Bcum, const double* logZ, float* P) {
    const int S = ipow4(state_len), q = S / 4;
    float* raw = (float*)malloc(sizeof(float) * S);
    float* nxt = (float*)malloc(sizeof(float) * S);
    if (!raw || !nxt) { free(raw); free(nxt); return -1; }
    for (int n = 0; n < N; ++n) {
        const float* bn = beta + (size_t)n * (T + 1) * S;
        const double* Bn = Bcum + (size_t)n * (T + 1);
        for (int s = 0; s < S; ++s) raw[s] = 0.0f;
        double A = 0.0;   /* sum of raw_v[0] for v <= u: alpha_u = alpha~_u + A_u */
        for (int t = 0; t < T; ++t) {
            const uint16_t* sc = scores + ((size_t)n * T + t) * 4 * S;
            const float ref = raw[0];
            A += (double)ref;
            for (int j = 0; j < S; ++j) {
                float acc = blank + (raw[j] - ref);
                for (int r = 0; r < 4; ++r)
                    acc = oracle_lse2(acc, h2f(sc[j * 4 + r]) + (raw[r * q + (j >> 2)] - ref));
                nxt[j] = acc;
            }
            float* tmp = raw; raw = nxt; nxt = tmp;
            /* boundary u = t+1: alpha_u[s] = (raw[s]-raw[0]) + raw[0] + A ; beta_u[s] = beta~_u[s] + B_u' where
               the backward pass defines beta_u = beta~_u + raw^b_u[0] + B_u and B_{u-1} = B_u + raw^b_u[0]. */
            const double norm = logZ[n] - (A + (double)raw[0]) - Bn[t];   /* Bn[t] = B_t = B_{t+1} + raw^b_{t+1}[0] */
            double cls[4] = {0, 0, 0, 0};
            for (int s = 0; s < S; ++s)
                cls[s & 3] += exp((double)(raw[s] - raw[0]) + (double)bn[(size_t)(t + 1) * S + s] - norm);
            for (int x = 0; x < 4; ++x) P[((size_t)n * T + t) * 4 + x] = (float)cls[x];
        }
    }
    free(raw); free(nxt);
    return 0;
}

/* ---- BS-2: deterministic exp / log (include/bh_bs2.h) and the linear-domain guide scan -------------------------------------------- */
#include "../include/bh_bs2.h"
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float oracle_bs2_exp(float x) {
    x = fminf(fmaxf(x, -BH_BS2_XMAX), BH_BS2_XMAX);
    const float t = fmaf(x, BH_BS2_LOG2E, BH_BS2_MAGIC);
    const float n = t - BH_BS2_MAGIC;
    const float f = fmaf(x, BH_BS2_LOG2E, -n);
    float p = BH_BS2_E5;
    p = fmaf(p, f, BH_BS2_E4);
    p = fmaf(p, f, BH_BS2_E3);
    p = fmaf(p, f, BH_BS2_E2);
    p = fmaf(p, f, BH_BS2_E1);
    p = fmaf(p, f, BH_BS2_E0);
    return u2f(f2u(p) + ((f2u(t) - (uint32_t)BH_BS2_MAGIC_BITS) << 23));
}

/* ln v for a positive NORMAL fp32 (absolute error ~1e-5: include/bh_bs2.h) */
float oracle_bs2_log(float v) {
    const uint32_t u = f2u(v);
    const float ef = (float)(int)(u >> 23);                         /* biased exponent field; the bias sits in BH_BS2_LOGC */
    const float z = u2f((u & 0x007fffffu) | 0x3f800000u) - 1.0f;    /* mantissa - 1 in [0, 1) */
    float q = BH_BS2_L5;
    q = fmaf(q, z, BH_BS2_L4);
    q = fmaf(q, z, BH_BS2_L3);
    q = fmaf(q, z, BH_BS2_L2);
    q = fmaf(q, z, BH_BS2_L1);
    q = fmaf(q, z, BH_BS2_L0);
    return fmaf(ef, BH_BS2_LN2, fmaf(z, q, BH_BS2_LOGC));
}

/* b [N][T+1][S] fp32: the linear-domain guide, every row scaled (exactly, by a power of two) so that its maximum lies in [1, 2) */
int oracle_bs2_backward(const uint16_t* scores, int N, int T, int state_len, float blank, float* b) {
    const int S = ipow4(state_len), sh = 2 * (state_len - 1);
    float* raw = (float*)malloc(sizeof(float) * S);
    if (!raw) return -1;
    const float eb = oracle_bs2_exp(blank);
    for (int n = 0; n < N; ++n) {
        float* bn = b + (size_t)n * (T + 1) * S;
        for (int s = 0; s < S; ++s) bn[(size_t)T * S + s] = 1.0f;
        for (int t = T - 1; t >= 0; --t) {
            const uint16_t* sc = scores + ((size_t)n * T + t) * 4 * S;
            const float* prev = bn + (size_t)(t + 1) * S;
            float mx = 0.0f;
            for (int s = 0; s < S; ++s) {
                float acc = fmaf(eb, prev[s], BH_BS2_TINY);
                const int lead = s >> sh;
                for (int x = 0; x < 4; ++x) {
                    const int s2 = ((s << 2) | x) & (S - 1);
                    acc = fmaf(oracle_bs2_exp(h2f(sc[s2 * 4 + lead])), prev[s2], acc);
                }
                raw[s] = acc;
                if (acc > mx) mx = acc;
            }
            const uint32_t eb23 = ((f2u(mx) >> 23) - 127u) << 23;          /* exponent of the maximum, in place */
            for (int s = 0; s < S; ++s) bn[(size_t)t * S + s] = u2f(f2u(raw[s]) - eb23);
        }
    }
    free(raw);
    return 0;
}

/* TRUE class posteriors P [N][T][4] (boundary u = t+1 stored at index t): forward and backward in fp64, log domain, libm. Scores (and
 * the blank score) are clamped to [-XMAX, XMAX] like the guide's: that is part of BS-2's definition; a trained head stays within +-5. */
static inline double clampx(float x) { return (double)fminf(fmaxf(x, -BH_BS2_XMAX), BH_BS2_XMAX); }
int oracle_crf_posteriors_f64(const uint16_t* scores, int N, int T, int state_len, float blank_in, float* P) {
    const int S = ipow4(state_len), sh = 2 * (state_len - 1), q = S / 4;
    const double blank = clampx(blank_in);
    double* beta = (double*)malloc(sizeof(double) * (size_t)(T + 1) * S);
    double* al = (double*)malloc(sizeof(double) * S);
    double* nx = (double*)malloc(sizeof(double) * S);
    if (!beta || !al || !nx) { free(beta); free(al); free(nx); return -1; }
    for (int n = 0; n < N; ++n) {
        for (int s = 0; s < S; ++s) beta[(size_t)T * S + s] = 0.0;
        for (int t = T - 1; t >= 0; --t) {
            const uint16_t* sc = scores + ((size_t)n * T + t) * 4 * S;
            const double* prev = beta + (size_t)(t + 1) * S;
            double* cur = beta + (size_t)t * S;
            for (int s = 0; s < S; ++s) {
                const int lead = s >> sh;
                double v[5], m;
                v[0] = blank + prev[s];
                m = v[0];
                for (int x = 0; x < 4; ++x) {
                    const int s2 = ((s << 2) | x) & (S - 1);
                    v[1 + x] = clampx(h2f(sc[s2 * 4 + lead])) + prev[s2];
                    if (v[1 + x] > m) m = v[1 + x];
                }
                double sum = 0.0;
                for (int k = 0; k < 5; ++k) sum += exp(v[k] - m);
                cur[s] = m + log(sum);
            }
        }
        for (int s = 0; s < S; ++s) al[s] = 0.0;
        for (int t = 0; t < T; ++t) {
            const uint16_t* sc = scores + ((size_t)n * T + t) * 4 * S;
            for (int j = 0; j < S; ++j) {
                double v[5], m;
                v[0] = blank + al[j];
                m = v[0];
                for (int r = 0; r < 4; ++r) {
                    v[1 + r] = clampx(h2f(sc[j * 4 + r])) + al[r * q + (j >> 2)];
                    if (v[1 + r] > m) m = v[1 + r];
                }
                double sum = 0.0;
                for (int k = 0; k < 5; ++k) sum += exp(v[k] - m);
                nx[j] = m + log(sum);
            }
            double* tmp = al; al = nx; nx = tmp;
            const double* bu = beta + (size_t)(t + 1) * S;
            double mx = -INFINITY;
            for (int s = 0; s < S; ++s) if (al[s] + bu[s] > mx) mx = al[s] + bu[s];
            double cls[4] = {0, 0, 0, 0}, tot = 0.0;
            for (int s = 0; s < S; ++s) { const double pv = exp(al[s] + bu[s] - mx); cls[s & 3] += pv; tot += pv; }
            for (int x = 0; x < 4; ++x) P[((size_t)n * T + t) * 4 + x] = (float)(cls[x] / tot);
        }
    }
    free(beta); free(al); free(nx);
    return 0;
}

static inline uint32_t bs_hash0(int s) { return ((uint32_t)s + 1u) * 2654435761u; }
static inline uint32_t bs_mix(uint32_t h, int x) {
    h = (h ^ ((uint32_t)x + 1u)) * 16777619u;
    return h ^ (h >> 15);
}
static inline uint32_t bs_ukey(float k) {
    uint32_t u;
    memcpy(&u, &k, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

#define BS_MAXW 32
typedef struct { int state; uint32_t hash; float score; } bs_elem;

/* Full decode.  sequence/qstring/moves: [N][T] int8 (0 where nothing is emitted).  qfloat: [N][T] fp32
 * un-rounded q at emitting steps (0 elsewhere) or NULL.  Returns 0 / -1. */
int oracle_beam_search(const uint16_t* scores, int N, int T, int state_len, int beam_width, float beam_cut,
                       float blank, float q_scale, float q_offset, int8_t* sequence, int8_t* qstring,
                       int8_t* moves, float* qfloat) {
    const int S = ipow4(state_len), sh = 2 * (state_len - 1);
    if (beam_width < 1 || beam_width > BS_MAXW) return -2;
    const int W = beam_width;
    float* beta = (float*)malloc(sizeof(float) * (size_t)N * (T + 1) * S);        /* the linear-domain guide b */
    float* P = (float*)malloc(sizeof(float) * (size_t)N * T * 4);
    uint8_t* bp = (uint8_t*)malloc((size_t)T * BS_MAXW);
    if (!beta || !P || !bp) { free(beta); free(P); free(bp); return -1; }
    if (oracle_bs2_backward(scores, N, T, state_len, blank, beta) || oracle_crf_posteriors_f64(scores, N, T, state_len, blank, P)) {
        free(beta); free(P); free(bp); return -1;
    }
    const float cut = logf(beam_cut);
    for (int n = 0; n < N; ++n) {
        const float* bn = beta + (size_t)n * (T + 1) * S;
        bs_elem beam[BS_MAXW];
        int nb = 0;
        /* init: top-W states by b_0 (ties: lower state), slots in state order */
        {
            char* used = (char*)calloc(S, 1);
            int take = W < S ? W : S;
            for (int k = 0; k < take; ++k) {
                int bi = -1;
                for (int s = 0; s < S; ++s)
                    if (!used[s] && (bi < 0 || bs_ukey(bn[s]) > bs_ukey(bn[bi]))) bi = s;
                used[bi] = 1;
            }
            for (int s = 0; s < S; ++s)
                if (used[s]) { beam[nb].state = s; beam[nb].hash = bs_hash0(s); beam[nb].score = 0.0f; ++nb; }
            free(used);
        }
        for (int t = 0; t < T; ++t) {
            const uint16_t* sc = scores + ((size_t)n * T + t) * 4 * S;
            const float* b1 = bn + (size_t)(t + 1) * S;
            /* candidates in THIS order (it decides ties and the slots of the new beam): the stays, c = e, then the moves, c = 32 + 4 e + x
               (round 5: the kernels keep the stays of all elements in one lane set and the 128 moves in two - a merge costs one lse2 per
               lane instead of three; rounds 1-4 interleaved them as c = 5 e + j) */
            int c_state[BS_MAXW * 5];
            uint32_t c_hash[BS_MAXW * 5];
            float c_score[BS_MAXW * 5], c_key[BS_MAXW * 5];
            uint8_t c_info[BS_MAXW * 5];   /* parent | move<<5 | base<<6 */
            char c_alive[BS_MAXW * 5];
            const int nc = BS_MAXW * 5;
            memset(c_alive, 0, sizeof(c_alive));
            for (int c = 0; c < nc; ++c) { c_state[c] = 0; c_hash[c] = 0; c_score[c] = -INFINITY; c_info[c] = 0; }
            for (int e = 0; e < nb; ++e) {
                const int s = beam[e].state, lead = s >> sh;
                c_state[e] = s; c_hash[e] = beam[e].hash; c_score[e] = beam[e].score + blank;
                c_info[e] = (uint8_t)e; c_alive[e] = 1;
                for (int x = 0; x < 4; ++x) {
                    const int c = BS_MAXW + e * 4 + x, s2 = ((s << 2) | x) & (S - 1);
                    c_state[c] = s2; c_hash[c] = bs_mix(beam[e].hash, x);
                    c_score[c] = beam[e].score + h2f(sc[s2 * 4 + lead]);
                    c_info[c] = (uint8_t)(e | (1 << 5) | (x << 6)); c_alive[c] = 1;
                }
            }
            /* merge a move into the stay that spells the same sequence (the first such stay) */
            for (int e = 0; e < nb; ++e)
                for (int x = 0; x < 4; ++x) {
                    const int c = BS_MAXW + e * 4 + x;
                    for (int d = 0; d < nb; ++d) {
                        if (c_hash[d] == c_hash[c] && c_state[d] == c_state[c]) {
                            const float stay_sc = beam[d].score + blank;   /* un-merged stay score */
                            if (c_score[c] > stay_sc) c_info[d] = c_info[c];
                            c_score[d] = oracle_lse2(stay_sc, c_score[c]);
                            c_alive[c] = 0;
                            break;
                        }
                    }
                }
            float best = -INFINITY;
            for (int c = 0; c < nc; ++c) {
                c_key[c] = c_alive[c] ? c_score[c] + oracle_bs2_log(b1[c_state[c]]) : -INFINITY;
                if (c_key[c] > best) best = c_key[c];
            }
            const float thr = best - cut;
            for (int c = 0; c < nc; ++c) if (c_alive[c] && c_key[c] < thr) { c_alive[c] = 0; c_key[c] = -INFINITY; }
            /* select top W by key (ties: lower index) */
            char sel[BS_MAXW * 5];
            memset(sel, 0, sizeof(sel));
            int nsel = 0;
            for (int k = 0; k < W; ++k) {
                int bi = -1;
                for (int c = 0; c < nc; ++c)
                    if (c_alive[c] && !sel[c] && (bi < 0 || bs_ukey(c_key[c]) > bs_ukey(c_key[bi]))) bi = c;
                if (bi < 0) break;
                sel[bi] = 1; ++nsel;
            }
            /* new beam in candidate-index order; renormalise by the best element's score */
            int bestc = -1;
            for (int c = 0; c < nc; ++c)
                if (sel[c] && (bestc < 0 || bs_ukey(c_key[c]) > bs_ukey(c_key[bestc]))) bestc = c;
            const float shift = c_score[bestc];
            nb = 0;
            for (int c = 0; c < nc; ++c)
                if (sel[c]) {
                    beam[nb].state = c_state[c]; beam[nb].hash = c_hash[c]; beam[nb].score = c_score[c] - shift;
                    bp[(size_t)t * BS_MAXW + nb] = c_info[c];
                    ++nb;
                }
            (void)nsel;
        }
        /* best final element (b_T = 1 -> key = score); ties: lower slot */
        int r = 0;
        for (int e = 1; e < nb; ++e) if (bs_ukey(beam[e].score) > bs_ukey(beam[r].score)) r = e;
        int8_t* sq = sequence + (size_t)n * T;
        int8_t* qs = qstring + (size_t)n * T;
        int8_t* mv = moves + (size_t)n * T;
        float* qf = qfloat ? qfloat + (size_t)n * T : NULL;
        const float* Pn = P + (size_t)n * T * 4;
        for (int t = T - 1; t >= 0; --t) {
            const uint8_t info = bp[(size_t)t * BS_MAXW + r];
            const int is_move = (info >> 5) & 1;
            mv[t] = (int8_t)is_move;
            sq[t] = is_move ? (int8_t)"ACGT"[info >> 6] : 0;
            qs[t] = 0;
            if (qf) qf[t] = 0.0f;
            r = info & 31;
        }
        for (int t = 0; t < T; ++t) {
            if (!mv[t]) continue;
            int t2 = t + 1;
            while (t2 < T && !mv[t2]) ++t2;          /* dwell covers boundaries u = t+1 .. t2 (indices t .. t2-1) */
            const int x = sq[t] == 'A' ? 0 : sq[t] == 'C' ? 1 : sq[t] == 'G' ? 2 : 3;
            double err = 0.0;
            for (int u = t; u < t2; ++u)
                for (int y = 0; y < 4; ++y) if (y != x) err += (double)Pn[(size_t)u * 4 + y];
            err /= (double)(t2 - t);
            if (err < 1e-10) err = 1e-10;
            float qv = (float)(-10.0 * log10(err)) * q_scale + q_offset;
            if (qv < 1.0f) qv = 1.0f;
            if (qv > 50.0f) qv = 50.0f;
            if (qf) qf[t] = qv;
            qs[t] = (int8_t)(33 + (int)floorf(qv + 0.5f));
        }
    }
    free(beta); free(P); free(bp);
    return 0;
}

/* ---- BS-1 (the beam search of rounds 1-4), kept as a QUALITY REFERENCE for BS-2 only (tests/test_oracle_crf.py, bench.py `parity`):
 * the same search with the Log-semiring table-lse2 guide (oracle_crf_backward) and the candidate order c = 5 e + j. Sequence and moves
 * only. The product decoder is BS-2; this function exists so that a redefinition of the decoder cannot quietly cost sequence quality
 * (review, round 5): oracle_seq_logprob_f64 below scores both answers with the model's exact path sum. */
int oracle_beam_search_bs1(const uint16_t* scores, int N, int T, int state_len, int beam_width, float beam_cut,
                           float blank, int8_t* sequence, int8_t* moves) {
    const int S = ipow4(state_len), sh = 2 * (state_len - 1);
    if (beam_width < 1 || beam_width > BS_MAXW) return -2;
    const int W = beam_width;
    float* beta = (float*)malloc(sizeof(float) * (size_t)N * (T + 1) * S);
    double* Bcum = (double*)malloc(sizeof(double) * (size_t)N * (T + 1));
    double* logZ = (double*)malloc(sizeof(double) * N);
    uint8_t* bp = (uint8_t*)malloc((size_t)T * BS_MAXW);
    if (!beta || !Bcum || !logZ || !bp) { free(beta); free(Bcum); free(logZ); free(bp); return -1; }
    oracle_crf_backward(scores, N, T, state_len, blank, beta, Bcum, logZ);
    const float cut = logf(beam_cut);
    for (int n = 0; n < N; ++n) {
        const float* bn = beta + (size_t)n * (T + 1) * S;
        bs_elem beam[BS_MAXW];
        int nb = 0;
        {
            char* used = (char*)calloc(S, 1);
            int take = W < S ? W : S;
            for (int k = 0; k < take; ++k) {
                int bi = -1;
                for (int s = 0; s < S; ++s)
                    if (!used[s] && (bi < 0 || bs_ukey(bn[s]) > bs_ukey(bn[bi]))) bi = s;
                used[bi] = 1;
            }
            for (int s = 0; s < S; ++s)
                if (used[s]) { beam[nb].state = s; beam[nb].hash = bs_hash0(s); beam[nb].score = 0.0f; ++nb; }
            free(used);
        }
        for (int t = 0; t < T; ++t) {
            const uint16_t* sc = scores + ((size_t)n * T + t) * 4 * S;
            const float* b1 = bn + (size_t)(t + 1) * S;
            int c_state[BS_MAXW * 5];
            uint32_t c_hash[BS_MAXW * 5];
            float c_score[BS_MAXW * 5], c_key[BS_MAXW * 5];
            uint8_t c_info[BS_MAXW * 5];
            char c_alive[BS_MAXW * 5];
            const int nc = nb * 5;
            for (int e = 0; e < nb; ++e) {
                const int s = beam[e].state, lead = s >> sh;
                c_state[e * 5] = s; c_hash[e * 5] = beam[e].hash; c_score[e * 5] = beam[e].score + blank;
                c_info[e * 5] = (uint8_t)e; c_alive[e * 5] = 1;
                for (int x = 0; x < 4; ++x) {
                    const int c = e * 5 + 1 + x, s2 = ((s << 2) | x) & (S - 1);
                    c_state[c] = s2; c_hash[c] = bs_mix(beam[e].hash, x);
                    c_score[c] = beam[e].score + h2f(sc[s2 * 4 + lead]);
                    c_info[c] = (uint8_t)(e | (1 << 5) | (x << 6)); c_alive[c] = 1;
                }
            }
            for (int e = 0; e < nb; ++e)
                for (int x = 0; x < 4; ++x) {
                    const int c = e * 5 + 1 + x;
                    for (int d = 0; d < nb; ++d) {
                        const int cs = d * 5;
                        if (c_hash[cs] == c_hash[c] && c_state[cs] == c_state[c]) {
                            const float stay_sc = beam[d].score + blank;
                            if (c_score[c] > stay_sc) c_info[cs] = c_info[c];
                            c_score[cs] = oracle_lse2(stay_sc, c_score[c]);
                            c_alive[c] = 0;
                            break;
                        }
                    }
                }
            float best = -INFINITY;
            for (int c = 0; c < nc; ++c) {
                c_key[c] = c_alive[c] ? c_score[c] + b1[c_state[c]] : -INFINITY;
                if (c_key[c] > best) best = c_key[c];
            }
            const float thr = best - cut;
            for (int c = 0; c < nc; ++c) if (c_alive[c] && c_key[c] < thr) { c_alive[c] = 0; c_key[c] = -INFINITY; }
            char sel[BS_MAXW * 5];
            memset(sel, 0, sizeof(sel));
            for (int k = 0; k < W; ++k) {
                int bi = -1;
                for (int c = 0; c < nc; ++c)
                    if (c_alive[c] && !sel[c] && (bi < 0 || bs_ukey(c_key[c]) > bs_ukey(c_key[bi]))) bi = c;
                if (bi < 0) break;
                sel[bi] = 1;
            }
            int bestc = -1;
            for (int c = 0; c < nc; ++c)
                if (sel[c] && (bestc < 0 || bs_ukey(c_key[c]) > bs_ukey(c_key[bestc]))) bestc = c;
            const float shift = c_score[bestc];
            nb = 0;
            for (int c = 0; c < nc; ++c)
                if (sel[c]) {
                    beam[nb].state = c_state[c]; beam[nb].hash = c_hash[c]; beam[nb].score = c_score[c] - shift;
                    bp[(size_t)t * BS_MAXW + nb] = c_info[c];
                    ++nb;
                }
        }
        int r = 0;
        for (int e = 1; e < nb; ++e) if (bs_ukey(beam[e].score) > bs_ukey(beam[r].score)) r = e;
        int8_t* sq = sequence + (size_t)n * T;
        int8_t* mv = moves + (size_t)n * T;
        for (int t = T - 1; t >= 0; --t) {
            const uint8_t info = bp[(size_t)t * BS_MAXW + r];
            const int is_move = (info >> 5) & 1;
            mv[t] = (int8_t)is_move;
            sq[t] = is_move ? (int8_t)"ACGT"[info >> 6] : 0;
            r = info & 31;
        }
    }
    free(beta); free(Bcum); free(logZ); free(bp);
    return 0;
}

/* ---- The model's EXACT log-probability of a called sequence, fp64, independent of every decoder here.
 * The CTC-CRF of bonito/crf/model.py:30-108 in the koi layout: alpha_0[s] = 0 for every k-mer state s; per step a path either stays
 * (score `blank`) or moves from s to s' = ((s << 2) | x) & (S-1) emitting base x (score sc[t][4 s' + (s >> 2(k-1))]).
 *     logp = ln  sum over start states and over every alignment that emits exactly seq[0..len)  of exp(path score)
 *     logz = ln  sum over ALL paths                                                              (CTC_CRF.logZ, crf/model.py:47-52)
 * so logp - logz is ln P(sequence | scores), the quantity a sequence-level beam search maximises. After i >= k emissions the state is
 * the last k bases of the prefix; before that, the unfilled leading digits are those of the start state: level i < k keeps a dense
 * [S] row (only states whose low i digits spell seq[0..i) are ever finite), level i >= k one scalar. Cost T (len + k S).
 * seq: base indices 0..3. Scores are used as they are (no clamp): for heads within +-5 that is the model itself. */
static inline double lse2_f64(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}
int oracle_seq_logprob_f64(const uint16_t* scores, int T, int state_len, float blank_in, const int8_t* seq, int len,
                           double* logp, double* logz) {
    const int S = ipow4(state_len), sh = 2 * (state_len - 1), k = state_len, q = S / 4;
    const double blank = (double)blank_in;
    if (len < 0 || len > T) { *logp = -INFINITY; }
    /* logZ */
    double* al = (double*)malloc(sizeof(double) * S);
    double* nx = (double*)malloc(sizeof(double) * S);
    /* dense levels 0 .. k-1 ([k][S]), scalar levels k .. len */
    double* dense = (double*)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1) * S);
    double* tail = (double*)malloc(sizeof(double) * (size_t)(len + 1));
    int* st = (int*)malloc(sizeof(int) * (size_t)(len + 1));      /* state after i >= k emissions */
    if (!al || !nx || !dense || !tail || !st) { free(al); free(nx); free(dense); free(tail); free(st); return -1; }
    for (int s = 0; s < S; ++s) al[s] = 0.0;
    for (int t = 0; t < T; ++t) {
        const uint16_t* sc = scores + (size_t)t * 4 * S;
        for (int j = 0; j < S; ++j) {
            double v[5], m;
            v[0] = blank + al[j];
            m = v[0];
            for (int r = 0; r < 4; ++r) {
                v[1 + r] = (double)h2f(sc[j * 4 + r]) + al[r * q + (j >> 2)];
                if (v[1 + r] > m) m = v[1 + r];
            }
            double sum = 0.0;
            for (int c = 0; c < 5; ++c) sum += exp(v[c] - m);
            nx[j] = m + log(sum);
        }
        double* tmp = al; al = nx; nx = tmp;
    }
    {
        double m = -INFINITY, sum = 0.0;
        for (int s = 0; s < S; ++s) if (al[s] > m) m = al[s];
        for (int s = 0; s < S; ++s) sum += exp(al[s] - m);
        *logz = m + log(sum);
    }
    if (len < 0 || len > T) { free(al); free(nx); free(dense); free(tail); free(st); return 0; }
    /* the sequence's own forward pass */
    for (int i = 0; i < k; ++i)
        for (int s = 0; s < S; ++s) dense[(size_t)i * S + s] = (i == 0) ? 0.0 : -INFINITY;
    for (int i = 0; i <= len; ++i) tail[i] = -INFINITY;
    {
        int cur = 0;
        for (int i = 1; i <= len; ++i) {
            cur = ((cur << 2) | seq[i - 1]) & (S - 1);
            st[i] = cur;          /* meaningful for i >= k */
        }
        st[0] = 0;
    }
    for (int t = 0; t < T; ++t) {
        const uint16_t* sc = scores + (size_t)t * 4 * S;
        const int hi = len < t + 1 ? len : t + 1;
        for (int i = hi; i >= 0; --i) {           /* descending: level i reads the OLD level i - 1 */
            if (i >= k) {
                double stay = tail[i] + blank, mv = -INFINITY;
                if (i >= 1) {
                    const int s2 = st[i];
                    if (i - 1 >= k) {
                        const int sp = st[i - 1];
                        mv = tail[i - 1] + (double)h2f(sc[s2 * 4 + (sp >> sh)]);
                    } else {                        /* i - 1 == k - 1: the four predecessors differ in their leading digit */
                        const double* dp = dense + (size_t)(i - 1) * S;
                        for (int r = 0; r < 4; ++r) {
                            const int sp = r * q + (s2 >> 2);
                            mv = lse2_f64(mv, dp[sp] + (double)h2f(sc[s2 * 4 + r]));
                        }
                    }
                }
                tail[i] = lse2_f64(stay, mv);
            } else {
                double* di = dense + (size_t)i * S;
                if (i == 0) {
                    for (int s = 0; s < S; ++s) di[s] += blank;
                } else {
                    const double* dp = dense + (size_t)(i - 1) * S;
                    const int x = seq[i - 1];
                    for (int s2 = 0; s2 < S; ++s2) {
                        double v = di[s2] + blank;
                        if ((s2 & 3) == x)
                            for (int r = 0; r < 4; ++r) {
                                const int sp = r * q + (s2 >> 2);
                                v = lse2_f64(v, dp[sp] + (double)h2f(sc[s2 * 4 + r]));
                            }
                        di[s2] = v;
                    }
                }
            }
        }
    }
    if (len >= k) *logp = tail[len];
    else {
        double acc = -INFINITY;
        const double* dl = dense + (size_t)len * S;
        for (int s = 0; s < S; ++s) acc = lse2_f64(acc, dl[s]);
        *logp = acc;
    }
    free(al); free(nx); free(dense); free(tail); free(st);
    return 0;
}

/* =================================================================================================
 * CTC prefix beam search ("PB-1").  Replaces fast_ctc_decode.beam_search(probs, alphabet, beam_size=5,
 * beam_cut_threshold=1e-3) (bonito/ctc/model.py:44; Rust crate, un-pinned in requirements.txt:3, absent
 * from /root/reference -> its tie-breaking/normalisation cannot be pinned: PARITY UNPINNED, this is the
 * textbook algorithm it implements, in log space with the deterministic table lse2).
 *   beam entry = (prefix-tree node, log p_blank, log p_nonblank); per step, in beam order:
 *     blank          : same prefix, p_b  += (p_b + p_nb) * P[0]
 *     label c >= thr : if c == last label: same prefix p_nb += p_nb * P[c], extended prefix p_nb += p_b * P[c]
 *                      else               extended prefix p_nb += (p_b + p_nb) * P[c]
 *   candidates with the same prefix are merged; keep the `beam_size` best by p_b + p_nb (ties: first created).
 * Output: label indices (1..C-1) of the best prefix and, per label, the step at which its node entered the beam.
 * ================================================================================================= */
#define PB_MAXB 16
#define PB_MAXC 8
int oracle_ctc_prefix_beam(const float* lp, int T, int C, int beam_size, float threshold, int8_t* labels_out,
                           int* path_out, int* count_out) {
    if (beam_size < 1 || beam_size > PB_MAXB || C < 2 || C > PB_MAXC) return -2;
    const int max_nodes = 1 + T * beam_size;
    int* parent = (int*)malloc(sizeof(int) * max_nodes);
    int8_t* label = (int8_t*)malloc(max_nodes);
    int* tstep = (int*)malloc(sizeof(int) * max_nodes);
    int* child = (int*)malloc(sizeof(int) * (size_t)max_nodes * (PB_MAXC - 1));
    if (!parent || !label || !tstep || !child) { free(parent); free(label); free(tstep); free(child); return -1; }
    for (size_t i = 0; i < (size_t)max_nodes * (PB_MAXC - 1); ++i) child[i] = -1;
    parent[0] = -1; label[0] = 0; tstep[0] = -1;
    int n_nodes = 1;
    int b_node[PB_MAXB];
    float b_pb[PB_MAXB], b_pnb[PB_MAXB];
    int nb = 1;
    b_node[0] = 0; b_pb[0] = 0.0f; b_pnb[0] = -INFINITY;
    const float lthr = logf(threshold);
    for (int t = 0; t < T; ++t) {
        const float* row = lp + (size_t)t * C;
        /* candidate key: node id >= 0, or a not-yet-allocated child encoded as -(parent * 8 + c) - 1 */
        long c_key[PB_MAXB * PB_MAXC];
        float c_pb[PB_MAXB * PB_MAXC], c_pnb[PB_MAXB * PB_MAXC];
        int nc = 0;
#define PB_FIND(KEY, IDX)                                              \
        do {                                                           \
            IDX = -1;                                                  \
            for (int q_ = 0; q_ < nc; ++q_) if (c_key[q_] == (KEY)) { IDX = q_; break; } \
            if (IDX < 0) { IDX = nc++; c_key[IDX] = (KEY); c_pb[IDX] = -INFINITY; c_pnb[IDX] = -INFINITY; } \
        } while (0)
        for (int e = 0; e < nb; ++e) {
            const int n = b_node[e];
            const float tot = oracle_lse2(b_pb[e], b_pnb[e]);
            int idx;
            PB_FIND((long)n, idx);
            c_pb[idx] = oracle_lse2(c_pb[idx], tot + row[0]);
            for (int c = 1; c < C; ++c) {
                if (!(row[c] >= lthr)) continue;
                float contrib;
                if (n != 0 && label[n] == c) {
                    PB_FIND((long)n, idx);
                    c_pnb[idx] = oracle_lse2(c_pnb[idx], b_pnb[e] + row[c]);
                    contrib = b_pb[e] + row[c];
                } else {
                    contrib = tot + row[c];
                }
                const int ch = child[(size_t)n * (PB_MAXC - 1) + (c - 1)];
                const long key = ch >= 0 ? (long)ch : -((long)n * 8 + c) - 1;
                PB_FIND(key, idx);
                c_pnb[idx] = oracle_lse2(c_pnb[idx], contrib);
            }
        }
        /* top beam_size by total score, ties: lower candidate index */
        float sc[PB_MAXB * PB_MAXC];
        char used[PB_MAXB * PB_MAXC];
        for (int i = 0; i < nc; ++i) { sc[i] = oracle_lse2(c_pb[i], c_pnb[i]); used[i] = 0; }
        int nn = 0;
        int s_node[PB_MAXB];
        float s_pb[PB_MAXB], s_pnb[PB_MAXB];
        for (int k = 0; k < beam_size && k < nc; ++k) {
            int bi = -1;
            for (int i = 0; i < nc; ++i)
                if (!used[i] && sc[i] > -INFINITY && (bi < 0 || sc[i] > sc[bi])) bi = i;
            if (bi < 0) break;
            used[bi] = 1;
            int node;
            if (c_key[bi] >= 0) node = (int)c_key[bi];
            else {
                const long pk = -(c_key[bi] + 1);
                const int pn = (int)(pk / 8), c = (int)(pk % 8);
                node = n_nodes++;
                parent[node] = pn; label[node] = (int8_t)c; tstep[node] = t;
                child[(size_t)pn * (PB_MAXC - 1) + (c - 1)] = node;
            }
            s_node[nn] = node; s_pb[nn] = c_pb[bi]; s_pnb[nn] = c_pnb[bi]; ++nn;
        }
        /* renormalise by the best total so that scores stay bounded */
        const float shift = nn ? oracle_lse2(s_pb[0], s_pnb[0]) : 0.0f;
        nb = nn;
        for (int i = 0; i < nn; ++i) { b_node[i] = s_node[i]; b_pb[i] = s_pb[i] - shift; b_pnb[i] = s_pnb[i] - shift; }
#undef PB_FIND
    }
    int n = nb ? b_node[0] : 0, len = 0;
    for (int m = n; m > 0; m = parent[m]) ++len;
    *count_out = len;
    for (int m = n, i = len - 1; m > 0; m = parent[m], --i) { labels_out[i] = label[m]; path_out[i] = tstep[m]; }
    free(parent); free(label); free(tstep); free(child);
    return 0;
}

/* Posterior decoding (SeqdistModel.decode_batch, bonito/crf/model.py:196-199): post = posteriors(scores) + 1e-8,
 * then the Max-semiring best path over log(post). Plain fp64 forward/backward here (independent of the table
 * LSE used by the kernel), koi layout. */
int oracle_crf_posterior_viterbi(const uint16_t* scores, int N, int T, int state_len, float blank, int8_t* moves,
                                 int8_t* path) {
    const int S = ipow4(state_len), q = S / 4;
    double* al = (double*)malloc(sizeof(double) * (size_t)(T + 1) * S);
    double* be = (double*)malloc(sizeof(double) * (size_t)(T + 1) * S);
    double* v0 = (double*)malloc(sizeof(double) * S);
    double* v1 = (double*)malloc(sizeof(double) * S);
    uint8_t* bp = (uint8_t*)malloc((size_t)T * S);
    if (!al || !be || !v0 || !v1 || !bp) { free(al); free(be); free(v0); free(v1); free(bp); return -1; }
    for (int n = 0; n < N; ++n) {
        const uint16_t* sn = scores + (size_t)n * T * 4 * S;
        for (int j = 0; j < S; ++j) { al[j] = 0.0; be[(size_t)T * S + j] = 0.0; }
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < S; ++j) {
                double v[5], m;
                for (int k = 0; k < 5; ++k) {
                    const int src = k == 0 ? j : (k - 1) * q + (j >> 2);
                    v[k] = al[(size_t)t * S + src] + ms(sn + (size_t)t * 4 * S, 0, blank, j, k);
                }
                m = v[0];
                for (int k = 1; k < 5; ++k) if (v[k] > m) m = v[k];
                double s = 0;
                for (int k = 0; k < 5; ++k) s += exp(v[k] - m);
                al[(size_t)(t + 1) * S + j] = m + log(s);
            }
        for (int t = T - 1; t >= 0; --t) {
            for (int i = 0; i < S; ++i) be[(size_t)t * S + i] = -INFINITY;
            for (int j = 0; j < S; ++j)
                for (int k = 0; k < 5; ++k) {
                    const int src = k == 0 ? j : (k - 1) * q + (j >> 2);
                    const double x = ms(sn + (size_t)t * 4 * S, 0, blank, j, k) + be[(size_t)(t + 1) * S + j];
                    double* d = &be[(size_t)t * S + src];
                    if (*d == -INFINITY) *d = x;
                    else { const double m = *d > x ? *d : x; *d = m + log(exp(*d - m) + exp(x - m)); }
                }
        }
        double lz;
        { double m = -INFINITY, s = 0; for (int j = 0; j < S; ++j) if (al[(size_t)T * S + j] > m) m = al[(size_t)T * S + j];
          for (int j = 0; j < S; ++j) s += exp(al[(size_t)T * S + j] - m); lz = m + log(s); }
        for (int j = 0; j < S; ++j) v0[j] = 0.0;
        for (int t = 0; t < T; ++t) {
            for (int j = 0; j < S; ++j) {
                double best = 0; int bk = 0;
                for (int k = 0; k < 5; ++k) {
                    const int src = k == 0 ? j : (k - 1) * q + (j >> 2);
                    const double lp = al[(size_t)t * S + src] + ms(sn + (size_t)t * 4 * S, 0, blank, j, k) +
                                      be[(size_t)(t + 1) * S + j] - lz;
                    const double c = log(exp(lp) + 1e-8) + v0[src];
                    if (k == 0 || c > best) { best = c; bk = k; }
                }
                v1[j] = best; bp[(size_t)t * S + j] = (uint8_t)bk;
            }
            double* tmp = v0; v0 = v1; v1 = tmp;
        }
        int st = 0;
        for (int j = 1; j < S; ++j) if (v0[j] > v0[st]) st = j;
        for (int t = T - 1; t >= 0; --t) {
            const int k = bp[(size_t)t * S + st];
            moves[(size_t)n * T + t] = (int8_t)(k != 0);
            path[(size_t)n * T + t] = (int8_t)(k != 0 ? 1 + (st & 3) : 0);
            if (k != 0) st = (k - 1) * q + (st >> 2);
        }
    }
    free(al); free(be); free(v0); free(v1); free(bp);
    return 0;
}
