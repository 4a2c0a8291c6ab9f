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

static inline float lse2(float a, float b) {
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
