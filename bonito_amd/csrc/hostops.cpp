// Host-side helpers of the basecalling pipeline (no device code; compiled as plain C++).
//
// The formatting and chunking stages of bonito's pipeline (/root/reference bonito/crf/basecall.py:48-82, util.py:142-161)
// are numpy / torch one-liners per read. At > 1e8 samples/s their interpreter time and the interpreter lock they hold while
// running become visible next to the GPU stages, so the two byte-pushing loops live here and are called through ctypes
// (which drops the lock for the duration of the call).
#include <cstdint>
#include <cstring>
#include <vector>
#include <immintrin.h>

#include "bonito_hip.h"

// koi.decode.to_str before the text decode (bonito/crf/basecall.py:48-55): the non-zero bytes of src[0..n) in order.
extern "C" long bh_host_compact(const int8_t* src, long n, char* dst) {
    if (!src || !dst || n <= 0) return 0;
    long k = 0;
    for (long i = 0; i < n; ++i) {
        dst[k] = (char)src[i];
        k += src[i] != 0;
    }
    return k;
}

namespace {

// fp32 -> fp16, round to nearest even (what torch's .to(float16) does)
__attribute__((target("avx2,f16c"))) void cvt_row_f16c(const float* src, uint16_t* dst, long n) {
    long i = 0;
    for (; i + 8 <= n; i += 8)
        _mm_storeu_si128((__m128i*)(dst + i), _mm256_cvtps_ph(_mm256_loadu_ps(src + i), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
    for (; i < n; ++i) {
        const __m128i h = _mm_cvtps_ph(_mm_set_ss(src[i]), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        dst[i] = (uint16_t)_mm_extract_epi16(h, 0);
    }
}

uint16_t cvt_one_soft(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u | ((x >> 13) & 0x3ffu) : 0u));   // inf / nan
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                  // rounds to inf
    if (x < 0x33000001u) return (uint16_t)sign;                               // rounds to zero
    if (x < 0x38800000u) {                                                    // half subnormal
        const int shift = 126 - (int)(x >> 23);                               // 14 .. 24
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        const uint32_t q = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        return (uint16_t)(sign | (q + ((rem > half) || (rem == half && (q & 1u)))));
    }
    const uint32_t q = (x - 0x38000000u) >> 13, rem = x & 0x1fffu;
    return (uint16_t)(sign | (q + ((rem > 0x1000u) || (rem == 0x1000u && (q & 1u)))));
}

void cvt_row(const float* src, uint16_t* dst, long n) {
    static const bool fast = __builtin_cpu_supports("f16c") && __builtin_cpu_supports("avx2");
    if (fast) { cvt_row_f16c(src, dst, n); return; }
    for (long i = 0; i < n; ++i) dst[i] = cvt_one_soft(src[i]);
}

}  // namespace

// Rows [row0, row0 + nrows) of util.chunk(signal[0..T), chunksize, overlap) for T >= chunksize (bonito/util.py:142-161: windows
// advance by chunksize - overlap from offset stub = (T - overlap) % step, and when stub > 0 an extra first chunk covers
// signal[0:chunksize]), cast to fp16 and written to dst[nrows][chunksize]. Returns the number of rows written (< 0: bad arguments).
extern "C" long bh_host_chunk_rows(const float* signal, long T, int chunksize, int overlap, long row0, long nrows, uint16_t* dst) {
    if (!signal || !dst || chunksize <= 0 || overlap < 0 || overlap >= chunksize || T < chunksize || row0 < 0 || nrows < 0) return -1;
    const long step = chunksize - overlap;
    const long stub = (T - overlap) % step;
    const long n_total = (T - stub - chunksize) / step + 1 + (stub > 0 ? 1 : 0);
    if (row0 + nrows > n_total) return -1;
    for (long k = 0; k < nrows; ++k) {
        const long row = row0 + k;
        const long start = stub > 0 ? (row == 0 ? 0 : stub + (row - 1) * step) : row * step;
        cvt_row(signal + start, dst + k * (long)chunksize, chunksize);
    }
    return nrows;
}

// ---------------------------------------------------------------------------------------------------------------------------
// One basecalled read -> its record text, in one call: stitch (bonito/util.py:164-183 via crf/basecall.py:13-24) + to_str
// (crf/basecall.py:48-55) + the FASTQ / FASTA / SAM line with the tags of bonito/io.py:135-166 (RG, qs, ns, ts, mv). The Python
// spelling of the same steps (unbatchify -> torch.cat -> reshape -> cat -> compact -> numpy -> str.join) cost ~1 ms per read on one
// host thread, i.e. it capped a rank at ~1e8 samples/s; this is ~20 us.
#include <cmath>
#include <cstdio>

namespace {

// Python's a[lo:hi] on a sequence of length n (negative bounds count from the end; -0 is 0, exactly like the slices of
// util.stitch / stitch_planes this mirrors, quirks included)
inline void py_slice(long lo, long hi, long n, long& a, long& b) {
    a = lo < 0 ? (n + lo < 0 ? 0 : n + lo) : (lo > n ? n : lo);
    b = hi < 0 ? (n + hi < 0 ? 0 : n + hi) : (hi > n ? n : hi);
    if (b < a) b = a;
}

struct Piece { const int8_t* base; long plane_stride, lo, rows; };

inline const int8_t* chunk_row(const Piece* pc, int n_pieces, long j, int plane, long T) {
    for (int i = 0; i < n_pieces; ++i) {
        if (j < pc[i].rows) return pc[i].base + plane * pc[i].plane_stride + (pc[i].lo + j) * T;
        j -= pc[i].rows;
    }
    return nullptr;
}

double mean_qscore(const unsigned long* hist) {
    // util.mean_qscore_from_qstring: mean error probability of the phred characters, as a q-score (floor 1e-4 on the mean)
    unsigned long n = 0;
    double sum = 0.0;
    for (int q = 0; q < 256; ++q)
        if (hist[q]) { n += hist[q]; sum += (double)hist[q] * std::exp((double)((q - 33) & 0xff) * (-std::log(10.0) / 10.0)); }
    if (!n) return 0.0;
    const double mean = sum / (double)n;
    return -10.0 * std::log10(mean > 1e-4 ? mean : 1e-4);
}

}  // namespace

extern "C" double bh_host_mean_qscore(const char* qstring, long n) {
    if (!qstring || n <= 0) return 0.0;
    unsigned long hist[256] = {0};
    for (long i = 0; i < n; ++i) ++hist[(unsigned char)qstring[i]];
    return mean_qscore(hist);
}

// mode: 0 fastq, 1 fasta, 2 sam. Returns the number of bytes written to `out`; 0 = the read is filtered out (empty sequence or
// mean q-score below min_qscore: seq_len / mean_q are still set); -1 = bad arguments; < -1 = -(bytes needed) when out_cap is short.
extern "C" long bh_host_format_read(const int8_t* const* base, const long* plane_stride, const long* lo, const long* rows, int n_pieces,
                                    long T, long length, int chunksize, int overlap, int stride, int reverse, int rna, int mode,
                                    double min_qscore, const char* read_id, const char* run_id, long num_samples, long trimmed_samples,
                                    char* out, long out_cap, long* seq_len, double* mean_q) {
    if (!base || !plane_stride || !lo || !rows || n_pieces <= 0 || T <= 0 || stride <= 0 || !read_id || !out || !seq_len || !mean_q) return -1;
    Piece pc_fixed[64];                       // a read's chunks usually sit in a handful of engine calls; any number is accepted
    std::vector<Piece> pc_big;
    Piece* pc = pc_fixed;
    if (n_pieces > 64) { pc_big.resize((size_t)n_pieces); pc = pc_big.data(); }
    long n_chunks = 0;
    for (int i = 0; i < n_pieces; ++i) { pc[i] = Piece{base[i], plane_stride[i], lo[i], rows[i]}; n_chunks += rows[i]; }
    if (n_chunks <= 0) return -1;
    // ---- which [a, b) of which chunk, in output order (stitch_planes) -----------------------------------------------------------
    struct Seg { long chunk, a, b; };
    Seg one[3];
    long n_edge = 0, mid_lo = 0, mid_hi = 0, mid_a = 0, mid_b = 0;         // edge segments + the run of middle chunks
    bool mid_flip = false;
    int mid_at = -1;                                                        // position of the middle run among the edge segments
    const long size = chunksize;
    if (length < size) {
        long a, b; py_slice(0, (long)std::floor((double)length / stride), T, a, b);
        one[n_edge++] = Seg{0, a, b};
    } else if (n_chunks == 1) {
        one[n_edge++] = Seg{0, 0, T};
    } else {
        const long semi = overlap / 2, start = semi / stride, end = (size - semi) / stride;
        const long stub = (length - overlap) % (size - overlap);
        const long first_end = stub > 0 ? (stub + semi) / stride : end;
        long a, b;
        if (reverse) {
            py_slice(0, -start, T, a, b); one[n_edge++] = Seg{n_chunks - 1, a, b};
            mid_at = 1; mid_lo = 1; mid_hi = n_chunks - 1; mid_flip = true; py_slice(-end, -start, T, mid_a, mid_b);
            py_slice(-first_end, T, T, a, b); one[n_edge++] = Seg{0, a, b};
        } else {
            py_slice(0, first_end, T, a, b); one[n_edge++] = Seg{0, a, b};
            mid_at = 1; mid_lo = 1; mid_hi = n_chunks - 1; py_slice(start, end, T, mid_a, mid_b);
            py_slice(start, T, T, a, b); one[n_edge++] = Seg{n_chunks - 1, a, b};
        }
    }
    long positions = 0;
    for (long i = 0; i < n_edge; ++i) positions += one[i].b - one[i].a;
    if (mid_at >= 0 && mid_hi > mid_lo) positions += (mid_hi - mid_lo) * (mid_b - mid_a);
    const long id_len = (long)strlen(read_id), rg_len = run_id && *run_id ? (long)strlen(run_id) : 7;
    // the record (at most 2 x positions for sequence + qstring, 2 x positions for the move table, the names) is assembled at the front
    // of `out`; sequence / qstring / moves are first collected in 3 x positions of scratch at its far end
    const long need = 4 * positions + 24 + 2 * id_len + rg_len + 160 + 3 * positions;
    if (need > out_cap) return -need;
    // ---- strings -----------------------------------------------------------------------------------------------------------------
    char* seq = out + out_cap - positions;
    char* qs = seq - positions;
    char* mvs = qs - positions;
    long n_seq = 0, n_qs = 0, n_mv = 0;
    unsigned long hist[256] = {0};
    auto emit = [&](long chunk, long a, long b) {
        const int8_t* s = chunk_row(pc, n_pieces, chunk, 0, T);
        const int8_t* q = chunk_row(pc, n_pieces, chunk, 1, T);
        const int8_t* m = chunk_row(pc, n_pieces, chunk, 2, T);
        for (long t = a; t < b; ++t) {
            seq[n_seq] = (char)s[t]; n_seq += s[t] != 0;
            qs[n_qs] = (char)q[t]; n_qs += q[t] != 0;
            mvs[n_mv++] = (char)('0' + m[t]);
        }
    };
    for (long i = 0; i < n_edge; ++i) {
        if (mid_at == i && mid_hi > mid_lo) {
            if (mid_flip) for (long c = mid_hi - 1; c >= mid_lo; --c) emit(c, mid_a, mid_b);
            else for (long c = mid_lo; c < mid_hi; ++c) emit(c, mid_a, mid_b);
        }
        emit(one[i].chunk, one[i].a, one[i].b);
    }
    if (rna) {
        for (long i = 0, j = n_seq - 1; i < j; ++i, --j) { const char t = seq[i]; seq[i] = seq[j]; seq[j] = t; }
        for (long i = 0, j = n_qs - 1; i < j; ++i, --j) { const char t = qs[i]; qs[i] = qs[j]; qs[j] = t; }
    }
    // io.format_record's conventions (tests fuzz the two against each other) = the reference's (bonito/io.py:431-433): the mean q-score
    // is always computed from the q-string - a decoded "*" beside one base is Q9, not a sentinel (this path never sees the sentinel);
    // an EMPTY q-string beside a sequence means "no qualities": mean 0.0, '!' per base in FASTQ, '*' in SAM
    const bool no_quals = n_qs == 0;
    for (long i = 0; i < n_qs; ++i) ++hist[(unsigned char)qs[i]];
    const double mq = n_qs ? mean_qscore(hist) : 0.0;
    *seq_len = n_seq;
    *mean_q = mq;
    if (mq < min_qscore || n_seq == 0) return 0;
    // ---- the record -----------------------------------------------------------------------------------------------------------------
    char* p = out;
    auto put = [&](const char* s, long n) { memcpy(p, s, (size_t)n); p += n; };
    auto tags = [&]() {
        p += sprintf(p, "RG:Z:%s\tqs:f:%0.2f\tns:i:%ld\tts:i:%ld", run_id && *run_id ? run_id : "unknown", mq, num_samples, trimmed_samples);
        if (n_mv) {
            p += sprintf(p, "\tmv:B:c,%d", stride);
            for (long i = 0; i < n_mv; ++i) { *p++ = ','; *p++ = mvs[i]; }
        }
    };
    if (mode == 1) {
        *p++ = '>'; put(read_id, id_len); *p++ = '\n'; put(seq, n_seq); *p++ = '\n';
    } else if (mode == 0) {
        *p++ = '@'; put(read_id, id_len); *p++ = ' '; tags(); *p++ = '\n';
        put(seq, n_seq); put("\n+\n", 3);
        if (no_quals) { memset(p, '!', (size_t)n_seq); p += n_seq; } else put(qs, n_qs);
        *p++ = '\n';
    } else {
        put(read_id, id_len); put("\t4\t*\t0\t0\t*\t*\t0\t0\t", 17); put(seq, n_seq); *p++ = '\t';
        if (no_quals) *p++ = '*'; else put(qs, n_qs);
        put("\tNM:i:0\t", 8); tags(); *p++ = '\n';
    }
    return (long)(p - out);
}

// ---- pod5 signal codec (bonito_amd/pod5.py) ------------------------------------------------------------------------------------------
// The inner layer of ONT's VBZ signal compression as published with the pod5 format (the outer layer, zstd, is undone by the caller):
// streamvbyte for 16-bit values ("svb16") over the zig-zag code of the first differences. `count` values: ceil(count / 8) key bytes (bit
// i % 8 of byte i / 8 says whether value i takes one byte or two), then the data bytes, little-endian. value -> delta = (v >> 1) ^ -(v & 1),
// sample_i = sample_{i-1} + delta (16-bit wrap-around, sample_{-1} = 0). Returns the bytes consumed, or -1 when the input is too short.
// Replaces pod5's compiled decoder on the path /root/reference bonito/pod5.py:52 (`read.signal`); the wheel is absent here: FORMAT UNPINNED.
namespace {
// per key byte: the offsets of its eight values inside the group's data bytes, and the group's length - breaks the pointer chase of the
// byte-at-a-time loop (eight independent loads per key byte; the running sum of the deltas is the only serial chain left)
struct Svb16Tab {
    uint8_t off[256][8];
    uint8_t len[256];
    Svb16Tab() {
        for (int kb = 0; kb < 256; ++kb) {
            int o = 0;
            for (int j = 0; j < 8; ++j) { off[kb][j] = (uint8_t)o; o += 1 + ((kb >> j) & 1); }
            len[kb] = (uint8_t)o;
        }
    }
};
const Svb16Tab g_svb16;
}  // namespace

extern "C" long bh_host_svb16_decode(const uint8_t* in, long n_in, long count, int16_t* out) {
    if (!in || !out || count < 0 || n_in < 0) return -1;
    const long nkeys = (count + 7) / 8;
    if (n_in < nkeys) return -1;
    const uint8_t* keys = in;
    const uint8_t* data = in + nkeys;
    const uint8_t* end = in + n_in;
    uint16_t prev = 0;
    long i = 0;
    // whole groups of eight while at least 16 data bytes remain (a group reads at most 16): table-driven, no pointer chase
    for (; i + 8 <= count && end - data >= 16; i += 8) {
        const unsigned kb = keys[i >> 3];
        const uint8_t* o = g_svb16.off[kb];
        uint16_t v[8];
        for (int j = 0; j < 8; ++j) {
            const uint8_t* p = data + o[j];
            const uint16_t two = (uint16_t)((kb >> j) & 1u);
            v[j] = (uint16_t)(p[0] | (uint16_t)((p[1] & (uint8_t)(0u - two)) << 8));
        }
        data += g_svb16.len[kb];
        for (int j = 0; j < 8; ++j) {
            const uint16_t delta = (uint16_t)((v[j] >> 1) ^ (uint16_t)(0u - (v[j] & 1u)));
            prev = (uint16_t)(prev + delta);
            out[i + j] = (int16_t)prev;
        }
    }
    for (; i < count; ++i) {                       // the tail (and inputs too short for the wide path): byte by byte, bounds checked
        const int two = (keys[i >> 3] >> (i & 7)) & 1;
        if (data + 1 + two > end) return -1;
        uint16_t v = data[0];
        if (two) v |= (uint16_t)data[1] << 8;
        data += 1 + two;
        const uint16_t delta = (uint16_t)((v >> 1) ^ (uint16_t)(0u - (v & 1u)));
        prev = (uint16_t)(prev + delta);
        out[i] = (int16_t)prev;
    }
    return (long)(data - in);
}
