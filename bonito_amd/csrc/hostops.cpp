// Host-side helpers of the basecalling pipeline (no device code; compiled as plain C++).
//
// The formatting and chunking stages of bonito's pipeline (/root/reference bonito/crf/basecall.py:48-82, util.py:142-161)
// are numpy / torch one-liners per read. At > 1e8 samples/s their interpreter time and the interpreter lock they hold while
// running become visible next to the GPU stages, so the two byte-pushing loops live here and are called through ctypes
// (which drops the lock for the duration of the call).
#include <cstdint>
#include <cstring>
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
