// Sliding-window multi-head attention with rotary embedding, and the DeepNorm residual RMSNorm, for
// gfx950.  Replaces flash_attn_qkvpacked_func(qkv, window_size=(l, r)) + RotaryEmbedding + the Triton
// RMSNorm(x, residual) on the reference's transformer path (/root/reference bonito/transformer/model.py:
// 42-79 MultiHeadAttention, 110-111,125-128 norm/residual; in-tree SDPA formulation at :58-66 and
// sliding_window_mask at :33-39 define the semantics: key j is visible to query i iff i-l <= j <= i+r).
//
// Attention kernel (head_dim 64):
//   workgroup = 8 waves = (chunk n, head h, 128 consecutive queries); the <= 128+l+r keys that block can
//   see are staged ONCE into LDS: K row-major with rotary applied (16-byte chunks XOR-swizzled by row&7
//   so fragment reads are conflict free), V transposed ([d][key], row padded by 4 halves).
//   Each wave owns 16 queries and computes the TRANSPOSED score tile S^T = K Q^T on MFMA 16x16x32 f16
//   (A = K fragment from LDS, B = rotated/scaled Q fragment in registers). In the accumulator layout a lane
//   then holds 4 consecutive keys of ONE query, so the softmax row statistics are an in-lane reduction plus
//   two cross-lane steps, and the normalised probabilities ARE the B fragment of O^T = V^T P^T -- no LDS
//   round trip for P (the contraction order over keys is permuted identically in A and B).
//   The whole visible key range is in registers (<= NT tiles), so the softmax is exact, not online.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace bh {

struct AttnArgs {
    const half_t* qkv;   // [N*T][3*D], q | k | v, head-major inside each
    half_t* out;         // [N*T][D]
    const float* cs;     // [T][32][2]  (cos, sin) of position * inv_freq
    int N, T, H;         // H heads, D = 64*H
    int wl, wr;          // window: i-wl <= j <= i+wr
    float scale;         // 1/sqrt(64)
};

constexpr int QB = 128;     // queries per workgroup
constexpr int HD = 64;

template <int NT>           // key tiles (of 16) per wave, even
__global__ __launch_bounds__(512) void attention_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KR = 7 * 16 + NT * 16;          // staged key rows
    constexpr int VS = KR + 4;                    // V^T row stride (halves)
    char* kl = smem;                              // [KR][128 B] swizzled
    half_t* vt = (half_t*)(smem + KR * 128);      // [64][VS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = p.H * HD;
    const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const int i0 = qb * QB;
    const int jbase = i0 - p.wl;                  // staged row r <-> key jbase + r
    const half_t* base = p.qkv + (long)n * p.T * 3 * D;

    // ---- stage K (rotary) : task = (row r, pair chunk c in 0..3) ---------------------------------
    for (int task = tid; task < KR * 4; task += 512) {
        const int r = task >> 2, c = task & 3;
        const int j = jbase + r;
        half8_t lo = {0, 0, 0, 0, 0, 0, 0, 0}, hi = lo;
        if (j >= 0 && j < p.T) {
            const half_t* kp = base + (long)j * 3 * D + D + h * HD;
            const half8_t x1 = *(const half8_t*)(kp + c * 8);
            const half8_t x2 = *(const half8_t*)(kp + 32 + c * 8);
            const float* cs = p.cs + ((long)j * 32 + c * 8) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float co = cs[2 * e], si = cs[2 * e + 1];
                const float a = (float)x1[e], b = (float)x2[e];
                lo[e] = (half_t)(a * co - b * si);
                hi[e] = (half_t)(a * si + b * co);
            }
        }
        *(half8_t*)(kl + r * 128 + ((c ^ (r & 7)) << 4)) = lo;
        *(half8_t*)(kl + r * 128 + (((c + 4) ^ (r & 7)) << 4)) = hi;
    }
    // ---- stage V transposed: task = (key pair kp, d chunk c in 0..7) -----------------------------
    for (int task = tid; task < (KR / 2) * 8; task += 512) {
        const int kp2 = task >> 3, c = task & 7;
        const int r = kp2 * 2;
        half8_t v0 = {0, 0, 0, 0, 0, 0, 0, 0}, v1 = v0;
        const int j = jbase + r;
        if (j >= 0 && j < p.T) v0 = *(const half8_t*)(base + (long)j * 3 * D + 2 * D + h * HD + c * 8);
        if (j + 1 >= 0 && j + 1 < p.T) v1 = *(const half8_t*)(base + (long)(j + 1) * 3 * D + 2 * D + h * HD + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            half2_t pr = {v0[e], v1[e]};
            *(half2_t*)(vt + (c * 8 + e) * VS + r) = pr;
        }
    }

    // ---- this wave's 16 queries: rotated, scaled Q fragments in registers -------------------------
    const int qi = i0 + wave * 16 + (lane & 15);
    const int g = lane >> 4;
    half8_t qf[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
    if (qi < p.T) {
        const half_t* qp = base + (long)qi * 3 * D + h * HD;
        const half8_t x1 = *(const half8_t*)(qp + g * 8);
        const half8_t x2 = *(const half8_t*)(qp + 32 + g * 8);
        const float* cs = p.cs + ((long)qi * 32 + g * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float co = cs[2 * e], si = cs[2 * e + 1];
            const float a = (float)x1[e], b = (float)x2[e];
            qf[0][e] = (half_t)((a * co - b * si) * p.scale);
            qf[1][e] = (half_t)((a * si + b * co) * p.scale);
        }
    }
    __syncthreads();

    // ---- S^T tiles: keys r0 + kt*16 .., r0 = 16*wave -----------------------------------------------
    const int r0 = wave * 16;
    float4_t s[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const int row = r0 + kt * 16 + (lane & 15);
        const char* rp = kl + row * 128;
        const half8_t a0 = *(const half8_t*)(rp + ((g ^ (row & 7)) << 4));
        const half8_t a1 = *(const half8_t*)(rp + (((g + 4) ^ (row & 7)) << 4));
        float4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma16(a0, qf[0], acc);
        acc = mfma16(a1, qf[1], acc);
        s[kt] = acc;
    }
    // ---- mask + exact softmax over the visible keys -------------------------------------------------
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = jbase + r0 + kt * 16 + g * 4 + e;
            const bool ok = j >= 0 && j < p.T && j >= qi - p.wl && j <= qi + p.wr;
            s[kt][e] = ok ? s[kt][e] : -INFINITY;
            m = fmaxf(m, s[kt][e]);
        }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    const float msafe = (m == -INFINITY) ? 0.0f : m;
    float sum = 0.0f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float pv = __expf(s[kt][e] - msafe);   // exp(-inf) = 0 for masked keys
            s[kt][e] = pv;
            sum += pv;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;

    // ---- O^T = V^T P^T ------------------------------------------------------------------------------
    float4_t o[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) o[mt] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NT / 2; ++c) {
        half8_t pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pb[e] = (half_t)(s[2 * c][e] * inv);
            pb[4 + e] = (half_t)(s[2 * c + 1][e] * inv);
        }
        const int kcol = r0 + c * 32 + g * 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const half_t* vp = vt + (mt * 16 + (lane & 15)) * VS + kcol;
            const half4_t va = *(const half4_t*)vp;
            const half4_t vb = *(const half4_t*)(vp + 16);
            half8_t af;
#pragma unroll
            for (int e = 0; e < 4; ++e) { af[e] = va[e]; af[4 + e] = vb[e]; }
            o[mt] = mfma16(af, pb, o[mt]);
        }
    }
    if (qi < p.T) {
        half_t* op = p.out + ((long)n * p.T + qi) * D + h * HD + g * 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            half4_t ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (half_t)o[mt][e];
            *(half4_t*)(op + mt * 16) = ov;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Persistent ring-buffer variant for PRE-ROTATED inputs (rotary and the 1/sqrt(d) scale are applied by the epilogue of the
// Wqkv GEMM, gemm.hip): a workgroup owns one (chunk, head) and walks its query blocks of 128 in order. K (row-major,
// swizzled) and V^T live in a 512-row LDS ring indexed by key & 511, so every key/value row is fetched from HBM and
// transposed exactly once per head (the block-per-workgroup kernel above re-stages ~3x the rows and re-applies the
// rotation each time), and the 128 new rows of the next block travel through registers while the current block computes.
// Block b (queries 128b..128b+127) reads keys 128b-128 .. 128b+271 (wave w: 18 tiles of 16 from 128b-128+16w), all tile
// boundaries are multiples of 16 so a tile never straddles the ring wrap. Needs wl + wr <= 256 and wl <= 128.
typedef float float8_t __attribute__((ext_vector_type(8)));
constexpr int RING = 512;
constexpr int RVS = RING + 4;                  // V^T row stride (halves): 4-bank skew between d rows

struct AttnRingArgs {
    const half_t* qkv;   // [N*T][3*D], q (rotated, scaled by log2(e)/sqrt(d)) | k (rotated) | v
    half_t* out;         // [N*T][D]
    int N, T, H;
    int wl, wr;
    int cpw;             // round-6 kernel: chunks per workgroup (its stream), pitch = T rounded up to 16
};

// WAVES = waves per workgroup = query tiles of 16 per block (8: blocks of 128 queries, the geometry of rounds 2-4; 12: blocks of 192 - the
// live rows of a block, 272 + 16 WAVES = 464, still fit the 512-row ring, so the same 130 KiB of LDS carry 12 waves per CU instead of 8:
// the kernel WAITS (two waves per SIMD, 49 % of the wave cycles), it does not issue - round 5)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void attention_ring_kernel(AttnRingArgs p) {
    constexpr int RQB = 16 * WAVES, NTHR = 64 * WAVES;
    static_assert(272 + RQB <= RING && RQB % 16 == 0, "the block's live rows must fit the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 18;
    char* kl = smem;                              // [RING][128 B] swizzled by slot & 7
    half_t* vt = (half_t*)(smem + RING * 128);    // [64][RVS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = p.H * HD;
    const int h = blockIdx.x, n = blockIdx.y;
    const half_t* base = p.qkv + (long)n * p.T * 3 * D;
    const int g = lane >> 4;

    // staging task: 16-byte chunk c of the key-row PAIR (first + 2*pp, first + 2*pp + 1), pp = pair index; pairs make the
    // transposing V^T writes 4-byte stores of two adjacent keys
    auto load_pair = [&](int first, int pp, int c, uint4_t (&kv)[2], uint4_t (&vv)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = first + 2 * pp + u;
            kv[u] = uint4_t{0, 0, 0, 0};
            vv[u] = kv[u];
            if (j >= 0 && j < p.T) {
                const half_t* rp = base + (long)j * 3 * D + D + h * HD + c * 8;
                kv[u] = *(const uint4_t*)rp;
                vv[u] = *(const uint4_t*)(rp + D);
            }
        }
    };
    auto store_pair = [&](int first, int pp, int c, const uint4_t (&kv)[2], const uint4_t (&vv)[2]) {
        const int slot = (first + 2 * pp) & (RING - 1);          // even; the pair never straddles the wrap
#pragma unroll
        for (int u = 0; u < 2; ++u) *(uint4_t*)(kl + (slot + u) * 128 + ((c ^ ((slot + u) & 7)) << 4)) = kv[u];
        const half8_t v0 = __builtin_bit_cast(half8_t, vv[0]), v1 = __builtin_bit_cast(half8_t, vv[1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const half2_t pr = {v0[e], v1[e]};
            *(half2_t*)(vt + (c * 8 + e) * RVS + slot) = pr;
        }
    };

    // ---- prologue: rows -128 .. RQB + 143 ((136 + RQB / 2) pairs x 8 chunks) ------------------------------------------------
    for (int t = tid; t < (136 + RQB / 2) * 8; t += NTHR) {
        uint4_t kv[2], vv[2];
        load_pair(-128, t >> 3, t & 7, kv, vv);
        store_pair(-128, t >> 3, t & 7, kv, vv);
    }
    const int nblk = (p.T + RQB - 1) / RQB;
    // this wave's 16 queries of a block as B fragments (already rotated and scaled); the next block's are requested one
    // block ahead together with its new key/value rows
    auto load_q = [&](int blk, half8_t (&q)[2]) {
        const int qi = blk * RQB + wave * 16 + (lane & 15);
        q[0] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        q[1] = q[0];
        if (qi < p.T) {
            const half_t* qp = base + (long)qi * 3 * D + h * HD;
            q[0] = *(const half8_t*)(qp + g * 8);
            q[1] = *(const half8_t*)(qp + 32 + g * 8);
        }
    };
    half8_t qn[2];
    load_q(0, qn);
    for (int b = 0; b < nblk; ++b) {
        const int i0 = b * RQB;
        const int jb = i0 - 128;
        const int qi = i0 + wave * 16 + (lane & 15);
        half8_t qf[2] = {qn[0], qn[1]};
        if (b + 1 < nblk) load_q(b + 1, qn);
        // the RQB rows the NEXT block adds (i0 + RQB + 144 ..): requested now, stored after this block's compute
        uint4_t nk[2], nv[2];
        const int nfirst = i0 + RQB + 144;
        const bool more = b + 1 < nblk;
        if (more) load_pair(nfirst, tid >> 3, tid & 7, nk, nv);        // RQB / 2 pairs x 8 chunks = one task per thread
        __syncthreads();                          // ring rows of this block are in place

        const int rel0 = wave * 16;               // this wave's first key tile, relative to jb
        float4_t s[NT];
        // ring row of this lane's key in tile kt: (jb + rel0 + 16 kt + (lane & 15)) & 511. Every term but the lane's is a multiple of 16,
        // so the swizzle (slot & 7) is the lane's own constant and the tile's offset is wave-uniform (round 5: two lane-constant base
        // pointers + a scalar offset per tile instead of re-deriving both addresses on the vector ALU)
        const int tile0 = __builtin_amdgcn_readfirstlane((jb + rel0) >> 4);          // wave-uniform ring tile of this wave's first keys
        const char* kl_lane0 = kl + ((lane & 15) << 7) + ((g ^ (lane & 7)) << 4);
        const char* kl_lane1 = kl + ((lane & 15) << 7) + (((g + 4) ^ (lane & 7)) << 4);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const int toff = ((tile0 + kt) & (RING / 16 - 1)) << 11;                   // scalar: 16 rows of 128 bytes per tile
            const half8_t a0 = *(const half8_t*)(kl_lane0 + toff);
            const half8_t a1 = *(const half8_t*)(kl_lane1 + toff);
            float4_t acc = {0.f, 0.f, 0.f, 0.f};
            acc = mfma16(a0, qf[0], acc);
            acc = mfma16(a1, qf[1], acc);
            s[kt] = acc;
        }
        // visible keys of query qi, relative to this lane's first key (jb + rel0 + 4g): [lo, hi]. Away from the ends of the
        // chunk only the first and the last two key tiles of a wave can hold invisible keys (tile kt spans 16kt..16kt+15 of
        // the wave's keys, query q of the wave sees q+1-(128-wl) .. q+128+wr): the 15 interior tiles skip the mask.
        const int kfirst = jb + rel0 + g * 4;
        const int lo = max(qi - p.wl, 0) - kfirst, hi = min(qi + p.wr, p.T - 1) - kfirst;
        const bool edge = jb < 0 || jb + 272 + RQB > p.T || p.wl != 127 || p.wr != 128;       // block-uniform
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (edge || kt == 0 || kt >= 16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rel = kt * 16 + e;
                    s[kt][e] = (rel >= lo && rel <= hi) ? s[kt][e] : -INFINITY;
                }
            }
            m = fmaxf(fmaxf(m, s[kt][0]), s[kt][1]);          // two v_max3_f32 per tile
            m = fmaxf(fmaxf(m, s[kt][2]), s[kt][3]);
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float msafe = (m == -INFINITY) ? 0.0f : m;
        // scores are in log2 units (the Wqkv epilogue folded log2(e) into the scale of q): p = 2^(s - m), one v_exp each.
        // P stays unnormalised (<= 1) on its way through the PV product; O is divided by the row sum at the end. The subtraction
        // and the row sum run two values per instruction (v_pk_add_f32).
        typedef float f2_t __attribute__((ext_vector_type(2)));
        const f2_t m2 = {msafe, msafe};
        f2_t sum2 = {0.0f, 0.0f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f2_t a = f2_t{s[kt][0], s[kt][1]} - m2, b = f2_t{s[kt][2], s[kt][3]} - m2;
            a.x = __builtin_amdgcn_exp2f(a.x); a.y = __builtin_amdgcn_exp2f(a.y);
            b.x = __builtin_amdgcn_exp2f(b.x); b.y = __builtin_amdgcn_exp2f(b.y);
            s[kt][0] = a.x; s[kt][1] = a.y; s[kt][2] = b.x; s[kt][3] = b.y;
            sum2 += a;
            sum2 += b;
        }
        float sum = sum2.x + sum2.y;
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;

        float4_t o[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) o[mt] = float4_t{0.f, 0.f, 0.f, 0.f};
        // V^T fragment reads: row (d index) mt * 16 + (lane & 15), columns slot .. slot + 3 and slot + 16 .. + 19; slot = the wave-uniform
        // tile start + 4 g, so a read is a lane-constant base + a scalar offset (+ an immediate for mt)
        const half_t* vlane = vt + (lane & 15) * RVS + g * 4;
#pragma unroll
        for (int c = 0; c < NT / 2; ++c) {
            const float8_t pf = {s[2 * c][0], s[2 * c][1], s[2 * c][2], s[2 * c][3],
                                 s[2 * c + 1][0], s[2 * c + 1][1], s[2 * c + 1][2], s[2 * c + 1][3]};
            const half8_t pb = __builtin_convertvector(pf, half8_t);
            const int u0 = ((tile0 + 2 * c) & (RING / 16 - 1)) << 4, u1 = ((tile0 + 2 * c + 1) & (RING / 16 - 1)) << 4;      // scalar
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const half4_t va = *(const half4_t*)(vlane + mt * 16 * RVS + u0);
                const half4_t vb = *(const half4_t*)(vlane + mt * 16 * RVS + u1);
                half8_t af;
#pragma unroll
                for (int e = 0; e < 4; ++e) { af[e] = va[e]; af[4 + e] = vb[e]; }
                o[mt] = mfma16(af, pb, o[mt]);
            }
        }
        if (qi < p.T) {
            half_t* op = p.out + ((long)n * p.T + qi) * D + h * HD + g * 4;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                half4_t ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (half_t)(o[mt][e] * inv);
                *(half4_t*)(op + mt * 16) = ov;
            }
        }
        __syncthreads();                          // everybody is done with the rows the next block overwrites
        if (more) store_pair(nfirst, tid >> 3, tid & 7, nk, nv);
    }
}

// ---------------------------------------------------------------------------------------------------
// Round 6 rebuild of the ring kernel (review: 0.12 of the MFMA peak, 8.6 vector instructions per MFMA, waves waiting 49 % of their cycles).
// Same ring, same S^T = K Q^T / O^T = V^T P^T formulation, same arithmetic per visible (query, key) pair; what changed:
//  * every key tile of a wave is classified ONCE per block with scalar arithmetic - EMPTY (no query of the wave sees any of its keys:
//    before the chunk, behind it, or outside the window), FULL (every query sees all 16 keys) or PARTIAL. Empty tiles cost nothing at all
//    (no fragment reads, no MFMAs, no exponentials, their PV pair is skipped when both halves are empty), full tiles are never masked;
//    only partial tiles pay the per-element compare (3 of 18 away from the chunk ends - the old kernel masked ALL tiles of every block
//    that touched a chunk end: 3 of the 6 blocks at T = 1000). Waves whose 16 queries all lie behind the chunk skip the block's arithmetic.
//  * V^T rows are 4 banks apart (row stride RING + 8 halves): the 16 rows x 2 column groups a ds_read_b64 half-wave touches fall on 32
//    distinct bank pairs (stride RING + 4: two-way conflicts on every PV fragment read). The staging tasks keep their coalesced (row pair,
//    16-byte piece) mapping (their transposing stores conflict eight ways - measured not to matter, see the V task below).
//  * ONE stream per workgroup (see STREAM below): no per-(chunk, head) prologue, no idle waves at a chunk's end.
//  * the exponentials of a tile pair sit in front of that pair's PV MFMAs instead of in one block ahead of all of them.
// EXPT (timing experiments only, wrong results on purpose; 0 in the product): bit 0 no exponentials, bit 1 no PV MFMAs, bit 2 no QK^T
// MFMAs, bit 3 no staging after the prologue, bit 4 no barriers inside the block loop.
constexpr int RVS2 = RING + 8;

template <int WAVES, int EXPT>
__global__ __launch_bounds__(64 * WAVES) void attention_ring2_kernel(AttnRingArgs p) {
    constexpr int RQB = 16 * WAVES, NTHR = 64 * WAVES;
    static_assert(272 + RQB <= RING && RQB % 16 == 0, "the block's live rows must fit the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 18;
    char* kl = smem;                              // [RING][128 B] swizzled by slot & 7
    half_t* vt = (half_t*)(smem + RING * 128);    // [64][RVS2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = p.H * HD;
    const int g = lane >> 4;
    // STREAM: this workgroup owns head h of the chunks [c0, c0 + nc) and treats them as ONE sequence of virtual rows, chunk i at rows
    // [i P, i P + T), P = T rounded up to 16. The ring never restarts: the prologue is paid once per workgroup instead of once per
    // (chunk, head) - 119 KB fetched with nothing to compute beside it, sixteen times per CU at 512 chunks - and the last block of a
    // chunk carries the first queries of the next one instead of idle waves (1000 tokens = 5.2 blocks of 192). A query never sees rows of
    // a neighbouring chunk: its visible range is clipped to [0, T) in CHUNK coordinates, and whatever the ring holds outside it is masked
    // or skipped like any other invisible key. Eight consecutive workgroups (one per XCD) walk the eight heads of the same chunks at the
    // same time, as before: a token's 3 KiB row is read by all of them while it is hot.
    const int h = blockIdx.x;
    const int c0 = blockIdx.y * p.cpw;
    const int nc = min(p.cpw, p.N - c0);
    const int P = (p.T + 15) & ~15;
    const long L = (long)nc * P;                  // virtual rows of this workgroup's stream
    const half_t* hb = p.qkv + h * HD;            // + (chunk * T + t) * 3 D [+ D | 2 D]

    // virtual row (chunk index ci, local row t) + d rows further on -> chunk index and local row; d >= 0
    auto advance = [&](int& ci, int& t, int d) {
        t += d;
        while (t >= P) { t -= P; ++ci; }
    };
    auto row_ptr = [&](int ci, int t) { return hb + ((long)(c0 + ci) * p.T + t) * 3 * D; };

    // K task: 16-byte piece c of the key-row pair at (ci, t), (ci, t + 1) [t even: a pair never straddles a chunk]: eight consecutive lanes
    // fetch one whole row (coalesced) and write eight distinct 16-byte slots of it (conflict free)
    auto load_k = [&](bool live, int ci, int t, int c, uint4_t (&kv)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            kv[u] = uint4_t{0, 0, 0, 0};
            if (live && ci < nc && t + u < p.T) kv[u] = *(const uint4_t*)(row_ptr(ci, t + u) + D + c * 8);
        }
    };
    auto store_k = [&](int vrow, int c, const uint4_t (&kv)[2]) {
        const int slot = vrow & (RING - 1);          // even; the pair never straddles the wrap
#pragma unroll
        for (int u = 0; u < 2; ++u) *(uint4_t*)(kl + (slot + u) * 128 + ((c ^ ((slot + u) & 7)) << 4)) = kv[u];
    };
    // V task: the same (pair, piece) mapping. Measured on the way and dropped: (i) consecutive lanes on consecutive pairs (conflict-free
    // transposing stores, but sixteen-byte loads from 64 different rows per instruction: 0.78 against 0.69 ms per 512 chunks); (ii) V^T
    // rows permuted so that the coalesced tasks' stores are conflict free as well (the same 0.69: the stores' eight-way conflicts do not
    // show, and the permutation costs the fragment reads their shared base - one address add per M tile instead of one per key-tile pair).
    auto load_v = [&](bool live, int ci, int t, int c, uint4_t (&vv)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            vv[u] = uint4_t{0, 0, 0, 0};
            if (live && ci < nc && t + u < p.T) vv[u] = *(const uint4_t*)(row_ptr(ci, t + u) + 2 * D + c * 8);
        }
    };
    auto store_v = [&](int vrow, int c, const uint4_t (&vv)[2]) {
        const int slot = vrow & (RING - 1);
        const half8_t v0 = __builtin_bit_cast(half8_t, vv[0]), v1 = __builtin_bit_cast(half8_t, vv[1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const half2_t pr = {v0[e], v1[e]};
            *(half2_t*)(vt + (c * 8 + e) * RVS2 + slot) = pr;
        }
    };

    // ---- prologue (once per workgroup): virtual rows -128 .. RQB + 143 ((136 + RQB / 2) pairs x 8 pieces); rows < 0 are zeros --------
    constexpr int PRO = 136 + RQB / 2;
    for (int t = tid; t < PRO * 8; t += NTHR) {
        uint4_t kv[2], vv[2];
        {
            const int vrow = -128 + 2 * (t >> 3);
            int ci = 0, tt = 0;
            if (vrow >= 0) advance(ci, tt, vrow);
            load_k(vrow >= 0, ci, tt, t & 7, kv);
            store_k(vrow, t & 7, kv);
        }
        {
            const int vrow = -128 + 2 * (t >> 3);
            int ci = 0, tt = 0;
            if (vrow >= 0) advance(ci, tt, vrow);
            load_v(vrow >= 0, ci, tt, t & 7, vv);
            store_v(vrow, t & 7, vv);
        }
    }
    const int nblk = (int)((L + RQB - 1) / RQB);
    // this wave's 16 queries of block `blk` start at virtual row blk RQB + 16 wave = (chunk index qc, local row qt): all 16 in one chunk (P % 16 == 0)
    auto load_q = [&](int qc, int qt, half8_t (&q)[2]) {
        const int t = qt + (lane & 15);
        q[0] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        q[1] = q[0];
        if (qc < nc && t < p.T) {
            const half_t* qp = row_ptr(qc, t);
            q[0] = *(const half8_t*)(qp + g * 8);
            q[1] = *(const half8_t*)(qp + 32 + g * 8);
        }
    };
    half8_t qn[2];
    int nqc = 0, nqt = 0;                         // (chunk index, local row) of this wave's first query in the NEXT block to run
    advance(nqc, nqt, wave * 16);
    load_q(nqc, nqt, qn);
    int sc = 0, st = 0;                           // (chunk index, local row) of virtual row RQB + 144: the first row the next block adds
    advance(sc, st, RQB + 144);
    const char* kl_lane0 = kl + ((lane & 15) << 7) + ((g ^ (lane & 7)) << 4);
    const char* kl_lane1 = kl + ((lane & 15) << 7) + (((g + 4) ^ (lane & 7)) << 4);
    const half_t* vlane = vt + (lane & 15) * RVS2 + g * 4;      // V^T fragment base of this lane: feature row lane & 15 (+ 16 mt: an immediate), column group g
    for (int b = 0; b < nblk; ++b) {
        const int v0 = __builtin_amdgcn_readfirstlane(b * RQB + wave * 16);          // virtual row of this wave's first query
        const int qc = __builtin_amdgcn_readfirstlane(nqc), qi0 = __builtin_amdgcn_readfirstlane(nqt);      // its chunk and local row
        const int qi = qi0 + (lane & 15);
        const int n = c0 + qc;
        half8_t qf[2] = {qn[0], qn[1]};
        const bool more = b + 1 < nblk && !(EXPT & 8);
        uint4_t nk[2], nv[2];
        const int nfirst = b * RQB + RQB + 144;
        // the global loads of the NEXT block (its queries, the RQB key / value rows it adds): requested now, consumed after this block
        auto issue_next = [&]() {
            if (b + 1 < nblk) {
                advance(nqc, nqt, RQB);
                nqc = __builtin_amdgcn_readfirstlane(nqc);          // (wave-uniform by construction: keep them on the scalar side)
                nqt = __builtin_amdgcn_readfirstlane(nqt);
                load_q(nqc, nqt, qn);
            }
            if (more) {
                int ci = sc, tt = st;
                advance(ci, tt, 2 * (tid >> 3));
                load_k(true, ci, tt, tid & 7, nk);
                load_v(true, ci, tt, tid & 7, nv);
                advance(sc, st, RQB);
                sc = __builtin_amdgcn_readfirstlane(sc);
                st = __builtin_amdgcn_readfirstlane(st);
            }
        };
        constexpr bool LATE = (EXPT & 32) != 0;      // (experiment) request them behind the QK^T phase instead of in front of the barrier
        if (!LATE) issue_next();
        if (!(EXPT & 16)) __syncthreads();        // ring rows of this block are in place

        if (qc < nc && qi0 < p.T) {               // (wave-uniform) a wave behind the end of the stream has nothing to compute
            const int kbase = qi0 - 128;          // local key of row 0 of this wave's tile 0; tile kt = keys kbase + 16 kt .. + 15
            // keys seen by ANY query of the wave [ulo, uhi] and by EVERY query [ilo, ihi] (queries behind the chunk included: never stored)
            const int ulo = max(qi0 - p.wl, 0), uhi = min(qi0 + 15 + p.wr, p.T - 1);
            const int ilo = max(qi0 + 15 - p.wl, 0), ihi = min(qi0 + p.wr, p.T - 1);
            const int k_first = max((ulo - kbase) >> 4, 0), k_last = min((uhi - kbase) >> 4, NT - 1);
            const int f_first = max((ilo - kbase + 15) >> 4, 0), f_last = min((ihi - kbase - 15) >> 4, NT - 1);
            const unsigned ne = k_first <= k_last ? ((2u << k_last) - 1u) & ~((1u << k_first) - 1u) : 0u;                 // non-empty tiles
            const unsigned full = f_first <= f_last ? ((2u << f_last) - 1u) & ~((1u << f_first) - 1u) & ne : 0u;         // never masked
            const int tile0 = __builtin_amdgcn_readfirstlane((v0 - 128) >> 4);      // ring tile of tile 0: VIRTUAL rows index the ring
            const int kfirst = kbase + g * 4;     // this lane's first key of tile 0
            const int lo = max(qi - p.wl, 0) - kfirst, hi = min(qi + p.wr, p.T - 1) - kfirst;
            // The pattern of a wave away from the chunk ends under the reference's window (127, 128): its 16 queries see 271 keys = tiles
            // 0 .. 16 (tile 17 is beyond every query's right edge), of which only the first and the last are partial. That case - 46 of the 63
            // working wave-blocks of a 1000-token chunk - runs as straight-line code with compile-time tile predicates (the scheduler can
            // then run fragment reads ahead of the MFMAs and MFMAs ahead of the vector work); every other case takes the same code with the
            // predicates read from the two scalar masks (a branch per tile).
            constexpr unsigned NE_STD = 0x1FFFFu, FULL_STD = 0x0FFFEu;
            auto compute = [&](auto fast_tag) {
                constexpr bool FAST = decltype(fast_tag)::value;
                float4_t s[NT];
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    if (FAST ? ((NE_STD >> kt) & 1u) != 0 : ((ne >> kt) & 1u) != 0) {
                        float4_t acc = {0.f, 0.f, 0.f, 0.f};
                        if (!(EXPT & 4)) {
                            // (measured and dropped: requesting the fragments four tiles ahead through a rotating register buffer - 0.700 against
                            //  0.690 ms per 512 chunks: the phase does not wait for its LDS reads)
                            const int toff = ((tile0 + kt) & (RING / 16 - 1)) << 11;
                            const half8_t a0 = *(const half8_t*)(kl_lane0 + toff);
                            const half8_t a1 = *(const half8_t*)(kl_lane1 + toff);
                            acc = mfma16(a0, qf[0], acc);
                            acc = mfma16(a1, qf[1], acc);
                        }
                        if (FAST ? ((FULL_STD >> kt) & 1u) == 0 : ((full >> kt) & 1u) == 0) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int rel = kt * 16 + e;
                                acc[e] = (rel >= lo && rel <= hi) ? acc[e] : -INFINITY;
                            }
                        }
                        s[kt] = acc;
                        m = fmaxf(fmaxf(m, acc[0]), acc[1]);
                        m = fmaxf(fmaxf(m, acc[2]), acc[3]);
                    }
                }
                if (LATE) issue_next();
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                const float msafe = (m == -INFINITY) ? 0.0f : m;
                typedef float f2_t __attribute__((ext_vector_type(2)));
                const f2_t m2 = {msafe, msafe};
                f2_t sum2 = {0.0f, 0.0f};
                float4_t o[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) o[mt] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NT / 2; ++c) {
                    if (FAST ? ((NE_STD >> (2 * c)) & 3u) != 0 : ((ne >> (2 * c)) & 3u) != 0) {
                        float4_t pr[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int kt = 2 * c + u;
                            if (FAST ? ((NE_STD >> kt) & 1u) != 0 : ((ne >> kt) & 1u) != 0) {
                                f2_t a = f2_t{s[kt][0], s[kt][1]} - m2, bb = f2_t{s[kt][2], s[kt][3]} - m2;
                                if (!(EXPT & 1)) {
                                    a.x = __builtin_amdgcn_exp2f(a.x); a.y = __builtin_amdgcn_exp2f(a.y);
                                    bb.x = __builtin_amdgcn_exp2f(bb.x); bb.y = __builtin_amdgcn_exp2f(bb.y);
                                }
                                sum2 += a;
                                sum2 += bb;
                                pr[u] = float4_t{a.x, a.y, bb.x, bb.y};
                            } else {
                                pr[u] = float4_t{0.f, 0.f, 0.f, 0.f};
                            }
                        }
                        const float8_t pf = {pr[0][0], pr[0][1], pr[0][2], pr[0][3], pr[1][0], pr[1][1], pr[1][2], pr[1][3]};
                        const half8_t pb = __builtin_convertvector(pf, half8_t);
                        if (!(EXPT & 2)) {
                            const int u0 = ((tile0 + 2 * c) & (RING / 16 - 1)) << 4, u1 = ((tile0 + 2 * c + 1) & (RING / 16 - 1)) << 4;
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt) {
                                const half4_t va = *(const half4_t*)(vlane + mt * 16 * RVS2 + u0);
                                const half4_t vb = *(const half4_t*)(vlane + mt * 16 * RVS2 + u1);
                                const half8_t af = __builtin_shufflevector(va, vb, 0, 1, 2, 3, 4, 5, 6, 7);
                                o[mt] = mfma16(af, pb, o[mt]);
                            }
                        } else {
                            o[c & 3][0] += (float)pb[0] + (float)pb[5];
                        }
                    }
                }
                float sum = sum2.x + sum2.y;
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
                if (qi < p.T) {
                    half_t* op = p.out + ((long)n * p.T + qi) * D + h * HD + g * 4;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        half4_t ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = (half_t)(o[mt][e] * inv);
                        *(half4_t*)(op + mt * 16) = ov;
                    }
                }
            };
            if (ne == NE_STD && full == FULL_STD) compute(std::true_type{});
            else compute(std::false_type{});
        } else if (LATE) {
            issue_next();
        }
        if (!(EXPT & 16)) __syncthreads();        // everybody is done with the rows the next block overwrites
        if (more) {
            store_k(nfirst + 2 * (tid >> 3), tid & 7, nk);
            store_v(nfirst + 2 * (tid >> 3), tid & 7, nv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// out[m][:] = rmsnorm(a[m][:] + alpha * x[m][:]) * w     (fp32 statistics, eps inside the sqrt)
struct NormArgs {
    const half_t* a;
    const half_t* x;
    const float* w;
    half_t* out;
    long M;
    int D;
    float alpha, eps;
};

__global__ __launch_bounds__(256) void rmsnorm_residual_kernel(NormArgs p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const half_t* a = p.a + row * p.D;
    const half_t* x = p.x != nullptr ? p.x + row * p.D : nullptr;     // null: `a` already holds the residual sum (fused into the producing GEMM)
    half_t* o = p.out + row * p.D;
    float z[2][8];     // D <= 1024: up to two 16-byte vectors per lane
    float ss = 0.0f;
    const int nv = p.D >> 3;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int idx = lane + 64 * v;
        if (idx < nv) {
            const half8_t av = *(const half8_t*)(a + idx * 8);
            half8_t xv = av;
            if (x != nullptr) xv = *(const half8_t*)(x + idx * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                z[v][e] = x != nullptr ? (float)av[e] + p.alpha * (float)xv[e] : (float)av[e];
                ss += z[v][e] * z[v][e];
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float r = rsqrtf(ss / (float)p.D + p.eps);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int idx = lane + 64 * v;
        if (idx < nv) {
            half8_t ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (half_t)(z[v][e] * r * p.w[idx * 8 + e]);
            *(half8_t*)(o + idx * 8) = ov;
        }
    }
}

}  // namespace bh

int bh_k_attention(const void* qkv, void* out, const float* cos_sin, int N, int T, int nhead, int head_dim,
                   int win_left, int win_right, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(head_dim == 64, "attention: only head_dim 64 is implemented (got %d)", head_dim);
    BH_REQUIRE(win_left >= 0 && win_right >= 0, "attention: a finite window (left, right) is required");
    BH_REQUIRE(N > 0 && T > 0 && nhead > 0, "attention: empty problem");
    const int need = (16 + win_left + win_right + 15) / 16;      // key tiles one wave can see
    AttnArgs a{(const half_t*)qkv, (half_t*)out, cos_sin, N, T, nhead, win_left, win_right, 0.125f};
    dim3 grid((T + QB - 1) / QB, nhead, N);
#define BH_ATTN(NT)                                                                                   \
    do {                                                                                              \
        const size_t lds = (size_t)(7 * 16 + NT * 16) * 128 + (size_t)64 * (7 * 16 + NT * 16 + 4) * 2; \
        if (lds > 64 * 1024)                                                                          \
            BH_CHECK_HIP(bh_max_lds((const void*)attention_kernel<NT>, (int)lds)); \
        hipLaunchKernelGGL(attention_kernel<NT>, grid, dim3(512), lds, stream, a);                    \
    } while (0)
    if (need <= 6) BH_ATTN(6);
    else if (need <= 10) BH_ATTN(10);
    else if (need <= 18) BH_ATTN(18);
    else if (need <= 26) BH_ATTN(26);
    else BH_REQUIRE(false, "attention: window %d+%d is too wide for the LDS-resident kernel", win_left, win_right);
#undef BH_ATTN
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// q (already rotated and scaled) | k (already rotated) | v  ->  attention output; see attention_ring_kernel.
int g_attn_waves = 0;      // bh_set_option("attn_waves", 0 | 8 | 12): 0 = automatic
int g_attn_version = 2;    // bh_set_option("attn_version", 1 | 2): 1 = the ring kernel of rounds 2-5, 2 = its round-6 rebuild (default)
int g_attn_expt = 0;       // bh_set_option("attn_expt", bits): timing experiments of the round-6 kernel (wrong results on purpose; 0 in the product)

int bh_k_attention_prerotated(const void* qkv, void* out, int N, int T, int nhead, int head_dim, int win_left, int win_right,
                              hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(head_dim == 64, "attention: only head_dim 64 is implemented (got %d)", head_dim);
    BH_REQUIRE(win_left >= 0 && win_right >= 0 && win_left <= 128 && win_left + win_right <= 256,
               "attention (ring): window (%d, %d) outside the supported range", win_left, win_right);
    BH_REQUIRE(N > 0 && T > 0 && nhead > 0, "attention: empty problem");
    AttnRingArgs a{(const half_t*)qkv, (half_t*)out, N, T, nhead, win_left, win_right, 1};
    // twelve waves (blocks of 192 queries) where the chunk is long enough to fill them; "attn_waves" 8 / 12 forces a geometry
    const int waves = g_attn_waves == 8 || g_attn_waves == 12 ? g_attn_waves : (T >= 384 ? 12 : 8);
    if (g_attn_version == 1) {
        const size_t lds = (size_t)RING * 128 + (size_t)64 * RVS * 2;
        if (waves == 12) {
            BH_CHECK_HIP(bh_max_lds((const void*)attention_ring_kernel<12>, (int)lds));
            hipLaunchKernelGGL(attention_ring_kernel<12>, dim3(nhead, N), dim3(768), lds, stream, a);
        } else {
            BH_CHECK_HIP(bh_max_lds((const void*)attention_ring_kernel<8>, (int)lds));
            hipLaunchKernelGGL(attention_ring_kernel<8>, dim3(nhead, N), dim3(512), lds, stream, a);
        }
        BH_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const size_t lds = (size_t)RING * 128 + (size_t)64 * RVS2 * 2;
    // chunks per workgroup: one workgroup per CU and head-column (eight columns = the eight XCDs walk the same chunks together)
    const int ncu = bh_cu_count() > 0 ? bh_cu_count() : 256;
    const int cols = ncu / nhead > 0 ? ncu / nhead : 1;
    a.cpw = (N + cols - 1) / cols;
    const int groups = (N + a.cpw - 1) / a.cpw;
#define BH_RING2(W, E)                                                                                          \
    do {                                                                                                        \
        BH_CHECK_HIP(bh_max_lds((const void*)attention_ring2_kernel<W, E>, (int)lds));                          \
        hipLaunchKernelGGL((attention_ring2_kernel<W, E>), dim3(nhead, groups), dim3(64 * W), lds, stream, a);   \
    } while (0)
    if (waves == 8) BH_RING2(8, 0);
    else if (g_attn_expt == 0) BH_RING2(12, 0);
#ifdef BH_ATTN_EXPT
    else if (g_attn_expt == 1) BH_RING2(12, 1);
    else if (g_attn_expt == 2) BH_RING2(12, 2);
    else if (g_attn_expt == 4) BH_RING2(12, 4);
    else if (g_attn_expt == 7) BH_RING2(12, 7);
    else if (g_attn_expt == 8) BH_RING2(12, 8);
    else if (g_attn_expt == 16) BH_RING2(12, 16);
    else if (g_attn_expt == 24) BH_RING2(12, 24);
    else if (g_attn_expt == 31) BH_RING2(12, 31);
    else if (g_attn_expt == 32) BH_RING2(12, 32);

#endif
    else BH_REQUIRE(false, "attention: timing experiment %d is not compiled into this library (BH_ATTN_EXPT)", g_attn_expt);
#undef BH_RING2
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

int bh_k_rmsnorm_residual(const void* a, const void* x, const float* w, void* out, long M, int D, float alpha,
                          float eps, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(D % 8 == 0 && D <= 1024 && D > 0, "rmsnorm: D must be a multiple of 8 and <= 1024 (got %d)", D);
    NormArgs na{(const half_t*)a, (const half_t*)x, w, (half_t*)out, M, D, alpha, eps};
    hipLaunchKernelGGL(rmsnorm_residual_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, na);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}
