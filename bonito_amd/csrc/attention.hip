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

int bh_k_attention_prerotated(const void* qkv, void* out, int N, int T, int nhead, int head_dim, int win_left, int win_right,
                              hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(head_dim == 64, "attention: only head_dim 64 is implemented (got %d)", head_dim);
    BH_REQUIRE(win_left >= 0 && win_right >= 0 && win_left <= 128 && win_left + win_right <= 256,
               "attention (ring): window (%d, %d) outside the supported range", win_left, win_right);
    BH_REQUIRE(N > 0 && T > 0 && nhead > 0, "attention: empty problem");
    AttnRingArgs a{(const half_t*)qkv, (half_t*)out, N, T, nhead, win_left, win_right};
    const size_t lds = (size_t)RING * 128 + (size_t)64 * RVS * 2;
    // twelve waves (blocks of 192 queries) where the chunk is long enough to fill them; "attn_waves" 8 / 12 forces a geometry
    const int waves = g_attn_waves == 8 || g_attn_waves == 12 ? g_attn_waves : (T >= 384 ? 12 : 8);
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

int bh_k_rmsnorm_residual(const void* a, const void* x, const float* w, void* out, long M, int D, float alpha,
                          float eps, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(D % 8 == 0 && D <= 1024 && D > 0, "rmsnorm: D must be a multiple of 8 and <= 1024 (got %d)", D);
    NormArgs na{(const half_t*)a, (const half_t*)x, w, (half_t*)out, M, D, alpha, eps};
    hipLaunchKernelGGL(rmsnorm_residual_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, na);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}
