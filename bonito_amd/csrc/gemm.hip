// fp16 "linear layer" GEMM on MFMA for gfx950:   out[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] )
//
// Replaces the cuBLAS calls behind torch.nn.Linear on the reference's hot path
// (bonito/nn.py:283-298 LinearCRFEncoder, nn.py:140-159 LinearUpsample, LSTM input projections
// nn.py:396-415, transformer Wqkv/out_proj/fc1/fc2 bonito/transformer/model.py:52-53,102-109).
//
// Design (MI355X-first, not a CUDA tiling):
//  * W is the MFMA *A* operand and X the *B* operand, so one lane's accumulators are
//    4 (reg) x 4 (feature tiles) = 16 CONSECUTIVE output features of one token: the epilogue
//    (bias, activation, scale, clamp, SwiGLU gating) is lane-local and stores two 16-byte
//    vectors per token instead of 2-byte scattered stores.
//  * 128(features) x 128(tokens) x 64(K) tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
//  * LDS tiles are [row][64 halves] with the 16-byte chunk index XOR-swizzled by (row & 7): the
//    ds_read_b128 fragment reads are bank-conflict free (see DESIGN.md "GEMM LDS layout").
//  * W rows are staged into LDS already permuted into MFMA fragment order so the read side is the
//    same conflict-free pattern for both operands.
//  * XCD-aware block remap: blocks that land on the same XCD (blockIdx % 8) walk consecutive
//    feature tiles of the same token tile, so the X tile is fetched into that XCD's L2 once.
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace bh {

struct GemmArgs {
    const half_t* X;   // [M][ldx]
    const half_t* W;   // [N][ldw]
    const float* bias; // [N] or null
    const half_t* res; // optional residual [M][ldres]: res_scale * res is added before the activation
    int ldres;
    float res_scale = 1.0f;
    half_t* out;       // rows remapped, see below; [.][ldo]
    int M, N, K;
    int ldx, ldw, ldo;
    float scale;       // applied after activation
    float clamp_lo, clamp_hi;
    int row_div;       // out_row = (m / row_div) * row_s_hi + (m % row_div) * row_s_lo
    long row_s_hi, row_s_lo;
    int row_lim;       // rows with (m % row_div) >= row_lim are not stored (batch padding)
    int n_ft, n_tt;    // tile counts
    int stagger = 0;   // persistent kernels: workgroup b starts ((b >> 3) & 7) * stagger * 1024 cycles late (see gemm_stagger_start)
    // optional rotary epilogue for the packed Wqkv projection (features [0, rot_nfeat) are heads of 64 rotated in place
    // by position m % rot_T; features [0, rot_qfeat) are scaled by rot_qscale afterwards)
    const float* rot_cs = nullptr;   // [rot_T][32][2] (cos, sin)
    int rot_T = 1, rot_nfeat = 0, rot_qfeat = 0;
    float rot_qscale = 1.0f;
    int w4_gf = 4;     // gemm_w4_kernel: feature tiles per block of its work order (1, 2, 4, 8, 16 or 32; see the kernel)
    int w4_order = 0;  // gemm_w4_kernel: 0 = token blocks fastest inside an XCD's share, 1 = feature groups fastest (see the kernel)
#ifdef BH_GEMM_STATS
    unsigned long long* dbg = nullptr;   // tools/gemm_lab.hip: cycle stamps of workgroup 0 (K loops, epilogues, total, real-time ticks, tiles)
#endif
};

constexpr int BF = 128, BT = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * (BK * 2) + ((chunk ^ (row & 7)) << 4);
}

typedef float float8_t __attribute__((ext_vector_type(8)));

// Epilogue. lane (c = r: token column, q = kg): features fbase + ft*4 + reg, ft,reg in 0..3; a wave owns NTT token tiles.
// With short K (384..512 on this path) the epilogue is a third of a tile's instructions, so everything that is uniform
// over the launch is branched on once (identity row map -> no integer division, no scale/clamp -> no extra VALU),
// residuals are fetched as two 16-byte loads and the fp32 -> fp16 conversion uses the packed form (v_cvt_pk_f16_f32).
template <int ACT, bool GATED, int NTT = 4, bool ALIGNED = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, float4_t (&acc)[4][NTT], int f0, int t0, int wf, int wt,
                                              int r, int kg) {
    const int fbase = f0 + wf * 64 + kg * 16;
    // this lane's 16 features all exist, or (ALIGNED: N % 16 == 0 guaranteed by the launcher) none of them does
    const bool full = fbase + 16 <= p.N;
    if (ALIGNED && !full) return;
    const bool ident = p.row_div == 1 && p.row_s_hi == 1;                   // out row == m, nothing dropped
    const bool plain = p.scale == 1.0f && p.clamp_lo == -INFINITY && p.clamp_hi == INFINITY;
    float8_t b0 = 0.0f, b1 = 0.0f;
    if (p.bias != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            b0[i] = fbase + i < p.N ? p.bias[fbase + i] : 0.0f;
            b1[i] = fbase + 8 + i < p.N ? p.bias[fbase + 8 + i] : 0.0f;
        }
    }
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int m = t0 + wt * (NTT * 16) + tt * 16 + r;
        if (m >= p.M) continue;
        long orow = m;
        if (!ident) {
            const int hi = m / p.row_div, lo = m - hi * p.row_div;
            if (lo >= p.row_lim) continue;
            orow = (long)hi * p.row_s_hi + (long)lo * p.row_s_lo;
        }
        float8_t v0, v1;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v0[ft * 4 + g] = acc[ft][tt][g];
                v1[ft * 4 + g] = acc[ft + 2][tt][g];
            }
        v0 += b0;
        v1 += b1;
        if (p.res != nullptr) {
            const half_t* rp = p.res + (long)m * p.ldres + fbase;
            if (ALIGNED || full) {
                v0 += p.res_scale * __builtin_convertvector(*(const half8_t*)rp, float8_t);
                v1 += p.res_scale * __builtin_convertvector(*(const half8_t*)(rp + 8), float8_t);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (fbase + i < p.N) v0[i] += p.res_scale * (float)rp[i];
                    if (fbase + 8 + i < p.N) v1[i] += p.res_scale * (float)rp[8 + i];
                }
            }
        }
        if constexpr (!GATED) {
            // Rotary embedding fused into the Wqkv projection (bonito/transformer/model.py:72-73 applies RotaryEmbedding to
            // q and k right after Wqkv): a wave's 64 features are one head, lane group kg holds dims 16kg..16kg+15, so the
            // rotation partner (dim +-32) of every value sits in lane ^ 32. Done on the fp32 accumulators, before the only
            // rounding to fp16. The head-level test is wave-uniform, so all 64 lanes take part in the exchange.
            if (p.rot_cs != nullptr && f0 + wf * 64 < p.rot_nfeat) {
                const int pos = m % p.rot_T;
                const float* cs = p.rot_cs + ((long)pos * 32 + (kg & 1) * 16) * 2;
                float8_t pa, pb;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    pa[i] = __shfl_xor(v0[i], 32);
                    pb[i] = __shfl_xor(v1[i], 32);
                }
                const float sgn = kg < 2 ? -1.0f : 1.0f;        // x1' = x1 cos - x2 sin ; x2' = x1 sin + x2 cos
                const float qs = f0 + wf * 64 < p.rot_qfeat ? p.rot_qscale : 1.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4_t c4 = *(const float4_t*)(cs + 4 * i);      // (cos, sin) of dims 2i, 2i+1 of this quarter
                    if (i < 4) {
                        v0[2 * i] = (v0[2 * i] * c4[0] + sgn * pa[2 * i] * c4[1]) * qs;
                        v0[2 * i + 1] = (v0[2 * i + 1] * c4[2] + sgn * pa[2 * i + 1] * c4[3]) * qs;
                    } else {
                        v1[2 * i - 8] = (v1[2 * i - 8] * c4[0] + sgn * pb[2 * i - 8] * c4[1]) * qs;
                        v1[2 * i - 7] = (v1[2 * i - 7] * c4[2] + sgn * pb[2 * i - 7] * c4[3]) * qs;
                    }
                }
            }
        }
        if constexpr (GATED) {
            // W rows were interleaved on the host: feature 2j = y_j, 2j+1 = gate_j
            // (flash_attn GatedMlp semantics: y, gate = fc1(x).chunk(2); y * silu(gate)).
            float8_t y;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y[i] = v0[2 * i] * swishf_(v0[2 * i + 1]);
                y[4 + i] = v1[2 * i] * swishf_(v1[2 * i + 1]);
            }
            const half8_t o = __builtin_convertvector(y, half8_t);
            const int fo = fbase >> 1;
            if (ALIGNED || fo + 8 <= (p.N >> 1)) *(half8_t*)(p.out + orow * p.ldo + fo) = o;
            else
                for (int i = 0; i < 8; ++i)
                    if (fo + i < (p.N >> 1)) p.out[orow * p.ldo + fo + i] = o[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v0[i] = apply_act<ACT>(v0[i]);
                v1[i] = apply_act<ACT>(v1[i]);
            }
            if (!plain) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v0[i] = fminf(fmaxf(v0[i] * p.scale, p.clamp_lo), p.clamp_hi);
                    v1[i] = fminf(fmaxf(v1[i] * p.scale, p.clamp_lo), p.clamp_hi);
                }
            }
            const half8_t o0 = __builtin_convertvector(v0, half8_t), o1 = __builtin_convertvector(v1, half8_t);
            half_t* dst = p.out + orow * p.ldo + fbase;
            if (ALIGNED || full) {
                *(half8_t*)dst = o0;
                *(half8_t*)(dst + 8) = o1;
            } else {
                for (int i = 0; i < 16; ++i)
                    if (fbase + i < p.N) dst[i] = i < 8 ? o0[i] : o1[i - 8];
            }
        }
    }
}

template <int ACT, bool GATED>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [buf][A tile | B tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wf = wave >> 1, wt = wave & 1;
    const int r = lane & 15, kg = lane >> 4;

    // XCD-aware bijective remap of the linear block id (guide T1).
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int tile_t = work / p.n_ft;
    const int tile_f = work - tile_t * p.n_ft;
    const int f0 = tile_f * BF, t0 = tile_t * BT;

    // staging assignment: 4 chunks per operand per thread
    const int sc = tid & 7;
    const int srow = tid >> 3;  // + 32*j
    const half_t* xsrc[4];
    const half_t* wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int row = srow + 32 * j;
        int tok = min(t0 + row, p.M - 1);
        xsrc[j] = p.X + (long)tok * p.ldx + sc * 8;
        int wfb = row >> 6, within = row & 63;
        int ft = within >> 4, rr = within & 15;
        int feat = min(f0 + wfb * 64 + (rr >> 2) * 16 + ft * 4 + (rr & 3), p.N - 1);
        wsrc[j] = p.W + (long)feat * p.ldw + sc * 8;
    }

    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    uint4_t xr[4], wr[4];
    auto gload = [&](int kt) {
        int kofs = kt * BK + sc * 8;
        bool ok = kofs < p.K;  // K % 8 == 0 is required by the host wrapper
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xr[j] = ok ? *(const uint4_t*)(xsrc[j] + kt * BK) : uint4_t{0, 0, 0, 0};
            wr[j] = ok ? *(const uint4_t*)(wsrc[j] + kt * BK) : uint4_t{0, 0, 0, 0};
        }
    };
    auto lstore = [&](int buf) {
        char* a = smem + buf * 2 * TILE_BYTES;
        char* b = a + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int row = srow + 32 * j;
            *(uint4_t*)(a + lds_off(row, sc)) = wr[j];
            *(uint4_t*)(b + lds_off(row, sc)) = xr[j];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const char* a = smem + (kt & 1) * 2 * TILE_BYTES;
        const char* b = a + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8_t af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *(const half8_t*)(a + lds_off(wf * 64 + i * 16 + r, ks * 4 + kg));
                bf[i] = *(const half8_t*)(b + lds_off(wt * 64 + i * 16 + r, ks * 4 + kg));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
        if (kt + 1 < nk) lstore((kt + 1) & 1);
        __syncthreads();
    }

    gemm_epilogue<ACT, GATED>(p, acc, f0, t0, wf, wt, r, kg);
}

// ---------------------------------------------------------------------------------------------------
// v2: direct global->LDS staging (global_load_lds, 16 B per lane, no VGPR round trip), BK = 32, two LDS
// buffers of 16 KiB -> 32 KiB per workgroup and <= 128 VGPRs, so 4 workgroups (16 waves) share a CU and
// hide each other's HBM / barrier latency. One barrier per K-step: wait(tile k) -> barrier -> issue the DMA
// of tile k+1 into the other buffer -> MFMAs of tile k.
// The DMA writes LDS linearly (wave-uniform base + lane*16), so the bank swizzle is applied on the SOURCE
// side: lane (row r = lane>>2, slot = lane&3) fetches global chunk slot ^ g[(r>>2)&3], g = {0,3,2,1}; a
// fragment read of chunk kg of row r goes to slot kg ^ g[(r>>2)&3]. With that permutation every 16-lane
// group of a ds_read_b128 fragment read touches 16 distinct 16-byte slots (derivation: DESIGN.md).
// Requires K % 32 == 0 (no K tail) -- the launcher falls back to gemm_kernel otherwise.
constexpr int BK2 = 32;
constexpr int TILE2 = 128 * BK2 * 2;   // 8 KiB per operand tile

__device__ __forceinline__ int swz2(int r) { return (0x1230 >> (((r >> 2) & 3) * 4)) & 3; }   // g = {0,3,2,1}

template <int ACT, bool GATED>
__global__ __launch_bounds__(256, 4) void gemm_glds_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [buf][A 8K | B 8K]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = wave >> 1, wt = wave & 1;
    const int r = lane & 15, kg = lane >> 4;

    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int tile_t = work / p.n_ft;
    const int tile_f = work - tile_t * p.n_ft;
    const int f0 = tile_f * BF, t0 = tile_t * BT;

    // DMA assignment: wave w, instruction j in {0,1} covers LDS rows (w*2+j)*16 .. +15 of each operand tile
    const int lr = lane >> 2, slot = lane & 3;
    const half_t* asrc[2];
    const half_t* bsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + lr;
        const int chunk = slot ^ swz2(lr);
        const int wfb = row >> 6, within = row & 63;
        const int ft = within >> 4, rr = within & 15;
        const int feat = min(f0 + wfb * 64 + (rr >> 2) * 16 + ft * 4 + (rr & 3), p.N - 1);
        asrc[j] = p.W + (long)feat * p.ldw + chunk * 8;
        const int tok = min(t0 + row, p.M - 1);
        bsrc[j] = p.X + (long)tok * p.ldx + chunk * 8;
    }
    auto dma = [&](int kt, int buf) {
        char* a = smem + buf * 2 * TILE2;
        char* b = a + TILE2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = (wave * 2 + j) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + kt * BK2),
                                             (__attribute__((address_space(3))) void*)(a + off), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[j] + kt * BK2),
                                             (__attribute__((address_space(3))) void*)(b + off), 16, 0, 0);
        }
    };

    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK2;
    const int fpos = (kg ^ swz2(r)) << 4;          // byte position of this lane's chunk inside a 64-byte row
    dma(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile kt has landed (this wave's part)
        __syncthreads();                                    // ... and everybody's; all reads of tile kt-1 are done
        if (kt + 1 < nk) dma(kt + 1, (kt + 1) & 1);
        const char* a = smem + (kt & 1) * 2 * TILE2;
        const char* b = a + TILE2;
        half8_t af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = *(const half8_t*)(a + (wf * 64 + i * 16 + r) * 64 + fpos);
            bf[i] = *(const half8_t*)(b + (wt * 64 + i * 16 + r) * 64 + fpos);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
    }
    gemm_epilogue<ACT, GATED>(p, acc, f0, t0, wf, wt, r, kg);
}

// ---------------------------------------------------------------------------------------------------
// v3: 256 (features) x 256 (tokens) x 64 (K) tile, 8 waves (4 along features x 2 along tokens; a wave owns
// 64 x 128 = 4 x 8 MFMA tiles, 128 accumulator registers), global_load_lds staging into two 64 KiB buffers with the
// XOR swizzle applied on the source address (LDS rows are 128 B: chunk ^= row & 7, the v1 read pattern), one barrier
// per K-tile. 2.7 MFMAs per ds_read_b128 instead of 2 and half the barriers of v2. The kernel is PERSISTENT over
// output tiles: with K = 384..2048 a tile is only 6..32 K-tiles, so the first DMA of the next tile is issued before
// the epilogue of the current one (LDS is idle while the accumulators drain).
// Requires K % 64 == 0 and N % 16 == 0; used when the problem has enough 256 x 256 tiles to fill the chip.
// Measured on the transformer shapes (M = 256000): +10..22 % over v2 (e.g. fc2 K=2048: 774 -> 946 TFLOP/s, gated fc1:
// 734 -> 864). A 256 x 128 tile with two workgroups per CU landed in between (~750 everywhere). A three-buffer BK=32 pipeline with counted vmcnt waits (two stages in flight) was tried and dropped:
// hipcc drains the vm counter in front of every ds_read it can see while an LDS-DMA is outstanding, and hiding the
// reads in inline asm pushed the persistent kernel over the register budget. A non-persistent four-buffer version (prefetch
// distance 3, builtin s_waitcnt, asm fragment reads, no spurious waits left) was measured too: 631-991 TFLOP/s on the same
// shapes, no better than this kernel -- the DMA latency is not what bounds it; with a barrier per stage both waves of a SIMD
// read LDS at the same time and issue MFMAs at the same time (the fix is a phase-staggered schedule, not more buffers).
// The persistent kernels run compute (K loop) and store (epilogue) phases of equal length on every CU; launched together, all 256 CUs
// reach their epilogues together and the output tile of every one of them (128 KiB) goes to HBM in one burst while the matrix cores
// of the whole chip idle - then nobody writes for a K loop. Starting the workgroups of an XCD in eight phase groups spreads the
// bursts over the tile period: stores of one group overlap the K loops of the others ("gemm_stagger" option, units of 1024 cycles).
__device__ __forceinline__ void gemm_stagger_start(int units) {
    const int slot = (blockIdx.x >> 3) & 7;
    for (int i = 0; i < slot * units; ++i) __builtin_amdgcn_s_sleep(16);
}

constexpr int BF3 = 256, BT3 = 256, BK3 = 64;
constexpr int TILE3 = 256 * BK3 * 2;    // 32 KiB per operand tile

template <int ACT, bool GATED>
__global__ __launch_bounds__(512, 1) void gemm_big_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [buf][A 32K | B 32K]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = wave >> 1, wt = wave & 1;
    const int r = lane & 15, kg = lane >> 4;
    const int n_tiles = p.n_ft * p.n_tt;
    const int nk = p.K / BK3;

    // DMA assignment: an instruction covers 8 LDS rows x 128 B; wave w issues rows (w*4 + j)*8 .. +7 of each operand, j < 4
    const int lr = lane >> 3, slot = lane & 7;
    const int chunk = slot ^ lr;                       // row & 7 == lr (8-row groups are aligned)
    unsigned aoff[4], boff[4];                         // element offsets of this lane's 16-byte chunk, per instruction
    // XCD-aware bijective remap of the linear work index (block b runs on XCD b % 8, and so does b + k * gridDim when
    // gridDim % 8 == 0): every XCD walks one contiguous eighth of the tile space, feature tiles fastest, so an X tile is
    // fetched into ONE L2 instead of up to eight
    const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
    auto set_tile = [&](int w, int& f0, int& t0) {
        const int xcd = w & 7, loc = w >> 3;
        const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
        const int tile_t = work / p.n_ft;
        const int tile_f = work - tile_t * p.n_ft;
        f0 = tile_f * BF3;
        t0 = tile_t * BT3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (wave * 4 + j) * 8 + lr;
            const int wfb = row >> 6, within = row & 63;
            const int ft = within >> 4, rr = within & 15;
            const int feat = min(f0 + wfb * 64 + (rr >> 2) * 16 + ft * 4 + (rr & 3), p.N - 1);
            aoff[j] = (unsigned)feat * (unsigned)p.ldw + chunk * 8;
            const int tok = min(t0 + row, p.M - 1);
            boff[j] = (unsigned)tok * (unsigned)p.ldx + chunk * 8;
        }
    };
    auto dma = [&](int kt, int buf) {
        char* a = smem + buf * 2 * TILE3;
        char* b = a + TILE3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = (wave * 4 + j) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.W + aoff[j] + kt * BK3),
                                             (__attribute__((address_space(3))) void*)(a + off), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.X + boff[j] + kt * BK3),
                                             (__attribute__((address_space(3))) void*)(b + off), 16, 0, 0);
        }
    };

    int work = blockIdx.x;
    if (work >= n_tiles) return;
    gemm_stagger_start(p.stagger);
    int f0, t0;
    set_tile(work, f0, t0);
    dma(0, 0);
    int buf = 0;
    while (true) {
        float4_t acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
        const int next = work + gridDim.x;
        int nf0 = 0, nt0 = 0;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // K-tile kt has landed (this wave's part)
            __syncthreads();                                    // ... everybody's; all reads of the other buffer are done
            if (kt + 1 < nk) dma(kt + 1, buf ^ 1);
            else if (next < n_tiles) {                          // first K-tile of the next output tile, under the epilogue
                const int cf0 = f0, ct0 = t0;
                set_tile(next, nf0, nt0);
                dma(0, buf ^ 1);
                f0 = cf0; t0 = ct0;
            }
            const char* a = smem + buf * 2 * TILE3;
            const char* b = a + TILE3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8_t af[4], bf[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const half8_t*)(a + lds_off(wf * 64 + i * 16 + r, ks * 4 + kg));
#pragma unroll
                for (int j = 0; j < 8; ++j) bf[j] = *(const half8_t*)(b + lds_off(wt * 128 + j * 16 + r, ks * 4 + kg));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
            }
            buf ^= 1;
        }
        gemm_epilogue<ACT, GATED, 8, true>(p, acc, f0, t0, wf, wt, r, kg);
        if (next >= n_tiles) break;
        work = next; f0 = nf0; t0 = nt0;
    }
}

// (Round 3, measured and not kept: "v4" = this kernel with the fragment reads software-pipelined across the barrier - barrier in the
// middle of a K-tile, the next phase's twelve fragments read during the first six MFMA groups of the current one, DMA from inline
// assembly with scalar bases, one instruction per MFMA group; the compiled K loop was clean (no spills, one lgkmcnt wait per phase).
// Bit-identical, and no faster: 634 vs 752 TFLOP/s at K = 512 (more per-tile overhead), 1030-1060 vs 978-985 at K = 2048, 917 vs 956
// at K = 1024. The SQ counters say why (profiles/r03_sq_counters_other_models.txt): the waves of v3 wait 40 % of their time, and
// SQ_WAIT_INST_LDS is 1.6 % - they wait for the global -> LDS stream (vmcnt(0) in front of the barrier: 64 KiB per K-tile and CU,
// ~6-9 TB/s out of the L2s chip-wide at these rates), not for fragment reads. The next step is a deeper DMA pipeline (three stages
// need BK = 32 or a 256 x 128 tile: 2 x 64 KiB is what fits beside nothing else), not a better MFMA schedule. Also measured: v3 with its
// DMA issued from inline assembly (scalar base, so that the compiler inserts no vmcnt drain in front of the fragment reads) and a counted
// vmcnt behind the epilogue (its sixteen stores are younger than the next tile's first DMA): 586 vs 758 TFLOP/s at K = 512, 728 vs 985 at
// K = 2048 - the eight asm statements pin the DMA issue (~50 cycles each) in front of the K-tile's fragment reads, where the builtin
// lets the scheduler spread them; the builtin form stays. And "deep": a FOUR-stage pipeline of 32-wide K slices (three requests in
// flight, counted vmcnt waits, conflict-free 64-byte-row planes with chunk ^= (row >> 2) & 3, no compiler waits left in the K loop,
// bit-identical): 674-930 TFLOP/s on the same shapes, within 3 % of v3 everywhere but Wqkv (787 vs 674) - the DMA latency is not the
// bound either. Three schedules of the same work (v3, v4, deep) run at the same rate, like the stream variants of the recurrent kernel
// (DESIGN.md 4a "the clock"): the kernel is most likely at the chip's power limit, and what separates it from the library's
// 1.15-1.24 PFLOP/s at K >= 1024 is energy per MFMA - 2.7 MFMAs per 1 KiB fragment read here; a 128 x 128 wave tile gives 4, and
// 32 x 32 x 16 MFMAs halve the operand reads again.)

// ---------------------------------------------------------------------------------------------------
// v5 ("w4", round 4; round 6 added the 16x16x32 stream below, now the default): the same 256 x 256 x 64 workgroup tile on FOUR waves - one per
// SIMD, each owning a 128 x 128 wave tile on v_mfma_f32_32x32x16_f16 (256 accumulator registers) - with the whole K-tile as ONE generated instruction stream
// (tools/gen_gemmstep.py -> gemm_ktile_mfma.inc): 64 MFMAs with the 32 fragment reads of the next k-steps, the wave's 16 LDS-DMA
// pieces of the K-tiles ahead and one barrier issued from inside the stream. What that changes against v3 (8 waves, 64 x 128
// wave tiles on 16 x 16 x 32, compiler-scheduled, everything of a K-tile - wait, barrier, DMA issue, fragment reads, MFMAs - one
// after the other with both waves of a SIMD in the same phase: matrix pipe busy 41-50 % per cycle):
//   * 4 MFMAs per KiB of fragment reads instead of 2.7, half the operand-register reads per FLOP, half the MFMA instructions:
//     less energy per FLOP on a kernel that sits at the board's power cap;
//   * DMA issue (~60-180 cycles each), fragment reads and the barrier hide behind MFMAs of the same wave instead of standing in
//     front of them;
//   * output tiles are written as FULL 128-byte lines: a lane's 32 consecutive features of a token go through a per-wave 8 KiB LDS
//     scratch (XOR-swizzled, no barrier: one wave) and leave as rows of 8 lanes x 16 B. v3's stores were 64 separate 16-byte
//     pieces per instruction - the "17 k cycles of turn-around per output tile" of round 3 were mostly that.
// LDS: two stages of (W tile 32 KiB | X tile 32 KiB) + 4 x 8 KiB epilogue scratch = 160 KiB. Rows are 128 B = eight 16-byte chunks;
// chunk c of row r sits at position c ^ ((r >> 1) & 7): a 32x32x16 fragment read takes rows l & 31 and chunk 2 ks + (l >> 5), so
// every 16-lane group of the ds_read_b128 (lanes of one half, sixteen distinct rows of which eight even, eight odd) touches
// sixteen distinct 16-byte slots of the 256-byte bank window (slot = (r & 1) * 8 + position). The DMA applies the permutation to
// its SOURCE chunk (the LDS image of a DMA is lane-linear), a row's eight lanes still fetch one whole 128-byte line.
// W rows are staged permuted (row rho of a 32-row MFMA tile i of a 64-feature pair holds feature 32 ((rho >> 2) & 1) + 16 (i & 1)
// + 4 (rho >> 3) + (rho & 3)), so that lane (h = l >> 5, col = l & 31) ends up with the 32 CONSECUTIVE features 32 h .. 32 h + 31
// of the pair for token col: the epilogue stays lane-local (bias, activation, SwiGLU pairs, the rotary partner in lane ^ 32).
// K-tile g lives in stage g & 1; K % 128 == 0 keeps that parity across output tiles, so the streams of a tile are unrolled by
// two with the stage as a compile-time choice of operands; the DMA cursor runs one (X) / two (W) K-tiles ahead and crosses into the
// NEXT output tile in the last two instances (persistent kernel: the first K-tile of the next tile lands under the epilogue).
#ifndef BH_GEMM_KTILE_INC
#define BH_GEMM_KTILE_INC "gemm_ktile_mfma.inc"      // (tools/gemm_lab.hip builds the kernel around other variants of the stream)
#endif
#include BH_GEMM_KTILE_INC
// The same K-tile on v_mfma_f32_16x16x32_f16 (round 6, gemm_w4_kernel<..., T16 = true>; tools/gen_gemmstep.py --tile16). Under the board's
// power cap the small tile is the cheaper instruction: a hand-scheduled 128 x 128 wave tile fed from LDS runs 1775 TFLOP/s on it against 1556
// on 32x32x16 (tools/gen_tile_probe.py, profiles/r06_mfma_tile_energy_lds.txt). What changes around the stream: the accumulators are 64
// float4 (acc[i][j]: token tile i, feature tile j; lane (g = l >> 4, col = l & 15) holds features 16 j + 4 g .. + 3 of token 16 i + col), the W
// rows are staged in NATURAL order (a lane's four accumulator registers are four consecutive features by themselves), fragment reads take
// rows l & 15 of a 16-row tile (immediate offset 2048 per tile) and chunk 4 s + (l >> 4) of slab s under the same XOR swizzle (every 16-lane
// group of the ds_read_b128 still touches sixteen distinct slots), and the epilogue writes its transposition scratch from that layout - the
// scratch image, and everything behind it, is the same.
#ifndef BH_GEMM_KTILE16_INC
#define BH_GEMM_KTILE16_INC "gemm_ktile16_mfma.inc"
#endif
#include BH_GEMM_KTILE16_INC

constexpr int W4_OP = 32768, W4_BRING = 3 * W4_OP, W4_LDS = 5 * W4_OP;     // A ring: three 32 KiB stages, B ring: two

__device__ __forceinline__ void w4_dma(unsigned m0v, unsigned voff, const char* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory");
}

// per-lane byte offsets of this wave's eight DMA pieces of each operand for the output tile (f0, t0): va = X (tokens, MFMA A), vb = W (features, B)
template <bool T16 = false>
__device__ __forceinline__ void w4_offsets(const GemmArgs& p, int f0, int t0, int wave, int lane, unsigned (&va)[8], unsigned (&vb)[8]) {
    const int lr = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int R = wave * 64 + n * 8 + lr;                 // LDS row of this lane's chunk
        const int chunk = slot ^ ((R >> 1) & 7);
        // token ring: row R = token t0 + R. Feature ring: row R = (wave half R >> 7, MFMA tile j = (R >> 5) & 3, tile row rho = R & 31) holds
        // feature 128 (R >> 7) + 64 (j >> 1) + 32 ((rho >> 2) & 1) + 16 (j & 1) + 4 (rho >> 3) + (rho & 3): accumulator register r of lane
        // half h is tile row (r & 3) + 8 (r >> 2) + 4 h, so the lane holds the 32 CONSECUTIVE features 32 h + 16 (j & 1) + r of the pair
        const int rho = R & 31, j = (R >> 5) & 3;
        // (T16: natural order - see the note at the include of the 16x16x32 stream)
        const int feat = T16 ? f0 + R : f0 + (R >> 7) * 128 + (j >> 1) * 64 + 32 * ((rho >> 2) & 1) + 16 * (j & 1) + 4 * (rho >> 3) + (rho & 3);
        va[n] = (unsigned)min(t0 + R, p.M - 1) * (unsigned)(p.ldx * 2) + chunk * 16;
        vb[n] = (unsigned)min(feat, p.N - 1) * (unsigned)(p.ldw * 2) + chunk * 16;
    }
}

// Epilogue of one wave. acc[i][j][r] of lane (h = l >> 5, col = l & 31): token tw + 32 i + col, feature fw + 64 (j >> 1) + 32 h + 16 (j & 1)
// + r - a lane holds 32 consecutive features of ONE token per pair of feature tiles, i.e. the tokens run along the lanes, and a store
// straight from this layout would touch 64 different rows per instruction. What the CU's store path costs is instructions, not bytes
// (measured: ~48 cycles per 64-lane dwordx2 store and ~74 per dwordx4 store whatever they write - 128 KiB per output tile took 12.4 k
// cycles as 256 dwordx2 stores of whole 256-byte row slices, 9.5 k as 128 dwordx4 stores of whole 128-byte lines; the chip-wide HBM
// rate is nowhere near its limit), so the tile leaves as the FEWEST, WIDEST stores: dwordx4, eight lanes per 128-byte line.
// That needs a transposition, through LDS: a block of 32 tokens x 64 features goes out as fp32 straight from the accumulation registers
// (ds_write_b128 with an AGPR data operand, no v_accvgpr_read) and comes back as lane (tl = l >> 3, q = l & 7) <- features 8 q .. 8 q + 7
// of token tl + 8 rr, rr < 4. Everything else happens in THAT layout: bias (8 values per lane and feature pair), residual (one
// coalesced 16-byte buffer load per row), rotary (the partner dims +-32 of a head of 64 sit in lane ^ 4), activation, scale / clamp,
// SwiGLU (four (y, gate) pairs per lane), ONE rounding to fp16. Stores and residual loads are buffer operations (tile base in a scalar
// resource, the lane's column in a loop-invariant voffset, the row in a scalar soffset: no vector address arithmetic); rows that are
// not stored (ragged last token tile, dropped batch-padding rows) are predicated in a second instance of the code that only such tiles run. The row map must be
// affine over a tile: identity, or a remap whose group size is a multiple of 256 (the launcher sends nothing else to this kernel).
// LDS: the block is 8 KiB = this wave's slice of the A-ring stage the last K-tile just vacated (`scratch`; the next instance's DMA into
// that slice is issued by this wave, behind its epilogue). Rows are 256 B = sixteen 16-byte pieces, piece k of token t at position
// k ^ (t & 15): writes (8-lane groups = 8 tokens x one piece) and reads (16-lane groups = 4 tokens x 4 lanes) are conflict free. Block
// b + 1 is written while block b's reads are in flight (the LDS operations of one wave execute in issue order).
// DEFERRED: the 32 finished rows of a full tile are not stored here but PARKED (park[4 b + rr], fp16, store layout) together with
// the tile's store parameters (W4Store); the K-tile instances of the NEXT output tile issue them a few at a time behind their
// barriers (gemm_ktile_st*: four per K-tile), the last tile's rows go out through w4_store_parked. The CU's store
// path needs ~74 cycles per dwordx4 store - 9.5 k cycles per output tile, a third of a K = 512 tile when the epilogue waits for
// it - and runs beside the matrix pipe for free when the stores are spread over the next K loop. Ragged tiles (returns false)
// are stored here, predicated.
struct W4Store {
    uint4_t srd;        // buffer resource of the tile's output rows (base = first row of the wave, first feature of the wave)
    unsigned voff;      // this lane's byte offset inside a block of 8 rows: (l >> 3) rows + (l & 7) pieces
    unsigned rowb;      // bytes between consecutive tokens
};

// MODE (compile time, so that the block loop is straight-line code the compiler can software-pipeline; with run-time flags every
// `if` inside it was a join with full s_waitcnt's: 9.5 k cycles per tile whatever the stores did): 1 = residual, 2 = rotary, 4 = scale / clamp
// ... and a token tile's whole row of eight quads. Why statements, and why several quads per statement: (i) as plain copies the reads are common
// to the full-tile and the ragged-tile instance of an epilogue, and the compiler hoists ALL of a tile's reads in front of that branch (256 live
// registers, every parked row spilled); (ii) the statements are scheduling boundaries, and an epilogue that reads quad by quad runs its
// ~20-instruction dependent chain (exp -> add -> rcp -> mul -> mul -> cvt) one quad at a time, at the vector ALU's latency (SwiGLU: 9.3 k
// cycles per tile that way, 7.6 k with a row per statement and the math written stage by stage)
__device__ __forceinline__ void w4_acc_read8(const float4_t (&a)[8], float4_t (&x)[8]) {
    float f[32];
    asm volatile("v_accvgpr_read_b32 %0, %32\n\tv_accvgpr_read_b32 %1, %33\n\tv_accvgpr_read_b32 %2, %34\n\tv_accvgpr_read_b32 %3, %35\n\t"
                 "v_accvgpr_read_b32 %4, %36\n\tv_accvgpr_read_b32 %5, %37\n\tv_accvgpr_read_b32 %6, %38\n\tv_accvgpr_read_b32 %7, %39\n\t"
                 "v_accvgpr_read_b32 %8, %40\n\tv_accvgpr_read_b32 %9, %41\n\tv_accvgpr_read_b32 %10, %42\n\tv_accvgpr_read_b32 %11, %43\n\t"
                 "v_accvgpr_read_b32 %12, %44\n\tv_accvgpr_read_b32 %13, %45\n\tv_accvgpr_read_b32 %14, %46\n\tv_accvgpr_read_b32 %15, %47\n\t"
                 "v_accvgpr_read_b32 %16, %48\n\tv_accvgpr_read_b32 %17, %49\n\tv_accvgpr_read_b32 %18, %50\n\tv_accvgpr_read_b32 %19, %51\n\t"
                 "v_accvgpr_read_b32 %20, %52\n\tv_accvgpr_read_b32 %21, %53\n\tv_accvgpr_read_b32 %22, %54\n\tv_accvgpr_read_b32 %23, %55\n\t"
                 "v_accvgpr_read_b32 %24, %56\n\tv_accvgpr_read_b32 %25, %57\n\tv_accvgpr_read_b32 %26, %58\n\tv_accvgpr_read_b32 %27, %59\n\t"
                 "v_accvgpr_read_b32 %28, %60\n\tv_accvgpr_read_b32 %29, %61\n\tv_accvgpr_read_b32 %30, %62\n\tv_accvgpr_read_b32 %31, %63"
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]),
                   "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15]), "=v"(f[16]), "=v"(f[17]), "=v"(f[18]), "=v"(f[19]), "=v"(f[20]),
                   "=v"(f[21]), "=v"(f[22]), "=v"(f[23]), "=v"(f[24]), "=v"(f[25]), "=v"(f[26]), "=v"(f[27]), "=v"(f[28]), "=v"(f[29]), "=v"(f[30]), "=v"(f[31])
                 : "a"(a[0][0]), "a"(a[0][1]), "a"(a[0][2]), "a"(a[0][3]), "a"(a[1][0]), "a"(a[1][1]), "a"(a[1][2]), "a"(a[1][3]), "a"(a[2][0]), "a"(a[2][1]),
                   "a"(a[2][2]), "a"(a[2][3]), "a"(a[3][0]), "a"(a[3][1]), "a"(a[3][2]), "a"(a[3][3]), "a"(a[4][0]), "a"(a[4][1]), "a"(a[4][2]), "a"(a[4][3]),
                   "a"(a[5][0]), "a"(a[5][1]), "a"(a[5][2]), "a"(a[5][3]), "a"(a[6][0]), "a"(a[6][1]), "a"(a[6][2]), "a"(a[6][3]), "a"(a[7][0]), "a"(a[7][1]),
                   "a"(a[7][2]), "a"(a[7][3]));
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = float4_t{f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3]};
}
// accumulation registers -> vector registers: four quads a[j0 .. j0 + 3] in one volatile statement ...
__device__ __forceinline__ void w4_acc_read4(const float4_t (&a)[8], int j0, float4_t (&x)[4]) {
    float f[16];
    asm volatile("v_accvgpr_read_b32 %0, %16\n\tv_accvgpr_read_b32 %1, %17\n\tv_accvgpr_read_b32 %2, %18\n\tv_accvgpr_read_b32 %3, %19\n\t"
                 "v_accvgpr_read_b32 %4, %20\n\tv_accvgpr_read_b32 %5, %21\n\tv_accvgpr_read_b32 %6, %22\n\tv_accvgpr_read_b32 %7, %23\n\t"
                 "v_accvgpr_read_b32 %8, %24\n\tv_accvgpr_read_b32 %9, %25\n\tv_accvgpr_read_b32 %10, %26\n\tv_accvgpr_read_b32 %11, %27\n\t"
                 "v_accvgpr_read_b32 %12, %28\n\tv_accvgpr_read_b32 %13, %29\n\tv_accvgpr_read_b32 %14, %30\n\tv_accvgpr_read_b32 %15, %31"
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7]), "=v"(f[8]), "=v"(f[9]), "=v"(f[10]),
                   "=v"(f[11]), "=v"(f[12]), "=v"(f[13]), "=v"(f[14]), "=v"(f[15])
                 : "a"(a[j0][0]), "a"(a[j0][1]), "a"(a[j0][2]), "a"(a[j0][3]), "a"(a[j0 + 1][0]), "a"(a[j0 + 1][1]), "a"(a[j0 + 1][2]), "a"(a[j0 + 1][3]),
                   "a"(a[j0 + 2][0]), "a"(a[j0 + 2][1]), "a"(a[j0 + 2][2]), "a"(a[j0 + 2][3]), "a"(a[j0 + 3][0]), "a"(a[j0 + 3][1]), "a"(a[j0 + 3][2]),
                   "a"(a[j0 + 3][3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = float4_t{f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3]};
}

// (The 16x16x32 stream uses this function for the fused residual only - MODE 1, ACC = float4_t[8][8]: its accumulators go straight from the
// AGPRs into the same scratch image; every other epilogue of that stream works in the accumulator layout: w4_epilogue16p / w4_epilogue16g.)
template <int ACT, bool GATED, int MODE, typename ACC, typename PARK, int NP>
__device__ __forceinline__ bool w4_epilogue(const GemmArgs& p, ACC& acc, int f0, int t0, int wa, int wb, int lane, char* scratch,
                                            PARK (&park)[NP], W4Store& st) {
    constexpr bool T16 = std::is_same_v<ACC, float4_t[8][8]>;       // the accumulator layout of the 16x16x32 stream
    constexpr int NOW = 32 - NP;                                    // rows stored here; the other NP are parked (W4_NPARK / W4_NPARK16)
    static_assert(NP == (T16 ? W4_NPARK16 : W4_NPARK) && NOW % 4 == 0, "parked rows: the split the K-tile streams were generated for");
    asm volatile("" : "+v"(lane));      // opaque: the lane constants below are recomputed per tile instead of living in registers through the K loop
    const int h = lane >> 5, col = lane & 31;
    const int tl = lane >> 3, q = lane & 7;
    const int fw = f0 + wb * 128, tw = t0 + wa * 128;
    const bool ident = p.row_div == 1 && p.row_s_hi == 1;
    constexpr bool plain = (MODE & 4) == 0;
    const int hi_u = ident ? 0 : t0 / p.row_div;
    const int mlim = ident ? p.M : min(p.M, hi_u * p.row_div + p.row_lim);       // tokens >= mlim of this tile are not stored
    const long o0 = ident ? (long)tw : (long)hi_u * p.row_s_hi + (long)(tw - hi_u * p.row_div) * p.row_s_lo;
    const int rowbytes = (ident ? 1 : (int)p.row_s_lo) * p.ldo * 2;              // bytes between consecutive tokens (< 16 MiB)
    half_t* const obase = p.out + o0 * p.ldo + (GATED ? fw >> 1 : fw);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, 0x7ffffff0, 0x00020000);
    const int ovoff = tl * rowbytes + (GATED ? 8 * q : 16 * q);
    {
        const unsigned long long ob = (unsigned long long)obase;
        st.srd = uint4_t{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ob),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(ob >> 32) & 0xffffu)), 0x7ffffff0u, 0x00020000u};
        st.voff = (unsigned)ovoff;
        st.rowb = (unsigned)__builtin_amdgcn_readfirstlane(rowbytes);
    }
    constexpr bool has_res = (MODE & 1) != 0;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_res ? p.res + (long)tw * p.ldres + fw : p.out), 0, 0x7ffffff0, 0x00020000);
    const int resbytes = p.ldres * 2;
    const int rvoff = tl * resbytes + 16 * q;
    float bq[2][8];
#pragma unroll
    for (int P = 0; P < 2; ++P) {
        float4_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (p.bias != nullptr) { b0 = *(const float4_t*)(p.bias + fw + P * 64 + 8 * q); b1 = *(const float4_t*)(p.bias + fw + P * 64 + 8 * q + 4); }
#pragma unroll
        for (int e = 0; e < 4; ++e) { bq[P][e] = b0[e]; bq[P][4 + e] = b1[e]; }
    }
    constexpr bool rot = !GATED && (MODE & 2) != 0;
    const float rsgn = q < 4 ? -1.0f : 1.0f;
    const int pos0 = rot ? (tw + tl) % p.rot_T : 0;                              // position of the lane's first row (rot_T >= 256: one wrap at most)
    const float* const cs0 = p.rot_cs + (q & 3) * 16;
    char* const wrow = scratch + col * 256;
    const int wx = col & 15;
    auto write_block = [&](int b) {           // block b = (32 tokens i = b >> 1, feature pair P = b & 1) into the scratch
        if constexpr (T16) {
            // lane (g = l >> 4, c16 = l & 15): piece 4 jj + g (features 16 jj + 4 g .. + 3 of the pair) of token 16 tt + c16; an 8-lane group of
            // the ds_write_b128 is eight tokens x one piece = eight positions piece ^ token: conflict free like the 32x32 layout's
            const int g = lane >> 4, c16 = lane & 15;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    *(float4_t*)(scratch + (16 * tt + c16) * 256 + (((4 * jj + g) ^ c16) << 4)) = acc[2 * (b >> 1) + tt][4 * (b & 1) + jj];
        } else {
            // the lane's 32 features of one token: tiles j = 2 P, 2 P + 1
            const float16_t& a0 = acc[b >> 1][2 * (b & 1)];
            const float16_t& a1 = acc[b >> 1][2 * (b & 1) + 1];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4_t v0 = {a0[4 * c], a0[4 * c + 1], a0[4 * c + 2], a0[4 * c + 3]};
                const float4_t v1 = {a1[4 * c], a1[4 * c + 1], a1[4 * c + 2], a1[4 * c + 3]};
                *(float4_t*)(wrow + (((8 * h + c) ^ wx) << 4)) = v0;
                *(float4_t*)(wrow + (((8 * h + 4 + c) ^ wx) << 4)) = v1;
            }
        }
    };
    auto blocks = [&](auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        write_block(0);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int i = b >> 1, P = b & 1;                // token tile, feature pair
            float4_t lo[4], hi[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int tr = tl + 8 * rr;
                lo[rr] = *(const float4_t*)(scratch + tr * 256 + (((2 * q) ^ (tr & 15)) << 4));
                hi[rr] = *(const float4_t*)(scratch + tr * 256 + (((2 * q + 1) ^ (tr & 15)) << 4));
            }
            uint4_t rres[4];
            if constexpr (has_res) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    rres[rr] = uint4_t{0u, 0u, 0u, 0u};
                    if (!MASKED || tw + 32 * i + 8 * rr + tl < mlim)
                        rres[rr] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, rvoff + P * 128, (32 * i + 8 * rr) * resbytes, 0);
                }
            }
            if (b + 1 < 8) write_block(b + 1);
            const bool rot_here = rot && fw + P * 64 < p.rot_nfeat;
            const float rqs = fw + P * 64 < p.rot_qfeat ? p.rot_qscale : 1.0f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[rr][e] + bq[P][e]; v[4 + e] = hi[rr][e] + bq[P][4 + e]; }
                if constexpr (has_res) {
                    const half8_t r8 = __builtin_bit_cast(half8_t, rres[rr]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += p.res_scale * (float)r8[e];
                }
                if constexpr (rot) {
                    if (rot_here) {
                        // rotary embedding of the packed Wqkv projection (see gemm_epilogue): the 64-feature pair is one head, lanes q < 4
                        // hold its first half, q >= 4 the second; the partner of dim d is dim d +- 32 = lane ^ 4, same index
                        int pos = pos0 + 32 * i + 8 * rr;
                        pos = pos >= p.rot_T ? pos - p.rot_T : pos;
                        const float* cs = cs0 + (long)pos * 64;
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const float4_t c4 = *(const float4_t*)(cs + 2 * e);      // (cos, sin) of dims e, e + 1 of this lane's eight
                            const float pa = __shfl_xor(v[e], 4), pb = __shfl_xor(v[e + 1], 4);
                            v[e] = (v[e] * c4[0] + rsgn * pa * c4[1]) * rqs;
                            v[e + 1] = (v[e + 1] * c4[2] + rsgn * pb * c4[3]) * rqs;
                        }
                    }
                }
                const bool live = !MASKED || tw + 32 * i + 8 * rr + tl < mlim;
                // (the row goes into the VECTOR offset: a buffer_store_dwordx4 with a scalar offset reads its data registers late, and
                // the two wait states hipcc leaves before the next row's arithmetic overwrites them were not enough on gfx950 - dword 1
                // of lanes 12-15 of every 16 came out of the NEXT row, run-to-run different; stores without a scalar offset never did)
                const int vo = ovoff;
                if constexpr (GATED) {
                    // W rows interleaved on the host: feature 2 k = y_k, 2 k + 1 = gate_k: four outputs per lane, 8 bytes
                    float4_t y;
#pragma unroll
                    for (int k = 0; k < 4; ++k) y[k] = v[2 * k] * swishf_(v[2 * k + 1]);
                    const uint2_t o2 = __builtin_bit_cast(uint2_t, __builtin_convertvector(y, half4_t));
                    if constexpr (!MASKED) { if (4 * b + rr >= NOW) park[4 * b + rr - NOW] = o2; }
                    if ((!MASKED && 4 * b + rr < NOW) || (MASKED && live)) __builtin_amdgcn_raw_buffer_store_b64(o2, orsrc, vo + P * 64 + (32 * i + 8 * rr) * rowbytes, 0, 0);
                } else {
                    float8_t f8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) f8[e] = apply_act<ACT>(v[e]);
                    if constexpr (!plain) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f8[e] = fminf(fmaxf(f8[e] * p.scale, p.clamp_lo), p.clamp_hi);
                    }
                    const uint4_t o4 = __builtin_bit_cast(uint4_t, __builtin_convertvector(f8, half8_t));
                    if constexpr (!MASKED) { if (4 * b + rr >= NOW) park[4 * b + rr - NOW] = o4; }
                    if ((!MASKED && 4 * b + rr < NOW) || (MASKED && live)) __builtin_amdgcn_raw_buffer_store_b128(o4, orsrc, vo + P * 128 + (32 * i + 8 * rr) * rowbytes, 0, 0);
                }
            }
        }
    };
    if (tw + 128 <= mlim) { blocks(std::false_type{}); return true; }
    blocks(std::true_type{});
    return false;
}

// SwiGLU epilogue of one wave around the 16x16x32 stream (round 6). The accumulator layout holds a lane's FOUR consecutive features of a token: the
// (y, gate) pairs are lane-local, so the activation runs BEFORE the transposition and what crosses the LDS is the final fp16 - 2 halves per
// accumulator quad, 16 KiB per wave tile each way where w4_epilogue moves 64 KiB of fp32 each way - and a token's 64 outputs are exactly one
// 128-byte line: 16 stores of 8 rows x 128 bytes per lane and tile instead of 32 stores of 8 rows x 64 bytes.
// Block blk = 32 tokens (token tiles 2 blk, 2 blk + 1) x the wave's 64 outputs = 4 KiB of the scratch (double buffered: 8 KiB). Lane (g = l >> 4,
// c = l & 15) writes outputs 8 j + 2 g, + 1 of token r = 16 tt + c as one dword at r * 128 + 16 (j ^ ((r >> 1) & 7)) + 4 g - the sixteen tokens
// x four g of a ds_write_b32 fall on 64 different banks - and lane (tl = l >> 3, q = l & 7) reads the 16 bytes of piece q of row tl + 8 rr.
// Unit u = 4 blk + rr (token row 32 blk + 8 rr of the wave): u < W4_NOWG is stored here, the rest parked (gemm_ktile16_st4g).
template <typename PARK>
__device__ __forceinline__ bool w4_epilogue16g(const GemmArgs& p, float4_t (&acc)[8][8], int f0, int t0, int wa, int wb, int lane, char* scratch,
                                               PARK (&park)[W4_NPARKG], W4Store& st) {
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, c16 = lane & 15;
    const int tl = lane >> 3, q = lane & 7;
    const int fw = f0 + wb * 128, tw = t0 + wa * 128;
    const bool ident = p.row_div == 1 && p.row_s_hi == 1;
    const int hi_u = ident ? 0 : t0 / p.row_div;
    const int mlim = ident ? p.M : min(p.M, hi_u * p.row_div + p.row_lim);
    const long o0 = ident ? (long)tw : (long)hi_u * p.row_s_hi + (long)(tw - hi_u * p.row_div) * p.row_s_lo;
    const int rowbytes = (ident ? 1 : (int)p.row_s_lo) * p.ldo * 2;
    half_t* const obase = p.out + o0 * p.ldo + (fw >> 1);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, 0x7ffffff0, 0x00020000);
    const int ovoff = tl * rowbytes + 16 * q;
    {
        const unsigned long long ob = (unsigned long long)obase;
        st.srd = uint4_t{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ob),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(ob >> 32) & 0xffffu)), 0x7ffffff0u, 0x00020000u};
        st.voff = (unsigned)ovoff;
        st.rowb = (unsigned)__builtin_amdgcn_readfirstlane(rowbytes);
    }
    float4_t bj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bj[j] = float4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) bj[j] = *(const float4_t*)(p.bias + fw + 16 * j + 4 * g);
    }
    auto write_block = [&](int blk, auto with_bias) {
        char* const base = scratch + (blk & 1) * 4096;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int r = 16 * tt + c16, sw = (c16 >> 1) & 7;
            float4_t row[8];
            w4_acc_read8(acc[2 * blk + tt], row);
            // W rows interleaved on the host: feature 2 k = y_k, 2 k + 1 = gate_k. Stage by stage over the row's sixteen outputs (independent
            // instructions back to back: the chain's latency is paid once per stage, not once per output)
            float yv[16], gv[16], ev[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float4_t a = row[j];
                if constexpr (decltype(with_bias)::value) a += bj[j];
                yv[2 * j] = a[0]; gv[2 * j] = a[1]; yv[2 * j + 1] = a[2]; gv[2 * j + 1] = a[3];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) ev[k] = gv[k] * -1.4426950408889634f;
#pragma unroll
            for (int k = 0; k < 16; ++k) ev[k] = __builtin_amdgcn_exp2f(ev[k]);
#pragma unroll
            for (int k = 0; k < 16; ++k) ev[k] = 1.0f + ev[k];
#pragma unroll
            for (int k = 0; k < 16; ++k) ev[k] = __builtin_amdgcn_rcpf(ev[k]);
#pragma unroll
            for (int k = 0; k < 16; ++k) yv[k] = yv[k] * (gv[k] * ev[k]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                typedef float float2_t __attribute__((ext_vector_type(2)));
                const float2_t y = {yv[2 * j], yv[2 * j + 1]};
                *(unsigned*)(base + r * 128 + ((j ^ sw) << 4) + 4 * g) = __builtin_bit_cast(unsigned, __builtin_convertvector(y, half2_t));
            }
        }
    };
    // (the epilogue is bound by its vector instructions - ~25 per accumulator quad, one wave per SIMD; bonito's GatedMlp has no bias, and its 256
    // additions of zero per lane and tile were 16 % of them: the full-tile instance exists with and without, the ragged one with)
    auto blocks = [&](auto masked, auto with_bias) {
        constexpr bool MASKED = decltype(masked)::value;
        write_block(0, with_bias);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            uint4_t o[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = tl + 8 * rr;
                o[rr] = *(const uint4_t*)(scratch + (blk & 1) * 4096 + row * 128 + ((q ^ ((row >> 1) & 7)) << 4));
            }
            if (blk + 1 < 4) write_block(blk + 1, with_bias);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int u = 4 * blk + rr;
                const bool live = !MASKED || tw + 32 * blk + 8 * rr + tl < mlim;
                if constexpr (!MASKED) { if (u >= W4_NOWG) park[u - W4_NOWG] = o[rr]; }
                if ((!MASKED && u < W4_NOWG) || (MASKED && live)) __builtin_amdgcn_raw_buffer_store_b128(o[rr], orsrc, ovoff + (32 * blk + 8 * rr) * rowbytes, 0, 0);
            }
        }
    };
    if (tw + 128 <= mlim) {
        if (p.bias != nullptr) blocks(std::false_type{}, std::true_type{});
        else blocks(std::false_type{}, std::false_type{});
        return true;
    }
    blocks(std::true_type{}, std::true_type{});
    return false;
}

// The same idea for the other epilogues of the 16x16x32 stream (ACT x MODE 0 / 2 / 4; the fused residual, MODE 1, stays on w4_epilogue): bias,
// rotary (lane-local partner, see w4_epilogue's note), activation, scale / clamp and the ONE rounding to fp16 happen in the accumulator
// layout, stage by stage over the eight quads of a block; 8 bytes per quad cross the LDS (32 KiB per wave tile each way instead of 64 KiB),
// and the units are w4_epilogue's (block b = 32 tokens x feature pair P, rr: row 32 i + 8 rr, 128 bytes at byte column 128 P), so the
// parked rows and gemm_ktile16_st4w do not change. Block b = 4 KiB of the scratch, double buffered. Lane (g, c) writes the four halves of
// quad (tt, jj) at r * 128 + 16 ((2 jj + (g >> 1)) ^ ((r >> 1) & 7)) + 8 (g & 1), r = 16 tt + c: the 64 lanes of a ds_write_b64 fall on 64
// different 8-byte slots.
template <int ACT, int MODE, typename PARK>
__device__ __forceinline__ bool w4_epilogue16p(const GemmArgs& p, float4_t (&acc)[8][8], int f0, int t0, int wa, int wb, int lane, char* scratch,
                                               PARK (&park)[W4_NPARK16], W4Store& st) {
    static_assert((MODE & 1) == 0, "the fused residual is read in the store layout: w4_epilogue");
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, c16 = lane & 15;
    const int tl = lane >> 3, q = lane & 7;
    const int fw = f0 + wb * 128, tw = t0 + wa * 128;
    const bool ident = p.row_div == 1 && p.row_s_hi == 1;
    constexpr bool plain = (MODE & 4) == 0;
    constexpr bool rot = (MODE & 2) != 0;
    const int hi_u = ident ? 0 : t0 / p.row_div;
    const int mlim = ident ? p.M : min(p.M, hi_u * p.row_div + p.row_lim);
    const long o0 = ident ? (long)tw : (long)hi_u * p.row_s_hi + (long)(tw - hi_u * p.row_div) * p.row_s_lo;
    const int rowbytes = (ident ? 1 : (int)p.row_s_lo) * p.ldo * 2;
    half_t* const obase = p.out + o0 * p.ldo + fw;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, 0x7ffffff0, 0x00020000);
    const int ovoff = tl * rowbytes + 16 * q;
    {
        const unsigned long long ob = (unsigned long long)obase;
        st.srd = uint4_t{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ob),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(ob >> 32) & 0xffffu)), 0x7ffffff0u, 0x00020000u};
        st.voff = (unsigned)ovoff;
        st.rowb = (unsigned)__builtin_amdgcn_readfirstlane(rowbytes);
    }
    float4_t bj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bj[j] = float4_t{0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) bj[j] = *(const float4_t*)(p.bias + fw + 16 * j + 4 * g);
    }
    const int pos16 = rot ? (tw + c16) % p.rot_T : 0;          // position of the lane's token in token tile 0 (rot_T >= 256: one wrap at most)
    auto write_block = [&](int b, auto with_bias) {
        const int P = b & 1;
        char* const base = scratch + (b & 1) * 4096;
        float4_t x[2][4];
        w4_acc_read4(acc[2 * (b >> 1)], 4 * P, x[0]);
        w4_acc_read4(acc[2 * (b >> 1) + 1], 4 * P, x[1]);
        if constexpr (decltype(with_bias)::value) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) x[tt][jj] += bj[4 * P + jj];
        }
        if constexpr (rot) {
            if (fw + P * 64 < p.rot_nfeat) {             // (wave uniform) this 64-feature pair is one rotated head: tiles jj = t (dims < 32) and t + 2
                const float rqs = fw + P * 64 < p.rot_qfeat ? p.rot_qscale : 1.0f;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    int pos = pos16 + 16 * (2 * (b >> 1) + tt);
                    pos = pos >= p.rot_T ? pos - p.rot_T : pos;
                    const float* cs = p.rot_cs + (long)pos * 64 + 8 * g;          // (cos, sin) of dims 4 g .. 4 g + 3; + 32 floats: dims 16 + 4 g ..
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float4_t ca = *(const float4_t*)(cs + 32 * t), cb = *(const float4_t*)(cs + 32 * t + 4);
                        const float cc[4] = {ca[0] * rqs, ca[2] * rqs, cb[0] * rqs, cb[2] * rqs}, ss[4] = {ca[1] * rqs, ca[3] * rqs, cb[1] * rqs, cb[3] * rqs};
                        const float4_t lo = x[tt][t], hi = x[tt][t + 2];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            x[tt][t][r] = lo[r] * cc[r] - hi[r] * ss[r];
                            x[tt][t + 2][r] = hi[r] * cc[r] + lo[r] * ss[r];
                        }
                    }
                }
            }
        }
        if constexpr (ACT != ACT_NONE) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[tt][jj][r] = apply_act<ACT>(x[tt][jj][r]);
        }
        if constexpr (!plain) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[tt][jj][r] = fminf(fmaxf(x[tt][jj][r] * p.scale, p.clamp_lo), p.clamp_hi);
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int r = 16 * tt + c16, sw = (c16 >> 1) & 7;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                *(uint2_t*)(base + r * 128 + (((2 * jj + (g >> 1)) ^ sw) << 4) + 8 * (g & 1)) =
                    __builtin_bit_cast(uint2_t, __builtin_convertvector(x[tt][jj], half4_t));
        }
    };
    auto blocks = [&](auto masked, auto with_bias) {
        constexpr bool MASKED = decltype(masked)::value;
        write_block(0, with_bias);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int i = b >> 1, P = b & 1;
            uint4_t o[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = tl + 8 * rr;
                o[rr] = *(const uint4_t*)(scratch + (b & 1) * 4096 + row * 128 + ((q ^ ((row >> 1) & 7)) << 4));
            }
            if (b + 1 < 8) write_block(b + 1, with_bias);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int u = 4 * b + rr;
                const bool live = !MASKED || tw + 32 * i + 8 * rr + tl < mlim;
                if constexpr (!MASKED) { if (u >= W4_NOW16) park[u - W4_NOW16] = o[rr]; }
                if ((!MASKED && u < W4_NOW16) || (MASKED && live))
                    __builtin_amdgcn_raw_buffer_store_b128(o[rr], orsrc, ovoff + P * 128 + (32 * i + 8 * rr) * rowbytes, 0, 0);
            }
        }
    };
    if (tw + 128 <= mlim) {
        if (p.bias != nullptr) blocks(std::false_type{}, std::true_type{});
        else blocks(std::false_type{}, std::false_type{});
        return true;
    }
    blocks(std::true_type{}, std::true_type{});
    return false;
}

// the parked rows of a tile that no K loop follows
template <bool GATED, bool T16, typename PARK, int NP>
__device__ __forceinline__ void w4_store_parked(const PARK (&park)[NP], const W4Store& st) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((unsigned long long)st.srd[1] << 32) | st.srd[0]), 0, 0x7ffffff0, 0x00020000);
    if constexpr (GATED && T16) {               // w4_epilogue16g: whole 128-byte rows
#pragma unroll
        for (int idx = 0; idx < NP; ++idx) __builtin_amdgcn_raw_buffer_store_b128(park[idx], rsrc, st.voff + w4_store_rowg(idx) * st.rowb, 0, 0);
        return;
    } else {
#pragma unroll
    for (int idx = 0; idx < NP; ++idx) {
        const int row = T16 ? w4_store_row16(idx) : w4_store_row(idx), col = T16 ? w4_store_col16(idx) : w4_store_col(idx);
        if constexpr (GATED) __builtin_amdgcn_raw_buffer_store_b64(park[idx], rsrc, st.voff + col * 64 + row * st.rowb, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(park[idx], rsrc, st.voff + col * 128 + row * st.rowb, 0, 0);
    }
    }
}

// one K-tile instance that also issues S parked rows (IDX0 .. IDX0 + S - 1, modulo W4_NPARK) of the previous tile
template <int IDX0, int S, bool WIDE, bool FIRST, typename PARK>
__device__ __forceinline__ void w4_inst_st(float16_t (&acc)[4][4], half8_t (&fa)[2][4], half8_t (&fb)[2][4], const unsigned (&rab)[4],
                                           const unsigned (&rbb)[4], unsigned sa, unsigned sb, unsigned san, unsigned sbn, const unsigned (&vd1)[8],
                                           const unsigned (&vd2)[8], const char* sd1, const char* sd2, unsigned md1, unsigned md2, unsigned,
                                           const PARK (&park)[W4_NPARK], const W4Store& st, std::false_type) {
    if constexpr (S == 4 && WIDE && !FIRST) gemm_ktile_st4w<IDX0>(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, park, st.srd, st.voff, st.rowb);
    else if constexpr (S == 4 && WIDE && FIRST) gemm_ktile_first_st4w<IDX0>(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, park, st.srd, st.voff, st.rowb);
    else if constexpr (S == 4 && !WIDE && !FIRST) gemm_ktile_st4n<IDX0>(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, park, st.srd, st.voff, st.rowb);
    else if constexpr (S == 4 && !WIDE && FIRST) gemm_ktile_first_st4n<IDX0>(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, park, st.srd, st.voff, st.rowb);
    else static_assert(S == 4, "tools/gen_gemmstep.py emits the four-stores-per-instance variants only");
}

template <int IDX0, int S, bool WIDE, bool FIRST>
__device__ __forceinline__ void w4_inst_st(float4_t (&acc)[8][8], half8_t (&fa)[8], half8_t (&fb)[2][8], const unsigned (&rab)[2],
                                           const unsigned (&rbb)[2], unsigned sa, unsigned sb, unsigned san, unsigned sbn, const unsigned (&vd1)[8],
                                           const unsigned (&vd2)[8], const char* sd1, const char* sd2, unsigned md1, unsigned md2, unsigned wv,
                                           const uint4_t (&park)[W4_NPARKG], const W4Store& st, std::true_type /* SwiGLU rows: w4_epilogue16g */) {
    static_assert(S == 4, "tools/gen_gemmstep.py emits the four-stores-per-instance variants only");
    if constexpr (!FIRST) gemm_ktile16_st4g<IDX0>(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, wv, park, st.srd, st.voff, st.rowb);
    else gemm_ktile16_first_st4g<IDX0>(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, wv, park, st.srd, st.voff, st.rowb);
}
template <int IDX0, int S, bool WIDE, bool FIRST, typename PARK>
__device__ __forceinline__ void w4_inst_st(float4_t (&acc)[8][8], half8_t (&fa)[8], half8_t (&fb)[2][8], const unsigned (&rab)[2],
                                           const unsigned (&rbb)[2], unsigned sa, unsigned sb, unsigned san, unsigned sbn, const unsigned (&vd1)[8],
                                           const unsigned (&vd2)[8], const char* sd1, const char* sd2, unsigned md1, unsigned md2, unsigned wv,
                                           const PARK (&park)[W4_NPARK16], const W4Store& st, std::false_type) {
    if constexpr (S == 4 && WIDE && !FIRST) gemm_ktile16_st4w<IDX0>(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, wv, park, st.srd, st.voff, st.rowb);
    else if constexpr (S == 4 && WIDE && FIRST) gemm_ktile16_first_st4w<IDX0>(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, wv, park, st.srd, st.voff, st.rowb);
    else if constexpr (S == 4 && !WIDE && !FIRST) gemm_ktile16_st4n<IDX0>(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, wv, park, st.srd, st.voff, st.rowb);
    else if constexpr (S == 4 && !WIDE && FIRST) gemm_ktile16_first_st4n<IDX0>(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, wv, park, st.srd, st.voff, st.rowb);
    else static_assert(S == 4, "tools/gen_gemmstep.py emits the four-stores-per-instance variants only");
}

// MFMA result -> vector ALU: the streams end on an MFMA and pad nothing (16 wait states in front of the epilogue's first accumulator read)
__device__ __forceinline__ void w4_settle(float16_t (&acc)[4][4]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3"
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]),
                   "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]),
                   "+a"(acc[3][2]), "+a"(acc[3][3]));
}
__device__ __forceinline__ void w4_settle(float4_t (&acc)[8][8]) {
#define W4_ROW(i) "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7])
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : W4_ROW(0), W4_ROW(1), W4_ROW(2), W4_ROW(3), W4_ROW(4), W4_ROW(5), W4_ROW(6), W4_ROW(7));
#undef W4_ROW
}
__device__ __forceinline__ void w4_ktile(float16_t (&acc)[4][4], half8_t (&fa)[2][4], half8_t (&fb)[2][4], const unsigned (&rab)[4], const unsigned (&rbb)[4],
                                         unsigned sa, unsigned sb, unsigned san, unsigned sbn, const unsigned (&vd1)[8], const unsigned (&vd2)[8],
                                         const char* sd1, const char* sd2, unsigned md1, unsigned md2, unsigned, bool first) {
    if (first) gemm_ktile_first(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2);
    else gemm_ktile(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2);
}
__device__ __forceinline__ void w4_ktile(float4_t (&acc)[8][8], half8_t (&fa)[8], half8_t (&fb)[2][8], const unsigned (&rab)[2], const unsigned (&rbb)[2],
                                         unsigned sa, unsigned sb, unsigned san, unsigned sbn, const unsigned (&vd1)[8], const unsigned (&vd2)[8],
                                         const char* sd1, const char* sd2, unsigned md1, unsigned md2, unsigned wv, bool first) {
    if (first) gemm_ktile16_first(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, wv);
    else gemm_ktile16(acc, fa, fb, rab, rbb, sa, sb, san, sbn, vd1, vd2, sd1, sd2, md1, md2, wv);
}

// S: parked output rows per K-tile instance; MODE: w4_epilogue; T16: the K-tile stream on 16x16x32 MFMAs (gemm_ktile16_mfma.inc)
template <int ACT, bool GATED, int S, int MODE, bool T16 = false>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [stage][X tile 32K | W tile 32K] x 2 (addressed by offset only)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;
    const int nk = p.K / 64;

    // fragment read addresses: rows l & 31 of tile i (immediate offset i * 4096), chunk (2 ks + (l >> 5)) ^ ((row >> 1) & 7); the stage
    // offsets (A ring: three stages from LDS 0, B ring: two from W4_BRING) are scalars added inside the stream
    // (T16: rows l & 15 of tile i (immediate offset i * 2048), chunk 4 s + (l >> 4) of slab s)
    unsigned rab[T16 ? 2 : 4], rbb[T16 ? 2 : 4];
    if constexpr (T16) {
        const int row = lane & 15, kg = lane >> 4, g = (row >> 1) & 7;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const unsigned off = (unsigned)(row * 128 + (((4 * sl + kg) ^ g) << 4));
            rab[sl] = wa * 16384 + off;
            rbb[sl] = wb * 16384 + off;
        }
    } else {
        const int row = lane & 31, hh = lane >> 5, g = (row >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned off = (unsigned)(row * 128 + (((2 * ks + hh) ^ g) << 4));
            rab[ks] = wa * 16384 + off;
            rbb[ks] = wb * 16384 + off;
        }
    }
    const unsigned wdma = wave * 8192;                       // this wave's eight pieces of a 32 KiB operand tile
    const char* const Ab = (const char*)p.X;      // MFMA A operand: tokens
    const char* const Bb = (const char*)p.W;      // MFMA B operand: features

    // Work order (persistent kernel, block b on XCD b % 8 - observed dispatch, for speed only). XCD x owns the token tiles
    // [tlo, tlo + NT); inside its share the order is: blocks of GF feature tiles x GT token tiles (GF * GT = 32 = one sweep of the
    // XCD's CUs: 32 workgroups share GF W tiles and GT X tiles, GF + GT L2 fills per K-tile instead of 32 + 32 / n_ft), then
    // (w4_order 1, round 6) feature groups fastest: the GT X tiles stay in the XCD's L2 while the W groups come round - or (w4_order 0)
    // token blocks fastest, so the GF W tiles stay in the XCD's L2 while the X tiles stream past once per group of GF feature tiles.
    // (Feature tiles fastest over ALL of them, as gemm_big_kernel walks, puts n_ft W tiles - 4 MiB at N = 4096, K = 512: the whole
    // L2 - in competition with the X stream: 2800-3600 cycles per K-tile at N = 4096 against 2250 at N = 512.) Slots that fall
    // outside the problem (ragged edges of the blocking) are skipped: every tile is visited exactly once for any grid.
    const int GF = p.w4_gf, GT = 32 / GF;
    const int tq = p.n_tt >> 3, tr = p.n_tt & 7;
    const int nbt = ((tq + (tr ? 1 : 0)) + GT - 1) / GT, nfg = (p.n_ft + GF - 1) / GF;
    const int slots = 8 * nfg * nbt * 32;
    auto tile_of = [&](int w, int& f0, int& t0) -> bool {
        const int xcd = w & 7, loc = w >> 3;
        const int NT = tq + (xcd < tr ? 1 : 0), tlo = xcd * tq + min(xcd, tr);
        const int blk = loc >> 5, within = loc & 31;
        const int fg = p.w4_order ? blk % nfg : blk / nbt, tg = p.w4_order ? blk / nfg : blk - fg * nbt;
        const int tf = fg * GF + within % GF, tt = tg * GT + within / GF;
        f0 = tf * 256;
        t0 = (tlo + tt) * 256;
        return tf < p.n_ft && tt < NT;
    };
    auto next_valid = [&](int w, int& f0, int& t0) -> int {       // first slot >= w (stride gridDim) that holds a tile; `slots` if none
        for (; w < slots; w += gridDim.x)
            if (tile_of(w, f0, t0)) return w;
        return slots;
    };

    int f0 = 0, t0 = 0;
    int work = next_valid(blockIdx.x, f0, t0);
    if (work >= slots) return;
    // Optional phase stagger ("gemm_stagger", as gemm_big_kernel's): the workgroups of an XCD start in eight groups p.stagger cycles apart.
    // Tried because all CUs reach their epilogues together; measured without effect here (an epilogue takes 9-12 k cycles at 1.7 and at
    // 2.4 GHz alike: it is bound by the CU's own store path - ~74 cycles per dwordx4 store instruction - not by a chip-wide burst).
    if (p.stagger > 0) {
        const unsigned long long until = __builtin_readcyclecounter() + (unsigned long long)(((blockIdx.x >> 3) & 7) * p.stagger);
        while (__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(32);
    }
    unsigned va[8], vb[8];
    w4_offsets<T16>(p, f0, t0, wave, lane, va, vb);

    // prologue: K-tiles 0 and 1 of both operands; fragments of k-step 0 of K-tile 0
    unsigned a0 = 0, a1 = W4_OP, a2 = 2 * W4_OP;             // A stage offsets of K-tiles G, G + 1, G + 2 (rotating)
    unsigned b0 = W4_BRING, b1 = W4_BRING + W4_OP;           // B stage offsets of K-tiles G, G + 1 (alternating)
#pragma unroll
    for (int n = 0; n < 8; ++n) w4_dma(a0 + wdma + n * 1024, va[n], Ab);
#pragma unroll
    for (int n = 0; n < 8; ++n) w4_dma(b0 + wdma + n * 1024, vb[n], Bb);
#pragma unroll
    for (int n = 0; n < 8; ++n) w4_dma(a1 + wdma + n * 1024, va[n], Ab + 128);
#pragma unroll
    for (int n = 0; n < 8; ++n) w4_dma(b1 + wdma + n * 1024, vb[n], Bb + 128);
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
    // fragment registers: 32x32x16 - double buffered by k-step, [parity][tile]; 16x16x32 - X: one ring of eight tiles, W: [slab][tile]
    std::conditional_t<T16, half8_t[8], half8_t[2][4]> fa;
    std::conditional_t<T16, half8_t[2][8], half8_t[2][4]> fb;
    if constexpr (T16) {
        // what an instance expects in place: X tiles 0, 1 and the eight W tiles of slab 0
        asm volatile("ds_read_b128 %0, %10 offset:0\n\tds_read_b128 %1, %10 offset:2048\n\t"
                     "ds_read_b128 %2, %11 offset:0\n\tds_read_b128 %3, %11 offset:2048\n\tds_read_b128 %4, %11 offset:4096\n\tds_read_b128 %5, %11 offset:6144\n\t"
                     "ds_read_b128 %6, %11 offset:8192\n\tds_read_b128 %7, %11 offset:10240\n\tds_read_b128 %8, %11 offset:12288\n\tds_read_b128 %9, %11 offset:14336\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(fa[0]), "=&v"(fa[1]), "=&v"(fb[0][0]), "=&v"(fb[0][1]), "=&v"(fb[0][2]), "=&v"(fb[0][3]), "=&v"(fb[0][4]), "=&v"(fb[0][5]),
                       "=&v"(fb[0][6]), "=&v"(fb[0][7])
                     : "v"(rab[0] + a0), "v"(rbb[0] + b0)
                     : "memory");
    } else {
    asm volatile("ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:8192\n\tds_read_b128 %3, %8 offset:12288\n\t"
                 "ds_read_b128 %4, %9 offset:0\n\tds_read_b128 %5, %9 offset:4096\n\tds_read_b128 %6, %9 offset:8192\n\tds_read_b128 %7, %9 offset:12288\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(fa[0][0]), "=&v"(fa[0][1]), "=&v"(fa[0][2]), "=&v"(fa[0][3]),
                   "=&v"(fb[0][0]), "=&v"(fb[0][1]), "=&v"(fb[0][2]), "=&v"(fb[0][3])
                 : "v"(rab[0] + a0), "v"(rbb[0] + b0)
                 : "memory");
    }

#ifdef BH_GEMM_STATS
    unsigned long long st_loop = 0, st_epi = 0, st_tiles = 0;
    const unsigned long long st_t0 = __builtin_readcyclecounter(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr bool G16 = GATED && T16;                       // SwiGLU outputs in whole 128-byte rows (w4_epilogue16g)
    using park_t = std::conditional_t<GATED && !T16, uint2_t, uint4_t>;
    constexpr int NPARK = G16 ? W4_NPARKG : T16 ? W4_NPARK16 : W4_NPARK;
    constexpr int NST = (NPARK + S - 1) / S;                 // K-tile instances that carry parked rows (nk >= NST: the launcher)
    park_t park[NPARK];
    W4Store pst;
    bool parked = false;
    const char* dA = Ab + 256;                               // DMA cursor: K-tile k + 2 of the instance of K-tile k ...
    const char* dB = Bb + 256;
    while (true) {
#ifdef BH_GEMM_STATS
        const unsigned long long st_a = __builtin_readcyclecounter();
#endif
        int nf0 = f0, nt0 = t0;
        int tf0 = 0, tt0 = 0;
        const int next = next_valid(work + gridDim.x, tf0, tt0);
        if (next < slots) { nf0 = tf0; nt0 = tt0; }         // (no next tile: the run-ahead DMAs re-fetch this one, nobody reads them)
        std::conditional_t<T16, float4_t[8][8], float16_t[4][4]> acc;
        // instance of K-tile k: reads A stage a0 / B stage b0 (k-step 0 of K-tile k + 1 from a1 / b1), D1 = A of K-tile k + 2 into A stage
        // a2, D2 = B of K-tile k + 2 into B stage b0. In the last two instances the DMA cursor is in the NEXT output tile: va / vb are
        // overwritten with its offsets in front of instance nk - 2 (nothing of this tile is fetched any more).
#define W4_ROTATE() do { const unsigned ta = a0; a0 = a1; a1 = a2; a2 = ta; const unsigned tb = b0; b0 = b1; b1 = tb; dA += 128; dB += 128; } while (0)
#define W4_CURSOR(k) do { if ((k) == nk - 2) { int lc = lane; asm volatile("" : "+v"(lc)); /* (opaque: nothing of it is hoisted and kept live) */ \
                                               w4_offsets<T16>(p, nf0, nt0, wave, lc, va, vb); dA = Ab; dB = Bb; } } while (0)
#define W4_ARGS acc, fa, fb, rab, rbb, a0, b0, a1, b1, va, vb, dA, dB, a2 + wdma, b0 + wdma, (unsigned)wave
        if (parked) {
            // the first NST instances also issue the previous tile's parked rows
#define W4_ST(T) if constexpr ((T) < NST) { W4_CURSOR(T); w4_inst_st<(T) * S, S, !GATED, (T) == 0>(W4_ARGS, park, pst, std::bool_constant<G16>{}); W4_ROTATE(); }
            W4_ST(0) W4_ST(1) W4_ST(2) W4_ST(3) W4_ST(4) W4_ST(5) W4_ST(6) W4_ST(7)
#undef W4_ST
            for (int k = NST; k < nk; ++k) {
                W4_CURSOR(k);
                w4_ktile(W4_ARGS, false);
                W4_ROTATE();
            }
        } else {
            w4_ktile(W4_ARGS, true);
            W4_ROTATE();
            for (int k = 1; k < nk; ++k) {
                W4_CURSOR(k);
                w4_ktile(W4_ARGS, false);
                W4_ROTATE();
            }
        }
#undef W4_ARGS
#undef W4_CURSOR
#undef W4_ROTATE
#ifdef BH_GEMM_STATS
        const unsigned long long st_b = __builtin_readcyclecounter();
#endif
        w4_settle(acc);
        // (a2: the A stage the last K-tile has just vacated - the next instance's D1 target - serves as the transposition scratch)
        if constexpr (G16) parked = w4_epilogue16g(p, acc, f0, t0, wa, wb, lane, smem + a2 + wdma, park, pst);
        else if constexpr (T16 && (MODE & 1) == 0) parked = w4_epilogue16p<ACT, MODE>(p, acc, f0, t0, wa, wb, lane, smem + a2 + wdma, park, pst);
        else parked = w4_epilogue<ACT, GATED, MODE>(p, acc, f0, t0, wa, wb, lane, smem + a2 + wdma, park, pst);
#ifdef BH_GEMM_STATS
        st_loop += st_b - st_a; st_epi += __builtin_readcyclecounter() - st_b; ++st_tiles;
#endif
        if (next >= slots) break;
        work = next; f0 = nf0; t0 = nt0;
    }
    if (parked) w4_store_parked<GATED, T16>(park, pst);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the run-ahead DMAs of the tile that does not exist
#ifdef BH_GEMM_STATS
    if (p.dbg != nullptr && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 133)) {
        unsigned long long* d = p.dbg + (blockIdx.x == 0 ? 0 : 8);
        d[0] = st_loop; d[1] = st_epi; d[2] = __builtin_readcyclecounter() - st_t0; d[3] = __builtin_amdgcn_s_memrealtime() - st_r0; d[4] = st_tiles;
    }
#endif
}

#ifdef BH_GEMM_STATS
unsigned long long* g_gemm_dbg = nullptr;
#endif
int g_w4_gf = 0;             // experiments: feature tiles per block of gemm_w4_kernel's work order (0 = default)
// bh_k_linear_order ("gemm_order"): GemmArgs::w4_order. 1 (default since round 6): an XCD finishes ALL feature groups of a block of token tiles
// before it moves on - the block's X tiles (GT x 256 KiB at K = 512) are fetched from HBM once and re-read from L2, the W tiles (4 MiB at
// N = 4096: Infinity-Cache resident) come round once per token block. 0 (rounds 4-5): token blocks fastest - X streamed from HBM once per
// feature group (fc1: four times; PMC 5.25 GB against 2.62 algorithmic). Lab, tools/gemm_lab_order.sh: +4 % where there is more than one
// feature group (Wqkv, fc1, the N = 4096 input projections), nothing elsewhere (profiles/r06_gemm_tile16.txt)
int g_w4_order = 1;
static int g_stagger = 0;    // bh_k_linear_stagger
// bh_k_linear_tile16 ("gemm_tile16"): gemm_w4_kernel's K-tile stream on 16x16x32 MFMAs (1, default since round 6: 5-9 % faster on every
// shape of tools/gemm_bench.py, sup 60.4 -> 59.2 ms per batch - profiles/r06_gemm_tile16.txt) or on 32x32x16 (0: rounds 4-5)
static int g_w4_t16 = 1;
static int g_force_v1 = 0;   // test / A-B hook (bh_k_linear_force_v1): 1 = v1 only, 2 = never v3 / v5, 3 = never v5 (v3 where it applies), 5 = v5 whenever the shape is legal (tests: small problems)

template <int ACT, bool GATED>
static int launch(const GemmArgs& a, hipStream_t s) {
    int grid = a.n_ft * a.n_tt;
    // v5 (gemm_w4_kernel) / v3 when the problem has at least ~2 waves of 256 x 256 tiles over the chip and no K tail
    {
        const int nf3 = (a.N + BF3 - 1) / BF3, nt3 = (a.M + BT3 - 1) / BT3;
        if (a.K % 128 == 0 && a.K >= 384 && (g_force_v1 == 0 || g_force_v1 == 5) && a.N % 256 == 0 && ((long)nf3 * nt3 >= 512 || g_force_v1 == 5) &&
            ((a.row_div == 1 && a.row_s_hi == 1) || a.row_div % 256 == 0) && (a.rot_cs == nullptr || a.rot_T >= 256) &&
            (long)a.ldo * 2 * (a.row_div == 1 ? 1 : a.row_s_lo) < (1l << 24) &&
            (long)a.M * a.ldx < (1l << 31) && (long)a.N * a.ldw < (1l << 31)) {
            const int cus = bh_cu_count();
            GemmArgs b = a;
            b.n_ft = nf3; b.n_tt = nt3;
            int gf = g_w4_gf > 0 ? g_w4_gf : 4;
            while (gf > 1 && (gf > nf3 || (g_w4_gf <= 0 && nf3 % gf != 0))) gf >>= 1;      // (a group that does not divide n_ft leaves slots empty: 6 tiles in groups of 4 wasted a quarter)
            b.w4_gf = gf;
            b.w4_order = g_w4_order;
            // "gemm_stagger" n > 0: phase groups n x 256 cycles apart (measured: no gain for this kernel - its epilogue is bound by the CU's
            // store path, not by a chip-wide burst - and the last groups finish up to 7 n x 256 cycles late); default off
            b.stagger = g_stagger > 0 ? g_stagger * 256 : 0;
            const int gt = 32 / gf, ntx = (nt3 >> 3) + ((nt3 & 7) ? 1 : 0);
            const long slots = 8l * ((nf3 + gf - 1) / gf) * ((ntx + gt - 1) / gt) * 32;
            // epilogue modes are compile-time (w4_epilogue): the combinations the engine uses are instantiated, anything else falls through
            // to the eight-wave kernel below
            const bool plain = a.scale == 1.0f && a.clamp_lo == -INFINITY && a.clamp_hi == INFINITY;
            const int mode = (a.res != nullptr ? 1 : 0) | (a.rot_cs != nullptr ? 2 : 0) | (plain ? 0 : 4);
            bool done = true;
// four parked rows per K-tile instance on every K (K = 384, six instances: six rows per instance over four of them measured 3 %
// slower than four rows over five - 0.879 against 0.853 ms on the hac CRF head)
#define W4_LAUNCH_T(A_, G_, MODE_, T16_)                                                                                            \
    do {                                                                                                                          \
        BH_CHECK_HIP(bh_max_lds((const void*)gemm_w4_kernel<A_, G_, 4, MODE_, T16_>, W4_LDS));                                    \
        hipLaunchKernelGGL((gemm_w4_kernel<A_, G_, 4, MODE_, T16_>), dim3(slots < cus ? (int)slots : cus), dim3(256), W4_LDS, s, b); \
    } while (0)
#define W4_LAUNCH(A_, G_, MODE_) do { if (g_w4_t16) W4_LAUNCH_T(A_, G_, MODE_, true); else W4_LAUNCH_T(A_, G_, MODE_, false); } while (0)
            if constexpr (GATED) {
                if (mode == 0) W4_LAUNCH(ACT_NONE, true, 0); else done = false;
            } else if constexpr (ACT == ACT_NONE) {
                if (mode == 0) W4_LAUNCH(ACT_NONE, false, 0);
                else if (mode == 1) W4_LAUNCH(ACT_NONE, false, 1);
                else if (mode == 2) W4_LAUNCH(ACT_NONE, false, 2);
                else if (mode == 4) W4_LAUNCH(ACT_NONE, false, 4);
                else done = false;
            } else if constexpr (ACT == ACT_TANH) {
                if (mode == 4) W4_LAUNCH(ACT_TANH, false, 4); else if (mode == 0) W4_LAUNCH(ACT_TANH, false, 0); else done = false;
            } else if constexpr (ACT == ACT_SWISH) {
                if (mode == 0) W4_LAUNCH(ACT_SWISH, false, 0); else done = false;
            } else {
                done = false;
            }
#undef W4_LAUNCH
#undef W4_LAUNCH_T
            if (done) return 0;
        }
        if (false) {
            return 0;
        }
        if (a.K % BK3 == 0 && (g_force_v1 == 0 || g_force_v1 == 3) && a.N >= 256 && a.N % 16 == 0 && (long)nf3 * nt3 >= 512 &&
            (long)a.M * a.ldx < (1l << 31) && (long)a.N * a.ldw < (1l << 31)) {
            const int cus = bh_cu_count();
            GemmArgs b = a;
            b.n_ft = nf3; b.n_tt = nt3;
            b.stagger = g_stagger;
            const int tiles = nf3 * nt3;
            BH_CHECK_HIP(bh_max_lds((const void*)gemm_big_kernel<ACT, GATED>, 4 * TILE3));
            hipLaunchKernelGGL((gemm_big_kernel<ACT, GATED>), dim3(tiles < cus ? tiles : cus), dim3(512), 4 * TILE3, s, b);
            return 0;
        }
    }
    if (a.K % BK2 == 0 && g_force_v1 != 1)
        hipLaunchKernelGGL((gemm_glds_kernel<ACT, GATED>), dim3(grid), dim3(256), 4 * TILE2, s, a);
    else
        hipLaunchKernelGGL((gemm_kernel<ACT, GATED>), dim3(grid), dim3(256), 4 * TILE_BYTES, s, a);
    return 0;
}

}  // namespace bh

void bh_k_linear_force_v1(int on) { bh::g_force_v1 = on; }
void bh_k_linear_stagger(int units) { bh::g_stagger = units; }
void bh_k_linear_tile16(int on) { bh::g_w4_t16 = on ? 1 : 0; }
void bh_k_linear_order(int order) { bh::g_w4_order = order ? 1 : 0; }
void bh_k_linear_gf(int gf) { bh::g_w4_gf = gf; }

int bh_k_linear(const void* X, const void* W, const float* bias, void* out, int M, int N, int K,
                int ldx, int ldw, int ldo, int act, float scale, float clamp_lo, float clamp_hi,
                int gated, int row_div, long row_s_hi, long row_s_lo, int row_lim, hipStream_t stream,
                const void* residual, int ldres, float res_scale) {
    using namespace bh;
    BH_REQUIRE(M > 0 && N > 0 && K > 0, "linear: empty problem M=%d N=%d K=%d", M, N, K);
    BH_REQUIRE(K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0, "linear: K/ldx/ldw must be multiples of 8 halves");
    BH_REQUIRE(ldo % 8 == 0, "linear: ldo must be a multiple of 8 halves");
    BH_REQUIRE(!gated || (N % 16 == 0), "linear: gated epilogue needs N %% 16 == 0");
    GemmArgs a;
    a.X = (const half_t*)X; a.W = (const half_t*)W; a.bias = bias; a.out = (half_t*)out;
    a.res = (const half_t*)residual; a.ldres = ldres; a.res_scale = res_scale;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldo = ldo;
    a.scale = scale; a.clamp_lo = clamp_lo; a.clamp_hi = clamp_hi;
    a.row_div = row_div > 0 ? row_div : 1;
    a.row_s_hi = row_div > 0 ? row_s_hi : 1;
    a.row_s_lo = row_div > 0 ? row_s_lo : 0;
    a.row_lim = (row_div > 0 && row_lim > 0) ? row_lim : 0x7fffffff;
    a.n_ft = (N + BF - 1) / BF; a.n_tt = (M + BT - 1) / BT;
#ifdef BH_GEMM_STATS
    a.dbg = g_gemm_dbg;
#endif
    int rc = 0;
    if (gated) { rc = launch<ACT_NONE, true>(a, stream); }
    else switch (act) {
        case ACT_NONE: rc = launch<ACT_NONE, false>(a, stream); break;
        case ACT_SWISH: rc = launch<ACT_SWISH, false>(a, stream); break;
        case ACT_TANH: rc = launch<ACT_TANH, false>(a, stream); break;
        case ACT_RELU: rc = launch<ACT_RELU, false>(a, stream); break;
        default: BH_REQUIRE(false, "linear: unknown activation %d", act);
    }
    if (rc) return rc;
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// Packed Wqkv projection with the rotary embedding (and the softmax scale of q) applied in the epilogue:
// out[m][0:D) = rot(q) * qscale, out[m][D:2D) = rot(k), out[m][2D:3D) = v; position of row m is m % T. cos_sin: [>=T][32][2].
int bh_k_linear_qkv_rotary(const void* X, const void* W, const float* bias, void* out, int M, int D, int K, const float* cos_sin,
                           int T, float qscale, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(M > 0 && D > 0 && K > 0 && T > 0 && cos_sin != nullptr, "linear_qkv_rotary: bad arguments");
    BH_REQUIRE(D % 64 == 0 && K % 8 == 0, "linear_qkv_rotary: d_model must be a multiple of 64 (heads of 64), K of 8");
    GemmArgs a;
    a.X = (const half_t*)X; a.W = (const half_t*)W; a.bias = bias; a.out = (half_t*)out;
    a.res = nullptr; a.ldres = 0;
    a.M = M; a.N = 3 * D; a.K = K; a.ldx = K; a.ldw = K; a.ldo = 3 * D;
    a.scale = 1.0f; a.clamp_lo = -INFINITY; a.clamp_hi = INFINITY;
    a.row_div = 1; a.row_s_hi = 1; a.row_s_lo = 0; a.row_lim = 0x7fffffff;
    a.n_ft = (a.N + BF - 1) / BF; a.n_tt = (M + BT - 1) / BT;
    a.rot_cs = cos_sin; a.rot_T = T; a.rot_nfeat = 2 * D; a.rot_qfeat = D; a.rot_qscale = qscale;
    if (int rc = launch<ACT_NONE, false>(a, stream)) return rc;
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}
