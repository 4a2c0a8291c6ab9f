// 8-bit recurrent path "Q8-1" for gfx950 -- the MI355X counterpart of what `--quantize` asks of koi.lstm.update_graph
// (/root/reference bonito/cli/basecaller.py:186-189, bonito/crf/model.py:240-246: `quantize` hands the LSTM stack to
// koi's int8 kernels; those are closed source, so the arithmetic is DEFINED here and in oracle/lstm_q8_ref.py, not copied).
//
//   W_ih, W_hh : int8 per output row, s_r = max|W[r,:]| / 127, q = clip(rint(W / s_r), -127, 127)
//   x_t, h_t   : int8 with static scales: h in (-1,1) -> rint(127 h16), x of the first layer -> rint(127 x / bound)
//   pre[r]     = (float(sum_k q_ih[r,k] xq[k]) * sx[r] + b[r]) + float(sum_k q_hh[r,k] hq[k]) * sh[r]
//                sx = s_ih * float(bound/127), sh = s_hh / 127; the two int32 sums are exact (v_mfma_i32_16x16x64_i8),
//                every fp32 operation is rounded separately (no contraction), so `pre` is bit-identical to the oracle's
//   cell       : the fp32 gate arithmetic of the fp16 kernels (lstm_cell), h published as fp16 and quantised from THAT value.
//
// Structure: the workgroup-shared kernel of lstm.hip (four slices of one ring per workgroup, lstm_layer_wgx_kernel) with three changes
// the 8-bit operands allow (this file had them first; the fp16 kernel took the ring buffer and the LDS-DMA x stream from here):
//   * weights: W_hh and W_ih tiles of a wave are 2 * MT * (H/64) * 4 registers (144 at H = 384, U = 12 units per wave:
//     half of the fp16 kernel's 288), MFMAs per step halve (K = 64 per instruction);
//   * the exchange no longer goes through the layer's output tensor. Activations travel as int8 in MFMA B-fragment order
//     ([k-step][lane][16 bytes]: what a consumer lane needs is one aligned 16-byte piece, a k-step of a ring is one
//     contiguous KiB), and the hand-off uses a small RING BUFFER of four time slots per ring (4 x H x 16 bytes; 768 KiB for
//     a 512-chunk batch: it never leaves the L2). Bytes are valid when they differ from the sentinel 0x80 (-128 is not a
//     quantised value). A producer re-arms its own bytes of slot (t+2)%4 at step t: every wave of the ring has published
//     h_{t-1} by then, so nobody reads h_{t-2} any more, and the re-arm store is two steps ahead of the data store that
//     replaces it. No sentinel pre-fill of the output tensor (the fp16 path's fill kernel and the first-touch fetch of the
//     polled lines from HBM are gone), no flags, no fences; every byte is validated on its own;
//   * the layer output for the next layer is a second, plain store of the same bytes (fragment order, [T][ring][k-step]);
//     only the last recurrent layer also writes fp16 rows for the CRF head.
// Everything else -- rings of 16 chunks, XCD agreement for the store policy, quarter polling + one workgroup barrier per
// step, input projection of step t+1 under the exchange round trip of step t, bounded spins raising *err -- is as in lstm.hip.
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace bh {

typedef int int4_t __attribute__((ext_vector_type(4)));
typedef unsigned short __attribute__((may_alias)) u16a_t;
typedef unsigned char __attribute__((may_alias)) u8a_t;
typedef unsigned int __attribute__((may_alias)) u32a_t;
typedef unsigned long long __attribute__((may_alias)) u64a_t;

struct LstmQ8Args {
    const int8_t* xq;      // layer input, fragment order [T][R][NK8][64 lanes][16]   (R = ring stride of the whole batch)
    int8_t* hq_out;        // layer output for the next recurrent layer, same layout (may be null)
    half_t* h16_out;       // [T][N][H] fp16 rows (may be null): what the linear layer after the recurrent stack reads
    int8_t* ex;            // exchange ring buffer [4][R][NK8][64][16], armed with 0x80
    const int8_t* wih;     // packed tiles [slice][m][k-step][lane][16]
    const int8_t* whh;
    const float* sx;       // [4H]  s_ih * float(bound / 127)   (torch gate-major rows)
    const float* sh;       // [4H]  s_hh / 127
    const float* bias;     // [4H]  b_ih + b_hh
    int T, N, H, R;
    int n_rings;           // rings served by this launch (pointers are already offset to the first one)
    int reverse;
    int* err;
    unsigned max_spins;
    int* xcc_ws;
    int force_slow;
    int tune;              // bit 2: per-wave cycle statistics into xcc_ws' tail
    int* dbg;              // test hook: int32 sums [T][N][4H][2] (x part, h part); null in the product path
};

__device__ __forceinline__ int xcc_id_q8() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xF); }

// any byte of the four words equal to 0x80?  (x ^ 0x80 has a zero byte)
__device__ __forceinline__ bool has_sentinel(const uint4_t& v) {
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned y = v[i] ^ 0x80808080u;
        r |= (y - 0x01010101u) & ~y & 0x80808080u;
    }
    return r != 0;
}

// same cell as the fp16 kernels (lstm.hip: lstm_cell), duplicated here because that one is file-local there
__device__ __forceinline__ float q8_cell(float ai, float af, float ag, float ao, float& c) {
    const float ei = __expf(-__builtin_amdgcn_fmed3f(ai, -25.0f, 25.0f));
    const float ef = __expf(-__builtin_amdgcn_fmed3f(af, -25.0f, 25.0f));
    const float eg = __expf(-2.0f * __builtin_amdgcn_fmed3f(ag, -12.5f, 12.5f));
    const float eo = __expf(-__builtin_amdgcn_fmed3f(ao, -25.0f, 25.0f));
    const float didg = __fmul_rn(1.0f + ei, 1.0f + eg);
    const float df = 1.0f + ef;
    const float num = __builtin_fmaf(c, didg, __fmul_rn(1.0f - eg, df));
    c = __fmul_rn(num, rcpf_(__fmul_rn(df, didg)));
    const float ec = __expf(-2.0f * __builtin_amdgcn_fmed3f(c, -12.5f, 12.5f));
    const float hv = __fmul_rn(1.0f - ec, rcpf_(__fmul_rn(1.0f + ec, 1.0f + eo)));
    return (fabsf(hv) <= 1.0f) ? hv : 0.0f;
}

// One 1 KiB global -> LDS DMA: lane l moves 16 bytes from g (per lane) to lds + 16 l (lds is wave-uniform). The compiler
// neither counts nor waits for it: the issuing wave covers it with its own s_waitcnt vmcnt(0) before a barrier publishes it.
__device__ __forceinline__ void dma16_q8(const char* g, char* lds) {
    const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory");
}

// byte offset of unit u (0 .. 64*NK8) of chunk c inside a ring tile in fragment order
__device__ __forceinline__ int frag_byte(int u, int c) { return (((u >> 6) * 64 + ((u >> 4) & 3) * 16 + c) << 4) + (u & 15); }

// LAST: this layer also writes fp16 rows (the layer after it is not an 8-bit recurrent layer); DBG: dump the integer sums.
// Template parameters rather than run-time tests so that the gate arithmetic of a wave's MT units is ONE basic block (the
// compiler interleaves the transcendentals of the three cells; behind per-unit branches they ran one after the other).
template <int NK8, int MT, int WPS, bool LAST, bool DBG>
__global__ __launch_bounds__(256, WPS) void lstm_layer_q8_kernel(LstmQ8Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int U = 4 * MT, KQ = (NK8 + 3) / 4, TILE = NK8 * 1024;
    constexpr bool EXACT = NK8 % 4 == 0;
    const int H = p.H;
    const int NSL = H / U, WPR = NSL / 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7;
    const int lwg = blockIdx.x >> 3;
    const int rl = lwg / WPR;
    // test hook ("lstm_tune" bit 5): spread the workgroups of every ring over all eight XCDs, so that the placement-independent
    // (write-through) hand-off really crosses XCDs
    const int ring = rl * 8 + ((p.tune & 32) ? ((xcd + (lwg - rl * WPR)) & 7) : xcd);
    const int slice = (lwg - rl * WPR) * 4 + wave;
    if (ring >= p.n_rings) return;                      // whole workgroup (same ring) leaves together

    char* hbuf = smem;                                  // [2][NK8][64][16]  B fragments of h_{t-1}
    char* xbuf = smem + 2 * TILE;                       // [3][NK8][64][16]  B fragments of x_t: consumed / landed / landing
    char* stage = smem + 5 * TILE + wave * (16 * U * 3);   // per wave: [16 chunks][U] bytes + [16][U] halves

    int4_t whh[MT][NK8], wih[MT][NK8];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int ks = 0; ks < NK8; ++ks) {
            const long o = ((((long)slice * MT + m) * NK8 + ks) * 64 + lane) * 16;
            whh[m][ks] = *(const int4_t*)(p.whh + o);
            wih[m][ks] = *(const int4_t*)(p.wih + o);
        }
    const int c = lane & 15, q = lane >> 4;
    float cst[MT];
    float4_t sx4[MT], sh4[MT], b4[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        cst[m] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * H + slice * U + q * MT + m;
            sx4[m][i] = p.sx[r]; sh4[m][i] = p.sh[r]; b4[m][i] = p.bias[r];
        }
    }
    bool dead = false;
    bool fast = false;
    {
        int* slot = p.xcc_ws + (long)ring * NSL;
        const int mine = xcc_id_q8();
        if (lane == 0) __hip_atomic_store(slot + slice, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        bool ok = false;
        while (true) {
            bool unset = false, other = false;
            for (int i = lane; i < NSL; i += 64) {
                const int v = __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unset |= v < 0; other |= v != mine;
            }
            if (!__any(unset)) { ok = !__any(other); break; }
            if (++spins > p.max_spins) break;          // not an error: the write-through policy is valid for any placement
            __builtin_amdgcn_s_sleep(4);
        }
        fast = ok && !p.force_slow;
    }

    int t = p.reverse ? p.T - 1 : 0;
    const int dt = p.reverse ? -1 : 1;
    const long t_stride = (long)p.R * TILE;             // one time step of xq / hq_out
    const long slot_stride = (long)p.R * TILE;          // one time slot of the exchange ring
    const int8_t* xptr = p.xq + (long)ring * TILE + lane * 16;
    char* exr = (char*)p.ex + (long)ring * TILE;        // this ring's tile inside slot 0
    const int lo = lane * 16;

    // the lanes that move this wave's U units out: lane -> (chunk cc, dword part): 4 consecutive units of one chunk
    constexpr int PARTS = U / 4;
    const bool mover = lane < 16 * PARTS;
    const int cc = lane / PARTS, part = lane - cc * PARTS;
    const int my_byte = frag_byte(slice * U + part * 4, cc);          // inside a ring tile

    uint4_t hq[KQ];
    int4_t xacc[MT];
    float4_t xpre[MT];          // float(input sum) * sx + b, formed right behind the input projection (inside the exchange round trip)

    auto x_phase = [&](const char* xb) {
#pragma unroll
        for (int m = 0; m < MT; ++m) xacc[m] = int4_t{0, 0, 0, 0};
        int4_t bf[NK8];
#pragma unroll
        for (int ks = 0; ks < NK8; ++ks) bf[ks] = *(const int4_t*)(xb + ks * 1024 + lo);
#pragma unroll
        for (int ks = 0; ks < NK8; ++ks)
#pragma unroll
            for (int m = 0; m < MT; ++m) xacc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wih[m][ks], bf[ks], xacc[m], 0, 0, 0);
        if constexpr (!DBG) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) xpre[m][i] = __fadd_rn(__fmul_rn((float)xacc[m][i], sx4[m][i]), b4[m][i]);
        }
    };
    // The x stream never touches registers: each wave moves its share of the k-steps of a tile with LDS-DMA (1 KiB per
    // instruction, lane l -> LDS base + 16 l, which IS the fragment order), so the compiler has no destination registers to
    // wait for and cannot put the fetch on the critical path. A wave's DMAs are covered by its own vmcnt(0) ahead of the
    // barrier that publishes the tile. Shares: the polls of the exchange go w, w+4; the x stream goes 3-w, 7-w, so every
    // wave moves three KiB per step at H = 384.
    auto x_dma = [&](int tt, int slot) {
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const int ks = (3 - wave) + 4 * kk;
            if (EXACT || ks < NK8) dma16_q8((const char*)(xptr + (long)tt * t_stride + ks * 1024), xbuf + (slot * NK8 + ks) * 1024);
        }
    };

    // ---- prologue: x_0 and x_1 -> LDS, input projection of step 0 -------------------------------------------------------
    x_dma(t, 0);
    x_dma(p.T > 1 ? t + dt : t, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the builtin (unlike inline asm) also clears the compiler's own scoreboard
    __syncthreads();
    x_phase(xbuf);
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) hq[kk] = uint4_t{0, 0, 0, 0};

    long long st_poll = 0, st_rounds = 0, st_first_ok = 0, st_bar = 0, st_rec = 0, st_x = 0, st_mf = 0, st_gate = 0;
    const long long st_t0 = __builtin_readcyclecounter();

    for (int step = 0; step < p.T; ++step, t += dt) {
        const int par = step & 1;
        // ---- B. my quarter of h_{t-1}: round one was issued right after the previous store --------------------------
        const long long pc0 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        if (step > 0) {
            // wave-uniform descriptor (ring tile of the slot), per-lane offset: a per-lane base would make the compiler wrap
            // every load in a 64-trip readfirstlane loop
            const char* src = exr + (long)((step - 1) & 3) * slot_stride;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, TILE, 0x00020000);
            unsigned spins = dead ? p.max_spins : 0u;
            unsigned pend = 0;
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int ks = wave + 4 * kk;
                if ((EXACT || ks < NK8) && __any(has_sentinel(hq[kk]))) pend |= (1u << kk);
            }
            unsigned rounds = 1;
            while (pend != 0) {
                if (++spins > p.max_spins) {
                    if (lane == 0 && !dead) atomicExch(p.err, 1);
                    dead = true;
                    break;
                }
                if (!(p.tune & 1)) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk)
                    if (pend & (1u << kk))
                        hq[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, (wave + 4 * kk) * 1024 + lo, 0, (int)0x80000010);
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk)
                    if ((pend & (1u << kk)) && !__any(has_sentinel(hq[kk]))) pend &= ~(1u << kk);
                ++rounds;
            }
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int ks = wave + 4 * kk;
                if (EXACT || ks < NK8) *(uint4_t*)(hbuf + (par * NK8 + ks) * 1024 + lo) = hq[kk];
            }
            if (p.tune & 4) { st_poll += __builtin_readcyclecounter() - pc0; st_rounds += rounds; st_first_ok += (rounds == 1); }
        }
        // ---- D. publish h_{t-1} (just written) and x_{t+1} (DMA issued a step ago) to the workgroup. The explicit vmcnt(0)
        //         covers this wave's share of the x tile and, for the exchange protocol, the re-arm store of the previous
        //         step: it is complete before this wave publishes anything newer ---------------------------------------
        const long long pc1 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the builtin (unlike inline asm) also clears the compiler's own scoreboard
        __syncthreads();
        const long long pc2 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        // ---- E. recurrent part, gates, publish h_t --------------------------------------------------------------------
        int4_t acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = int4_t{0, 0, 0, 0};
        if (step > 0) {
            const char* hb = hbuf + par * TILE + lo;
            int4_t bf[NK8];
#pragma unroll
            for (int ks = 0; ks < NK8; ++ks) bf[ks] = *(const int4_t*)(hb + ks * 1024);
#pragma unroll
            for (int ks = 0; ks < NK8; ++ks)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(whh[m][ks], bf[ks], acc[m], 0, 0, 0);
        }
        const long long pcm = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        // ---- X. request x_{t+2} into the slot x_{t-1} was consumed from; it is needed behind the NEXT barrier. Issued between
        //         the MFMAs and the gates: older than this step's stores and polls in the memory queue, and behind the
        //         compiler's own vmcnt(0) at the head of this phase (which would otherwise wait for it) ---------------------
        if (step + 2 < p.T) x_dma(t + 2 * dt, (step + 2) % 3);
        if constexpr (DBG) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long r = ((long)t * p.N + ring * 16 + c) * 4 * H + i * H + slice * U + q * MT + m;
                    p.dbg[2 * r] = xacc[m][i];
                    p.dbg[2 * r + 1] = acc[m][i];
                }
        }
        u8a_t* sg8 = (u8a_t*)stage + c * U + q * MT;
        u16a_t* sg16 = (u16a_t*)(stage + 16 * U) + c * U + q * MT;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float pre[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                pre[i] = __fadd_rn(DBG ? __fadd_rn(__fmul_rn((float)xacc[m][i], sx4[m][i]), b4[m][i]) : xpre[m][i],
                                   __fmul_rn((float)acc[m][i], sh4[m][i]));
            const float hv = q8_cell(pre[0], pre[1], pre[2], pre[3], cst[m]);
            const half_t h16 = (half_t)hv;
            const float hq_f = __builtin_rintf(__fmul_rn((float)h16, 127.0f));       // |h16| <= 1 -> |hq| <= 127, never the sentinel
            sg8[m] = (unsigned char)(signed char)(int)hq_f;
            if constexpr (LAST) sg16[m] = __builtin_bit_cast(unsigned short, h16);
        }
        const long long pcg = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        if (mover) {
            const unsigned packed = *(const u32a_t*)((const char*)stage + cc * U + part * 4);
            unsigned* dst = (unsigned*)(exr + (long)(step & 3) * slot_stride + my_byte);
            if (fast) *dst = packed;                                                                    // stays in this XCD's L2
            else __hip_atomic_store(dst, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);               // sc1 write-through
            // ---- R. re-arm my bytes of slot (step+2)&3. It holds h_{t-2}; behind this step's barrier the workgroup has seen
            //         every k-step of h_{t-1}, i.e. EVERY wave of the ring has published h_{t-1} and so finished reading
            //         h_{t-2} (a wave publishes only after its workgroup's barrier, which follows all four quarter polls). The
            //         data store into this slot follows two steps from now; the vmcnt(0) ahead of the next barrier completes
            //         the re-arm before this wave publishes h_{t+1}, and nobody polls the slot for h_{t+2} before having seen
            //         everybody's h_{t+1}: no consumer can find the old bytes -------------------------------------------
            if (step >= 2 && step + 2 < p.T) {
                unsigned* ra = (unsigned*)(exr + (long)((step + 2) & 3) * slot_stride + my_byte);
                if (fast) *ra = 0x80808080u;
                else __hip_atomic_store(ra, 0x80808080u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!LAST || p.hq_out != nullptr) *(unsigned*)(p.hq_out + (long)t * t_stride + (long)ring * TILE + my_byte) = packed;
            if constexpr (LAST) {
                const unsigned long long h4 = *(const u64a_t*)((const char*)stage + 16 * U + (cc * U + part * 4) * 2);
                *(unsigned long long*)(p.h16_out + ((long)t * p.N + ring * 16 + cc) * H + slice * U + part * 4) = h4;
            }
        }
        // ---- F. first poll round for h_t goes out now; it is checked after the input projection -----------------------
        if (step + 1 < p.T) {
            const char* src = exr + (long)(step & 3) * slot_stride;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, TILE, 0x00020000);
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int ks = wave + 4 * kk;
                if (EXACT || ks < NK8) hq[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, ks * 1024 + lo, 0, (int)0x80000010);
            }
        }
        const long long pc3 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        // ---- G. input projection of step t+1 from the x tile published at D --------------------------------------------
        x_phase(xbuf + ((step + 1) % 3) * TILE);
        if (p.tune & 4) {
            const long long now = __builtin_readcyclecounter();
            st_bar += pc2 - pc1; st_rec += pc3 - pc2; st_x += now - pc3; st_mf += pcm - pc2; st_gate += pcg - pcm;
        }
    }
    if ((p.tune & 4) && lane == 0) {
        long long* st = (long long*)((char*)p.xcc_ws + (((size_t)p.n_rings * NSL * sizeof(int) + 64 + 7) & ~(size_t)7)) + ((long)ring * NSL + slice) * 16;
        st[0] = __builtin_readcyclecounter() - st_t0;
        st[1] = st_poll; st[2] = st_rounds; st[3] = st_first_ok; st[4] = st_x; st[5] = st_bar; st[6] = st_rec;
        st[7] = st_mf; st[8] = st_gate;
    }
}

// fp16 rows [T][N][H] (time-major activations of the layer before the recurrent stack) -> int8 fragment order
// [T][R][NK8][64][16], q = clip(rint(x * scale), -127, 127), scale = float(127 / bound). One thread per 16-byte piece.
__global__ void quantise_rows_kernel(const half_t* x, int8_t* out, long pieces, int N, int H, int NK8, int R, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pieces) return;
    // piece index = ((t * R + ring) * NK8 + ks) * 64 + lane
    const int lane = (int)(i & 63);
    long r = i >> 6;
    const int ks = (int)(r % NK8); r /= NK8;
    const int ring = (int)(r % R);
    const long t = r / R;
    const int c = lane & 15, q = lane >> 4;
    const int u0 = ks * 64 + q * 16;
    const int n = ring * 16 + c;
    uint4_t o = {0, 0, 0, 0};
    if (n < N && u0 < H) {          // H % 16 == 0: a piece is either all real units or all padding
        const half_t* src = x + ((long)t * N + n) * H + u0;
        half8_t a = *(const half8_t*)src, b = *(const half8_t*)(src + 8);
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned v = 0;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const int e = j * 4 + k2;
                const float f = (float)(e < 8 ? a[e] : b[e - 8]);
                const float qf = __builtin_amdgcn_fmed3f(__builtin_rintf(__fmul_rn(f, scale)), -127.0f, 127.0f);
                v |= ((unsigned)(unsigned char)(signed char)(int)qf) << (8 * k2);
            }
            w[j] = v;
        }
        o = uint4_t{w[0], w[1], w[2], w[3]};
    }
    *(uint4_t*)(out + i * 16) = o;
}

}  // namespace bh

// Units per wave of the 8-bit kernel for hidden size H (0: not covered). `variant` 1 asks for the single-tile geometry
// (U = 4, three workgroups per CU) where it is instantiated.
// Hidden units per wave of the 8-bit kernel, 0 = this width has no instance (the layer then keeps the fp16 kernels). Must name
// exactly the (k-steps, M tiles) pairs instantiated in bh_k_lstm_layer_q8 below: a width this accepted without an instance
// (48, 240, 320, 432, 448, 480) made every forward of a quantised model fail instead of falling back (advisor finding, round 2).
int bh_k_lstm_q8_units(int H, int variant) {
    if (H % 16 != 0 || H > 512 || H <= 0) return 0;
    if (variant == 1 && H == 384) return 4;
    const int nk8 = (H + 63) / 64;
    if (H % 48 == 0) return (nk8 == 2 || nk8 == 3 || nk8 == 5 || nk8 == 6) ? 12 : 0;      // 96, 144, 192, 288, 336, 384
    if (H % 64 == 0) return (nk8 == 1 || nk8 == 2 || nk8 == 4 || nk8 == 8) ? 16 : 0;      // 64, 128, 256, 512
    return 0;
}
size_t bh_k_lstm_q8_tile_bytes(int H) { return (size_t)((H + 63) / 64) * 1024; }

// W [4H][K = H] fp32 (torch gate order) -> int8 tiles [slice][m][k-step][lane][16] + per-row scales (s[r] = max|row| / 127).
// Row r of tile m of a slice is (unit slice*U + (r>>2)*MT + m, gate r&3); lane l holds row l&15, columns ks*64 + (l>>4)*16 + j.
int bh_k_lstm_q8_pack(const float* w, int H, int U, int8_t* packed, float* scale) {
    BH_REQUIRE(w && packed && scale && U > 0 && H % (4 * U) == 0, "lstm_q8_pack: bad arguments");
    const int MT = U / 4, nk8 = (H + 63) / 64, nsl = H / U;
    for (int r = 0; r < 4 * H; ++r) {
        float mx = 0.0f;
        for (int k = 0; k < H; ++k) mx = fmaxf(mx, fabsf(w[(size_t)r * H + k]));
        scale[r] = mx > 0.0f ? mx / 127.0f : 1.0f;
    }
    for (int s = 0; s < nsl; ++s)
        for (int m = 0; m < MT; ++m)
            for (int ks = 0; ks < nk8; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 16; ++j) {
                        const int rr = lane & 15;
                        const int row = (rr & 3) * H + s * U + (rr >> 2) * MT + m;
                        const int col = ks * 64 + (lane >> 4) * 16 + j;
                        int8_t v = 0;
                        if (col < H) {
                            float qf = rintf(w[(size_t)row * H + col] / scale[row]);       // round half to even (default mode)
                            qf = fminf(127.0f, fmaxf(-127.0f, qf));
                            v = (int8_t)qf;
                        }
                        packed[((((size_t)s * MT + m) * nk8 + ks) * 64 + lane) * 16 + j] = v;
                    }
    return 0;
}

int bh_k_quantise_rows(const void* x, void* out, int T, int N, int H, int R, float bound, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(H % 16 == 0, "quantise_rows: H must be a multiple of 16");
    const int nk8 = (H + 63) / 64;
    const long pieces = (long)T * R * nk8 * 64;
    const float scale = (float)(127.0 / (double)bound);
    hipLaunchKernelGGL(quantise_rows_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, (const half_t*)x, (int8_t*)out,
                       pieces, N, H, nk8, R, scale);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// Arm the exchange ring buffer of a layer ([4][R][tile]) with the sentinel; padding units of the last k-step (H % 64 != 0)
// are never produced and must not read as the sentinel, so they are zeroed. Once per layer, before its launches.
int bh_k_lstm_q8_arm(void* ex, int R, int H, hipStream_t stream) {
    const size_t tile = bh_k_lstm_q8_tile_bytes(H);
    BH_CHECK_HIP(hipMemsetAsync(ex, 0x80, 4 * (size_t)R * tile, stream));
    if (H % 64 != 0) {
        const int ks = H / 64, first_q = (H % 64) / 16;
        BH_CHECK_HIP(hipMemset2DAsync((char*)ex + (size_t)ks * 1024 + (size_t)first_q * 256, tile, 0, (size_t)(4 - first_q) * 256,
                                      4 * (size_t)R, stream));
    }
    return 0;
}

// One launch serves the rings whose workgroups fit the device together (co-residency, as the fp16 wg kernel). `ex` must hold
// 4 * R * tile bytes and be armed (bh_k_lstm_q8_arm). R = ring stride of the tensors (N / 16 of the whole batch), n_rings = rings of this launch.
int bh_k_lstm_layer_q8(const void* xq, const void* wih, const void* whh, const float* sx, const float* sh, const float* bias,
                       void* hq_out, void* h16_out, void* ex, int T, int N, int H, int R, int n_rings, int reverse, int* err_flag,
                       hipStream_t stream, int* xcc_ws, int flags, int variant, int* dbg, unsigned max_spins) {
    using namespace bh;
    const int U = bh_k_lstm_q8_units(H, variant);
    BH_REQUIRE(U != 0, "lstm_q8: hidden size %d is not covered by the 8-bit kernel", H);
    BH_REQUIRE(N % 16 == 0 && n_rings > 0 && n_rings <= R, "lstm_q8: bad batch geometry (N=%d, rings=%d of %d)", N, n_rings, R);
    int dev = 0, cus = 0;
    BH_CHECK_HIP(hipGetDevice(&dev));
    BH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int nsl = H / U, wpr = nsl / 4, nk8 = (H + 63) / 64;
    const int per_cu = U == 4 ? 3 : (variant == 2 && H == 384) ? 2 : 1;
    const int rl = (n_rings + 7) / 8;
    const int grid = 8 * rl * wpr;
    BH_REQUIRE(grid <= cus * per_cu, "lstm_q8: %d workgroups must be co-resident but the device holds %d; split the batch", grid, cus * per_cu);
    BH_REQUIRE(xcc_ws != nullptr && ex != nullptr, "lstm_q8: missing workspace");
    BH_CHECK_HIP(hipMemsetAsync(xcc_ws, 0xFF, (size_t)n_rings * nsl * sizeof(int), stream));
    const size_t tile = (size_t)nk8 * 1024;
    LstmQ8Args a{(const int8_t*)xq, (int8_t*)hq_out, (half_t*)h16_out, (int8_t*)ex, (const int8_t*)wih, (const int8_t*)whh, sx, sh, bias,
                 T, N, H, R, n_rings, reverse, err_flag, max_spins, xcc_ws, flags & 1, flags >> 8, dbg};
    const size_t lds = 5 * tile + 4 * (size_t)(16 * U * 3);
    const bool last = h16_out != nullptr;
    BH_REQUIRE(last || hq_out != nullptr, "lstm_q8: no output buffer");
#define BH_Q8_LAUNCH(NK8, MT, WPS, LAST, DBG)                                                                            \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            BH_CHECK_HIP(bh_max_lds((const void*)lstm_layer_q8_kernel<NK8, MT, WPS, LAST, DBG>, (int)lds));                    \
        hipLaunchKernelGGL((lstm_layer_q8_kernel<NK8, MT, WPS, LAST, DBG>), dim3(grid), dim3(256), lds, stream, a);       \
    } while (0)
#define BH_Q8(NK8, MT, WPS)                                                                                              \
    if (nk8 == NK8 && U == 4 * MT && per_cu == WPS) {                                                                                  \
        if (dbg) BH_Q8_LAUNCH(NK8, MT, WPS, true, true);                                                                 \
        else if (last) BH_Q8_LAUNCH(NK8, MT, WPS, true, false);                                                          \
        else BH_Q8_LAUNCH(NK8, MT, WPS, false, false);                                                                   \
    } else
    BH_Q8(6, 3, 1) BH_Q8(6, 3, 2) BH_Q8(6, 1, 3) BH_Q8(2, 3, 1) BH_Q8(3, 3, 1) BH_Q8(5, 3, 1) BH_Q8(1, 4, 1) BH_Q8(2, 4, 1) BH_Q8(4, 4, 1) BH_Q8(8, 4, 1)
    { BH_REQUIRE(false, "lstm_q8: no kernel instance for H=%d", H); }
#undef BH_Q8_LAUNCH
#undef BH_Q8
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}
