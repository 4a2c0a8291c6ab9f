// Persistent weight-stationary LSTM layer for gfx950 -- the dominant kernel of the fast/hac models.
// Replaces koi.lstm.update_graph's fused CUDA LSTM (call site /root/reference bonito/crf/model.py:240-246)
// and torch.nn.LSTM under bonito.nn.LSTM / RNNWrapper (bonito/nn.py:353-415: single layer,
// unidirectional, gate order i,f,g,o, h0 = c0 = 0, `reverse` = run the time loop backwards).
//
// Decomposition (MI355X-first):
//   * The input projection x_t W_ih^T + b for ALL t is one big MFMA GEMM (gemm.hip) -> G[T][N][4H].
//   * Only h_{t-1} W_hh^T and the gate math are inside the time loop, which is strictly serial in t.
//     The batch is cut into "rings" of 16 chunks (one MFMA column tile). A ring is served by H/16
//     waves; wave (ring, slice) keeps the 64 rows of W_hh that produce hidden units
//     [16*slice, 16*slice+16) x {i,f,g,o} RESIDENT IN REGISTERS for the whole layer (4 gate tiles x
//     H/32 k-steps of 16x16x32 f16 A-fragments), so the recurrent weights are read from HBM once.
//   * Per step a wave needs all H values of h_{t-1} for its 16 chunks (written by the other waves of
//     its ring, which live on other CUs) and produces 16 hidden units x 16 chunks of h_t.
//     The exchange buffer IS the layer output tensor h[T][N][H]: it is pre-filled with the fp16 bit
//     pattern 0xFFFF, producers write h_t with write-through (sc1) 8-byte stores and consumers poll
//     the data itself with L1-bypassing (sc1) 16-byte loads until no sentinel is left. |h| <= 1 so a
//     valid value never has bit 14 set: one OR-reduction + one mask test validates 8 halves. No
//     flags, no fences, no atomics, and every 2-byte element is checked individually so the protocol
//     does not depend on store granularity, dispatch order or workgroup->XCD placement.
//   * Rings are laid out so that the waves of one ring sit on one XCD when the dispatcher places
//     block b on XCD b % 8 (speed only; correctness never relies on it). At kernel start every wave
//     publishes its HW_REG_XCC_ID and the ring agrees on ONE of two store policies:
//       - all members on one XCD  -> plain stores (the line stays in that XCD's L2; the sc1 polls are
//         L2 hits, ~3x lower hand-off latency and no fabric traffic from polling);
//       - anything else (or the agreement times out) -> write-through sc1 stores, valid for any placement.
//     Every member reads the same published ids, so the choice is identical across the ring.
//   * Polling re-issues only the k-steps that still contain a sentinel (wave-uniform pending mask).
//   * W as the MFMA A operand: accumulator rows are hidden units, columns are chunks, so one lane
//     holds i,f,g,o for 4 consecutive hidden units of one chunk -> lane-local gate math, c_t kept in
//     fp32 registers for the whole layer, 8-byte packed h stores.
// All spins are bounded; on timeout the kernel raises *err_flag and keeps going (never hangs).
#include <string.h>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace bh {

struct LstmArgs {
    const half_t* G;    // [T][N][4H]  x W_ih^T + b_ih + b_hh, gate-major columns (torch order)
    const half_t* whh;  // packed fragments, see bh_pack_whh
    half_t* h;          // [T][N][H], pre-filled with 0xFFFF
    int T, N, H;
    int n_rings;        // N / 16
    int reverse;
    int* err;
    unsigned max_spins;
    int* xcc_ws;        // [n_rings][H/16], pre-set to -1: XCD agreement
    int force_slow;     // test hook: always use the placement-independent write-through policy
    int tune;           // experiment bits: 1 = spin without s_sleep, 2 = single-k-step canary poll before the full read
};

__device__ __forceinline__ int xcc_id() {
    return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xF);   // hwreg(HW_REG_XCC_ID, 0, 4)
}

// XCD agreement of a ring: every member publishes its XCC id and reads the ring's ids; the ring uses plain stores (the lines stay
// in that XCD's L2, the sc1 polls are L2 hits) iff all members sit on one XCD. A timeout is not an error: the write-through
// policy is valid for any placement. Every member reads the same published ids, so the choice is identical across the ring.
__device__ __forceinline__ bool ring_store_policy(const LstmArgs& p, int ring, int slice, int nsl, int lane) {
    int* slot = p.xcc_ws + (long)ring * nsl;
    const int mine = xcc_id();
    if (lane == 0) __hip_atomic_store(slot + slice, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    bool ok = false;
    while (true) {
        bool any_unset = false, any_other = false;
        for (int i = lane; i < nsl; i += 64) {
            const int v = __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            any_unset |= v < 0;
            any_other |= v != mine;
        }
        if (!__any(any_unset)) { ok = !__any(any_other); break; }
        if (++spins > p.max_spins) break;
        __builtin_amdgcn_s_sleep(4);
    }
    return ok && !p.force_slow;
}

constexpr unsigned SENTINEL_MASK = 0x40004000u;

// LDS output transpose: 2-byte writes, 8-byte reads of the same bytes. Plain half_t / unsigned long long accesses are
// different types for the compiler's alias analysis and were reordered (a wave then published stale LDS contents, which may
// look like the exchange sentinel): both sides go through may_alias types.
typedef unsigned short __attribute__((may_alias)) u16_alias_t;
typedef unsigned long long __attribute__((may_alias)) u64_alias_t;

// Gate-math tanh: the exp form has absolute error ~1e-7 everywhere, which is all the recurrence needs
// (h and c are consumed at fp16 / additive precision); the relative-accuracy branch of common.h's tanhf_ is skipped.
__device__ __forceinline__ float tanh_gate(float x) { return __builtin_fmaf(-2.0f, rcpf_(__expf(2.0f * x) + 1.0f), 1.0f); }

// One cell update from the four gate pre-activations (torch order i, f, g, o). The contractions are spelled out so
// that every kernel variant of this file produces the same bits for the same pre-activations; the returned h is
// forced into [-1, 1] so that a non-finite value can never alias the exchange sentinel.
__device__ __forceinline__ float lstm_cell(float ai, float af, float ag, float ao, float& c) {
    // Transcendental issue (quarter rate) is what the gate arithmetic costs, so the five activations share
    // reciprocals: with E_x = exp(-x),
    //     c' = c * sig(f) + sig(i) * tanh(g) = (c * Di * Dg + (1 - Eg2) * Df) / (Df * Di * Dg),   D_x = 1 + E_x, Eg2 = exp(-2g)
    //     h  = sig(o) * tanh(c')             = (1 - Ec2) / ((1 + Ec2) * Do)
    // i.e. 5 exp + 2 rcp instead of 5 + 5. Pre-activations are clamped to +-25 (sig(-25) = 1.4e-11, far below what h
    // can resolve) so that no product of three (1 + E) factors leaves the fp32 range.
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

// STREAM = false: W_hh fragments of the wave's slice are register-resident, a workgroup serves one slice of FOUR rings.
// STREAM = true (hidden sizes the register file cannot hold, H > 512: the 768-wide old-style r9.4.1 models, the 1024-wide v4.3
// `sup`, when the stationary wide kernel is switched off or does not cover H): the fragments are re-read from L2 / Infinity Cache
// every step, a workgroup serves four slices of ONE ring, so a ring costs only NSL/4 workgroups and any H % 64 == 0 fits.
template <int NKS, bool STREAM>
__global__ __launch_bounds__(256, 1) void lstm_layer_kernel(LstmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int H = NKS * 32;
    constexpr int NSL = H / 16;  // slices == waves per ring
    // block -> (xcd, ring group, slice) and wave -> ring inside the group; streaming: block -> (xcd, ring, group of 4 slices)
    const int xcd = blockIdx.x & 7;
    const int lwg = blockIdx.x >> 3;
    constexpr int WPR = NSL / 4;          // streaming: workgroups per ring
    const int rg = STREAM ? lwg / WPR : lwg / NSL;
    const int slice = STREAM ? (lwg - rg * WPR) * 4 + wave : lwg - rg * NSL;
    const int ring = STREAM ? rg * 8 + xcd : (rg * 4 + wave) * 8 + xcd;
    if (ring >= p.n_rings) return;
    const half_t* wbase = p.whh + ((long)slice * 4 * NKS * 64 + lane) * 8;

    // ---- recurrent weights -> registers (once per layer) -------------------------------------
    half8_t w[STREAM ? 1 : 4][STREAM ? 1 : NKS];
    if constexpr (!STREAM) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) w[g][ks] = *(const half8_t*)(wbase + (long)(g * NKS + ks) * 512);
    }

    const int c = lane & 15, q = lane >> 4;
    const int n = ring * 16 + c;
    const int hu0 = slice * 16 + q * 4;
    const long row_bytes = (long)p.N * H * 2;           // one time step of h
    const unsigned voff = (unsigned)(((ring * 16 + c) * H + q * 8) * 2);
    float cst[4] = {0.f, 0.f, 0.f, 0.f};
    bool dead = false;

    const bool fast = ring_store_policy(p, ring, slice, NSL, lane);

    const long g_row = (long)p.N * 4 * H;
    const half_t* gptr = p.G + (long)n * 4 * H + hu0;
    int t = p.reverse ? p.T - 1 : 0;
    const int dt = p.reverse ? -1 : 1;

    half4_t gin[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) gin[g] = *(const half4_t*)(gptr + (long)t * g_row + g * H);

    for (int step = 0; step < p.T; ++step, t += dt) {
        // prefetch next step's input projection (independent of the recurrence)
        half4_t gnx[4];
        {
            int tn = (step + 1 < p.T) ? t + dt : t;
#pragma unroll
            for (int g = 0; g < 4; ++g) gnx[g] = *(const half4_t*)(gptr + (long)tn * g_row + g * H);
        }
        float4_t acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = float4_t{0.f, 0.f, 0.f, 0.f};

        if (step > 0) {
            const char* base = (const char*)p.h + (long)(t - dt) * row_bytes;
            __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)row_bytes, 0x00020000);
            uint4_t hf[NKS];
            unsigned spins = dead ? p.max_spins : 0u;   // after one timeout never wait again
            unsigned pend = (NKS >= 32) ? 0xffffffffu : ((1u << NKS) - 1u);   // wave-uniform
            while (true) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    if (pend & (1u << ks))
                        hf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + ks * 64, 0, (int)0x80000010 /*sc1 + volatile*/);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    if (pend & (1u << ks)) {
                        unsigned orv = hf[ks].x | hf[ks].y | hf[ks].z | hf[ks].w;
                        if (!__any((orv & SENTINEL_MASK) != 0)) pend &= ~(1u << ks);
                    }
                if (pend == 0) break;
                if (++spins > p.max_spins) {
                    if (lane == 0 && !dead) atomicExch(p.err, 1);
                    dead = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                half8_t b = __builtin_bit_cast(half8_t, hf[ks]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (STREAM) acc[g] = mfma16(*(const half8_t*)(wbase + (long)(g * NKS + ks) * 512), b, acc[g]);
                    else acc[g] = mfma16(w[g][ks], b, acc[g]);
                }
            }
        }

        half4_t ho;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // a non-finite or out-of-range h can never be published (keeps the sentinel space clean)
            const float hv = lstm_cell(acc[0][i] + (float)gin[0][i], acc[1][i] + (float)gin[1][i],
                                       acc[2][i] + (float)gin[2][i], acc[3][i] + (float)gin[3][i], cst[i]);
            ho[i] = (half_t)hv;
        }
        unsigned long long packed = __builtin_bit_cast(unsigned long long, ho);
        unsigned long long* dst =
            (unsigned long long*)(p.h + ((long)t * p.N + n) * H + hu0);
        if (fast) *dst = packed;                                                       // stays in this XCD's L2
        else __hip_atomic_store(dst, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 write-through
#pragma unroll
        for (int g = 0; g < 4; ++g) gin[g] = gnx[g];
    }
}

// ---------------------------------------------------------------------------------------------------
// Fused variant: the input projection x_t W_ih^T + b is computed INSIDE the recurrence instead of by a
// separate GEMM. x_t (the previous layer's output, available for all t up front) is prefetched one step
// ahead and its 4 x NKS MFMAs run while the wave would otherwise be waiting for h_{t-1} from the other
// CUs, so they are free; the G tensor (T*N*4H fp16 = 2.6 GB per hac layer) is never written or read and
// one GEMM launch per layer disappears. W_ih's fragments for this workgroup's 16 hidden units live in
// LDS (4*NKS KiB, shared by the 4 waves of the workgroup, which serve 4 different rings of the SAME slice);
// W_hh stays in registers. Requires input size == hidden size (true for every LSTM layer of the CRF models).
struct LstmFusedArgs {
    const half_t* x;     // [T][N][H] layer input (time-major)
    const half_t* wih;   // packed fragments like whh
    const float* bias;   // [4H] b_ih + b_hh
    LstmArgs a;          // G unused
};

// MFMA whose A operand (a stationary W_hh fragment) lives in the accumulation-register half of the unified
// register file: the 48 fragments of a slice (192 registers at H=384) would otherwise crowd the 256 architected
// VGPRs and serialise every LDS read behind its consumer. The compiler does not track MFMA hazards through
// inline asm: callers keep >= 19 wait states between the last mfma16_a() and the first VALU read of `c`
// (mfma_settle()) and never issue one directly after a VALU write of its operands.
__device__ __forceinline__ void mfma16_a(const half8_t& a_agpr, const half8_t& b, float4_t& c) {
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "a"(a_agpr), "v"(b));
}
__device__ __forceinline__ void mfma_settle(float4_t& c0, float4_t& c1, float4_t& c2, float4_t& c3) {
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
}

template <int NKS>
__global__ __launch_bounds__(256, 1) void lstm_layer_fused_kernel(LstmFusedArgs fp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LstmArgs& p = fp.a;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int H = NKS * 32;
    constexpr int NSL = H / 16;
    const int xcd = blockIdx.x & 7;
    const int lwg = blockIdx.x >> 3;
    const int rg = lwg / NSL;
    const int slice = lwg - rg * NSL;
    const int ring = (rg * 4 + wave) * 8 + xcd;

    // ---- W_ih fragments of this slice -> LDS (all 256 threads, before any wave may leave) -------------
    {
        const uint4_t* src = (const uint4_t*)(fp.wih + (long)slice * 4 * NKS * 64 * 8);
        uint4_t* dst = (uint4_t*)smem;
        for (int i = threadIdx.x; i < 4 * NKS * 64; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    if (ring >= p.n_rings) return;

    half8_t w[4][NKS];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            w[g][ks] = *(const half8_t*)(p.whh + ((((long)slice * 4 + g) * NKS + ks) * 64 + lane) * 8);

    const int c = lane & 15, q = lane >> 4;
    const int n = ring * 16 + c;
    const int hu0 = slice * 16 + q * 4;
    const long row_bytes = (long)p.N * H * 2;
    const unsigned voff = (unsigned)(((ring * 16 + c) * H + q * 8) * 2);
    float cst[4] = {0.f, 0.f, 0.f, 0.f};
    bool dead = false;
    float4_t bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[g][i] = fp.bias[g * H + hu0 + i];

    const bool fast = ring_store_policy(p, ring, slice, NSL, lane);

    int t = p.reverse ? p.T - 1 : 0;
    const int dt = p.reverse ? -1 : 1;
    const half_t* xptr = fp.x + ((long)(ring * 16 + c) * H + q * 8);
    const long x_row = (long)p.N * H;
    const char* wl = smem + lane * 16;

    uint4_t xf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) xf[ks] = *(const uint4_t*)(xptr + (long)t * x_row + ks * 32);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(xf[ks]));      // landed (see the step loop)
    long long st_poll = 0, st_rounds = 0, st_first_ok = 0, st_x = 0, st_re = 0, st_rec = 0, st_hist = 0;
    const long long st_t0 = __builtin_readcyclecounter();

    for (int step = 0; step < p.T; ++step, t += dt) {
        float4_t acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = bias4[g];
        uint4_t hf[NKS];
        // ---- 1. ask for h_{t-1} (first poll round): the x_t loads of the previous step are the only thing ahead
        //         of these in the memory queue, and they have had a whole recurrent phase to land -----------------
        const char* base = (const char*)p.h + (long)(t - dt) * row_bytes;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(step > 0 ? base : (const char*)p.h), 0,
                                                                      (int)row_bytes, 0x00020000);
        const long long pc0 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        if (step > 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                hf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + ks * 64, 0, (int)0x80000010);
        }
        // ---- 2. input projection of this step while the exchange is in flight (W_ih fragments stream from LDS,
        //         one k-step ahead of the MFMAs that consume them) -----------------------------------------------
        {
            half8_t a_cur[4], a_nxt[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) a_cur[g] = *(const half8_t*)(wl + (g * NKS) * 1024);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks + 1 < NKS) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) a_nxt[g] = *(const half8_t*)(wl + (g * NKS + ks + 1) * 1024);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                }
                const half8_t xb = __builtin_bit_cast(half8_t, xf[ks]);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = mfma16(a_cur[g], xb, acc[g]);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) a_cur[g] = a_nxt[g];
            }
        }
        const long long pc1 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        // ---- 3. h_{t-1} must be complete: check the first round, re-poll only the k-steps still holding a
        //         sentinel ---------------------------------------------------------------------------------------
        if (step > 0) {
            unsigned spins = dead ? p.max_spins : 0u;
            unsigned pend = 0;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                unsigned orv = hf[ks].x | hf[ks].y | hf[ks].z | hf[ks].w;
                if (__any((orv & SENTINEL_MASK) != 0)) pend |= (1u << ks);
            }
            unsigned rounds = 1;
            const long long pc2 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
            while (pend != 0) {
                if (++spins > p.max_spins) {
                    if (lane == 0 && !dead) atomicExch(p.err, 1);
                    dead = true;
                    break;
                }
                if (!(p.tune & 1)) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    if (pend & (1u << ks))
                        hf[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + ks * 64, 0, (int)0x80000010);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    if (pend & (1u << ks)) {
                        unsigned orv = hf[ks].x | hf[ks].y | hf[ks].z | hf[ks].w;
                        if (!__any((orv & SENTINEL_MASK) != 0)) pend &= ~(1u << ks);
                    }
                ++rounds;
            }
            if (p.tune & 4) {
                const long long now = __builtin_readcyclecounter();
                st_poll += now - pc0; st_rounds += rounds; st_first_ok += (rounds == 1);
                st_x += pc1 - pc0; st_re += now - pc2;
                st_hist += 1ll << (16 * (rounds > 3 ? 3 : rounds - 1));     // 4 x 16-bit histogram: 1,2,3,>=4 rounds
            }
        }
        const long long pc3 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        // ---- 4. fetch x of the next step (behind the poll loads in the queue; lands during the recurrent phase) --
        {
            const int tn = (step + 1 < p.T) ? t + dt : t;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) xf[ks] = *(const uint4_t*)(xptr + (long)tn * x_row + ks * 32);
        }
        // ---- 5. recurrent part ------------------------------------------------------------------------------
        if (step > 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                half8_t b = __builtin_bit_cast(half8_t, hf[ks]);
#pragma unroll
                for (int g = 0; g < 4; ++g) mfma16_a(w[g][ks], b, acc[g]);
            }
            mfma_settle(acc[0], acc[1], acc[2], acc[3]);
        }
        half4_t ho;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float hv = lstm_cell(acc[0][i], acc[1][i], acc[2][i], acc[3][i], cst[i]);
            ho[i] = (half_t)hv;
        }
        // x_{t+1} (requested before the recurrent phase) must have landed BEFORE h_t is published: with nothing but
        // the h_t store ahead of them, the next step's poll loads are the only entries of the in-order memory
        // counter the input-projection has to look at.
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(xf[ks]));
        unsigned long long packed = __builtin_bit_cast(unsigned long long, ho);
        unsigned long long* dst = (unsigned long long*)(p.h + ((long)t * p.N + n) * H + hu0);
        if (fast) *dst = packed;
        else __hip_atomic_store(dst, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.tune & 4) st_rec += __builtin_readcyclecounter() - pc3;
    }
    if ((p.tune & 4) && lane == 0) {
        long long* st = (long long*)((char*)p.xcc_ws + (((size_t)p.n_rings * NSL * sizeof(int) + 64 + 7) & ~(size_t)7)) + ((long)ring * NSL + slice) * 8;
        st[0] = __builtin_readcyclecounter() - st_t0;
        st[1] = st_poll;
        st[2] = st_rounds;
        st[3] = st_first_ok;
        st[4] = st_x;
        st[5] = st_re;
        st[6] = st_rec;
        st[7] = st_hist;
    }
}

// ---------------------------------------------------------------------------------------------------
// Workgroup-shared kernels (lstm_layer_wgx_kernel / lstm_layer_wgx2_kernel below): the four waves of a workgroup are four SLICES OF
// THE SAME RING. (Round 1's lstm_layer_wg_kernel - this structure with the hand-off through the sentinel-filled output tensor - was
// removed in round 4: superseded by the ring-buffer exchange since round 2; lstm_layer_fused_kernel is the bit-identity reference that
// still exchanges through the output tensor, "lstm_exchange" 0 selects it.)
// Every wave owns U = 4*MT hidden units (MT M-tiles whose 16 rows are 4 units x 4 gates, so a lane ends up with all
// four gate pre-activations of MT units), holds BOTH its W_hh and W_ih fragments in registers (H=384, U=12:
// 2*144 registers of the 512-entry unified file) and the workgroup shares h_{t-1} and x_{t+1} through LDS:
// each wave polls / fetches only a QUARTER of the k-steps from global memory and deposits it as ready-made B
// fragments, one workgroup barrier per step publishes them. Compared with lstm_layer_fused_kernel this cuts the
// L2 read traffic of the exchange and of the x stream by 4x (it was 768 waves x 12 KB x ~2.5 per step at hac
// size, i.e. L2-bandwidth-bound polling), needs no LDS for weights and keeps all 256 CUs busy at N=512, H=384.
// XCD agreement and bounded spins as in lstm_layer_fused_kernel.
__device__ __forceinline__ void mfma16_av(const half8_t& a_agpr, const half8_t& b, float4_t& c) {
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a_agpr), "v"(b));
}
__device__ __forceinline__ void mfma16_vv(const half8_t& a, const half8_t& b, float4_t& c) {
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int MT>
__device__ __forceinline__ void mfma_settle_v(float4_t (&c)[MT]) {
    if constexpr (MT == 3) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]));
    else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
}

// ---------------------------------------------------------------------------------------------------
// "wgx": the workgroup-shared structure above with the exchange and the x stream of the 8-bit kernel (lstm_q8.hip). Same
// arithmetic, same accumulation order, same lstm_cell() -> the same bytes as every other fp16 variant (tested); what changes is
// where the bytes travel:
//   * the hand-off no longer goes through the layer's output tensor. h_t is published into a small RING BUFFER of four time
//     slots per ring, laid out as ready-made MFMA B fragments ([k-step][lane][16 bytes]: a consumer's poll of a k-step is one
//     fully coalesced KiB, 8 whole cache lines instead of 16 half lines of a row-major tensor). 4 x 12 KiB per ring, 1.5 MiB for
//     a 512-chunk batch: it lives in the L2. Sentinel = fp16 0xFFFF as before (|h| <= 1 never has bit 14 set). A producer
//     re-arms its own bytes of slot (t+2)%4 at step t, behind the workgroup barrier that proves every wave of the ring has
//     published h_{t-1} (hence finished reading h_{t-2}); its vmcnt(0) ahead of the next barrier completes the re-arm before it
//     publishes anything newer, so no consumer can find stale bytes. Consequences: no sentinel pre-fill of the output tensor
//     (the fill kernel, 0.66 GB of writes per layer, is gone), no first-touch fetch of 0.66 GB of polled sentinel lines from
//     HBM per layer, no third activation buffer;
//   * the layer output is a second, plain 8-byte store of the same values into [T][N][H] (next layer's input / the linear layer);
//   * x_{t+2} is fetched with LDS-DMA (global_load_lds_dwordx4: lane l -> LDS base + 16 l, which is the B-fragment order) by the
//     four waves in shares, straight into a three-slot LDS ring: no registers, no compiler-placed waits on the x stream.
// (round 2's cells3_mfma.inc - the gate arithmetic woven into the input-projection MFMAs only - is no longer part of the library; it is
// the baseline variant of tools/stream_bench.hip)
// The whole arithmetic of a ring step as one stream (round 3): recurrent MFMAs TILE-major, so that the gate arithmetic of a tile has
// the next tile's recurrent MFMAs (and a share of the next step's input projection) to hide behind; generated by
// tools/gen_ringstep.py, statement cuts audited on the compiled ISA by tools/audit_ringstep.py (tests/test_abi.py).
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
#include "ringstep3_mfma.inc"
// The same stream for the two-ring section of the fast path, carrying the section's vector-memory work as well (gen_ringstep.py
// --polls-at --xdma-at --validate-at): the OTHER ring's three poll DMAs, this ring's three x-stream DMAs, and near the end the
// read-back of my quarter of the other ring's h tile (`bad`: lanes that still found the exchange sentinel).
#include "ringstep3p_mfma.inc"
template <int NKS, int MT>
__device__ __forceinline__ void ring_stream(float4_t (&xacc)[MT], float (&cst)[MT], float (&hv)[MT], const half8_t (&whh)[MT][NKS],
                                            const half8_t (&wih)[MT][NKS], const float4_t (&bias)[MT], unsigned hb, unsigned xb) {
    if constexpr (NKS == 12 && MT == 3) ringstep3_mfma(xacc, cst, hv, whh, wih, bias, hb, xb);      // the only generated width so far
}
template <int NKS, int MT, int KQ>
__device__ __forceinline__ void ring_stream_paired(float4_t (&xacc)[MT], float (&cst)[MT], float (&hv)[MT], const half8_t (&whh)[MT][NKS],
                                                   const half8_t (&wih)[MT][NKS], const float4_t (&bias)[MT], unsigned hb, unsigned xb,
                                                   unsigned long long& bad, unsigned pm0, const char* exo, const unsigned (&vp)[KQ],
                                                   unsigned xm0, const char* xsrc, unsigned vx, unsigned hbo, unsigned sga, unsigned rda,
                                                   unsigned vmy, unsigned vh, const char* exs, const char* exa, const char* hrow, unsigned fast) {
    if constexpr (NKS == 12 && MT == 3)
        ringstep3p_mfma(xacc, cst, hv, whh, wih, bias, hb, xb, bad, pm0, exo, vp, xm0, xsrc, vx, hbo, sga, rda, vmy, vh, exs, exa, hrow, fast);
}
// The paired stream once more for the main loop of the fast path, which is unrolled over four steps (exchange slot = step & 3, tile
// parity = step & 1): which h tile / x slot an instruction addresses is a template constant added to a loop-invariant base
// (gen_ringstep.py --preset unrolled; the same instructions in the same order otherwise)
#include "ringstep3u_mfma.inc"
template <int NKS, int MT, int KQ, int HOFF, int XOFF, int VOFF, int PMO, int XMO>
__device__ __forceinline__ void ring_stream_unrolled(float4_t (&xacc)[MT], float (&cst)[MT], float (&hv)[MT], const half8_t (&whh)[MT][NKS],
                                                     const half8_t (&wih)[MT][NKS], const float4_t (&bias)[MT], unsigned hb, unsigned xb,
                                                     unsigned long long& bad, unsigned pm0, const char* exo, const unsigned (&vp)[KQ],
                                                     unsigned xm0, const char* xsrc, unsigned vx, unsigned hbo, unsigned sga, unsigned rda,
                                                     unsigned vmy, unsigned vh, const char* exs, const char* exa, const char* hrow, unsigned fast) {
    if constexpr (NKS == 12 && MT == 3)
        ringstep3u_mfma<HOFF, XOFF, VOFF, PMO, XMO>(xacc, cst, hv, whh, wih, bias, hb, xb, bad, pm0, exo, vp, xm0, xsrc, vx, hbo, sga, rda, vmy, vh, exs,
                                                    exa, hrow, fast);
}
// The recurrent half alone (36 MFMAs tile-major + the gate arithmetic) for the single-ring kernel, whose input projection runs behind
// the publish: gen_ringstep.py --preset rec
#include "ringstep3r_mfma.inc"
template <int NKS, int MT>
__device__ __forceinline__ void ring_stream_rec(float4_t (&xacc)[MT], float (&cst)[MT], float (&hv)[MT], const half8_t (&whh)[MT][NKS],
                                                const half8_t (&wih)[MT][NKS], const float4_t (&bias)[MT], unsigned hb) {
    if constexpr (NKS == 12 && MT == 3) ringstep3r_mfma(xacc, cst, hv, whh, wih, bias, hb, 0u);
}
__device__ __forceinline__ void dma16_wgx(const char* g, char* lds) {
    const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory");
}

struct LstmWgxArgs {
    LstmFusedArgs f;
    char* ex;             // exchange ring buffer [4][R][NKS][64][16], armed with 0xFF
    int R;                // ring stride of `ex` (rings of the whole batch)
};

// Vector-memory operations of the H = 384 fast path with SCALAR addressing: a wave-uniform 64-bit base in an SGPR pair plus a 32-bit
// per-lane byte offset that is loop-invariant (the compiler's per-lane 64-bit address arithmetic, the generic -> LDS pointer
// conversions in front of every M0 write and the spilled scalars behind them were ~55 instructions for the three x-stream DMAs of a
// ring step alone). IMM: immediate byte offset (13-bit signed) - keep it 0 for LDS-DMA: the hardware adds the instruction offset to
// the LDS address as well as to the global one. The DMA lands at LDS address lds_a + lane * 16.
template <int IMM, bool POLL>
__device__ __forceinline__ void dma16_s(const char* sbase, unsigned voff, unsigned lds_a) {
    if constexpr (POLL)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3 sc0 sc1" ::"s"(lds_a), "v"(voff), "s"(sbase), "i"(IMM) : "memory");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(lds_a), "v"(voff), "s"(sbase), "i"(IMM) : "memory");
}
template <bool SC1>
__device__ __forceinline__ void store8_s(const char* sbase, unsigned voff, unsigned long long v) {
    if constexpr (SC1) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}

template <int NKS, int MT>
__global__ __launch_bounds__(256, 1) void lstm_layer_wgx_kernel(LstmWgxArgs wp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LstmFusedArgs& fp = wp.f;
    const LstmArgs& p = fp.a;
    constexpr int H = NKS * 32, U = 4 * MT, NSL = H / U, WPR = NSL / 4, KQ = (NKS + 3) / 4, TILE = NKS * 1024;
    constexpr bool EXACT = NKS % 4 == 0;
    static_assert(H % (4 * U) == 0, "four slices per workgroup");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7;
    const int lwg = blockIdx.x >> 3;
    const int rl = lwg / WPR;
    // test hook ("lstm_tune" bit 5): spread the workgroups of every ring over all eight XCDs, so that the placement-independent
    // (write-through) hand-off really crosses XCDs
    const int ring = rl * 8 + ((p.tune & 32) ? ((xcd + (lwg - rl * WPR)) & 7) : xcd);
    const int slice = (lwg - rl * WPR) * 4 + wave;
    if (ring >= p.n_rings) return;

    char* hbuf = smem;                                  // [2][NKS][64][16]  B fragments of h_{t-1}
    char* xbuf = smem + 2 * TILE;                       // [3][NKS][64][16]  B fragments of x_t: consumed / landed / landing
    char* stage = smem + 5 * TILE + wave * (16 * U * 2);

    half8_t whh[MT][NKS], wih[MT][NKS];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const long o = ((((long)slice * MT + m) * NKS + ks) * 64 + lane) * 8;
            whh[m][ks] = *(const half8_t*)(p.whh + o);
            wih[m][ks] = *(const half8_t*)(fp.wih + o);
        }
    const int c = lane & 15, q = lane >> 4;
    float cst[MT];
    float4_t bias4[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        cst[m] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[m][i] = fp.bias[i * H + slice * U + q * MT + m];
    }
    bool dead = false;
    const bool fast = ring_store_policy(p, ring, slice, NSL, lane);

    int t = p.reverse ? p.T - 1 : 0;
    const int dt = p.reverse ? -1 : 1;
    // x_t rows are row-major [T][N][H]; lane (c, q) of k-step ks needs halves [ks*32 + q*8, +8) of chunk ring*16 + c
    const half_t* xptr = fp.x + ((long)(ring * 16 + c) * H + q * 8);
    const long x_row = (long)p.N * H;
    const int lo = lane * 16;
    const long slot_stride = (long)wp.R * TILE;
    char* exr = wp.ex + (long)ring * TILE;

    // lanes that move this wave's U units out: (chunk cc, 4 consecutive units = 8 bytes)
    constexpr int PARTS = U / 4;
    const bool mover = lane < 16 * PARTS;
    const int cc = lane / PARTS, part = lane - cc * PARTS;
    const int u0 = slice * U + part * 4;
    const int my_byte = (((u0 >> 5) * 64 + ((u0 >> 3) & 3) * 16 + cc) << 4) + (u0 & 7) * 2;      // inside a ring tile (fragment order)

    uint4_t hq[KQ];
    float4_t xacc[MT];
    auto x_phase = [&](const char* xb) {
#pragma unroll
        for (int m = 0; m < MT; ++m) xacc[m] = bias4[m];
        half8_t b_cur = *(const half8_t*)(xb + lo), b_nxt = b_cur;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 1 < NKS) b_nxt = *(const half8_t*)(xb + (ks + 1) * 1024 + lo);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m < MT - 1) mfma16_av(wih[m][ks], b_cur, xacc[m]);
                else mfma16_vv(wih[m][ks], b_cur, xacc[m]);
            }
            b_cur = b_nxt;
        }
        mfma_settle_v<MT>(xacc);
    };
    // x stream: the polls of the exchange go to k-steps w, w+4, ..; the DMA shares are 3-w, 7-w, .. (balanced over the waves).
    // H = 384: scalar-addressed (a wave-uniform row base + a loop-invariant 32-bit per-lane offset, LDS addresses as integers) - the
    // per-lane 64-bit address arithmetic and the generic -> LDS pointer conversions were ~18 instructions per DMA
    const unsigned smem_a = lds_addr(smem);
    const unsigned vx_off = (unsigned)(((ring * 16 + c) * H + q * 8) * 2 + (3 - wave) * 64);
    const unsigned vh_off = (unsigned)(((ring * 16 + cc) * H + slice * U + part * 4) * 2);
    const long row_bytes = x_row * 2;
    auto x_dma = [&](int tt, int slot) {
        if constexpr (NKS == 12 && MT == 3) {
            const char* src = (const char*)fp.x + (long)tt * row_bytes;
            const unsigned m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_a + (2 + slot) * TILE + (3 - wave) * 1024));
            dma16_s<0, false>(src, vx_off, m0);
            dma16_s<0, false>(src + 256, vx_off, m0 + 4096);
            dma16_s<0, false>(src + 512, vx_off, m0 + 8192);
        } else {
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int ks = (3 - wave) + 4 * kk;
                if (EXACT || ks < NKS) dma16_wgx((const char*)(xptr + (long)tt * x_row + ks * 32), xbuf + (slot * NKS + ks) * 1024);
            }
        }
    };

    x_dma(t, 0);
    x_dma(p.T > 1 ? t + dt : t, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the builtin (unlike inline asm) also clears the compiler's own scoreboard
    __syncthreads();
    x_phase(xbuf);
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) hq[kk] = uint4_t{0, 0, 0, 0};

    long long st_poll = 0, st_rounds = 0, st_first_ok = 0, st_x = 0, st_rec = 0, st_bar = 0, st_hist = 0;
    long long st_mf = 0, st_gate = 0, st_store = 0;
    const long long st_t0 = __builtin_readcyclecounter();
    const long long st_r0 = (p.tune & 4) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;

    for (int step = 0; step < p.T; ++step, t += dt) {
        const int par = step & 1;
        float4_t acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = xacc[m];
        // ---- B. my quarter of h_{t-1}: round one went out right after the previous store --------------------------------
        const long long pc0 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        if (step > 0) {
            // wave-uniform descriptor, per-lane offset (a per-lane base would turn every load into a 64-trip readfirstlane loop)
            const char* src = exr + (long)((step - 1) & 3) * slot_stride;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, TILE, 0x00020000);
            unsigned spins = dead ? p.max_spins : 0u;
            unsigned pend = 0;
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int ks = wave + 4 * kk;
                if (EXACT || ks < NKS) {
                    unsigned orv = hq[kk].x | hq[kk].y | hq[kk].z | hq[kk].w;
                    if (__any((orv & SENTINEL_MASK) != 0)) pend |= (1u << kk);
                }
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
                    if (pend & (1u << kk)) {
                        unsigned orv = hq[kk].x | hq[kk].y | hq[kk].z | hq[kk].w;
                        if (!__any((orv & SENTINEL_MASK) != 0)) pend &= ~(1u << kk);
                    }
                ++rounds;
            }
            if (p.tune & 4) {
                st_poll += __builtin_readcyclecounter() - pc0; st_rounds += rounds; st_first_ok += (rounds == 1);
                st_hist += 1ll << (16 * (rounds > 3 ? 3 : rounds - 1));
            }
        }
        constexpr bool STREAM = NKS == 12 && MT == 3;      // H = 384: the recurrent MFMAs and the gate arithmetic are one generated stream
        if (step > 0 || STREAM) {                          // (step 0 of the stream multiplies a zero tile: hq starts as zeros)
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int ks = wave + 4 * kk;
                if (EXACT || ks < NKS) *(uint4_t*)(hbuf + (par * NKS + ks) * 1024 + lo) = hq[kk];
            }
        }
        // ---- D. publish h_{t-1} and x_{t+1} (DMA issued a step ago) to the workgroup; the vmcnt(0) covers this wave's DMA share
        //         and completes its re-arm store of the previous step before it publishes anything newer ----------------------
        const long long pc1 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the builtin (unlike inline asm) also clears the compiler's own scoreboard
        __syncthreads();
        const long long pc2 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        half_t ho[MT];
        long long pcm = 0;
        if constexpr (STREAM) {
            // ---- X / E. x_{t+2} requested first, then the recurrent MFMAs tile by tile with the gate arithmetic of tile m behind the
            //      MFMAs of tile m + 1 (ringstep3r_mfma, round 3: the bare MFMA phase + the bare gate phase were 1.5 k cycles) ---------
            if (step + 2 < p.T) x_dma(t + 2 * dt, (step + 2) % 3);
            float hv[MT];
            ring_stream_rec<NKS, MT>(acc, cst, hv, whh, wih, bias4, lds_addr(hbuf + par * TILE + lo));
#pragma unroll
            for (int m = 0; m < MT; ++m) ho[m] = (half_t)hv[m];
            pcm = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        } else {
            // ---- E. recurrent part ------------------------------------------------------------------------------------------
            if (step > 0) {
                const char* hb = hbuf + par * TILE + lo;
                half8_t hb_f[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) hb_f[ks] = *(const half8_t*)(hb + ks * 1024);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int m = 0; m < MT; ++m) mfma16_av(whh[m][ks], hb_f[ks], acc[m]);
                mfma_settle_v<MT>(acc);
            }
            pcm = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
            // ---- X. request x_{t+2} into the LDS slot x_{t-1} was consumed from; needed behind the NEXT barrier ----------------
            if (step + 2 < p.T) x_dma(t + 2 * dt, (step + 2) % 3);
#pragma unroll
            for (int m = 0; m < MT; ++m) ho[m] = (half_t)lstm_cell(acc[m][0], acc[m][1], acc[m][2], acc[m][3], cst[m]);
        }
        const long long pcg = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        {
            u16_alias_t* sg = (u16_alias_t*)stage + c * U + q * MT;
#pragma unroll
            for (int m = 0; m < MT; ++m) sg[m] = __builtin_bit_cast(unsigned short, ho[m]);
            if (mover) {
                const unsigned long long packed = *(const u64_alias_t*)((half_t*)stage + cc * U + part * 4);
                if constexpr (STREAM) {                    // scalar-addressed stores (uniform slot base + my byte offset)
                    const char* exs = exr + (long)(step & 3) * slot_stride;
                    if (fast) store8_s<false>(exs, (unsigned)my_byte, packed); else store8_s<true>(exs, (unsigned)my_byte, packed);
                    if (step >= 2 && step + 2 < p.T) {     // re-arm my bytes of slot (step+2)&3 (it holds h_{t-2}; see the header)
                        const char* exa = exr + (long)((step + 2) & 3) * slot_stride;
                        if (fast) store8_s<false>(exa, (unsigned)my_byte, ~0ull); else store8_s<true>(exa, (unsigned)my_byte, ~0ull);
                    }
                    store8_s<false>((const char*)p.h + (long)t * row_bytes, vh_off, packed);
                } else {
                unsigned long long* dst = (unsigned long long*)(exr + (long)(step & 3) * slot_stride + my_byte);
                if (fast) *dst = packed;
                else __hip_atomic_store(dst, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // re-arm my bytes of slot (step+2)&3 (it holds h_{t-2}; see the header)
                if (step >= 2 && step + 2 < p.T) {
                    unsigned long long* ra = (unsigned long long*)(exr + (long)((step + 2) & 3) * slot_stride + my_byte);
                    if (fast) *ra = ~0ull;
                    else __hip_atomic_store(ra, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // the layer output proper (next layer's x rows / the linear layer's input)
                *(unsigned long long*)(p.h + ((long)t * p.N + ring * 16 + cc) * H + slice * U + part * 4) = packed;
                }
            }
        }
        // ---- F. first poll round for h_t ----------------------------------------------------------------------------------
        if (step + 1 < p.T) {
            const char* src = exr + (long)(step & 3) * slot_stride;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, TILE, 0x00020000);
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int ks = wave + 4 * kk;
                if (EXACT || ks < NKS) hq[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, ks * 1024 + lo, 0, (int)0x80000010);
            }
        }
        const long long pc3 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        // ---- G. input projection of step t+1 --------------------------------------------------------------------------------
        x_phase(xbuf + ((step + 1) % 3) * TILE);
        if (p.tune & 4) {
            const long long now = __builtin_readcyclecounter();
            st_bar += pc2 - pc1; st_rec += pc3 - pc2; st_x += now - pc3;
            st_mf += pcm - pc2; st_gate += pcg - pcm; st_store += pc3 - pcg;
        }
    }
    if ((p.tune & 4) && lane == 0) {
        long long* st = (long long*)((char*)p.xcc_ws + (((size_t)p.n_rings * NSL * sizeof(int) + 64 + 7) & ~(size_t)7)) + ((long)ring * NSL + slice) * 16;
        st[0] = __builtin_readcyclecounter() - st_t0;
        st[1] = st_poll; st[2] = st_rounds; st[3] = st_first_ok; st[4] = st_x; st[5] = st_bar; st[6] = st_rec; st[7] = st_hist;
        st[8] = 0; st[9] = 0; st[10] = st_mf; st[11] = st_gate; st[12] = st_store;
        st[13] = (long long)__builtin_amdgcn_s_memrealtime() - st_r0;
    }
}

// ---------------------------------------------------------------------------------------------------
// "wgx2": lstm_layer_wgx_kernel serving TWO rings per workgroup with ONE register-resident copy of the weights. A batch of
// more than 32 rings (N > 512 chunks at H = 384) does not fit the chip one ring per 8 workgroups; instead of a second launch
// the workgroups of ring p also carry ring p + n_pairs and alternate between them: while the hand-off of one ring is in flight
// (publish -> L2 -> poll) the wave runs the whole step of the other ring, so nothing waits for the exchange. Arithmetic,
// accumulation order and lstm_cell() are those of the single-ring kernel: same bytes.
//   * Nothing of the exchange lives in registers while in flight: the first poll round is an LDS-DMA (global_load_lds, sc0 sc1)
//     straight into the NEXT parity of the ring's h tile; the wave later reads its quarter back to validate it. (With polls into
//     VGPRs the compiler copies / re-uses the in-flight destination registers across the loop back edge and waits for them there.)
//   * Every wait on the vector-memory queue is explicit: the wave counts the operations it has issued behind a ring's polls (the
//     other ring's x-stream DMA, its three stores, its polls) and waits with exactly that vmcnt - the queue retires in order.
//   * One workgroup barrier per ring step. LDS: 2 rings x (2 h tiles + 2 x tiles): the x slot consumed by step s is refilled
//     (x_{s+2}) behind the ring's next barrier, two ring steps before it is needed.
__device__ __forceinline__ void dma16_poll(const char* g, char* lds) {
    const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 sc1" ::"s"(l), "v"(g) : "memory");
}
__device__ __forceinline__ void wait_vm(int n) {       // s_waitcnt vmcnt(n), n wave-uniform; a smaller count than necessary is always safe
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
        case 1: __builtin_amdgcn_s_waitcnt(0x0F71); break;
        case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break;
        case 3: __builtin_amdgcn_s_waitcnt(0x0F73); break;
        case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
        case 5: __builtin_amdgcn_s_waitcnt(0x0F75); break;
        case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break;
        case 7: __builtin_amdgcn_s_waitcnt(0x0F77); break;
        case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
        case 9: __builtin_amdgcn_s_waitcnt(0x0F79); break;
        case 10: __builtin_amdgcn_s_waitcnt(0x0F7A); break;
        case 11: __builtin_amdgcn_s_waitcnt(0x0F7B); break;
        default: __builtin_amdgcn_s_waitcnt(0x0F7C); break;     // 12
    }
    asm volatile("" ::: "memory");
}

template <int NKS, int MT, bool STATS = false>
__global__ __launch_bounds__(256, 1) void lstm_layer_wgx2_kernel(LstmWgxArgs wp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LstmFusedArgs& fp = wp.f;
    const LstmArgs& p = fp.a;
    constexpr int H = NKS * 32, U = 4 * MT, NSL = H / U, WPR = NSL / 4, KQ = (NKS + 3) / 4, TILE = NKS * 1024;
    constexpr bool EXACT = NKS % 4 == 0;
    static_assert(H % (4 * U) == 0, "four slices per workgroup");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7;
    const int lwg = blockIdx.x >> 3;
    const int rl = lwg / WPR;
    const int n_pairs = (p.n_rings + 1) >> 1;
    const int pair = rl * 8 + ((p.tune & 32) ? ((xcd + (lwg - rl * WPR)) & 7) : xcd);          // bit 5: test hook, see lstm_layer_wgx_kernel
    const int slice = (lwg - rl * WPR) * 4 + wave;
    if (pair >= n_pairs) return;
    const int ring_of[2] = {pair, pair + n_pairs};
    const bool two = ring_of[1] < p.n_rings;            // odd ring count: the last workgroups carry one ring (uniform over the ring)

    char* hbuf = smem;                                  // [2 rings][2 parities][NKS][64][16]  B fragments of h_{t-1}
    char* xbuf = smem + 4 * TILE;                       // [2 rings][2 slots][NKS][64][16]     B fragments of x_t
    char* stage = smem + 8 * TILE + wave * (16 * U * 2);

    half8_t whh[MT][NKS], wih[MT][NKS];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const long o = ((((long)slice * MT + m) * NKS + ks) * 64 + lane) * 8;
            whh[m][ks] = *(const half8_t*)(p.whh + o);
            wih[m][ks] = *(const half8_t*)(fp.wih + o);
        }
    const int c = lane & 15, q = lane >> 4;
    float4_t bias4[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[m][i] = fp.bias[i * H + slice * U + q * MT + m];
    bool dead = false;
    const int dt = p.reverse ? -1 : 1;
    const int t0 = p.reverse ? p.T - 1 : 0;
    const long x_row = (long)p.N * H;
    const int lo = lane * 16;
    const long slot_stride = (long)wp.R * TILE;
    constexpr int PARTS = U / 4;
    const int ml = lane < 16 * PARTS ? lane : lane - 16 * PARTS;      // lanes beyond the movers mirror one (see the stores)
    const int cc = ml / PARTS, part = ml - cc * PARTS;
    const int u0 = slice * U + part * 4;
    const int my_byte = (((u0 >> 5) * 64 + ((u0 >> 3) & 3) * 16 + cc) << 4) + (u0 & 7) * 2;      // inside a ring tile (fragment order)
    int n_poll = 0, n_dma = 0;                          // instructions of one poll round / one x-stream share of this wave
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
        n_poll += (EXACT || wave + 4 * kk < NKS) ? 1 : 0;
        n_dma += (EXACT || (3 - wave) + 4 * kk < NKS) ? 1 : 0;
    }

    // per-ring state
    float cst[2][MT];
    float4_t xacc[2][MT];
    bool fast[2];
    const half_t* xptr[2];
    char* exr[2];
    int since[2] = {0, 0};                              // vector-memory operations this wave has issued behind ring r's first poll round
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ring = (r == 0 || two) ? ring_of[r] : ring_of[0];
        fast[r] = (r == 0 || two) ? ring_store_policy(p, ring, slice, NSL, lane) : false;
        xptr[r] = fp.x + ((long)(ring * 16 + c) * H + q * 8);
        exr[r] = wp.ex + (long)ring * TILE;
#pragma unroll
        for (int m = 0; m < MT; ++m) cst[r][m] = 0.f;
    }

    // H = 384 fast path (two rings): loop-invariant per-lane byte offsets for the scalar-addressed vector-memory operations
    constexpr bool FASTPATH = NKS == 12 && MT == 3;
    const unsigned smem_a = lds_addr(smem);             // hbuf at +0, xbuf at +4 TILE
    unsigned vx_off[2], vh_off[2], vp_off[KQ];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ring = (r == 0 || two) ? ring_of[r] : ring_of[0];
        vx_off[r] = (unsigned)(((ring * 16 + c) * H + q * 8) * 2 + (3 - wave) * 64);       // x row of my chunk, my first k-step share
        vh_off[r] = (unsigned)(((ring * 16 + cc) * H + slice * U + part * 4) * 2);        // my 8 bytes of the layer output row
    }
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) vp_off[kk] = (unsigned)(lo + (wave + 4 * kk) * 1024);  // my k-steps inside an exchange slot
    const unsigned stage_a = lds_addr(stage);           // this wave's transpose row
    unsigned fast_u[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)              // (a scalar the compiler cannot fold back into a zero-extended bool, which it keeps in a VGPR)
        asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(fast_u[r]) : "s"(__builtin_amdgcn_ballot_w64(fast[r])) : "scc");
    const long row_bytes = x_row * 2;
    const char* xrow2 = (const char*)fp.x + (long)(p.T > 2 ? t0 + 2 * dt : t0) * row_bytes;   // row of x_{t+2} (uniform), advanced per step
    const char* xrow_any = (const char*)fp.x + (long)t0 * row_bytes;
    char* hrow = (char*)p.h + (long)t0 * row_bytes;                                          // row of h_t in the layer output

    auto x_phase = [&](const char* xb, float4_t (&xa)[MT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) xa[m] = bias4[m];
        half8_t b_cur = *(const half8_t*)(xb + lo), b_nxt = b_cur;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 1 < NKS) b_nxt = *(const half8_t*)(xb + (ks + 1) * 1024 + lo);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m < MT - 1) mfma16_av(wih[m][ks], b_cur, xa[m]);
                else mfma16_vv(wih[m][ks], b_cur, xa[m]);
            }
            b_cur = b_nxt;
        }
        mfma_settle_v<MT>(xa);
    };
    auto x_dma = [&](const half_t* xp, char* xb, int tt, int slot) {
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const int ks = (3 - wave) + 4 * kk;
            if (EXACT || ks < NKS) dma16_wgx((const char*)(xp + (long)tt * x_row + ks * 32), xb + (slot * NKS + ks) * 1024);
        }
    };
    // first poll round of ring r for the h published into exchange slot `slot`: my k-steps, straight into parity `par` of the h tile
    auto poll_dma = [&](const char* exbase, int slot, char* hb, unsigned mask) {
        const char* src = exbase + (long)slot * slot_stride + lo;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const int ks = wave + 4 * kk;
            if ((EXACT || ks < NKS) && (mask & (1u << kk))) dma16_poll(src + ks * 1024, hb + ks * 1024);
        }
    };

#pragma unroll
    for (int r = 0; r < 2; ++r)
        if (r == 0 || two) {
            x_dma(xptr[r], xbuf + r * 2 * TILE, t0, 0);
            x_dma(xptr[r], xbuf + r * 2 * TILE, p.T > 1 ? t0 + dt : t0, 1);
        }
    if constexpr (NKS == 12 && MT == 3) {              // h_{-1} = 0: step 0 runs the same stream as every other step, on a zero tile
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk)
                *(uint4_t*)(hbuf + r * 2 * TILE + (wave + 4 * kk) * 1024 + lo) = uint4_t{0u, 0u, 0u, 0u};
    }
    wait_vm(0);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r)
        if (r == 0 || two) x_phase(xbuf + r * 2 * TILE, xacc[r]);

    long long st_poll = 0, st_bar = 0, st_slow = 0, st_sec[6] = {0, 0, 0, 0, 0, 0};      // lstm_tune bit 2: cycles per section (tools/lstm_stats.py)
    const long long st_t0 = __builtin_readcyclecounter();
    const long long st_r0 = (p.tune & 4) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;

    // validation of this wave's quarter of ring rr's h tile for step `st` (published in exchange slot (st-1)&3, landing in parity
    // st&1): begin = wait for the first poll round + read it back, end = test it, re-poll what is still armed
    auto check_begin = [&](int rr, int st, uint4_t (&chk)[KQ]) {
        wait_vm(since[rr]);
        const char* hb = hbuf + (rr * 2 + (st & 1)) * TILE;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const int ks = wave + 4 * kk;
            if (EXACT || ks < NKS) chk[kk] = *(const uint4_t*)(hb + ks * 1024 + lo);
        }
    };
    auto check_end = [&](int rr, int st, uint4_t (&chk)[KQ]) {
        char* hb = hbuf + (rr * 2 + (st & 1)) * TILE;
        unsigned spins = dead ? p.max_spins : 0u;
        while (true) {
            unsigned pend = 0;
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int ks = wave + 4 * kk;
                if ((EXACT || ks < NKS) && __any(((chk[kk].x | chk[kk].y | chk[kk].z | chk[kk].w) & SENTINEL_MASK) != 0)) pend |= (1u << kk);
            }
            if (pend == 0) break;
            if (++spins > p.max_spins) {
                if (lane == 0 && !dead) atomicExch(p.err, 1);
                dead = true;
                break;
            }
            if (!(p.tune & 1)) __builtin_amdgcn_s_sleep(1);
            poll_dma(exr[rr], (st - 1) & 3, hb, pend);
            wait_vm(0);
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int ks = wave + 4 * kk;
                if (EXACT || ks < NKS) chk[kk] = *(const uint4_t*)(hb + ks * 1024 + lo);
            }
        }
    };

    // one time step of ring r (r and TWO are compile-time constants: the per-ring state stays in registers)
    auto ring_step = [&](auto rc, auto two_c, int step, int t) {
        constexpr int r = decltype(rc)::value;
        constexpr bool TWO = decltype(two_c)::value;
        if constexpr (FASTPATH && TWO) {
            // ---- H = 384, two rings: one barrier, one instruction stream, three stores. The stream (ringstep3p_mfma) carries the
            //      section's other vector-memory work at chosen points of its MFMA sequence: the OTHER ring's first poll round (not
            //      before its publishes are visible, ~0.45 k cycles behind them; early enough to be back by the end of the stream),
            //      this ring's x-stream DMAs behind them, and at the end - behind a literal vmcnt(3): only the x stream is younger than
            //      the polls - the read-back of my quarter of the other ring's h tile, OR-ed into `bad`. All of it scalar-addressed;
            //      the operation count per section is a constant (polls 3, x stream 3, stores 3) ---------------------------------------
            constexpr int o = r ^ 1;
            const int par = step & 1;
            const long long q0 = STATS ? __builtin_readcyclecounter() : 0;
            __syncthreads();                      // the h tile (all quarters validated in the other ring's section) and the x tile of step t+1 are complete
            const long long q1 = STATS ? __builtin_readcyclecounter() : 0;
            // Where there is nothing to poll (first / last section) the three DMAs land in the parity of the other ring's h tile that
            // nobody reads and the validation's verdict is ignored: the operation count stays constant
            const int step_o = r == 0 ? step : step + 1;
            const bool chk_o = step_o >= 1 && step_o < p.T;
            const char* exo = exr[o] + (long)((step_o - 1) & 3) * slot_stride;
            // (readfirstlane: the expression shares its terms with the per-lane address below and would otherwise be built in a VGPR)
            const unsigned pm0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_a + (o * 2 + ((step_o & 1) ^ (chk_o ? 0 : 1))) * TILE + wave * 1024));
            const unsigned hbo = smem_a + (o * 2 + (step_o & 1)) * TILE + wave * 1024 + lo;
            const char* xsrc = (step + 2 < p.T) ? xrow2 : xrow_any;          // past the end: a valid row, into a slot nobody reads
            const unsigned xm0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_a + (4 + r * 2 + par) * TILE + (3 - wave) * 1024));
            unsigned long long bad;
            float hv[MT];
            const char* exs = exr[r] + (long)(step & 3) * slot_stride;           // the exchange slot of h_t (uniform)
            const char* exa = exr[r] + (long)((step + 2) & 3) * slot_stride;     // the slot re-armed for h_{t+2}
            ring_stream_paired<NKS, MT, KQ>(xacc[r], cst[r], hv, whh, wih, bias4, smem_a + (r * 2 + par) * TILE + lo,
                                            smem_a + (4 + r * 2 + (par ^ 1)) * TILE + lo, bad, pm0, exo, vp_off, xm0, xsrc, vx_off[r], hbo,
                                            stage_a + (unsigned)((c * U + q * MT) * 2), stage_a + (unsigned)((cc * U + part * 4) * 2),
                                            (unsigned)my_byte, vh_off[r], exs, exa, hrow, (unsigned)__builtin_amdgcn_readfirstlane((int)fast_u[r]));
            const long long q3 = STATS ? __builtin_readcyclecounter() : 0;
            if (__builtin_expect(chk_o && (bad != 0 || dead), 0)) {              // some element had not arrived: re-poll it, bounded
                uint4_t chk[KQ];
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk) chk[kk] = *(const uint4_t*)(hbuf + (o * 2 + (step_o & 1)) * TILE + (wave + 4 * kk) * 1024 + lo);
                check_end(o, step_o, chk);
                if constexpr (STATS) { st_poll += __builtin_readcyclecounter() - q3; ++st_slow; }
            }
            const long long q4 = STATS ? __builtin_readcyclecounter() : 0;
            if constexpr (STATS) {
                const long long q5 = __builtin_readcyclecounter();
                st_bar += q1 - q0; st_sec[2] += q3 - q1; st_sec[4] += q4 - q3; st_sec[3] += q5 - q4;
            }
            return;
        }
        char* hb_r = hbuf + (r * 2 + (step & 1)) * TILE;
        char* xb_r = xbuf + r * 2 * TILE;
        float4_t acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = xacc[r][m];
        // ---- my quarter of h_{t-1}. Two rings: it was validated in the middle of the other ring's section (below); a lone ring
        //      does it here, with the round trip of the polls exposed ---------------------------------------------------------------
        const long long pc0 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        if constexpr (!TWO) {
            if (step > 0) {
                uint4_t chk[KQ];
                check_begin(r, step, chk);
                check_end(r, step, chk);
            } else {
                wait_vm(0);
            }
        }
        const long long pc1 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        __syncthreads();                         // the h tile (all quarters validated) and the x tile of this step are complete
        if constexpr (!TWO) __syncthreads();     // (a lone ring: keeps the sections of consecutive steps apart like the other ring's barrier would)
        const long long pc2 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        if (p.tune & 4) { st_poll += pc1 - pc0; st_bar += pc2 - pc1; }
        // ---- x_{t+2} into the slot whose x_t the input projection of the previous step consumed -------------------------------
        if (step + 2 < p.T) {
            x_dma(xptr[r], xb_r, t + 2 * dt, step & 1);
            since[0] += n_dma; since[1] += n_dma;
        }
        const long long pc3 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        constexpr bool WOVEN = NKS == 12 && MT == 3;
        // ---- recurrent part (H = 384: part of the stream below; step 0 multiplies the zero-filled h tile) ---------------------------
        if (!WOVEN && step > 0) {
            half8_t hb_f[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) hb_f[ks] = *(const half8_t*)(hb_r + lo + ks * 1024);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int m = 0; m < MT; ++m) mfma16_av(whh[m][ks], hb_f[ks], acc[m]);
            mfma_settle_v<MT>(acc);
        }
        const long long pc4 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        // ---- gate arithmetic of this step; at H = 384 ONE hand-scheduled stream (ringstep3_mfma): the recurrent MFMAs tile by tile,
        //      the gate arithmetic of tile m behind the recurrent MFMAs of tile m + 1, the input projection of the next step
        //      (independent work for the matrix core) threaded through both, B fragments read from LDS inside the stream ---------
        half_t ho[MT];
        if constexpr (WOVEN) {
            float hv[MT];
            ring_stream<NKS, MT>(xacc[r], cst[r], hv, whh, wih, bias4, lds_addr(hb_r + lo), lds_addr(xb_r + ((step + 1) & 1) * TILE + lo));
#pragma unroll
            for (int m = 0; m < MT; ++m) ho[m] = (half_t)hv[m];
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) ho[m] = (half_t)lstm_cell(acc[m][0], acc[m][1], acc[m][2], acc[m][3], cst[r][m]);
        }
        const long long pc5 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        {
            u16_alias_t* sg = (u16_alias_t*)stage + c * U + q * MT;
#pragma unroll
            for (int m = 0; m < MT; ++m) sg[m] = __builtin_bit_cast(unsigned short, ho[m]);
            // every lane stores (lanes beyond the movers repeat a mover's store: same address, same bytes) and the re-arm is
            // unconditional (before step 2 it re-arms slots that are still armed, in the last two steps slots nobody reads any
            // more): always three stores per section, so the operation count behind a poll round is a constant
            const unsigned long long packed = *(const u64_alias_t*)((half_t*)stage + cc * U + part * 4);
            unsigned long long* dst = (unsigned long long*)(exr[r] + (long)(step & 3) * slot_stride + my_byte);
            unsigned long long* ra = (unsigned long long*)(exr[r] + (long)((step + 2) & 3) * slot_stride + my_byte);
            if (fast[r]) { *dst = packed; *ra = ~0ull; }
            else {
                __hip_atomic_store(dst, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ra, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *(unsigned long long*)(p.h + ((long)t * p.N + ring_of[r] * 16 + cc) * H + slice * U + part * 4) = packed;
            since[0] += 3; since[1] += 3;
        }
        const long long pc6 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        // ---- input projection of step t+1; around it the validation of the OTHER ring's quarter for its next section: its polls
        //      went out more than half a section ago, and the LDS latency of reading them back hides behind these MFMAs ------------
        constexpr int o = r ^ 1;
        const int step_o = r == 0 ? step : step + 1;
        const bool chk_o = TWO && step_o >= 1 && step_o < p.T;
        uint4_t chk[KQ];
        if (chk_o) check_begin(o, step_o, chk);
        if constexpr (!WOVEN) x_phase(xb_r + ((step + 1) & 1) * TILE, xacc[r]);
        if (chk_o) check_end(o, step_o, chk);
        const long long pc7 = (p.tune & 4) ? __builtin_readcyclecounter() : 0;
        // ---- first poll round for h_t, last in the section: the publishes of the other workgroups are visible by now, and it is
        //      looked at only after the other ring's section. It lands in the OTHER parity of the h tile: the waves of this
        //      workgroup may still be reading the current one ---------------------------------------------------------------------
        if (step + 1 < p.T) {
            poll_dma(exr[r], step & 3, hbuf + (r * 2 + ((step + 1) & 1)) * TILE, 0xFFu);
            since[r] = 0;
            since[r ^ 1] += n_poll;
        }
        if (p.tune & 4) {
            const long long pc8 = __builtin_readcyclecounter();
            st_sec[0] += pc3 - pc2; st_sec[1] += pc4 - pc3; st_sec[2] += pc5 - pc4; st_sec[3] += pc6 - pc5; st_sec[4] += pc7 - pc6; st_sec[5] += pc8 - pc7;
        }
    };

    // ---- the fast path's main loop, unrolled over four steps. With step & 3 a compile-time constant every exchange slot is one of
    //      eight loop-invariant pointers, every LDS tile a constant offset, and what is left between two streams is the barrier, one
    //      test of `bad`, and the two row pointers moving on: ~12 instructions where the generic section code above (runtime slot
    //      products, parity selects, the tests for the first / last steps) has ~60, each an issue slot of the only wave of its SIMD.
    //      Valid for 1 <= step and step + 2 < T (every poll / x fetch / validation is real); the steps around run the code above -----
    const char* ex_slot[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) ex_slot[r][k] = exr[r] + (long)k * slot_stride;
    const unsigned lds_h = smem_a + lo;                                  // + tile offset: a fragment row of an h tile
    const unsigned lds_x = smem_a + 4 * TILE + lo;                       // the same for the x slots
    const unsigned lds_v = smem_a + wave * 1024 + lo;                    // my quarter of an h tile (validation read-back)
    const unsigned m0_poll = (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_a + wave * 1024));
    const unsigned m0_x = (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_a + (3 - wave) * 1024));
    unsigned long long dead_m = 0;                                       // (uniform) all ones once the ring is given up
    auto fast_step = [&](auto rc, auto phc, int step) {
        constexpr int r = decltype(rc)::value, ph = decltype(phc)::value, o = r ^ 1;
        constexpr int par = ph & 1, par_o = r == 0 ? par : par ^ 1;     // parity of the other ring's h tile its next section reads
        __syncthreads();
        unsigned long long bad;
        float hv[MT];
        ring_stream_unrolled<NKS, MT, KQ, (r * 2 + par) * TILE, (r * 2 + (par ^ 1)) * TILE, (o * 2 + par_o) * TILE, (o * 2 + par_o) * TILE,
                             (4 + r * 2 + par) * TILE>(
            xacc[r], cst[r], hv, whh, wih, bias4, lds_h, lds_x, bad, m0_poll, ex_slot[o][r == 0 ? (ph + 3) & 3 : ph], vp_off, m0_x, xrow2, vx_off[r],
            lds_v, stage_a + (unsigned)((c * U + q * MT) * 2), stage_a + (unsigned)((cc * U + part * 4) * 2), (unsigned)my_byte, vh_off[r],
            ex_slot[r][ph], ex_slot[r][(ph + 2) & 3], hrow, fast_u[r]);
        if (__builtin_expect((bad | dead_m) != 0, 0)) {
            const int step_o = r == 0 ? step : step + 1;
            uint4_t chk[KQ];
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) chk[kk] = *(const uint4_t*)(hbuf + (o * 2 + par_o) * TILE + (wave + 4 * kk) * 1024 + lo);
            check_end(o, step_o, chk);
            dead_m = __builtin_amdgcn_readfirstlane(dead ? 1 : 0) ? ~0ull : 0ull;
            if constexpr (STATS) ++st_slow;
        }
    };

    auto run = [&](auto two_c) {
        int t = t0;
        const long drow = dt * row_bytes;
        auto generic_steps = [&](int from, int to) {
            for (int step = from; step < to; ++step, t += dt) {
                ring_step(std::integral_constant<int, 0>{}, two_c, step, t);
                if constexpr (decltype(two_c)::value) ring_step(std::integral_constant<int, 1>{}, two_c, step, t);
                xrow2 += drow;
                hrow += drow;
            }
        };
        if constexpr (FASTPATH && decltype(two_c)::value) {
            int step = p.T < 4 ? p.T : 4;
            generic_steps(0, step);
            // (lstm_tune bit 6: generic section code throughout - the comparison the tests make; a ring spread over several XCDs - write-through
            //  stores, see ring_store_policy - runs the generic code as well: the unrolled stream publishes with plain stores)
            if (!(p.tune & 64) && fast_u[0] != 0 && fast_u[1] != 0) {
                for (; step + 4 <= p.T - 2; step += 4) {
                    fast_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, step);
                    fast_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, step);
                    xrow2 += drow; hrow += drow;
                    fast_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, step + 1);
                    fast_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, step + 1);
                    xrow2 += drow; hrow += drow;
                    fast_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, step + 2);
                    fast_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, step + 2);
                    xrow2 += drow; hrow += drow;
                    fast_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, step + 3);
                    fast_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{}, step + 3);
                    xrow2 += drow; hrow += drow;
                }
                t = t0 + step * dt;
            }
            generic_steps(step, p.T);
        } else {
            generic_steps(0, p.T);
        }
    };
    if (two) run(std::true_type{});
    else run(std::false_type{});
    wait_vm(0);                                         // (the unconditional x-stream DMA of the last steps must have landed before the LDS is released)
    if ((p.tune & 4) && lane == 0) {
        long long* st = (long long*)((char*)p.xcc_ws + (((size_t)p.n_rings * NSL * sizeof(int) + 64 + 7) & ~(size_t)7)) + ((long)ring_of[0] * NSL + slice) * 16;
        st[0] = __builtin_readcyclecounter() - st_t0;
        st[1] = st_poll; st[2] = st_slow; st[3] = 0; st[4] = 0; st[5] = st_bar; st[6] = 0; st[7] = st_sec[5];
        st[8] = st_sec[0]; st[9] = st_sec[1]; st[10] = st_sec[2]; st[11] = st_sec[3]; st[12] = st_sec[4];
        st[13] = (long long)__builtin_amdgcn_s_memrealtime() - st_r0;
    }
}

// ---------------------------------------------------------------------------------------------------
// Ring-in-a-workgroup variant ("cta") for narrow layers (H = 64 / 96 / 128, e.g. the `fast` models): all H/U slices
// of a ring are waves of ONE workgroup, both weight sets are register-resident, and h never leaves the CU on its way
// to the next step - every wave drops its MT units straight into the LDS tile in B-fragment order, one workgroup
// barrier per step publishes it. No sentinel pre-fill, no polling, no co-residency constraint: grid = rings.
// The h tile is written to global memory (for the next layer) from the fragments the waves read back anyway,
// 16 bytes per lane, off the critical path.
template <int NKS, int MT>
__global__ __launch_bounds__(64 * (NKS * 32 / (4 * MT)), 1) void lstm_layer_cta_kernel(LstmFusedArgs fp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LstmArgs& p = fp.a;
    constexpr int H = NKS * 32, U = 4 * MT, NSL = H / U;
    static_assert(NSL >= NKS && NSL <= 16, "one workgroup per ring");
    const int lane = threadIdx.x & 63;
    const int slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave index == slice
    const int ring = blockIdx.x;
    char* hbuf = smem;                          // [2][NKS][64][16 B]
    char* xbuf = smem + 2 * NKS * 1024;         // [2][NKS][64][16 B]

    half8_t whh[MT][NKS], wih[MT][NKS];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const long o = ((((long)slice * MT + m) * NKS + ks) * 64 + lane) * 8;
            whh[m][ks] = *(const half8_t*)(p.whh + o);
            wih[m][ks] = *(const half8_t*)(fp.wih + o);
        }
    const int c = lane & 15, q = lane >> 4;
    const int lo = lane * 16;
    float cst[MT];
    float4_t bias4[MT];
    int hoff[MT];                               // where this lane's units live inside a B-fragment tile
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        cst[m] = 0.f;
        const int u = slice * U + q * MT + m;
#pragma unroll
        for (int i = 0; i < 4; ++i) bias4[m][i] = fp.bias[i * H + u];
        hoff[m] = (((u >> 5) * 64 + ((u >> 3) & 3) * 16 + c) * 16) + (u & 7) * 2;
    }
    int t = p.reverse ? p.T - 1 : 0;
    const int dt = p.reverse ? -1 : 1;
    const bool loader = slice < NKS;            // waves 0..NKS-1 move k-step `slice` of the x / h tiles
    const half_t* xptr = fp.x + ((long)(ring * 16 + c) * H + slice * 32 + q * 8);
    half_t* hptr = p.h + ((long)(ring * 16 + c) * H + slice * 32 + q * 8);
    const long row = (long)p.N * H;

    uint4_t xq = {0, 0, 0, 0}, xr = {0, 0, 0, 0};
    float4_t xacc[MT];
    auto x_phase = [&](const char* xb) {
#pragma unroll
        for (int m = 0; m < MT; ++m) xacc[m] = bias4[m];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const half8_t b = *(const half8_t*)(xb + ks * 1024 + lo);
#pragma unroll
            for (int m = 0; m < MT; ++m) xacc[m] = mfma16(wih[m][ks], b, xacc[m]);
        }
    };
    // x tiles are published one step EARLIER than they are consumed (slot (t+1)&1 holds x_{t+1} during step t), so the
    // input projection of step t+1 has no barrier of its own: it is issued right behind the h-MFMAs of step t and its
    // MFMAs run under the gate arithmetic.
    if (loader) {
        const int t1 = p.T > 1 ? t + dt : t, t2 = p.T > 2 ? t + 2 * dt : t, t3 = p.T > 3 ? t + 3 * dt : t;
        *(uint4_t*)(xbuf + slice * 1024 + lo) = *(const uint4_t*)(xptr + (long)t * row);
        *(uint4_t*)(xbuf + (NKS + slice) * 1024 + lo) = *(const uint4_t*)(xptr + (long)t1 * row);
        xq = *(const uint4_t*)(xptr + (long)t2 * row);
        xr = *(const uint4_t*)(xptr + (long)t3 * row);
    }
    __syncthreads();
    x_phase(xbuf);
    __syncthreads();                                    // slot 0 is rewritten by step 0: every wave must be done with x_0

    for (int step = 0; step < p.T; ++step, t += dt) {
        const int par = step & 1;
        if (loader) {                                   // x_{t+2} -> slot par (x_t was consumed a step ago), rotate, request x_{t+4}
            const int t4 = (step + 4 < p.T) ? t + 4 * dt : t;
            *(uint4_t*)(xbuf + (par * NKS + slice) * 1024 + lo) = xq;
            xq = xr;
            xr = *(const uint4_t*)(xptr + (long)t4 * row);
        }
        float4_t acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = xacc[m];
        if (step > 0) {
            const char* hb = hbuf + par * NKS * 1024 + lo;
            half8_t hb_f[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) hb_f[ks] = *(const half8_t*)(hb + ks * 1024);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma16(whh[m][ks], hb_f[ks], acc[m]);
            // h_{t-1} leaves for global memory from the fragment this wave just read (k-step == wave index)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                if (ks == slice) *(uint4_t*)(hptr + (long)(t - dt) * row) = __builtin_bit_cast(uint4_t, hb_f[ks]);
        }
        x_phase(xbuf + (par ^ 1) * NKS * 1024);         // projection of step t+1 (tile published by the previous barrier)
        char* hn = hbuf + (par ^ 1) * NKS * 1024;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float hv = lstm_cell(acc[m][0], acc[m][1], acc[m][2], acc[m][3], cst[m]);
            *(half_t*)(hn + hoff[m]) = (half_t)hv;
        }
        __syncthreads();                                // h_t tile and x_{t+2} tile complete
    }
    // last h tile
    if (loader) {
        const int par = p.T & 1;
        *(uint4_t*)(hptr + (long)(t - dt) * row) = *(const uint4_t*)(hbuf + (par * NKS + slice) * 1024 + lo);
    }
}

// ---------------------------------------------------------------------------------------------------
// Wide layers (512 < H <= 1024: the 768-wide old-style r9.4.1 models, the 1024-wide v4.3 `sup`): W_hh alone is 4.7 / 8.4 MB,
// so one copy per 16-chunk ring does not fit the chip's registers. Here a ring is NB = 2 column tiles (32 chunks) sharing one
// set of register-resident W_hh tiles: a wave owns U = 8 units (MT = 2 tiles, 64 fragments = all 256 accumulation
// registers at H = 1024), a workgroup is four slices of one ring, a ring spans H/32 workgroups (32 at H = 1024 = one whole XCD),
// and batch 256 fills 256 CUs. The input projection comes from a GEMM whose output columns are permuted so that a lane's
// eight pre-activations (4 gates x 2 units) are one 16-byte load: G[t][n][(slice*4 + q)*8 + gate*2 + m].
// Workgroup sharing of the h tile through LDS as in lstm_layer_wgx_kernel (each wave polls a quarter); the exchange goes through the
// ring buffer (RX) or, with "lstm_exchange" 0, through the sentinel-filled output tensor.
// Replaces the weight-streaming kernel on these shapes: 139 -> 14.4 ms per layer launch at H=1024, N=256, T=3334.
struct LstmWideArgs {
    const half_t* G;      // [T][N][4H], columns permuted as above, bias included
    LstmArgs a;
    char* ex;             // RX: exchange ring buffer [4][R][NB*NKS][64][16], armed with 0xFF
    int R;                // RX: ring stride of `ex`
};

// RX (round 2, default): hand-off through an L2-resident ring buffer in fragment order, exactly as lstm_layer_wgx_kernel (the
// protocol and its argument are stated there): a workgroup's poll of its ring tile (64 KiB at H = 1024) becomes 64 coalesced KiB
// instead of 1024 half-line pieces of a row-major tensor, the sentinel fill of the output tensor disappears, and the output
// tensor is written by a second plain store. Same arithmetic -> same bytes as RX = false (tested).
template <int NKS, bool RX, bool STATS = false>
__global__ __launch_bounds__(256, 1) void lstm_layer_wide_kernel(LstmWideArgs wp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LstmArgs& p = wp.a;
    constexpr int MT = 2, NB = 2, H = NKS * 32, U = 4 * MT, NSL = H / U, WPR = NSL / 4;
    constexpr int NF = NB * NKS;                 // B fragments of a ring's h tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7;
    const int lwg = blockIdx.x >> 3;
    const int rl = lwg / WPR;
    // test hook ("lstm_tune" bit 5): spread the workgroups of every ring over all eight XCDs, so that the placement-independent
    // (write-through) hand-off really crosses XCDs
    const int ring = rl * 8 + ((p.tune & 32) ? ((xcd + (lwg - rl * WPR)) & 7) : xcd);
    const int slice = (lwg - rl * WPR) * 4 + wave;
    if (ring >= p.n_rings) return;

    char* hbuf = smem;                                          // [2][NB][NKS][64 lanes][16 B]
    char* stage = smem + 2 * NF * 1024 + wave * (16 * U * 2);   // per-wave [16 chunks][U] output transpose

    half8_t whh[MT][NKS];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            whh[m][ks] = *(const half8_t*)(p.whh + ((((long)slice * MT + m) * NKS + ks) * 64 + lane) * 8);

    const int c = lane & 15, q = lane >> 4;
    const int lo = lane * 16;
    const long row_bytes = (long)p.N * H * 2;
    float cst[MT][NB];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) cst[m][nb] = 0.f;
    bool dead = false;
    const bool fast = ring_store_policy(p, ring, slice, NSL, lane);

    int t = p.reverse ? p.T - 1 : 0;
    const int dt = p.reverse ? -1 : 1;
    constexpr int TILE = NF * 1024;
    const long slot_stride = RX ? (long)wp.R * TILE : 0;
    char* exr = RX ? wp.ex + (long)ring * TILE : nullptr;
    // fragment f = nb*NKS + ks of the ring tile: chunk = ring*32 + nb*16 + c, units ks*32 + q*8 ..
    auto frag_voff = [&](int f) -> unsigned {
        if constexpr (RX) return (unsigned)(f * 1024 + lo);
        const int nb = f / NKS, ks = f - nb * NKS;
        return (unsigned)((((ring * NB + nb) * 16 + c) * H + ks * 32 + q * 8) * 2);
    };
    // gate pre-activations of this lane: 16 bytes per column tile
    const half_t* gptr = wp.G + ((long)(ring * NB * 16 + c) * 4 * H + (slice * 4 + q) * 8);
    const long g_row = (long)p.N * 4 * H;
    // Round 4: the two column tiles of a ring are INDEPENDENT chunks that merely share the weights, so a step is two half-steps, one
    // per tile, each with its own poll check, barrier, 2 * NKS MFMAs, gates, publish and first poll round: while tile 0's h_t travels
    // (publish -> L2 -> poll, ~1.5 k cycles) the wave does tile 1's half-step (~1.5 k cycles of MFMAs and gates) and vice versa. In
    // lock step (one barrier, both tiles' MFMAs, both tiles' gates, then the hand-off of both) the round trip was exposed in every step:
    // 6.5 k cycles per step for 2.0 k of MFMAs. Same arithmetic per cell -> the same bytes (tests).
    constexpr int KH = (NKS + 3) / 4;            // poll pieces of a wave per tile
    constexpr bool EXH = NKS % 4 == 0;
    uint4_t gq[NB], gr[NB], hq[NB][KH];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int t1 = p.T > 1 ? t + dt : t;
        gq[nb] = *(const uint4_t*)(gptr + (long)t * g_row + (long)nb * 16 * 4 * H);
        gr[nb] = *(const uint4_t*)(gptr + (long)t1 * g_row + (long)nb * 16 * 4 * H);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(gq[nb]), "+v"(gr[nb]));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) hq[nb][kk] = uint4_t{0, 0, 0, 0};

    // STATS ("lstm_tune" bit 2, H = 1024 with the ring-buffer exchange only): per-wave cycle stamps into the workspace tail
    // (tools/lstm_wide_stats.py). A template parameter: as a run-time flag the ten untaken branches per step cost the product kernel 6 %.
    constexpr bool stats = STATS;
    long long st_poll = 0, st_bar = 0, st_mf = 0, st_gate = 0, st_rounds = 0, st_wait = 0;
    const long long st_t0 = __builtin_readcyclecounter(), st_r0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int step = 0; step < p.T; ++step, t += dt) {
        const int par = step & 1;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const long long c0 = stats ? __builtin_readcyclecounter() : 0;
            // ---- B. my quarter of tile nb of the ring's h_{t-1} (round one went out right after this tile's previous store) --------
            if (step > 0) {
                const char* base = RX ? exr + (long)((step - 1) & 3) * slot_stride : (const char*)p.h + (long)(t - dt) * row_bytes;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, RX ? TILE : (int)row_bytes, 0x00020000);
                unsigned spins = dead ? p.max_spins : 0u;
                unsigned pend = 0;
#pragma unroll
                for (int kk = 0; kk < KH; ++kk) {
                    if (EXH || wave + 4 * kk < NKS) {
                        unsigned orv = hq[nb][kk].x | hq[nb][kk].y | hq[nb][kk].z | hq[nb][kk].w;
                        if (__any((orv & SENTINEL_MASK) != 0)) pend |= (1u << kk);
                    }
                }
                if (stats) st_wait += __builtin_readcyclecounter() - c0;
                while (pend != 0) {
                    if (++spins > p.max_spins) {
                        if (lane == 0 && !dead) atomicExch(p.err, 1);
                        dead = true;
                        break;
                    }
                    ++st_rounds;
                    if (!(p.tune & 1)) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int kk = 0; kk < KH; ++kk)
                        if (pend & (1u << kk))
                            hq[nb][kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, frag_voff(nb * NKS + wave + 4 * kk), 0, (int)0x80000010);
#pragma unroll
                    for (int kk = 0; kk < KH; ++kk)
                        if (pend & (1u << kk)) {
                            unsigned orv = hq[nb][kk].x | hq[nb][kk].y | hq[nb][kk].z | hq[nb][kk].w;
                            if (!__any((orv & SENTINEL_MASK) != 0)) pend &= ~(1u << kk);
                        }
                }
#pragma unroll
                for (int kk = 0; kk < KH; ++kk) {
                    const int ks = wave + 4 * kk;
                    if (EXH || ks < NKS) *(uint4_t*)(hbuf + (par * NF + nb * NKS + ks) * 1024 + lo) = hq[nb][kk];
                }
            }
            // ---- C. gate pre-activations of tile nb: this step's are in gq; rotate and request step t+2 ------------------------------
            float4_t acc[MT];
            {
                const half8_t g8 = __builtin_bit_cast(half8_t, gq[nb]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[m][i] = (float)g8[i * MT + m];
                const int t2 = (step + 2 < p.T) ? t + 2 * dt : t;
                gq[nb] = gr[nb];
                gr[nb] = *(const uint4_t*)(gptr + (long)t2 * g_row + (long)nb * 16 * 4 * H);
            }
            const long long c1 = stats ? __builtin_readcyclecounter() : 0;
            __syncthreads();                            // tile nb of h_{t-1} is complete in LDS (and the stage buffer is free again)
            const long long c2 = stats ? __builtin_readcyclecounter() : 0;
            // ---- E. recurrent part of tile nb, gates, publish ------------------------------------------------------------------------
            // The first poll round for the OTHER tile goes out from inside this tile's MFMAs, a quarter of the way in: that tile was
            // published ~0.6 k cycles earlier by every wave of the ring at about the same time, so the round finds complete data (issued
            // right behind the publish it sampled the L2 before the partners' stores were visible and a second, exposed round followed:
            // 12.9 ms per launch instead of 10.2 on that box), and it has landed when that tile's next half-step checks it (1-2 % of the
            // steps need a second round). Tile 0 polls tile 1's h_{t-1}, tile 1 polls tile 0's h_t.
            // Same box, sup-LSTM 256 x 20000, ms per launch: lock step 11.40-11.44, this 11.04-11.11, polls an eighth of the way in
            // 11.25-11.29. Also measured (tools/lstm_wide_stats.py, cycles per step incl. ~1.1 k of stamps; base 7290): fragment reads
            // 2 / 4 k-steps ahead 7240 / 7260; the gate prefetch behind the polls 7900; a counted vmcnt in front of the publish 7430; even /
            // odd k-steps in separate accumulators 8280. Round 5 (profiles/r05_lstm_wide_elimination.txt): NOT LDS bandwidth, as round 4
            // read it - without any fragment read the section loses 260 cycles per step; the 64 MFMAs of a half-step are 1056 cycles in any
            // order (the s_nop hipcc puts between them is free), the eight poll loads cost ~900 cycles of blocked issue (~116 for quiet
            // data) and the two add wherever the polls are placed (burst, one every 1-3 k-steps, reads 6-10 ahead, own quarter from the
            // poll registers, rotated per workgroup: all within 3 %). With no MFMAs at all the layer keeps 81 % of its time.
            const int ob = 1 - nb;                      // (a constant after unrolling)
            const bool want = nb == 1 ? step + 1 < p.T : step > 0;
            auto other_polls = [&]() __attribute__((always_inline)) {
                const int dstep = nb == 1 ? step : step - 1;
                const char* base = RX ? exr + (long)(dstep & 3) * slot_stride : (const char*)p.h + (long)(nb == 1 ? t : t - dt) * row_bytes;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, RX ? TILE : (int)row_bytes, 0x00020000);
#pragma unroll
                for (int kk = 0; kk < KH; ++kk) {
                    const int ks = wave + 4 * kk;
                    if (EXH || ks < NKS) hq[ob][kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, frag_voff(ob * NKS + ks), 0, (int)0x80000010);
                }
            };
            if (step > 0) {
                const char* hb = hbuf + (par * NF + nb * NKS) * 1024 + lo;
                half8_t b_cur = *(const half8_t*)hb, b_nxt = b_cur;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    if (ks + 1 < NKS) b_nxt = *(const half8_t*)(hb + (ks + 1) * 1024);
#pragma unroll
                    for (int m = 0; m < MT; ++m) mfma16_av(whh[m][ks], b_cur, acc[m]);
                    b_cur = b_nxt;
                    if (ks == NKS / 4 && want) other_polls();
                }
                asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
            } else if (want) other_polls();
            const long long c3 = stats ? __builtin_readcyclecounter() : 0;
            {
                u16_alias_t* sg = (u16_alias_t*)stage + c * U + q * MT;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    sg[m] = __builtin_bit_cast(unsigned short, (half_t)lstm_cell(acc[m][0], acc[m][1], acc[m][2], acc[m][3], cst[m][nb]));
                if (lane < 32) {                    // 16 chunks x 2 parts of 4 units: 8-byte stores
                    const int cc = lane >> 1, part = lane & 1;
                    const unsigned long long packed = *(const u64_alias_t*)((half_t*)stage + cc * U + part * 4);
                    unsigned long long* dst =
                        (unsigned long long*)(p.h + ((long)t * p.N + (ring * NB + nb) * 16 + cc) * H + slice * U + part * 4);
                    if constexpr (RX) {
                        const int u0 = slice * U + part * 4;
                        const int my_byte = (((nb * NKS + (u0 >> 5)) * 64 + ((u0 >> 3) & 3) * 16 + cc) << 4) + (u0 & 7) * 2;
                        // the re-arm store of this tile's previous step must be complete before anything newer of it is published
                        // (lstm_layer_wgx_kernel); what else is in flight - the other tile's polls - went out a half-step ago
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                        unsigned long long* xd = (unsigned long long*)(exr + (long)(step & 3) * slot_stride + my_byte);
                        if (fast) *xd = packed;
                        else __hip_atomic_store(xd, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (step >= 2 && step + 2 < p.T) {
                            unsigned long long* ra = (unsigned long long*)(exr + (long)((step + 2) & 3) * slot_stride + my_byte);
                            if (fast) *ra = ~0ull;
                            else __hip_atomic_store(ra, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        *dst = packed;              // the layer output proper
                    } else {
                        if (fast) *dst = packed;
                        else __hip_atomic_store(dst, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (stats) { const long long c4 = __builtin_readcyclecounter(); st_poll += c1 - c0; st_bar += c2 - c1; st_mf += c3 - c2; st_gate += c4 - c3; }
        }
    }
    if (stats && lane == 0) {
        long long* st = (long long*)((char*)p.xcc_ws + (((size_t)p.n_rings * NSL * sizeof(int) + 64 + 7) & ~(size_t)7)) + ((long)ring * NSL + slice) * 16;
        st[0] = __builtin_readcyclecounter() - st_t0;
        st[1] = st_poll; st[2] = st_bar; st[3] = st_mf; st[4] = st_gate; st[5] = st_rounds;
        st[6] = (long long)__builtin_amdgcn_s_memrealtime() - st_r0; st[7] = st_wait;
    }
}

// (Round 4 built the pairing of lstm_layer_wgx2_kernel for this kernel as well - two rings per workgroup on one register-resident copy of
// the weights, bit-identical in five geometries - and measured it SLOWER than two launches of single rings: 27.3 against 2 x 10.9 ms per
// 512 chunks x 3334 steps at H = 1024. A wide ring step is already 128 MFMAs against ~1.3 k cycles of hand-off, every register holds
// weights or accumulators, so the second ring's polls, validation and publish run one after the other in the same wave. Removed again;
// git: 113fefb "paired wide recurrent kernel".)
__global__ void fill_u16_kernel(uint16_t* dst, uint16_t v, size_t count) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
    const unsigned vv = (unsigned)v | ((unsigned)v << 16);
    for (; i + 8 <= count; i += stride) *(uint4_t*)(dst + i) = uint4_t{vv, vv, vv, vv};
    if (i < count && i + 8 > count)
        for (size_t j = i; j < count; ++j) dst[j] = v;
}

}  // namespace bh

// Bound of every exchange spin loop (poll rounds). Process-wide; lowered by tests to provoke the timeout path.
static unsigned g_max_spins = 1000000u;
int bh_k_lstm_set_option(const char* name, int value) {
    if (strcmp(name, "lstm_max_spins") != 0) return -1;
    g_max_spins = value >= 0 ? (unsigned)value : 1000000u;     // 0: the first incomplete poll round is a timeout
    return 0;
}

unsigned bh_k_lstm_max_spins() { return g_max_spins; }

size_t bh_k_lstm_packed_bytes(int H) { return (size_t)4 * H * H * 2; }
size_t bh_k_lstm_ws_bytes(int N, int H) {
    // XCD agreement slots + (tune bit 4) per-wave statistics: up to 16 x int64 per (ring, slice)
    const size_t waves = (size_t)((N + 15) / 16) * ((H + 7) / 8);        // up to H/8 slices per ring (wide variant)
    return waves * sizeof(int) + 64 + waves * 16 * sizeof(long long) + 64;
}

int bh_k_fill_u16(void* dst, uint16_t value, size_t count, hipStream_t stream) {
    using namespace bh;
    BH_REQUIRE(((uintptr_t)dst & 15) == 0, "fill_u16: destination must be 16-byte aligned");
    size_t vecs = (count + 7) / 8;
    int blocks = (int)((vecs + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fill_u16_kernel, dim3(blocks), dim3(256), 0, stream, (uint16_t*)dst, value, count);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// One launch serves at most (CUs / (8 * H/16)) * 4 * 8 rings co-resident; the caller (engine.cpp)
// splits larger batches by offsetting the base pointers by 16*ring0 columns: N stays the row stride of
// G / h and n_rings is the number of 16-chunk rings this launch runs. N %% 16 == 0 (engine pads).
int bh_k_lstm_layer(const void* gates_in, const void* whh_packed, void* h_out, int T, int N, int H,
                    int reverse, int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws, int force_slow) {
    using namespace bh;
    BH_REQUIRE(N % 16 == 0, "lstm: batch must be padded to a multiple of 16 (N=%d)", N);
    BH_REQUIRE(H % 32 == 0 && H >= 32 && H <= 512, "lstm: register-resident kernel needs H%%32==0, 32<=H<=512 (H=%d)", H);
    int dev = 0, cus = 0;
    BH_CHECK_HIP(hipGetDevice(&dev));
    BH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int nsl = H / 16;
    BH_REQUIRE(n_rings > 0 && n_rings <= N / 16, "lstm: n_rings=%d outside 1..%d", n_rings, N / 16);
    const int rl = (n_rings + 7) / 8;          // rings per XCD
    const int groups = (rl + 3) / 4;           // 4 rings (waves) per workgroup
    const int grid = 8 * groups * nsl;
    BH_REQUIRE(grid <= cus, "lstm: %d workgroups must be co-resident but the device has %d CUs; split the batch", grid, cus);
    BH_REQUIRE(xcc_ws != nullptr, "lstm: missing XCD agreement workspace");
    BH_CHECK_HIP(hipMemsetAsync(xcc_ws, 0xFF, (size_t)n_rings * nsl * sizeof(int), stream));
    LstmArgs a{(const half_t*)gates_in, (const half_t*)whh_packed, (half_t*)h_out, T, N, H, n_rings,
               reverse, err_flag, g_max_spins, xcc_ws, force_slow & 1, force_slow >> 8};
#define BH_LSTM_CASE(NKS) \
    case NKS: hipLaunchKernelGGL((lstm_layer_kernel<NKS, false>), dim3(grid), dim3(256), 0, stream, a); break;
    switch (H / 32) {
        BH_LSTM_CASE(1) BH_LSTM_CASE(2) BH_LSTM_CASE(3) BH_LSTM_CASE(4) BH_LSTM_CASE(5) BH_LSTM_CASE(6)
        BH_LSTM_CASE(7) BH_LSTM_CASE(8) BH_LSTM_CASE(9) BH_LSTM_CASE(10) BH_LSTM_CASE(11) BH_LSTM_CASE(12)
        BH_LSTM_CASE(13) BH_LSTM_CASE(14) BH_LSTM_CASE(15) BH_LSTM_CASE(16)
        default: BH_REQUIRE(false, "lstm: unsupported H=%d", H);
    }
#undef BH_LSTM_CASE
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

int bh_k_lstm_layer_stream(const void* gates_in, const void* whh_packed, void* h_out, int T, int N, int H,
                           int reverse, int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws, int force_slow) {
    using namespace bh;
    BH_REQUIRE(N % 16 == 0, "lstm: batch must be padded to a multiple of 16 (N=%d)", N);
    BH_REQUIRE(H % 64 == 0 && H >= 64 && H <= 1024, "lstm: streaming kernel needs H%%64==0, 64<=H<=1024 (H=%d)", H);
    int dev = 0, cus = 0;
    BH_CHECK_HIP(hipGetDevice(&dev));
    BH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int nsl = H / 16, wpr = nsl / 4;
    BH_REQUIRE(n_rings > 0 && n_rings <= N / 16, "lstm: n_rings=%d outside 1..%d", n_rings, N / 16);
    const int rl = (n_rings + 7) / 8;
    const int grid = 8 * rl * wpr;
    BH_REQUIRE(grid <= cus, "lstm: %d workgroups must be co-resident but the device has %d CUs; split the batch", grid, cus);
    BH_REQUIRE(xcc_ws != nullptr, "lstm: missing XCD agreement workspace");
    BH_CHECK_HIP(hipMemsetAsync(xcc_ws, 0xFF, (size_t)n_rings * nsl * sizeof(int), stream));
    LstmArgs a{(const half_t*)gates_in, (const half_t*)whh_packed, (half_t*)h_out, T, N, H, n_rings,
               reverse, err_flag, g_max_spins, xcc_ws, force_slow & 1, force_slow >> 8};
    switch (H / 32) {
        case 2: hipLaunchKernelGGL((lstm_layer_kernel<2, true>), dim3(grid), dim3(256), 0, stream, a); break;
        case 4: hipLaunchKernelGGL((lstm_layer_kernel<4, true>), dim3(grid), dim3(256), 0, stream, a); break;
        case 8: hipLaunchKernelGGL((lstm_layer_kernel<8, true>), dim3(grid), dim3(256), 0, stream, a); break;
        case 12: hipLaunchKernelGGL((lstm_layer_kernel<12, true>), dim3(grid), dim3(256), 0, stream, a); break;
        case 16: hipLaunchKernelGGL((lstm_layer_kernel<16, true>), dim3(grid), dim3(256), 0, stream, a); break;
        case 20: hipLaunchKernelGGL((lstm_layer_kernel<20, true>), dim3(grid), dim3(256), 0, stream, a); break;
        case 24: hipLaunchKernelGGL((lstm_layer_kernel<24, true>), dim3(grid), dim3(256), 0, stream, a); break;
        case 28: hipLaunchKernelGGL((lstm_layer_kernel<28, true>), dim3(grid), dim3(256), 0, stream, a); break;
        case 32: hipLaunchKernelGGL((lstm_layer_kernel<32, true>), dim3(grid), dim3(256), 0, stream, a); break;
        default: BH_REQUIRE(false, "lstm: unsupported H=%d for the streaming kernel", H);
    }
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

int bh_k_lstm_layer_fused(const void* x, const void* wih_packed, const float* bias, const void* whh_packed, void* h_out,
                          int T, int N, int H, int reverse, int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws,
                          int force_slow) {
    using namespace bh;
    BH_REQUIRE(N % 16 == 0, "lstm: batch must be padded to a multiple of 16 (N=%d)", N);
    BH_REQUIRE(H % 32 == 0 && H >= 32 && H <= 512, "lstm: register-resident kernel needs H%%32==0, 32<=H<=512 (H=%d)", H);
    BH_REQUIRE(x != h_out, "lstm: fused layer cannot run in place");
    int dev = 0, cus = 0;
    BH_CHECK_HIP(hipGetDevice(&dev));
    BH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int nsl = H / 16;
    BH_REQUIRE(n_rings > 0 && n_rings <= N / 16, "lstm: n_rings=%d outside 1..%d", n_rings, N / 16);
    const int rl = (n_rings + 7) / 8;
    const int groups = (rl + 3) / 4;
    const int grid = 8 * groups * nsl;
    BH_REQUIRE(grid <= cus, "lstm: %d workgroups must be co-resident but the device has %d CUs; split the batch", grid, cus);
    BH_REQUIRE(xcc_ws != nullptr, "lstm: missing XCD agreement workspace");
    BH_CHECK_HIP(hipMemsetAsync(xcc_ws, 0xFF, (size_t)n_rings * nsl * sizeof(int), stream));
    LstmFusedArgs a{(const half_t*)x, (const half_t*)wih_packed, bias,
                    LstmArgs{nullptr, (const half_t*)whh_packed, (half_t*)h_out, T, N, H, n_rings, reverse, err_flag, g_max_spins,
                             xcc_ws, force_slow & 1, force_slow >> 8}};
    const size_t lds = (size_t)(H / 32) * 4096;
#define BH_LSTM_CASE(NKS)                                                                                        \
    case NKS:                                                                                                    \
        if (lds > 64 * 1024)                                                                                     \
            BH_CHECK_HIP(bh_max_lds((const void*)lstm_layer_fused_kernel<NKS>, (int)lds));            \
        hipLaunchKernelGGL(lstm_layer_fused_kernel<NKS>, dim3(grid), dim3(256), lds, stream, a);                 \
        break;
    switch (H / 32) {
        BH_LSTM_CASE(1) BH_LSTM_CASE(2) BH_LSTM_CASE(3) BH_LSTM_CASE(4) BH_LSTM_CASE(5) BH_LSTM_CASE(6)
        BH_LSTM_CASE(7) BH_LSTM_CASE(8) BH_LSTM_CASE(9) BH_LSTM_CASE(10) BH_LSTM_CASE(11) BH_LSTM_CASE(12)
        BH_LSTM_CASE(13) BH_LSTM_CASE(14) BH_LSTM_CASE(15) BH_LSTM_CASE(16)
        default: BH_REQUIRE(false, "lstm: unsupported H=%d", H);
    }
#undef BH_LSTM_CASE
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// Workgroup-shared variant: usable when 4*U | H (U = 12 or 16 units per wave) and both weight sets fit the
// register file. Returns the units-per-wave it would use, 0 if the shape is not covered.
int bh_k_lstm_wg_units(int H) {
    if (H % 32 != 0) return 0;
    const int nks = H / 32;
    if (H % 48 == 0 && nks <= 12) return 12;
    if (H % 64 == 0 && nks <= 8) return 16;
    return 0;
}

// Ring-buffer exchange variant of the workgroup-shared kernel (lstm_layer_wgx_kernel). `ex`: 4 * R * (H/32) KiB, armed here
// with 0xFF when `arm` is set (once per layer: launches of one layer share it, each with its ring offset applied by the caller).
size_t bh_k_lstm_wgx_ex_bytes(int N, int H) { return (size_t)4 * (N / 16) * (H / 32) * 1024; }
int bh_k_lstm_layer_wgx(const void* x, const void* wih_packed, const float* bias, const void* whh_packed, void* h_out, void* ex,
                        int T, int N, int H, int R, int reverse, int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws,
                        int force_slow, int arm) {
    using namespace bh;
    BH_REQUIRE(N % 16 == 0, "lstm: batch must be padded to a multiple of 16 (N=%d)", N);
    const int U = bh_k_lstm_wg_units(H);
    BH_REQUIRE(U != 0, "lstm: workgroup-shared kernel does not cover H=%d", H);
    BH_REQUIRE(x != h_out && ex != nullptr, "lstm: fused layer cannot run in place / missing exchange buffer");
    int dev = 0, cus = 0;
    BH_CHECK_HIP(hipGetDevice(&dev));
    BH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int nsl = H / U, wpr = nsl / 4;
    BH_REQUIRE(n_rings > 0 && n_rings <= R, "lstm: n_rings=%d outside 1..%d", n_rings, R);
    const int rl = (n_rings + 7) / 8;
    const int grid = 8 * rl * wpr;
    BH_REQUIRE(grid <= cus, "lstm: %d workgroups must be co-resident but the device has %d CUs; split the batch", grid, cus);
    BH_REQUIRE(xcc_ws != nullptr, "lstm: missing XCD agreement workspace");
    BH_CHECK_HIP(hipMemsetAsync(xcc_ws, 0xFF, (size_t)n_rings * nsl * sizeof(int), stream));
    const int nks = H / 32;
    if (arm) BH_CHECK_HIP(hipMemsetAsync(ex, 0xFF, (size_t)4 * R * nks * 1024, stream));
    LstmWgxArgs a{LstmFusedArgs{(const half_t*)x, (const half_t*)wih_packed, bias,
                                LstmArgs{nullptr, (const half_t*)whh_packed, (half_t*)h_out, T, N, H, n_rings, reverse, err_flag,
                                         g_max_spins, xcc_ws, force_slow & 1, force_slow >> 8}},
                  (char*)ex, R};
    const size_t lds = (size_t)5 * nks * 1024 + 4 * 16 * U * 2;
#define BH_LSTM_WGX(NKS, MT)                                                                                     \
    if (nks == NKS && U == 4 * MT) {                                                                             \
        if (lds > 64 * 1024)                                                                                     \
            BH_CHECK_HIP(bh_max_lds((const void*)lstm_layer_wgx_kernel<NKS, MT>, (int)lds));            \
        hipLaunchKernelGGL((lstm_layer_wgx_kernel<NKS, MT>), dim3(grid), dim3(256), lds, stream, a);              \
    } else
    BH_LSTM_WGX(3, 3) BH_LSTM_WGX(6, 3) BH_LSTM_WGX(9, 3) BH_LSTM_WGX(12, 3)
    BH_LSTM_WGX(2, 4) BH_LSTM_WGX(4, 4) BH_LSTM_WGX(8, 4)
    { BH_REQUIRE(false, "lstm: workgroup-shared kernel has no instance for H=%d", H); }
#undef BH_LSTM_WGX
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// Two rings per workgroup (lstm_layer_wgx2_kernel): n_rings may be up to twice what one ring per 8 * H/(4U) workgroups allows.
int bh_k_lstm_layer_wgx2(const void* x, const void* wih_packed, const float* bias, const void* whh_packed, void* h_out, void* ex,
                         int T, int N, int H, int R, int reverse, int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws,
                         int force_slow, int arm) {
    using namespace bh;
    BH_REQUIRE(N % 16 == 0, "lstm: batch must be padded to a multiple of 16 (N=%d)", N);
    const int U = bh_k_lstm_wg_units(H);
    BH_REQUIRE(U != 0, "lstm: workgroup-shared kernel does not cover H=%d", H);
    BH_REQUIRE(x != h_out && ex != nullptr, "lstm: fused layer cannot run in place / missing exchange buffer");
    int dev = 0, cus = 0;
    BH_CHECK_HIP(hipGetDevice(&dev));
    BH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int nsl = H / U, wpr = nsl / 4;
    BH_REQUIRE(n_rings > 0 && n_rings <= R, "lstm: n_rings=%d outside 1..%d", n_rings, R);
    const int n_pairs = (n_rings + 1) / 2;
    const int rl = (n_pairs + 7) / 8;
    const int grid = 8 * rl * wpr;
    BH_REQUIRE(grid <= cus, "lstm: %d workgroups must be co-resident but the device has %d CUs; split the batch", grid, cus);
    BH_REQUIRE(xcc_ws != nullptr, "lstm: missing XCD agreement workspace");
    BH_CHECK_HIP(hipMemsetAsync(xcc_ws, 0xFF, (size_t)n_rings * nsl * sizeof(int), stream));
    const int nks = H / 32;
    if (arm) BH_CHECK_HIP(hipMemsetAsync(ex, 0xFF, (size_t)4 * R * nks * 1024, stream));
    LstmWgxArgs a{LstmFusedArgs{(const half_t*)x, (const half_t*)wih_packed, bias,
                                LstmArgs{nullptr, (const half_t*)whh_packed, (half_t*)h_out, T, N, H, n_rings, reverse, err_flag,
                                         g_max_spins, xcc_ws, force_slow & 1, force_slow >> 8}},
                  (char*)ex, R};
    const size_t lds = (size_t)8 * nks * 1024 + 4 * 16 * U * 2;
    if (nks == 12 && U == 12 && ((force_slow >> 8) & 4)) {          // lstm_tune bit 2: the instance with section stamps (tools/lstm_stats2.py)
        BH_CHECK_HIP(bh_max_lds((const void*)lstm_layer_wgx2_kernel<12, 3, true>, (int)lds));
        hipLaunchKernelGGL((lstm_layer_wgx2_kernel<12, 3, true>), dim3(grid), dim3(256), lds, stream, a);
        BH_CHECK_HIP(hipGetLastError());
        return 0;
    }
#define BH_LSTM_WGX2(NKS, MT)                                                                                    \
    if (nks == NKS && U == 4 * MT) {                                                                             \
        if (lds > 64 * 1024)                                                                                     \
            BH_CHECK_HIP(bh_max_lds((const void*)lstm_layer_wgx2_kernel<NKS, MT>, (int)lds));            \
        hipLaunchKernelGGL((lstm_layer_wgx2_kernel<NKS, MT>), dim3(grid), dim3(256), lds, stream, a);             \
    } else
    BH_LSTM_WGX2(3, 3) BH_LSTM_WGX2(6, 3) BH_LSTM_WGX2(9, 3) BH_LSTM_WGX2(12, 3)
    BH_LSTM_WGX2(2, 4) BH_LSTM_WGX2(4, 4) BH_LSTM_WGX2(8, 4)
    { BH_REQUIRE(false, "lstm: workgroup-shared kernel has no instance for H=%d", H); }
#undef BH_LSTM_WGX2
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// Ring-in-a-workgroup kernel: covers the narrow layers whose two weight sets fit the registers of H/U <= 16 waves.
int bh_k_lstm_cta_units(int H) {
    if (H == 96) return 12;
    if (H == 64 || H == 128) return 16;
    return 0;
}

int bh_k_lstm_layer_cta(const void* x, const void* wih_tiles, const float* bias, const void* whh_tiles, void* h_out, int T, int N,
                        int H, int reverse, hipStream_t stream, int n_rings) {
    using namespace bh;
    BH_REQUIRE(N % 16 == 0, "lstm: batch must be padded to a multiple of 16 (N=%d)", N);
    const int U = bh_k_lstm_cta_units(H);
    BH_REQUIRE(U != 0, "lstm: ring-in-a-workgroup kernel does not cover H=%d", H);
    BH_REQUIRE(x != h_out, "lstm: fused layer cannot run in place");
    BH_REQUIRE(n_rings > 0 && n_rings <= N / 16, "lstm: n_rings=%d outside 1..%d", n_rings, N / 16);
    LstmFusedArgs a{(const half_t*)x, (const half_t*)wih_tiles, bias,
                    LstmArgs{nullptr, (const half_t*)whh_tiles, (half_t*)h_out, T, N, H, n_rings, reverse, nullptr, 0u, nullptr, 0, 0}};
    const int nks = H / 32, nsl = H / U;
    const size_t lds = (size_t)4 * nks * 1024;
    if (H == 96) hipLaunchKernelGGL((lstm_layer_cta_kernel<3, 3>), dim3(n_rings), dim3(64 * nsl), lds, stream, a);
    else if (H == 64) hipLaunchKernelGGL((lstm_layer_cta_kernel<2, 4>), dim3(n_rings), dim3(64 * nsl), lds, stream, a);
    else hipLaunchKernelGGL((lstm_layer_cta_kernel<4, 4>), dim3(n_rings), dim3(64 * nsl), lds, stream, a);
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}

// Wide layers: stationary W_hh, rings of 32 chunks; the caller provides G with permuted columns (bh_k_lstm_wide_permute).
int bh_k_lstm_wide_ok(int H) { return H > 512 && H <= 1024 && H % 128 == 0; }

size_t bh_k_lstm_wide_ex_bytes(int N, int H) { return (size_t)4 * (N / 32) * 2 * (H / 32) * 1024; }

// ex != nullptr: ring-buffer exchange (R = rings of the whole batch, `arm` = fill it with the sentinel first: once per layer)
int bh_k_lstm_layer_wide(const void* gates_perm, const void* whh_tiles, void* h_out, int T, int N, int H, int reverse,
                         int* err_flag, hipStream_t stream, int n_rings, int* xcc_ws, int force_slow, void* ex, int R, int arm) {
    using namespace bh;
    BH_REQUIRE(bh_k_lstm_wide_ok(H), "lstm: wide kernel does not cover H=%d", H);
    BH_REQUIRE(N % 32 == 0, "lstm: wide kernel needs the batch padded to a multiple of 32 (N=%d)", N);
    int dev = 0, cus = 0;
    BH_CHECK_HIP(hipGetDevice(&dev));
    BH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int nsl = H / 8, wpr = nsl / 4;
    BH_REQUIRE(n_rings > 0 && n_rings <= N / 32, "lstm: n_rings=%d outside 1..%d", n_rings, N / 32);
    const int rl = (n_rings + 7) / 8;
    const int grid = 8 * rl * wpr;
    BH_REQUIRE(grid <= cus, "lstm: %d workgroups must be co-resident but the device has %d CUs; split the batch", grid, cus);
    BH_REQUIRE(xcc_ws != nullptr, "lstm: missing XCD agreement workspace");
    BH_CHECK_HIP(hipMemsetAsync(xcc_ws, 0xFF, (size_t)n_rings * nsl * sizeof(int), stream));
    LstmWideArgs a{(const half_t*)gates_perm,
                   LstmArgs{nullptr, (const half_t*)whh_tiles, (half_t*)h_out, T, N, H, n_rings, reverse, err_flag, g_max_spins, xcc_ws,
                            force_slow & 1, force_slow >> 8},
                   (char*)ex, R};
    const int nks = H / 32;
    if (ex && arm) BH_CHECK_HIP(hipMemsetAsync(ex, 0xFF, (size_t)4 * R * 2 * nks * 1024, stream));
    const size_t lds = (size_t)2 * 2 * nks * 1024 + 4 * 16 * 8 * 2;
#define BH_LSTM_WIDE(NKS)                                                                                               \
    if (nks == NKS && ex) {                                                                                             \
        BH_CHECK_HIP(bh_max_lds((const void*)lstm_layer_wide_kernel<NKS, true>, (int)lds));                                                                    \
        hipLaunchKernelGGL((lstm_layer_wide_kernel<NKS, true>), dim3(grid), dim3(256), lds, stream, a);                  \
    } else if (nks == NKS) {                                                                                            \
        BH_CHECK_HIP(bh_max_lds((const void*)lstm_layer_wide_kernel<NKS, false>, (int)lds));                                                                    \
        hipLaunchKernelGGL((lstm_layer_wide_kernel<NKS, false>), dim3(grid), dim3(256), lds, stream, a);                 \
    } else
    if (nks == 32 && ex && ((force_slow >> 8) & 4)) {
        BH_CHECK_HIP(bh_max_lds((const void*)lstm_layer_wide_kernel<32, true, true>, (int)lds));
        hipLaunchKernelGGL((lstm_layer_wide_kernel<32, true, true>), dim3(grid), dim3(256), lds, stream, a);
    } else
    BH_LSTM_WIDE(20) BH_LSTM_WIDE(24) BH_LSTM_WIDE(28) BH_LSTM_WIDE(32)
    { BH_REQUIRE(false, "lstm: wide kernel has no instance for H=%d", H); }
#undef BH_LSTM_WIDE
    BH_CHECK_HIP(hipGetLastError());
    return 0;
}
