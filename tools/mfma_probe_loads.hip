// what do eight 1-KiB buffer loads per wave cost beside 64 MFMAs (one wave per SIMD, four waves per CU, every CU)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

#define MF(c) "v_mfma_f32_16x16x32_f16 %" #c ", %2, %3, %" #c "\n\t"
#define MF8 MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) MF(0) MF(1)
// MODE 0: MFMAs only; 1: burst of 8 loads behind the first 16 MFMAs (sc0 sc1); 2: the same, plain loads; 3: one load behind every 6 MFMAs (sc0 sc1);
// 4: one load behind every 6 MFMAs, plain; 5: burst, loads only on wave 0 (the other waves none); 6: LDS-DMA burst (sc0 sc1);
// 7: burst (sc0 sc1), the 32 workgroups of an XCD read THE SAME 32 KiB tile in the same order (what a ring's workgroups do with its h tile);
// 8: the same tile, every workgroup starts at another fragment (order rotated by the workgroup's index); 9: as 7 with plain loads
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const char* buf, float* res, long long* cyc, int reps) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * ((threadIdx.x + j) % 23) - 0.1f); b[j] = (_Float16)(0.02f * ((threadIdx.x * 5 + j) % 17) - 0.15f); }
    float4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    uint4_t q[8];
    for (int i = 0; i < 8; ++i) q[i] = uint4_t{0, 0, 0, 0};
    const bool shared_tile = MODE >= 7;
    const char* base = buf + (size_t)(shared_tile ? (blockIdx.x & 7) : blockIdx.x) * 65536;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 65536, 0x00020000);
    unsigned acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        const unsigned off = ((r & 1) * 32768) + wave * 8192 + lane * 16;
        if (MODE == 0) {
            asm volatile(MF8 MF8 MF8 MF8 MF8 MF8 MF8 MF8 : "+v"(c0), "+v"(c1) : "a"(a), "v"(b));
        } else if (MODE == 7 || MODE == 8 || MODE == 9) {
            asm volatile(MF8 MF8 : "+v"(c0), "+v"(c1) : "a"(a), "v"(b));
            const int rot = MODE == 8 ? (int)(blockIdx.x >> 3) : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned frag = (unsigned)(wave + 4 * ((i + rot) & 7));      // fragment wave + 4 kk of the tile, kk rotated
                q[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((r & 1) * 32768) + frag * 1024 + lane * 16, 0, MODE == 9 ? 0 : (int)0x80000010);
            }
            asm volatile(MF8 MF8 MF8 MF8 MF8 MF8 : "+v"(c0), "+v"(c1) : "a"(a), "v"(b));
        } else if (MODE == 1 || MODE == 2 || MODE == 5) {
            asm volatile(MF8 MF8 : "+v"(c0), "+v"(c1) : "a"(a), "v"(b));
            if (MODE != 5 || wave == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + i * 1024, 0, MODE == 2 ? 0 : (int)0x80000010);
            }
            asm volatile(MF8 MF8 MF8 MF8 MF8 MF8 : "+v"(c0), "+v"(c1) : "a"(a), "v"(b));
        } else if (MODE == 3 || MODE == 4) {
            asm volatile(MF8 MF8 : "+v"(c0), "+v"(c1) : "a"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                q[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + i * 1024, 0, MODE == 4 ? 0 : (int)0x80000010);
                asm volatile(MF(0) MF(1) MF(0) MF(1) MF(0) MF(1) : "+v"(c0), "+v"(c1) : "a"(a), "v"(b));
            }
        } else if (MODE == 6) {
            asm volatile(MF8 MF8 : "+v"(c0), "+v"(c1) : "a"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned ldsb = (unsigned)(size_t)(lds + wave * 8192 + i * 1024);
                asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen sc0 sc1 lds" :: "s"(__builtin_amdgcn_readfirstlane(ldsb)), "v"(off + i * 1024), "s"(rs) : "memory", "m0");
            }
            asm volatile(MF8 MF8 MF8 MF8 MF8 MF8 : "+v"(c0), "+v"(c1) : "a"(a), "v"(b));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc |= q[i].x | q[i].w;
        for (int i = 0; i < 4; ++i) { c0[i] *= 0.5f; c1[i] *= 0.5f; }
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
    if (blockIdx.x == 0) res[threadIdx.x] = c0[0] + c1[1] + (float)(acc & 1) + (MODE == 6 ? (float)lds[threadIdx.x] : 0.f);
}

template <int MODE>
void run(const char* name, const char* dbuf, float* dres, long long* dcyc, int grid) {
    const int reps = 4000;
    probe<MODE><<<grid, 256>>>(dbuf, dres, dcyc, reps);
    hipError_t e = hipDeviceSynchronize();
    long long c[4];
    (void)hipMemcpy(c, dcyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-74s grid %3d: %7.0f cycles per round of 64 MFMAs (wave 0; waves 1-3: %.0f %.0f %.0f)%s\n", name, grid, c[0] / (double)reps, c[1] / (double)reps,
           c[2] / (double)reps, c[3] / (double)reps, e == hipSuccess ? "" : " ERROR");
}

int main() {
    char* dbuf; float* dres; long long* dcyc;
    (void)hipMalloc(&dbuf, 256 * 65536); (void)hipMemset(dbuf, 1, 256 * 65536); (void)hipMalloc(&dres, 1024); (void)hipMalloc(&dcyc, 64);
    for (int grid : {1, 256}) {
        run<0>("64 MFMAs", dbuf, dres, dcyc, grid);
        run<1>("+ a burst of 8 x 1 KiB loads per wave, sc0 sc1", dbuf, dres, dcyc, grid);
        run<2>("+ a burst of 8 x 1 KiB loads per wave, plain", dbuf, dres, dcyc, grid);
        run<3>("+ 8 loads, one behind every 6 MFMAs, sc0 sc1", dbuf, dres, dcyc, grid);
        run<4>("+ 8 loads, one behind every 6 MFMAs, plain", dbuf, dres, dcyc, grid);
        run<5>("+ a burst of 8 loads on wave 0 only, sc0 sc1", dbuf, dres, dcyc, grid);
        run<6>("+ a burst of 8 LDS-DMA pieces per wave, sc0 sc1", dbuf, dres, dcyc, grid);
        run<7>("+ burst, the 32 workgroups of an XCD read the SAME tile, same order", dbuf, dres, dcyc, grid);
        run<8>("+ burst, the same tile, order rotated by the workgroup index", dbuf, dres, dcyc, grid);
        run<9>("+ burst, the same tile, same order, plain loads", dbuf, dres, dcyc, grid);
    }
    return 0;
}
