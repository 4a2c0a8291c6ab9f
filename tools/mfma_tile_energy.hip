// Round 6 (review item 4): does the 32x32x16 tile buy throughput under the board's power cap where the operands already sit in registers -
// the premise of a "32-chunk ring" for the recurrent kernel? One wave per SIMD on every CU (the recurrent kernel's occupancy), A in the
// accumulation file, B in VGPRs, long enough for the power management to settle (each variant runs ~1.5 s of back-to-back launches):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_tile_energy tools/mfma_tile_energy.hip && /tmp/mfma_tile_energy
// Variants: bare MFMAs of either shape (the same FLOPs per round), and the same with the recurrent kernel's vector side work beside them
// (119 vector instructions per 72 MFMAs of 16x16x32 = 1.65 per MFMA, as fmas on private registers).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <int SHAPE, int VALU>      // SHAPE 0: 16x16x32 (16 per round), 1: 32x32x16 (8 per round); VALU: fmas per round
__global__ __launch_bounds__(256, 1) void probe(float* res, long long* cyc, int rounds) {
    half8_t a[4], b;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.01f * ((threadIdx.x * 7 + i * 3 + j) % 23) - 0.1f);
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.02f * ((threadIdx.x * 5 + j) % 17) - 0.15f);
    float4_t c0 = {0.5f, 0.25f, 0.125f, 1.f}, c1 = {0.1f, 0.2f, 0.3f, 0.4f};
    float16_t d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0.01f * i; d1[i] = 0.02f * i; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        if (SHAPE == 0) {
#define M16(c, i) "v_mfma_f32_16x16x32_f16 %" #c ", %" #i ", %6, %" #c "\n\t"
            asm volatile(M16(0,2) M16(1,3) M16(0,4) M16(1,5) M16(0,2) M16(1,3) M16(0,4) M16(1,5) M16(0,2) M16(1,3) M16(0,4) M16(1,5) M16(0,2) M16(1,3) M16(0,4) M16(1,5)
                         : "+v"(c0), "+v"(c1) : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]), "v"(b));
        } else {
#define M32(c, i) "v_mfma_f32_32x32x16_f16 %" #c ", %" #i ", %6, %" #c "\n\t"
            asm volatile(M32(0,2) M32(1,3) M32(0,4) M32(1,5) M32(0,2) M32(1,3) M32(0,4) M32(1,5)
                         : "+v"(d0), "+v"(d1) : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]), "v"(b));
        }
#pragma unroll
        for (int k = 0; k < VALU; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 0.999f, 1e-3f);
        if ((r & 63) == 63) {          // keep the accumulators bounded
            for (int i = 0; i < 4; ++i) { c0[i] *= 1e-3f; c1[i] *= 1e-3f; }
            for (int i = 0; i < 16; ++i) { d0[i] *= 1e-3f; d1[i] *= 1e-3f; }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = c0[0] + c1[1] + d0[3] + d1[7];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (blockIdx.x == 0) res[threadIdx.x] = s;
}

template <int SHAPE, int VALU>
void run(const char* name, float* dres, long long* dcyc, int grid) {
    const int rounds = 20000;                                   // 20000 x 16 MFMA-16 equivalents = 5.2e9 FLOP per wave per launch
    const double flop_per_launch = (double)rounds * 16 * 16384.0 * 4 * grid;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) probe<SHAPE, VALU><<<grid, 256>>>(dres, dcyc, rounds);      // settle (clock / power management)
    (void)hipDeviceSynchronize();
    int launches = 0;
    const auto h0 = std::chrono::steady_clock::now();
    (void)hipEventRecord(e0);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count() < 1.5) {
        for (int k = 0; k < 10; ++k) probe<SHAPE, VALU><<<grid, 256>>>(dres, dcyc, rounds);
        launches += 10;
        (void)hipStreamSynchronize(0);
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    (void)hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    const double per_launch_s = ms * 1e-3 / launches;
    printf("%-58s grid %3d: %7.1f TFLOP/s, %.2f cycles per 16x16x32-equivalent, kernel clock %.2f GHz (%d launches)\n", name, grid,
           flop_per_launch / per_launch_s / 1e12, c / (16.0 * rounds), c / per_launch_s / 1e9, launches);
    fflush(stdout);
}

int main() {
    float* dres; long long* dcyc;
    (void)hipMalloc(&dres, 4096); (void)hipMalloc(&dcyc, 64);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int grid : {1, cus}) {
        run<0, 0>("16x16x32 f16, bare", dres, dcyc, grid);
        run<1, 0>("32x32x16 f16, bare", dres, dcyc, grid);
        run<0, 26>("16x16x32 f16 + 1.65 vector fmas per MFMA", dres, dcyc, grid);
        run<1, 26>("32x32x16 f16 + the same vector work per FLOP", dres, dcyc, grid);
    }
    return 0;
}
