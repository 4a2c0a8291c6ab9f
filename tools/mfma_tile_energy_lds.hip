// Companion of tools/mfma_tile_energy.hip: the same question for a GEMM-shaped inner loop - a 128 x 128 wave tile fed from LDS, one wave
// per SIMD, 256 accumulator registers: per 32-deep slab 16 fragment reads (ds_read_b128, 1 KiB each) and either 64 MFMAs 16x16x32 or
// 32 MFMAs 32x32x16 (the LDS traffic per FLOP is the same for both shapes: it is set by the wave tile). Under the board's power cap,
// which shape delivers more FLOP/s?      hipcc --offload-arch=gfx950 -O3 -o /tmp/mte_lds tools/mfma_tile_energy_lds.hip && /tmp/mte_lds
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void probe(float* res, long long* cyc, int rounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 64 KiB of operand data
    for (int i = threadIdx.x; i < 65536 / 2; i += 256) ((_Float16*)smem)[i] = (_Float16)(0.01f * ((i * 7) % 23) - 0.1f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const char* base = smem + lane * 16;
    float4_t c16[8][8];
    float16_t c32[4][4];
    if (SHAPE == 0) { for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) c16[i][j] = float4_t{0.f, 0.f, 0.f, 0.f}; }
    else { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c32[i][j][e] = 0.f; }
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        const char* p = base + (r & 3) * 16384;
        if (SHAPE == 0) {
            half8_t a[8], b[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] = *(const half8_t*)(p + i * 1024); b[i] = *(const half8_t*)(p + 8192 + i * 1024); }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) c16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], c16[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8_t a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[i] = *(const half8_t*)(p + ks * 4096 + i * 1024); b[i] = *(const half8_t*)(p + 8192 + ks * 4096 + i * 1024); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) c32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], c32[i][j], 0, 0, 0);
            }
        }
        if ((r & 31) == 31) {
            if (SHAPE == 0) { for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) c16[i][j] *= 1e-3f; }
            else { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c32[i][j] *= 1e-3f; }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = 0.f;
    if (SHAPE == 0) { for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) s += c16[i][j][0] + c16[i][j][3]; }
    else { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += c32[i][j][0] + c32[i][j][7]; }
    if (blockIdx.x == 0) res[threadIdx.x] = s;
}

template <int SHAPE>
void run(const char* name, float* dres, long long* dcyc, int grid) {
    const int rounds = 6000;                                    // x 64 MFMA-16 equivalents = 6.3e9 FLOP per wave per launch
    const double flop_per_launch = (double)rounds * 64 * 16384.0 * 4 * grid;
    (void)hipFuncSetAttribute((const void*)probe<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) probe<SHAPE><<<grid, 256, 65536>>>(dres, dcyc, rounds);
    (void)hipDeviceSynchronize();
    int launches = 0;
    const auto h0 = std::chrono::steady_clock::now();
    (void)hipEventRecord(e0);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count() < 2.0) {
        for (int k = 0; k < 10; ++k) probe<SHAPE><<<grid, 256, 65536>>>(dres, dcyc, rounds);
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
    printf("%-44s grid %3d: %7.1f TFLOP/s, %.2f cycles per 16x16x32-equivalent, kernel clock %.2f GHz\n", name, grid,
           flop_per_launch / per_launch_s / 1e12, c / (64.0 * rounds), c / per_launch_s / 1e9);
    fflush(stdout);
}

int main() {
    float* dres; long long* dcyc;
    (void)hipMalloc(&dres, 4096); (void)hipMalloc(&dcyc, 64);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    for (int rep = 0; rep < 2; ++rep)
        for (int grid : {prop.multiProcessorCount}) {
            run<0>("128x128 wave tile from LDS, 16x16x32 f16", dres, dcyc, grid);
            run<1>("128x128 wave tile from LDS, 32x32x16 f16", dres, dcyc, grid);
        }
    return 0;
}
