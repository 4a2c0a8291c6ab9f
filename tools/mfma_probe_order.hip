// are MFMAs with alternating accumulators (c0, c1, c0, c1 ...; SrcC = the result of the MFMA before the previous one) interlocked by the
// hardware, i.e. bit-identical to the tile-major order (all of c0, then all of c1), and what does an s_nop between them cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

#define M(c, i) "v_mfma_f32_16x16x32_f16 %" #c ", %" #i ", %10, %" #c "\n\t"
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* res, long long* cyc, int reps) {
    half8_t a[8], b;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.01f * ((threadIdx.x * 7 + i * 3 + j) % 23) - 0.1f);
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.02f * ((threadIdx.x * 5 + j) % 17) - 0.15f);
    float4_t c0 = {0.5f, 0.25f, 0.125f, 1.f}, c1 = {0.1f, 0.2f, 0.3f, 0.4f};
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0)        // tile-major: c0 x 8, then c1 x 8
            asm volatile(M(0,2) M(0,3) M(0,4) M(0,5) M(0,6) M(0,7) M(0,8) M(0,9) M(1,2) M(1,3) M(1,4) M(1,5) M(1,6) M(1,7) M(1,8) M(1,9)
                         : "+v"(c0), "+v"(c1) : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]), "a"(a[4]), "a"(a[5]), "a"(a[6]), "a"(a[7]), "v"(b));
        else if (MODE == 1)   // alternating
            asm volatile(M(0,2) M(1,2) M(0,3) M(1,3) M(0,4) M(1,4) M(0,5) M(1,5) M(0,6) M(1,6) M(0,7) M(1,7) M(0,8) M(1,8) M(0,9) M(1,9)
                         : "+v"(c0), "+v"(c1) : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]), "a"(a[4]), "a"(a[5]), "a"(a[6]), "a"(a[7]), "v"(b));
        else if (MODE == 2)   // pairs
            asm volatile(M(0,2) M(0,3) M(1,2) M(1,3) M(0,4) M(0,5) M(1,4) M(1,5) M(0,6) M(0,7) M(1,6) M(1,7) M(0,8) M(0,9) M(1,8) M(1,9)
                         : "+v"(c0), "+v"(c1) : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]), "a"(a[4]), "a"(a[5]), "a"(a[6]), "a"(a[7]), "v"(b));
        else if (MODE == 3)   // alternating with an s_nop 0 behind every pair (what hipcc emits between inline-asm MFMAs)
            asm volatile(M(0,2) M(1,2) "s_nop 0\n\t" M(0,3) M(1,3) "s_nop 0\n\t" M(0,4) M(1,4) "s_nop 0\n\t" M(0,5) M(1,5) "s_nop 0\n\t" M(0,6) M(1,6) "s_nop 0\n\t" M(0,7) M(1,7) "s_nop 0\n\t" M(0,8) M(1,8) "s_nop 0\n\t" M(0,9) M(1,9) "s_nop 0\n\t"
                         : "+v"(c0), "+v"(c1) : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]), "a"(a[4]), "a"(a[5]), "a"(a[6]), "a"(a[7]), "v"(b));
        else if (MODE == 4)   // alternating with a ds_read-like filler: s_waitcnt + v_mov behind every pair
            asm volatile(M(0,2) M(1,2) "s_waitcnt lgkmcnt(0)\n\t" M(0,3) M(1,3) "s_waitcnt lgkmcnt(0)\n\t" M(0,4) M(1,4) "s_waitcnt lgkmcnt(0)\n\t" M(0,5) M(1,5) "s_waitcnt lgkmcnt(0)\n\t" M(0,6) M(1,6) "s_waitcnt lgkmcnt(0)\n\t" M(0,7) M(1,7) "s_waitcnt lgkmcnt(0)\n\t" M(0,8) M(1,8) "s_waitcnt lgkmcnt(0)\n\t" M(0,9) M(1,9) "s_waitcnt lgkmcnt(0)\n\t"
                         : "+v"(c0), "+v"(c1) : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]), "a"(a[4]), "a"(a[5]), "a"(a[6]), "a"(a[7]), "v"(b));
        else if (MODE == 5)   // s_nop 0 between the two MFMAs of a pair (in front of an independent one)
            asm volatile(M(0,2) "s_nop 0\n\t" M(1,2) M(0,3) "s_nop 0\n\t" M(1,3) M(0,4) "s_nop 0\n\t" M(1,4) M(0,5) "s_nop 0\n\t" M(1,5) M(0,6) "s_nop 0\n\t" M(1,6) M(0,7) "s_nop 0\n\t" M(1,7) M(0,8) "s_nop 0\n\t" M(1,8) M(0,9) "s_nop 0\n\t" M(1,9)
                         : "+v"(c0), "+v"(c1) : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]), "a"(a[4]), "a"(a[5]), "a"(a[6]), "a"(a[7]), "v"(b));
        // keep the values bounded: scale down (same in every mode)
        for (int i = 0; i < 4; ++i) { c0[i] *= 0.5f; c1[i] *= 0.5f; }
    }
    long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    if (blockIdx.x == 0) for (int i = 0; i < 4; ++i) { res[threadIdx.x * 8 + i] = c0[i]; res[threadIdx.x * 8 + 4 + i] = c1[i]; }
}

template <int MODE>
void run(const char* name, float* dres, long long* dcyc, float* ref, int grid) {
    const int reps = 4000;
    probe<MODE><<<grid, 256>>>(dres, dcyc, reps);
    (void)hipDeviceSynchronize();
    static float h[2048]; long long c[4];
    (void)hipMemcpy(h, dres, sizeof(h), hipMemcpyDeviceToHost); (void)hipMemcpy(c, dcyc, sizeof(c), hipMemcpyDeviceToHost);
    if (MODE == 0) memcpy(ref, h, sizeof(h));
    printf("%-62s grid %3d: %6.2f cycles per MFMA; results %s the tile-major order (sample %.6g)\n", name, grid, c[0] / (16.0 * reps),
           memcmp(ref, h, sizeof(h)) == 0 ? "BIT-IDENTICAL to" : "DIFFER from", h[5]);
}

int main() {
    float* dres; long long* dcyc; static float ref[2048];
    (void)hipMalloc(&dres, 2048 * 4); (void)hipMalloc(&dcyc, 64);
    for (int grid : {1, 256}) {
        run<0>("tile-major (c0 x 8, c1 x 8)", dres, dcyc, ref, grid);
        run<1>("alternating (c0, c1, c0, c1 ...)", dres, dcyc, ref, grid);
        run<2>("pairs (c0, c0, c1, c1, ...)", dres, dcyc, ref, grid);
        run<3>("alternating + s_nop 0 behind every pair", dres, dcyc, ref, grid);
        run<4>("alternating + s_waitcnt behind every pair", dres, dcyc, ref, grid);
        run<5>("alternating + s_nop 0 inside every pair", dres, dcyc, ref, grid);
    }
    return 0;
}
