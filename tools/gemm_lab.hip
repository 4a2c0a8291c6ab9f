// Stand-alone laboratory for the linear-layer GEMM kernels (tools/gemm_lab.py builds one binary per variant of the generated K-tile
// stream): compiles bonito_amd/csrc/gemm.hip as it is, with BH_GEMM_STATS (cycle stamps of workgroups 0 and 133 of gemm_w4_kernel),
// runs bh_k_linear on the shapes of tools/gemm_bench.py with random operands and prints time, TFLOP/s, the cycles per K-tile and per
// epilogue and the shader clock (cycle counter / 100 MHz real-time counter). Nothing here is part of the library.
#define BH_GEMM_STATS 1
#include "../bonito_amd/csrc/gemm.hip"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

void bh_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
}

// (the library's helpers of engine.cpp, restated for the stand-alone binary)
hipError_t bh_max_lds(const void* fn, int bytes) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
int bh_cu_count() {
    static int n = 0;
    if (!n) { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 256; n = prop.multiProcessorCount; }
    return n;
}

__global__ void fill_kernel(bh::half_t* p, size_t n, float scale, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        float acc = 0.0f;
        for (int k = 0; k < 4; ++k) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; acc += (float)(h & 0xffff) * (1.0f / 65536.0f); }
        p[i] = (bh::half_t)((acc - 2.0f) * 1.732f * scale);      // sum of four uniforms: ~N(0, scale^2)
    }
}

int main(int argc, char** argv) {
    struct Shape { long M; int N, K, gated; };
    std::vector<Shape> shapes = {{256000, 1536, 512, 0}, {256000, 512, 512, 0}, {256000, 4096, 512, 1}, {256000, 512, 2048, 0},
                                 {256000, 1024, 512, 0}, {512000, 4096, 512, 0}, {853504, 1024, 384, 0}, {853504, 4096, 1024, 0}};
    if (argc > 1) {
        shapes.clear();
        for (int i = 1; i + 3 < argc + 0; i += 4) shapes.push_back({atol(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3])});
    }
    const int path = getenv("LAB_PATH") ? atoi(getenv("LAB_PATH")) : 0;
    bh_k_linear_force_v1(path);
    if (getenv("LAB_GF")) bh::g_w4_gf = atoi(getenv("LAB_GF"));
    if (getenv("LAB_STAGGER")) bh_k_linear_stagger(atoi(getenv("LAB_STAGGER")));
    if (getenv("LAB_ORDER")) bh_k_linear_order(atoi(getenv("LAB_ORDER")));
    if (getenv("LAB_T16")) bh_k_linear_tile16(atoi(getenv("LAB_T16")));        // gemm_w4_kernel around the 16x16x32 K-tile stream
    const int reps = getenv("LAB_REPS") ? atoi(getenv("LAB_REPS")) : 40, warm = getenv("LAB_WARM") ? atoi(getenv("LAB_WARM")) : 25;
    unsigned long long* dbg = nullptr;
    hipMalloc((void**)&dbg, 16 * 8);
    bh::g_gemm_dbg = dbg;
    for (auto& s : shapes) {
        bh::half_t *x, *w, *out;
        const int ncol = s.gated ? s.N / 2 : s.N;
        hipMalloc((void**)&x, (size_t)s.M * s.K * 2); hipMalloc((void**)&w, (size_t)s.N * s.K * 2); hipMalloc((void**)&out, (size_t)s.M * ncol * 2);
        fill_kernel<<<2048, 256>>>(x, (size_t)s.M * s.K, 0.5f, 1u);
        fill_kernel<<<2048, 256>>>(w, (size_t)s.N * s.K, 0.2f, 7u);
        auto run = [&]() {
            return bh_k_linear(x, w, nullptr, out, (int)s.M, s.N, s.K, s.K, s.K, ncol, 0, 1.0f, -INFINITY, INFINITY, s.gated, 0, 0, 0, 0, nullptr, nullptr, 0);
        };
        for (int i = 0; i < warm; ++i) if (run()) return 1;      // (the clock needs ~10 ms of load to settle)
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < reps; ++i) run();
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        unsigned long long h[16] = {0};
        hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        const double nk = s.K / 64.0;
        printf("M=%ld N=%d K=%d g=%d: %.3f ms %.0f TF/s", s.M, s.N, s.K, s.gated, ms, 2.0 * s.M * s.N * s.K / ms / 1e9);
        for (int b = 0; b < 2; ++b) {
            const unsigned long long* d = h + 8 * b;
            if (d[4]) printf(" | wg%d: %.0f cyc/K-tile, epilogue %.0f, total %.0f/tile, %.2f GHz, %llu tiles", b ? 133 : 0, d[0] / (double)d[4] / nk,
                             d[1] / (double)d[4], d[2] / (double)d[4], d[2] / (double)d[3] * 0.1, d[4]);
        }
        printf("\n"); fflush(stdout);
        hipFree(x); hipFree(w); hipFree(out);
    }
    return 0;
}
