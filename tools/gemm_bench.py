#!/usr/bin/env python3
"""Per-shape throughput of bh_linear for the transformer / CRF-head shapes: the automatic path (the four-wave kernel where it applies)
vs the eight-wave 256-tile kernel ("gemm_path" 3), the 128-tile kernels and the vendor library behind torch.matmul."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonito_amd import _lib, decode
INF = float("inf")
dev = torch.device("cuda", 0)
shapes = [(256000, 1536, 512, 0), (256000, 512, 512, 0), (256000, 4096, 512, 1), (256000, 512, 2048, 0),
          (256000, 1024, 512, 0), (512000, 4096, 512, 0), (853504, 1024, 384, 0), (853504, 4096, 1024, 0)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
lib = _lib.lib()
for M, N, K, gated in shapes:
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    w = (torch.randn(N, K, device=dev) * 0.2).half()
    ncol = N // 2 if gated else N
    out = torch.empty((M, ncol), dtype=torch.float16, device=dev)
    res = {}
    for path in (0, 32, 3, 2):
        decode.set_option("gemm_path", 0 if path == 32 else path)
        decode.set_option("gemm_tile16", 0 if path == 32 else 1)      # 32: the four-wave kernel on its 32x32x16 stream (rounds 4-5)
        def run():
            _lib.check(lib.bh_linear(_lib.ptr(x), _lib.ptr(w), None, _lib.ptr(out), M, N, K, K, K, ncol, 0, 1.0, -INF, INF,
                                     gated, 0, 0, 0, 0, _lib.stream_ptr()), "bh_linear")
        for _ in range(15): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        res[path] = (ms, 2.0 * M * N * K / ms / 1e9)
    decode.set_option("gemm_path", 0); decode.set_option("gemm_stagger", 0); decode.set_option("gemm_tile16", 1)
    # yardstick: the vendor library behind torch.matmul (hipBLASLt / rocBLAS), plain GEMM without the fused epilogue
    full = torch.empty((M, N), dtype=torch.float16, device=dev)
    for _ in range(15): torch.matmul(x, w.t(), out=full)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): torch.matmul(x, w.t(), out=full)
    e1.record(); torch.cuda.synchronize()
    lib_ms = e0.elapsed_time(e1) / 40
    del full
    lib_tf = 2.0 * M * N * K / lib_ms / 1e9
    print("M=%d N=%d K=%d gated=%d: auto %.3f ms %.0f TF/s (%.2f x lib) | 32x32x16 stream %.3f ms %.0f TF/s | 8-wave 256-tile %.3f ms %.0f TF/s | 128-tile %.3f ms %.0f TF/s | torch.matmul %.3f ms %.0f TF/s" % (
        M, N, K, gated, res[0][0], res[0][1], res[0][1] / lib_tf, res[32][0], res[32][1], res[3][0], res[3][1], res[2][0], res[2][1], lib_ms, lib_tf), flush=True)
    del x, w, out
