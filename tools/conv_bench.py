#!/usr/bin/env python3
"""conv3 of the hac model (16 -> 384 channels, 19 taps, stride 6) alone: time-major (what the LSTM consumes) vs
chunk-major output layout. Separates the cost of the scattered 768-byte row writes from the cost of the arithmetic."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bonito_amd import _lib
lib = _lib.lib()
def run(N, L, Cin, Cout, K, stride, pad, act, tag):
    Lout = (L + 2 * pad - K) // stride + 1
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, K)) * 0.1).astype(np.float32)
    pk = np.zeros(lib.bh_conv1d_packed_halves(Cin, Cout, K), np.uint16)
    _lib.check(lib.bh_conv1d_pack(w.ctypes.data_as(C.c_void_p), Cin, Cout, K, pk.ctypes.data_as(C.c_void_p)), "pack")
    wpk = torch.from_numpy(pk.view(np.int16)).cuda()
    bias = torch.zeros(Cout, device="cuda")
    x = torch.randn(N, L, Cin, device="cuda").half()
    out = torch.empty(N * Lout * Cout, device="cuda", dtype=torch.half)
    for name, os_n, os_t in (("time-major", Cout, N * Cout), ("chunk-major", Lout * Cout, Cout)):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rep in range(3):
            ev[0].record()
            for _ in range(5):
                _lib.check(lib.bh_conv1d(_lib.ptr(x), _lib.ptr(wpk), _lib.ptr(bias), _lib.ptr(out), N, L, Cin, Cout, K, stride, pad,
                                         act, C.c_float(-1e30), C.c_float(1e30), os_n, os_t, _lib.stream_ptr()), "conv")
            ev[1].record(); torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 5
        print("%-34s %-12s %.3f ms per launch (%.2f TB/s of output)" % (tag, name, ms, N * Lout * Cout * 2 / (ms * 1e-3) / 1e12))


run(512, 10000, 16, 384, 19, 6, 9, 1, "conv3 (swish)")
run(512, 10000, 16, 384, 19, 6, 9, 0, "conv3, no activation")
run(512, 10000, 16, 96, 19, 6, 9, 1, "conv3 with 96 output channels")
run(512, 10000, 16, 384, 5, 6, 2, 1, "conv3 with 5 taps")
run(512, 10000, 16, 16, 5, 1, 2, 1, "conv2 (16 -> 16, 5 taps, stride 1)")
