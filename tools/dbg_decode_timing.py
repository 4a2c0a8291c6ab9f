import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonito_amd import decode, _lib
N, T, C = 512, 1667, 256
sc = (torch.randn(N, T, C, device="cuda") * 2.5).clamp(-5, 5).half()
dec = decode.CRFDecoder(N, T, C, "cuda:0", mode="beam")
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
lib = _lib.lib()
def kernels_only():
    out = dec.dev_out
    _lib.check(lib.bh_beam_search(_lib.ptr(sc), N, T, dec.sl, 32, 100.0, 2.0, 1.0, 0.0, _lib.ptr(dec.ws), _lib.ptr(out[0]),
                                  _lib.ptr(out[1]), _lib.ptr(out[2]), None, _lib.stream_ptr("cuda:0")))
print("kernels only (ctx buffers)   %.2f ms" % t(kernels_only))
print("pinned copy_ non_blocking    %.2f ms" % t(lambda: dec.host_out.copy_(dec.dev_out, non_blocking=True)))
print("pinned sliced copy_          %.2f ms" % t(lambda: dec.host_out[:, :N].copy_(dec.dev_out[:, :N], non_blocking=True)))
print(".cpu()                       %.2f ms" % t(lambda: dec.dev_out.cpu()))
ws2 = torch.empty_like(dec.ws)
def kernels_ws2():
    out = dec.dev_out
    _lib.check(lib.bh_beam_search(_lib.ptr(sc), N, T, dec.sl, 32, 100.0, 2.0, 1.0, 0.0, _lib.ptr(ws2), _lib.ptr(out[0]),
                                  _lib.ptr(out[1]), _lib.ptr(out[2]), None, _lib.stream_ptr("cuda:0")))
print("kernels only (fresh ws)      %.2f ms" % t(kernels_ws2))
print("alloc api                    %.2f ms" % t(lambda: decode.beam_search(sc)))
print("ctx submit+result            %.2f ms" % t(lambda: dec.submit(sc).result()))
print("ws bytes", dec.ws.numel(), "ptr align", dec.ws.data_ptr() % 256, ws2.data_ptr() % 256)
