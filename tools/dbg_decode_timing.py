import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bonito_amd import decode, _lib
N, T, C = 512, 1667, 256
sc = (torch.randn(N, T, C, device="cuda") * 2.5).clamp(-5, 5).half()
dec = decode.CRFDecoder(N, T, C, "cuda:0", mode="beam")
lib = _lib.lib()
def now():
    return time.perf_counter()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = now()
    tk = dec.submit(sc)
    t1 = now()
    dec.done.synchronize()
    t2 = now()
    r = tk.result()
    t3 = now()
    print("submit(host) %.2f  wait %.2f  result %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
# variant: same but without the pinned copy
for rep in range(2):
    torch.cuda.synchronize(); t0 = now()
    out = dec.dev_out
    _lib.check(lib.bh_beam_search(_lib.ptr(sc), N, T, dec.sl, 32, 100.0, 2.0, 1.0, 0.0, _lib.ptr(dec.ws), _lib.ptr(out[0]),
                                  _lib.ptr(out[1]), _lib.ptr(out[2]), None, _lib.stream_ptr("cuda:0")))
    t1 = now()
    dec.host_out[:, :N].copy_(out[:, :N], non_blocking=True)
    t2 = now()
    dec.done.record(torch.cuda.current_stream())
    t3 = now()
    dec.done.synchronize()
    t4 = now()
    print("launch %.2f  copy-enqueue %.2f  record %.2f  wait %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
