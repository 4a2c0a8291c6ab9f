import sys, torch
sys.path.insert(0, ".")
from bonito_amd import _lib, decode
INF = float("inf")
dev = torch.device("cuda", 0)
def lin(x, w, b, act=0, gated=0):
    M, K = x.shape; N = w.shape[0]; ncol = N // 2 if gated else N
    out = torch.zeros((M, ncol), dtype=torch.float16, device=dev)
    _lib.check(_lib.lib().bh_linear(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), M, N, K, K, K, ncol, act, 1.0, -INF, INF, gated, 0, 0, 0, 0, _lib.stream_ptr()), "lin")
    torch.cuda.synchronize(); return out
for (M, N, K) in [(300, 256, 256), (256, 256, 256), (512, 256, 256), (300, 256, 512), (1007, 512, 384), (4096, 512, 512)]:
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * 0.5).half().to(dev); w = (torch.randn(N, K, generator=g) * 0.2).half().to(dev); b = torch.randn(N, generator=g).to(dev)
    decode.set_option("gemm_path", 5)
    outs = [lin(x, w, b) for _ in range(6)]
    decode.set_option("gemm_path", 2)
    ref = lin(x, w, b)
    decode.set_option("gemm_path", 0)
    for i, o in enumerate(outs):
        bad = ((o.float() - ref.float()).abs() > 0.05).nonzero()
        print(M, N, K, "run", i, "bad elements vs 128-tile kernel:", bad.shape[0], "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:24])
