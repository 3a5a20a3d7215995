"""Does a weight-gradient GEMM group hide under a persistent recurrence launch?  Times the layer-0 forward recurrence
(stream A), a Listener-layer dW group (stream B), a plain torch elementwise kernel (stream B) alone and concurrently."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import _cabi
L = _cabi.lib()
B, T, H = int(os.environ.get("B", 32)), int(os.environ.get("T", 400)), 256
g = torch.Generator().manual_seed(0)
w = [((torch.rand(4 * H, H, generator=g) * 2 - 1) / np.sqrt(H)).cuda() for _ in range(2)]
pre = torch.randn(2 * B * T * 4 * H, generator=g).cuda() * 0.1
gates = pre.clone(); out = torch.empty(B, T, 2 * H, device="cuda")
cbuf = torch.empty(2 * B * T * H, device="cuda"); hprev = torch.empty_like(cbuf)
xbuf = torch.empty(L.las_rec_xbuf_bytes(B, H) // 4 + 4, device="cuda"); err = _cabi.err_word("cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def rec():
    _cabi.check(L.las_pblstm_rec_fwd(_cabi.ptr(gates), _cabi.ptr(w[0]), _cabi.ptr(w[1]), _cabi.ptr(out), _cabi.ptr(cbuf), _cabi.ptr(hprev),
                                     B, T, H, _cabi.ptr(xbuf), _cabi.ptr(err), _cabi.FLAG_STASH, sa.cuda_stream))

K = B * T
probs = [(1024, 512, K), (1024, 256, K), (1024, 512, K), (1024, 256, K)]
descs = (_cabi.GemmDescC * len(probs))(); keep = []
for i, (M, N, Kk) in enumerate(probs):
    A = torch.randn(Kk, M, device="cuda"); Bm = torch.randn(Kk, N, device="cuda"); C = torch.zeros(M, N, device="cuda"); keep += [A, Bm, C]
    d = descs[i]; d.A, d.B, d.C, d.A2, d.B2 = A.data_ptr(), Bm.data_ptr(), C.data_ptr(), None, None
    d.M, d.N, d.K, d.K1 = M, N, Kk, 0; d.lda, d.ldb, d.ldc = M, N, N; d.a_kc, d.b_kc, d.accumulate, d.c_zeroed = 0, 0, 1, 0

def gemm():
    _cabi.check(L.las_gemm_f32_group(descs, len(probs), sb.cuda_stream))

big = torch.randn(64 << 20, device="cuda")
def elem():
    with torch.cuda.stream(sb):
        big.mul_(1.0001)

def wall(fns, n=10):
    for f in fns: f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        for f in fns: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

for name, fns in [("rec", [rec]), ("gemm", [gemm]), ("elem", [elem]), ("gemm+rec", [gemm, rec]), ("rec+gemm", [rec, gemm]), ("elem+rec", [elem, rec]),
                  ("rec+elem", [rec, elem]), ("gemm+elem", [gemm, elem])]:
    print(f"{name:10s} {wall(fns):.3f} ms per round", flush=True)
print("err", int(err[0]))
