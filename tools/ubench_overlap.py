"""Does a latency-bound persistent recurrence launch overlap with an MFMA GEMM on another stream?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import _cabi
L = _cabi.lib()
B, T, H = 32, 400, 256
g = torch.Generator().manual_seed(0)
w = [((torch.rand(4 * H, H, generator=g) * 2 - 1) / np.sqrt(H)).cuda() for _ in range(2)]
pre = torch.randn(2 * B * T * 4 * H, generator=g).cuda()
gates = pre.clone(); out = torch.empty(B, T, 2 * H, device="cuda")
cbuf = torch.empty(2 * B * T * H, device="cuda"); hprev = torch.empty_like(cbuf)
xbuf = torch.empty(L.las_rec_xbuf_bytes(B, H) // 4 + 4, device="cuda"); err = _cabi.err_word("cuda")
M, N, K = 1024, 1024, 6400
A = torch.randn(M * K, device="cuda"); Bm = torch.randn(N * K, device="cuda"); C = torch.zeros(M * N, device="cuda")
def rec(stream):
    _cabi.check(L.las_pblstm_rec_fwd(_cabi.ptr(gates), _cabi.ptr(w[0]), _cabi.ptr(w[1]), _cabi.ptr(out), _cabi.ptr(cbuf), _cabi.ptr(hprev),
                                     B, T, H, _cabi.ptr(xbuf), _cabi.ptr(err), 1, stream.cuda_stream))
def gemm(stream, n=3):
    for _ in range(n):
        _cabi.check(L.las_gemm_f32(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), None, None, M, N, K, M, N, N, 0, 0, 1, 0, 0, 0, 1, 0, 0, stream.cuda_stream))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
t_rec = timeit(lambda: rec(s1))
t_gemm = timeit(lambda: gemm(s2))
t_both = timeit(lambda: (rec(s1), gemm(s2)))
t_both2 = timeit(lambda: (gemm(s2), rec(s1)))
print(f"rec alone {t_rec:.3f} ms, 3 GEMMs alone {t_gemm:.3f} ms, concurrent (rec first) {t_both:.3f} ms, (gemm first) {t_both2:.3f} ms, err={int(err[0])}")
