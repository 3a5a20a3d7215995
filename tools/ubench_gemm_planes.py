"""Pre-split-operand GEMM (las_gemm_planes) against the in-kernel split (las_gemm_f32, arithmetic mode 1) on the large GEMM shapes of
one P-config training step, plus the cost of las_split_planes itself.  One process, alternating: same box, same clocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from las_pytorch_amd import _cabi
if os.environ.get("LAS_ABL_LIB"): _cabi.LIB_PATH = os.path.abspath(os.environ["LAS_ABL_LIB"])      # an ablation build (-DLAS_PLANES_ABL=mask)
L = _cabi.lib()
L.las_gemm_set_arith(1)
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
def planes(X):
    R, C = X.shape
    out = torch.empty(L.las_planes_bytes(R, C), dtype=torch.uint8, device="cuda")
    _cabi.check(L.las_split_planes(X.data_ptr(), C, R, C, out.data_ptr(), C, _cabi.stream_ptr()))
    return out
def run(name, M, N, K, a_kc, b_kc, batch=1, splitk=0):
    A = torch.randn(batch * (M if a_kc else K), K if a_kc else M, device="cuda")
    Bm = torch.randn(batch * (N if b_kc else K), K if b_kc else N, device="cuda")
    C = torch.zeros(batch * M * N, device="cuda")
    lda = K if a_kc else M; ldb = K if b_kc else N
    Ap, Bp = planes(A), planes(Bm)
    f32 = lambda: _cabi.check(L.las_gemm_f32(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), None, None, M, N, K, lda, ldb, N, int(a_kc), int(b_kc),
                                             batch, M * K, N * K, M * N, splitk, 0, 0, _cabi.stream_ptr()))
    pl = lambda: _cabi.check(L.las_gemm_planes(Ap.data_ptr(), Bp.data_ptr(), C.data_ptr(), None, None, M, N, K, lda, ldb, N, int(a_kc), int(b_kc),
                                               batch, M * K, N * K, M * N, splitk, 0, 0, _cabi.stream_ptr()))
    sa = lambda: _cabi.check(L.las_split_planes(A.data_ptr(), A.shape[1], A.shape[0], A.shape[1], Ap.data_ptr(), A.shape[1], _cabi.stream_ptr()))
    sb = lambda: _cabi.check(L.las_split_planes(Bm.data_ptr(), Bm.shape[1], Bm.shape[0], Bm.shape[1], Bp.data_ptr(), Bm.shape[1], _cabi.stream_ptr()))
    t = [timed(f32), timed(pl), timed(f32), timed(pl)]
    us0, us1 = min(t[0], t[2]), min(t[1], t[3])
    fl = 2.0 * batch * M * N * K / 1e6
    print(f"{name:<26} M={M:<6} N={N:<5} K={K:<6} b={batch} akc={int(a_kc)} bkc={int(b_kc)}: split {us0:7.1f} us {fl/us0:6.1f} TF | planes {us1:7.1f} us {fl/us1:6.1f} TF"
          f" | split_planes A {timed(sa):6.1f} B {timed(sb):6.1f} us", flush=True)
if os.environ.get("QUICK"):
    print(os.environ.get("LAS_ABL_LIB", "product"))
    run("L1 fwd proj (2 dirs)", 6400, 1024, 1024, True, True, batch=2, splitk=1)
    run("L1 dW_ih", 1024, 1024, 6400, False, False)
    run("L1 dX", 6400, 1024, 2048, True, False, splitk=1)
    run("4096^3 NT", 4096, 4096, 4096, True, True, splitk=1)
    run("4096^3 TN", 4096, 4096, 4096, False, False, splitk=1)
    sys.exit(0)
for l, (BT, D) in enumerate([(12800, 160), (6400, 1024), (3200, 1024)]):
    run(f"L{l} fwd proj (2 dirs)", BT, 1024, D, True, True, batch=2, splitk=1)
    run(f"L{l} dW_ih", 1024, D, BT, False, False)
    if l > 0: run(f"L{l} dX", BT, D, 2048, True, False, splitk=1)
run("P = feat W_ctx^T", 3200, 2048, 512, True, True, splitk=1)
run("dctx = dG0 W_ctx", 4096, 512, 2048, True, False, splitk=1)
run("spl dW_ih1", 2048, 512, 4096, False, False)
run("4096^3 NT", 4096, 4096, 4096, True, True, splitk=1)
run("4096^3 NN", 4096, 4096, 4096, True, False, splitk=1)
run("4096^3 TN", 4096, 4096, 4096, False, False, splitk=1)
