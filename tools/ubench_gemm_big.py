"""256-tile GEMM (gemm_big.hip, option GEMM_BIG) against the 128-tile kernel on the GEMM launches of one P-config training step and on large
squares; one process, alternating (same box, same clocks).  GEMM_BIG: 0 never, 1 automatic (the product's choice), 2 whenever legal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from las_pytorch_amd import _cabi
if os.environ.get("LAS_ABL_LIB"): _cabi.LIB_PATH = os.path.abspath(os.environ["LAS_ABL_LIB"])
L = _cabi.lib()
L.las_gemm_set_arith(1)
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
def run(name, M, N, K, a_kc, b_kc, batch=1, splitk=0):
    A = torch.randn(batch * M * K, device="cuda"); Bm = torch.randn(batch * N * K, device="cuda"); C = torch.zeros(batch * M * N, device="cuda")
    lda = K if a_kc else M; ldb = K if b_kc else N
    call = lambda: _cabi.check(L.las_gemm_f32(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), None, None, M, N, K, lda, ldb, N, int(a_kc), int(b_kc),
                                              batch, M * K, N * K, M * N, splitk, 0, 0, _cabi.stream_ptr()))
    res = {}
    for rnd in range(2):
        for big in (0, 1, 2):
            _cabi.set_option("GEMM_BIG", big)
            us = timed(call)
            res[big] = min(res.get(big, (1e9, ""))[0], us), _cabi.last_path(_cabi.PATH_GEMM)
    fl = 2.0 * batch * M * N * K / 1e6
    print(f"{name:<24} M={M:<6} N={N:<5} K={K:<6} b={batch} {int(a_kc)}{int(b_kc)}: " +
          " | ".join(f"BIG={b} {res[b][0]:7.1f} us {fl/res[b][0]:6.1f} TF ({res[b][1]})" for b in (0, 1, 2)), flush=True)
def run_group(name, probs):
    """probs: (M, N, K) of row-contiguous weight-gradient problems, outputs pre-zeroed"""
    bufs, descs = [], (_cabi.GemmDescC * len(probs))()
    for i, (M, N, K) in enumerate(probs):
        A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); C = torch.zeros(M, N, device="cuda")
        bufs.append((A, B, C))
        descs[i] = _cabi.GemmDescC(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, None, M, N, K, 0, M, N, N, 0, 0, 1, 0, 0)
    call = lambda: _cabi.check(L.las_gemm_f32_group(descs, len(probs), _cabi.stream_ptr()))
    res = {}
    for rnd in range(2):
        for big in (0, 1):
            _cabi.set_option("GEMM_BIG", big)
            us = timed(call)
            res[big] = min(res.get(big, (1e9, ""))[0], us), _cabi.last_path(_cabi.PATH_GEMM)
    fl = sum(2.0 * M * N * K for M, N, K in probs) / 1e6
    print(f"{name:<24} {len(probs)} GEMMs: " + " | ".join(f"BIG={b} {res[b][0]:7.1f} us {fl/res[b][0]:6.1f} TF ({res[b][1]})" for b in (0, 1)), flush=True)
for l, (BT, D) in enumerate([(12800, 160), (6400, 1024), (3200, 1024)]):
    run(f"L{l} fwd proj (2 dirs)", BT, 1024, D, True, True, batch=2, splitk=1)
    if l > 0: run(f"L{l} dX", BT, D, 2048, True, False, splitk=1)
    run_group(f"L{l} dW group", [(1024, D, BT), (1024, 256, BT)] * 2)
run("P = feat W_ctx^T", 3200, 2048, 512, True, True, splitk=1)
run("dctx = dG0 W_ctx", 4096, 512, 2048, True, False, splitk=1)
run_group("speller dW group", [(2048, 512, 4096), (2048, 32, 4096), (2048, 512, 4064), (2048, 512, 4096), (2048, 512, 4064), (32, 512, 4096), (32, 512, 4096), (64, 512, 4096)])
if not os.environ.get("QUICK"):
    run("B=128 L1 fwd proj", 200 * 128, 1024, 1024, True, True, batch=2, splitk=1)
    run("4096^3 NT", 4096, 4096, 4096, True, True, splitk=1)
    run("4096^3 NN", 4096, 4096, 4096, True, False, splitk=1)
    run("4096^3 TN", 4096, 4096, 4096, False, False, splitk=1)
_cabi.set_option("GEMM_BIG", 1)
