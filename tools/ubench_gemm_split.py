"""fp32-MFMA GEMM (arith 0) against the split-operand bf16-MFMA GEMM (arith 1): error of both against float64 and time, on the
operand layouts and shapes of the training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from las_pytorch_amd import _cabi
L = _cabi.lib()

def gemm(A, B, C, M, N, K, a_kc, b_kc, splitk=0, batch=1):
    lda = K if a_kc else M; ldb = K if b_kc else N
    _cabi.check(L.las_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, None, M, N, K, lda, ldb, N, int(a_kc), int(b_kc),
                               batch, M * K, N * K, M * N, splitk, 0, 0, _cabi.stream_ptr()))

def run(name, M, N, K, a_kc, b_kc, splitk=0, reps=20, check=True, scale=None):
    g = torch.Generator(device="cuda"); g.manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M * K, device="cuda", generator=g); B = torch.randn(N * K, device="cuda", generator=g)
    if scale is not None:       # wide dynamic range: per-element exponent jitter
        A = A * torch.exp2(torch.randint(-scale, scale, (M * K,), device="cuda", generator=g).float())
        B = B * torch.exp2(torch.randint(-scale, scale, (N * K,), device="cuda", generator=g).float())
    out = {}
    for mode in (0, 1):
        L.las_gemm_set_arith(mode)
        C = torch.zeros(M * N, device="cuda")
        for _ in range(3): gemm(A, B, C, M, N, K, a_kc, b_kc, splitk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): gemm(A, B, C, M, N, K, a_kc, b_kc, splitk)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        err = float("nan")
        if check:
            A2 = (A.view(M, K) if a_kc else A.view(K, M).t()).double(); B2 = (B.view(N, K).t() if b_kc else B.view(K, N)).double()
            ref = A2 @ B2
            # error in units of the fp32 rounding of the magnitude sum |A||B| (what a forward error bound is stated against)
            mag = A2.abs() @ B2.abs()
            err = ((C.view(M, N).double() - ref).abs() / mag).max().item() / 2.0 ** -24
        out[mode] = (us, err)
    t0, e0_ = out[0]; t1, e1_ = out[1]
    print(f"{name:<26} M={M:<6} N={N:<5} K={K:<6} akc={int(a_kc)} bkc={int(b_kc)}: fp32 {t0:7.1f} us {2.0*M*N*K/t0/1e6:6.1f} TF err {e0_:6.3f} | "
          f"split {t1:7.1f} us {2.0*M*N*K/t1/1e6:6.1f} TF err {e1_:6.3f}   (err: max |C-ref| / (|A||B|) in units of 2^-24)")
    return t0, t1

tot0 = tot1 = 0.0
def acc(m, r):
    global tot0, tot1
    tot0 += m * r[0]; tot1 += m * r[1]
for l, (BT, D) in enumerate([(12800, 160), (6400, 1024), (3200, 1024)]):
    acc(2, run(f"L{l} fwd proj", BT, 1024, D, True, True, splitk=1))
    acc(2, run(f"L{l} dW_ih", 1024, D, BT, False, False))
    acc(2, run(f"L{l} dW_hh", 1024, 256, BT, False, False))
    if l > 0: acc(2, run(f"L{l} dX", BT, D, 1024, True, False, splitk=1))
acc(1, run("P = feat W_ctx^T", 3200, 2048, 512, True, True, splitk=1))
acc(1, run("dctx = dG0 W_ctx", 4096, 512, 2048, True, False, splitk=1))
acc(1, run("spl dW_ih0", 2048, 512, 4096, False, False))
acc(2, run("spl dW_hh", 2048, 512, 4064, False, False))
acc(1, run("spl dW_ih1", 2048, 512, 4096, False, False))
print(f"sum of the listed launches: fp32 {tot0/1e3:.3f} ms, split {tot1/1e3:.3f} ms")
run("4096^3 NT", 4096, 4096, 4096, True, True, splitk=1, reps=5)
run("4096^3 NN", 4096, 4096, 4096, True, False, splitk=1, reps=5)
run("4096^3 TN", 4096, 4096, 4096, False, False, splitk=1, reps=5)
run("4096^3 TT", 4096, 4096, 4096, False, True, splitk=1, reps=5)
run("wide range NT", 1024, 1024, 1024, True, True, splitk=1, scale=20)
run("wide range TN", 1024, 1024, 1024, False, False, splitk=1, scale=20)
run("odd edge NT", 1000, 900, 1000, True, True, splitk=1)
L.las_gemm_set_arith(0)
