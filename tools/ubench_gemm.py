"""Times las_gemm_f32 on the GEMM shapes of one P-config training step (B=32, T=800, U=128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import _cabi
L = _cabi.lib()
def run(name, M, N, K, a_kc, b_kc, batch=1, splitk=0, reps=20):
    A = torch.randn(batch * M * K, device="cuda"); Bm = torch.randn(batch * N * K, device="cuda"); C = torch.zeros(batch * M * N, device="cuda")
    lda = K if a_kc else M; ldb = K if b_kc else N
    def call():
        _cabi.check(L.las_gemm_f32(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), None, None, M, N, K, lda, ldb, N, int(a_kc), int(b_kc),
                                   batch, M * K, N * K, M * N, splitk, 0, 0, _cabi.stream_ptr()))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:<34} M={M:<6} N={N:<5} K={K:<6} b={batch:<3} akc={int(a_kc)} bkc={int(b_kc)} sk={splitk}: {us:8.1f} us  {2.0*batch*M*N*K/us/1e6:7.1f} TF")
    return us
tot = 0
for l, (BT, D) in enumerate([(12800, 160), (6400, 1024), (3200, 1024)]):
    tot += 2 * run(f"L{l} fwd proj", BT, 1024, D, True, True, splitk=1)
    tot += 2 * run(f"L{l} dW_ih", 1024, D, BT, False, False)
    tot += 2 * run(f"L{l} dW_hh", 1024, 256, BT, False, False)
    if l > 0: tot += 2 * run(f"L{l} dX", BT, D, 1024, True, False, splitk=1)
tot += run("keys", 3200, 64, 512, True, True, splitk=1)
tot += run("spl dW_ih0[:, V:]", 2048, 512, 4096, False, False)
tot += run("spl dW_ih0[:, :V]", 2048, 30, 4096, False, False)
tot += 2 * run("spl dW_hh", 2048, 512, 4064, False, False)
tot += run("spl dW_ih1", 2048, 512, 4096, False, False)
tot += 2 * run("spl dW_c half", 30, 512, 4096, False, False)
tot += run("spl dW_phi", 64, 512, 4096, False, False)
tot += run("spl dW_psi", 64, 512, 3200, False, False)
tot += run("spl dfeat += dK Wpsi", 3200, 512, 64, True, False, splitk=1)
tot += run("spl dfeat (batched)", 100, 512, 128, False, False, batch=32, splitk=1)
tot += run("spl dK (batched)", 100, 64, 128, False, False, batch=32, splitk=1)
print(f"sum over one training step: {tot/1e3:.3f} ms")
if os.environ.get("BIG"):
    run("4096^3 NT", 4096, 4096, 4096, True, True, splitk=1, reps=5)
    run("4096^3 NN", 4096, 4096, 4096, True, False, splitk=1, reps=5)
    run("4096^3 TN", 4096, 4096, 4096, False, False, splitk=1, reps=5)
    run("8192x1024x1024 NT", 8192, 1024, 1024, True, True, splitk=1, reps=10)
if os.environ.get("TORCH_REF"):
    # measuring stick only (never on the product path): what the vendor library reaches on the same shapes
    for (M, N, K) in [(6400, 1024, 1024), (3200, 1024, 1024), (12800, 1024, 160), (1024, 1024, 6400), (2048, 512, 4096), (4096, 4096, 4096)]:
        a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
        for _ in range(3): c = a @ b.t()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): c = a @ b.t()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"torch.matmul NT {M}x{N}x{K}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF")
