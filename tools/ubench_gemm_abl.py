"""Times a few GEMM shapes in split-operand arithmetic; LAS_ABL_LIB=<path> loads an ablation build (tools/abl/, built with
-DLAS_SPLIT_ABL=<mask>) instead of the product library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from las_pytorch_amd import _cabi
if os.environ.get("LAS_ABL_LIB"): _cabi.LIB_PATH = os.path.abspath(os.environ["LAS_ABL_LIB"])
L = _cabi.lib()
L.las_gemm_set_arith(int(os.environ.get("ARITH", "1")))
SH = [("L1 fwd NT", 6400, 1024, 1024, 1, 1, 1), ("L1 dX NN", 6400, 1024, 1024, 1, 0, 1), ("L1 dW TN", 1024, 1024, 6400, 0, 0, 0),
      ("spl dW TN", 2048, 512, 4096, 0, 0, 0), ("4096^3 NT", 4096, 4096, 4096, 1, 1, 1), ("4096^3 TN", 4096, 4096, 4096, 0, 0, 1)]
out = []
for name, M, N, K, a_kc, b_kc, sk in SH:
    A = torch.randn(M * K, device="cuda"); B = torch.randn(N * K, device="cuda"); C = torch.zeros(M * N, device="cuda")
    def call():
        _cabi.check(L.las_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, None, M, N, K, K if a_kc else M, K if b_kc else N, N,
                                   a_kc, b_kc, 1, 0, 0, 0, sk, 0, 0, _cabi.stream_ptr()))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    out.append(f"{name} {us:7.1f}us {2.0*M*N*K/us/1e6:6.1f}TF")
print(os.environ.get("LAS_ABL_LIB", "product"), " | ".join(out))
