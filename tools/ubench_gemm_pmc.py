"""A few GEMM shapes of the training step, three launches each, for rocprofv3 --pmc passes (shapes are told apart by grid size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from las_pytorch_amd import _cabi
L = _cabi.lib()
SHAPES = [("L1 fwd proj", 6400, 1024, 1024, 1, 1, 1), ("L1 dX", 6400, 1024, 1024, 1, 0, 1), ("L1 dW_ih", 1024, 1024, 6400, 0, 0, 0),
          ("spl dW_ih1", 2048, 512, 4096, 0, 0, 0), ("4096^3 NT", 4096, 4096, 4096, 1, 1, 1)]
for name, M, N, K, a_kc, b_kc, sk in SHAPES:
    A = torch.randn(M * K, device="cuda"); B = torch.randn(N * K, device="cuda"); C = torch.zeros(M * N, device="cuda")
    for _ in range(3):
        _cabi.check(L.las_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, None, M, N, K, K if a_kc else M, K if b_kc else N, N,
                                   a_kc, b_kc, 1, 0, 0, 0, sk, 0, 0, _cabi.stream_ptr()))
    torch.cuda.synchronize()
    print(name, M, N, K)
