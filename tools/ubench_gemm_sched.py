"""Schedule sweep of las_gemm_f32 inside ONE process (same box, same clocks): persistent stream-K vs classic grid vs split-K,
per shape, for the arithmetic mode in ARITH (default 1 = split-operand)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from las_pytorch_amd import _cabi
L = _cabi.lib()
L.las_gemm_set_arith(int(os.environ.get("ARITH", "1")))
SH = [("L1 fwd NT b2", 6400, 1024, 1024, 1, 1, 2), ("L2 fwd NT b2", 3200, 1024, 1024, 1, 1, 2), ("L0 fwd NT b2", 12800, 1024, 160, 1, 1, 2),
      ("L1 dX NN K2048", 6400, 1024, 2048, 1, 0, 1), ("L2 dX NN K2048", 3200, 1024, 2048, 1, 0, 1),
      ("P NT", 3200, 2048, 512, 1, 1, 1), ("dctx NN", 4096, 512, 2048, 1, 0, 1), ("L1 dW_ih TN", 1024, 1024, 6400, 0, 0, 1),
      ("L1 dW_hh TN", 1024, 256, 6400, 0, 0, 1), ("spl dW TN", 2048, 512, 4096, 0, 0, 1), ("1024^3 NT", 1024, 1024, 1024, 1, 1, 1)]
# (label, streamk, sk_min_tiles, split_below, split_target, splitk argument)
CFG = [("streamK any", 1, 0, 128, 512, 0), ("streamK>=W", 1, 512, 128, 512, 0), ("classic nosplit", 0, 0, 0, 512, 1),
       ("split<256 t512", 0, 0, 256, 512, 0), ("split<256 t1024", 0, 0, 256, 1024, 0), ("split<512 t512", 0, 0, 512, 512, 0), ("split<512 t1024", 0, 0, 512, 1024, 0)]
for name, M, N, K, a_kc, b_kc, batch in SH:
    A = torch.randn(batch * M * K, device="cuda"); B = torch.randn(batch * N * K, device="cuda"); C = torch.zeros(batch * M * N, device="cuda")
    res = []
    for label, sk, mint, below, target, splitk in CFG:
        for k, v in enumerate((sk, mint, below, target)): L.las_gemm_set_tuning(k, v)
        def call():
            _cabi.check(L.las_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, None, M, N, K, K if a_kc else M, K if b_kc else N, N,
                                       a_kc, b_kc, batch, M * K, N * K, M * N, splitk, 0, 0, _cabi.stream_ptr()))
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        res.append((label, e0.elapsed_time(e1) * 50))
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
    print(f"{name:<16} tiles={tiles:<4} kt={K//16:<4} " + " | ".join(f"{l} {us:6.1f}" for l, us in res))
