"""Times the backward of one pBLSTM layer (las_pblstm_bwd: recurrence + weight-gradient GEMMs) at large batch with the matrix-pipe
recurrence kernels on and off, and prints the recurrence kernel's own time from the library's kernel timer when available."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import _cabi
from las_pytorch_amd.model.las_model import pBLSTMLayer
B, T, H, D = int(os.environ.get("B", 128)), int(os.environ.get("T", 800)), int(os.environ.get("H", 256)), int(os.environ.get("D", 80))
torch.manual_seed(0)
layer = pBLSTMLayer(D, H, rnn_unit="LSTM").cuda()
x = torch.randn(B, T, D, device="cuda", requires_grad=True)
for mfma in (1, 0, 1, 0):
    _cabi.set_option("REC_MFMA", mfma)
    ts = []
    for it in range(6):
        y = layer(x); y = y[0] if isinstance(y, tuple) else y
        g = torch.ones_like(y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y.backward(g); e1.record(); torch.cuda.synchronize()
        if it >= 2: ts.append(e0.elapsed_time(e1))
    print(f"B={B} T={T} H={H} REC_MFMA={mfma}: layer backward {np.median(ts):.3f} ms")
_cabi.set_option("REC_MFMA", 1)
if os.environ.get("TRACE"):      # -DRM_TRACE build: phase stamps of workgroup 0 of the backward matrix-pipe kernel
    import ctypes
    L = _cabi.lib()
    y = layer(x); y = y[0] if isinstance(y, tuple) else y; y.backward(torch.ones_like(y)); torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * (256 * 8))()
    L.las_debug_rm_bwd_trace.argtypes = [ctypes.c_void_p]; L.las_debug_rm_bwd_trace.restype = None
    L.las_debug_rm_bwd_trace(ctypes.addressof(buf))
    tr = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.float64)[3:200]
    d = lambda a, b: (tr[:, b] - tr[:, a]).mean() / 100.0
    print("  bwd trace wg0: partial sums in %.2f | LDS + barrier %.2f | cell backward + planes + stash %.2f | barrier %.2f | MFMA %.2f | publish %.2f | period %.2f us" % (
        d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), (tr[1:, 0] - tr[:-1, 0]).mean() / 100.0))
