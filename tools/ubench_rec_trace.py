"""Per-phase timeline of the backward recurrence kernel.  Needs a profiling build of the library:
   make -C las_pytorch_amd/csrc clean all CXXFLAGS_EXTRA=-DLAS_REC_TRACE   (the stamps slow the kernel down; use for ratios)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import pBLSTMLayer, _cabi
B, T, D, H = int(os.environ.get("B", 32)), int(os.environ.get("T", 800)), int(os.environ.get("D", 80)), int(os.environ.get("H", 256))
torch.manual_seed(0)
layer = pBLSTMLayer(D, H).cuda()
x = torch.randn(B, T, D, device="cuda", requires_grad=True)
L = _cabi.lib()
L.las_debug_rec_trace.argtypes = [ctypes.c_void_p]; L.las_debug_rec_trace.restype = None
trace = torch.zeros(2 * 4096 * 8, dtype=torch.int64, device="cuda")
for it in range(3):
    out, _ = layer(x); loss = out.square().mean()
    if it == 2: L.las_debug_rec_trace(trace.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); loss.backward(); e1.record(); torch.cuda.synchronize()
L.las_debug_rec_trace(None)
n = T // 2
t = trace.cpu().numpy().reshape(2, 4096, 8)[:, 5:n - 5].astype(np.float64) / 100.0     # microseconds
a, w = t[0], t[1]
print(f"backward call {e0.elapsed_time(e1):.3f} ms (with stamps); step period {np.diff(a[:, 0]).mean():.3f} us")
print("thread 0   (cell wave):   dh -> dG %.3f | barrier wait %.3f | mat-vec + reduce %.3f | publish %.3f | barrier wait %.3f" % (
    (a[:, 1] - a[:, 0]).mean(), (a[:, 2] - a[:, 1]).mean(), (a[:, 3] - a[:, 2]).mean(), (a[:, 4] - a[:, 3]).mean(), (a[:, 5] - a[:, 4]).mean()))
print("thread 512 (factor wave): wait for dG %.3f | mat-vec + reduce %.3f | publish + next step's factors + prefetch %.3f | barrier wait %.3f" % (
    (w[:, 2] - w[:, 0]).mean(), (w[:, 3] - w[:, 2]).mean(), (w[:, 4] - w[:, 3]).mean(), (w[:, 5] - w[:, 4]).mean()))
