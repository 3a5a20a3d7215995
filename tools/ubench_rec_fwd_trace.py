"""Per-phase timeline of the multi-utterance forward recurrence (profiling build: make clean all CXXFLAGS_EXTRA=-DLAS_REC_TRACE)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import _cabi
if os.environ.get("LAS_ABL_LIB"): _cabi.LIB_PATH = os.path.abspath(os.environ["LAS_ABL_LIB"])      # the -DLAS_REC_TRACE build
L = _cabi.lib()
L.las_debug_rec_trace.argtypes = [ctypes.c_void_p]; L.las_debug_rec_trace.restype = None
B, T, H = int(os.environ.get("B", 128)), int(os.environ.get("T", 400)), int(os.environ.get("H", 256))
g = torch.Generator().manual_seed(0)
w = [((torch.rand(4 * H, H, generator=g) * 2 - 1) / np.sqrt(H)).cuda() for _ in range(2)]
pre = torch.randn(2 * B * T * 4 * H, generator=g).cuda() * 0.5
gates = torch.empty_like(pre); out = torch.empty(B, T, 2 * H, device="cuda")
cbuf = torch.empty(2 * B * T * H, device="cuda"); hprev = torch.empty_like(cbuf)
xbuf = torch.empty(L.las_rec_xbuf_bytes(B, H) // 4 + 4, device="cuda"); err = _cabi.err_word("cuda")
trace = torch.zeros(2 * 4096 * 8, dtype=torch.int64, device="cuda")
for it in range(3):
    gates.copy_(pre)
    if it == 2: L.las_debug_rec_trace(trace.data_ptr())
    _cabi.check(L.las_pblstm_rec_fwd(_cabi.ptr(gates), _cabi.ptr(w[0]), _cabi.ptr(w[1]), _cabi.ptr(out), _cabi.ptr(cbuf), _cabi.ptr(hprev),
                                     B, T, H, _cabi.ptr(xbuf), _cabi.ptr(err), _cabi.FLAG_STASH, _cabi.stream_ptr()))
    torch.cuda.synchronize()
L.las_debug_rec_trace(None)
t = trace.cpu().numpy().reshape(2, 4096, 8)[:, 5:T - 5].astype(np.float64) / 100.0
for name, a in (("thread 0 (cell wave)", t[0]), ("thread 512 (poller)", t[1])):
    if name.startswith("thread 0") and a[:, 5].any():
        print(f"B={B} cell wave: barrier release -> h published {(a[:,5]-a[:,2]).mean():.3f} us")
    print(f"B={B} {name}: period {np.diff(a[:, 0]).mean():.3f} us | matvec {(a[:,1]-a[:,0]).mean():.3f} | barrier {(a[:,2]-a[:,1]).mean():.3f} | cell/poll {(a[:,3]-a[:,2]).mean():.3f} | barrier {(a[:,4]-a[:,3]).mean():.3f}")
