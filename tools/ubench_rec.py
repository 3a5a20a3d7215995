"""Times the forward recurrence kernel alone (layer-0 shape of the bench workload)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import _cabi
if os.environ.get("LAS_ABL_LIB"): _cabi.LIB_PATH = os.path.abspath(os.environ["LAS_ABL_LIB"])      # an experiment build instead of the product library
L = _cabi.lib()
B, T, H = int(os.environ.get("B", 32)), int(os.environ.get("T", 400)), int(os.environ.get("H", 256))
g = torch.Generator().manual_seed(0)
w = [((torch.rand(4 * H, H, generator=g) * 2 - 1) / np.sqrt(H)).cuda() for _ in range(2)]
pre = torch.randn(2 * B * T * 4 * H, generator=g).cuda()
gates = torch.empty_like(pre); out = torch.empty(B, T, 2 * H, device="cuda")
cbuf = torch.empty(2 * B * T * H, device="cuda"); hprev = torch.empty_like(cbuf)
xbuf = torch.empty(L.las_rec_xbuf_bytes(B, H) // 4 + 4, device="cuda"); err = _cabi.err_word("cuda")
flags = _cabi.FLAG_STASH | (_cabi.FLAG_FORCE_GENERIC if os.environ.get("GENERIC") else 0)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for it in range(13):
    gates.copy_(pre)
    if it >= 3: ev[it - 3][0].record()
    _cabi.check(L.las_pblstm_rec_fwd(_cabi.ptr(gates), _cabi.ptr(w[0]), _cabi.ptr(w[1]), _cabi.ptr(out), _cabi.ptr(cbuf), _cabi.ptr(hprev),
                                     B, T, H, _cabi.ptr(xbuf), _cabi.ptr(err), flags, _cabi.stream_ptr()))
    if it >= 3: ev[it - 3][1].record()
torch.cuda.synchronize()
ms = np.median([a.elapsed_time(b) for a, b in ev])
print(f"dbg={os.environ.get('LAS_REC_DBG', '0'):>3} B={B} T={T} H={H}: {ms:.3f} ms  {ms * 1e3 / T:.3f} us/step  err={int(err[0])}")
