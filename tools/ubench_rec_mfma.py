"""Large-batch forward recurrence: the matrix-pipe kernel (pblstm_rec_mfma.hip) against the generic kernel (values) and against the
multi-utterance VALU kernels (time).  B, T, H from the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import _cabi
L = _cabi.lib()
T, H = int(os.environ.get("T", 400)), int(os.environ.get("H", 256))
g = torch.Generator().manual_seed(0)
w = [((torch.rand(4 * H, H, generator=g) * 2 - 1) / np.sqrt(H)).cuda() for _ in range(2)]
err = _cabi.err_word("cuda")
def run(B, flags, mfma, iters=5, check=None):
    _cabi.set_option("REC_MFMA", mfma)
    _cabi.set_option("REC_TRACE", 1 if os.environ.get("TRACE") else 0)
    pre = torch.randn(2 * B * T * 4 * H, generator=torch.Generator().manual_seed(1)).cuda()
    gates = torch.empty_like(pre); out = torch.empty(B, T, 2 * H, device="cuda")
    cbuf = torch.empty(2 * B * T * H, device="cuda"); hprev = torch.empty_like(cbuf)
    xbuf = torch.empty(L.las_rec_xbuf_bytes(B, H) // 4 + 4, device="cuda")
    ts = []
    for it in range(iters + 2):
        gates.copy_(pre)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _cabi.check(L.las_pblstm_rec_fwd(_cabi.ptr(gates), _cabi.ptr(w[0]), _cabi.ptr(w[1]), _cabi.ptr(out), _cabi.ptr(cbuf), _cabi.ptr(hprev),
                                         B, T, H, _cabi.ptr(xbuf), _cabi.ptr(err), flags, _cabi.stream_ptr()))
        e1.record(); torch.cuda.synchronize()
        if it >= 2: ts.append(e0.elapsed_time(e1))
    _cabi.check_device_errors()
    if os.environ.get("TRACE") and mfma == 2:
        tr = xbuf.view(torch.int64)[4096:4096 + 256 * 8].cpu().numpy().reshape(256, 8).astype(np.float64)[3:min(T, 256) - 1]
        d = lambda a, b: (tr[:, b] - tr[:, a]).mean() / 100.0
        dn = lambda a, b: (tr[1:, b] - tr[:-1, a]).mean() / 100.0      # stamp a of step s -> stamp b of step s + 1
        print("  pipeline trace wg0/batch0 (us): L wait+poll %.2f | L copy %.2f | L post -> M has planes %.2f | M mfma %.2f | M exchange+cell+publish %.2f | "
              "M publish -> next L tile in %.2f | M: start -> PF/SF seen + acc init %.2f -> planes seen %.2f | period %.2f" % (
                  d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), dn(5, 1), d(6, 7), d(7, 3), (tr[1:, 0] - tr[:-1, 0]).mean() / 100.0))
    if os.environ.get("TRACE") and mfma == 3:
        tr = xbuf.view(torch.int64)[4096:4096 + 256 * 8].cpu().numpy().reshape(256, 8).astype(np.float64)[2:min(T, 256) - 1]
        d = lambda a, b: (tr[:, b] - tr[:, a]).mean() / 100.0
        print("  trace wg0: canary wait %.2f | tile load + LDS + barrier %.2f | split + MFMA %.2f | red write + barrier %.2f | cell + publish %.2f | period %.2f us" % (
            d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), (tr[1:, 0] - tr[:-1, 0]).mean() / 100.0))
    return float(np.median(ts)), out.cpu().numpy(), gates.cpu().numpy(), cbuf.cpu().numpy(), hprev.cpu().numpy()
for B in [int(v) for v in os.environ.get("BS", "128,512").split(",")]:
    t1, o1, g1, c1, h1 = run(B, _cabi.FLAG_STASH, 3)
    t0, o0, g0, c0, h0 = run(B, _cabi.FLAG_STASH, 0)
    t2, o2, g2, c2, h2 = run(B, _cabi.FLAG_STASH, 2)
    d = [float(np.abs(a - b).max()) for a, b in ((o1, o0), (g1, g0), (c1, c0), (h1, h0))]
    d2 = [float(np.abs(a - b).max()) for a, b in ((o2, o0), (g2, g0), (c2, c0), (h2, h0))]
    print(f"B={B} T={T} H={H}: mfma {t1:.3f} ms ({t1 * 1e3 / T:.2f} us/step)  valu {t0:.3f} ms   max|diff| out {d[0]:.2e} gates {d[1]:.2e} c {d[2]:.2e} hprev {d[3]:.2e}")
    print(f"            pipeline (REC_MFMA=2) {t2:.3f} ms ({t2 * 1e3 / T:.2f} us/step)   max|diff| out {d2[0]:.2e} gates {d2[1]:.2e} c {d2[2]:.2e} hprev {d2[3]:.2e}", flush=True)
_cabi.set_option("REC_MFMA", 1)
