"""Placement of the XCD-partitioned launches of a small-batch training step (LAS_FLAG_DEFER_DW): does block b run on XCD b % 8 when the confined
backward recurrence (main stream) and the partitioned weight-gradient GEMM group (side stream) are dispatched at the same time?
    B=8 T=3000 python tools/xcd_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from las_pytorch_amd import _cabi, dp, synth
from las_pytorch_amd.optim import FusedClipAdam
B, T = int(os.environ.get("B", 8)), int(os.environ.get("T", 3000))
dev = torch.device("cuda", 0)
las, c, _ = bench.build_model("P", 128, dev)
x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=17)).to(dev)
idx, lens = synth.make_labels(B, 128, c["V"], seed=17)
lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).to(dev)
red = dp.FlatGradAllReducer(las, direct=True, defer_dw=True)
opt = FusedClipAdam(red, lr=2e-4)
step = bench.make_train_step(las, x, lab, red, opt)
for _ in range(3): step()
buf = torch.zeros(2048, dtype=torch.int32, device=dev)
L = _cabi.lib()

L.las_debug_xcd_probe(buf.data_ptr())
step(); torch.cuda.synchronize()
L.las_debug_xcd_probe(None)
h = buf.cpu().numpy()
for name, part in (("confined recurrence (last launch)", h[:1024]), ("partitioned GEMM group (last launch)", h[1024:])):
    b = np.flatnonzero(part)
    if b.size == 0:
        print(name, ": no launch recorded"); continue
    xcc = part[b] - 1
    ok = int((xcc == (b & 7)).sum())
    print(f"{name}: {b.size} blocks, {ok} on XCD b % 8; histogram of (actual - b%8) mod 8:", np.bincount((xcc - (b & 7)) % 8, minlength=8).tolist(),
          "| actual XCD histogram:", np.bincount(xcc, minlength=8).tolist())
_cabi.check_device_errors()
