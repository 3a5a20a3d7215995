"""A/B of one library option on the headline training step inside ONE process (same box, same clocks): alternates the two values
several times and prints the step time of every block.   python tools/ab_step_option.py GEMM_SK_FIXUP 1 0 [steps] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from las_pytorch_amd import _cabi, dp, synth
from las_pytorch_amd.optim import FusedClipAdam
key, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 4
B = int(os.environ.get("B", 32))
T = int(os.environ.get("T", 800))
dev = torch.device("cuda", 0)
las, c, _ = bench.build_model("P", 128, dev)
x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=17)).to(dev)
idx, lens = synth.make_labels(B, 128, c["V"], seed=17)
lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).to(dev)
red = dp.FlatGradAllReducer(las, direct=True, defer_dw=(key.upper().replace("LAS_", "") == "DEFER_DW"))
opt = FusedClipAdam(red, lr=2e-4)
step = bench.make_train_step(las, x, lab, red, opt)
for _ in range(10): step()
res = {va: [], vb: []}
for r in range(rounds):
    for v in (va, vb):
        _cabi.set_option(key, v)
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / steps * 1e3)
_cabi.check_device_errors()
for v in (va, vb):
    print(f"{key}={v}: " + " ".join(f"{t:.3f}" for t in res[v]) + f"  | median {np.median(res[v]):.3f} ms")
