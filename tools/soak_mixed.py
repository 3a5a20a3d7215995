"""Soak of the round-5 paths: N solver steps on the headline batch with teacher_force_rate 0.5 (so that teacher-forced and free-running
training steps alternate at random: PRE kernels both ways for both) and a validation decode every 50 steps (forward-only free-running kernel),
through solver.batch_iterator with the fused optimizer.  Counts hand-off timeouts (the solver re-runs such a step on the generic kernels and
warns); with the 43 ms spin limit of round 5 the count must stay 0 on an otherwise idle GPU.   python tools/soak_mixed.py [steps]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from las_pytorch_amd import dp, synth, _cabi
from las_pytorch_amd.optim import FusedClipAdam
from las_pytorch_amd.solver.solver import batch_iterator
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
las, c, _ = bench.build_model("P", 128, dev)
B, T, U = 32, 800, 128
x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=17)).to(dev)
idx, lens = synth.make_labels(B, U, c["V"], seed=17)
lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).to(dev)
red = dp.FlatGradAllReducer(las, direct=True)
opt = FusedClipAdam(red, lr=2e-4)
np.random.seed(0)
paths, losses, timeouts = {}, [], 0
t0 = time.perf_counter()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for s in range(N):
        loss, ler = batch_iterator(x, lab, las, opt, tf_rate=0.5, is_training=True, max_label_len=U, label_smoothing=0.1, use_gpu=True)
        key = (_cabi.last_path(_cabi.PATH_DECODE_FWD), _cabi.last_path(_cabi.PATH_DECODE_BWD))
        paths[key] = paths.get(key, 0) + 1
        losses.append(float(loss))
        if s % 50 == 49:
            vloss, vler = batch_iterator(x, lab, las, opt, tf_rate=0.0, is_training=False, max_label_len=U, label_smoothing=0.1, use_gpu=True)
            vk = ("validation", _cabi.last_path(_cabi.PATH_DECODE_FWD))
            paths[vk] = paths.get(vk, 0) + 1
    timeouts = sum(1 for m in w if "hand-off timeout" in str(m.message))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert np.isfinite(losses).all()
print(f"{N} solver steps (tf_rate 0.5) + {N // 50} validation decodes in {dt:.1f} s ({dt / N * 1e3:.2f} ms per step incl. the device LER and a host read of the loss); loss {losses[0]:.4f} -> {losses[-1]:.4f}; "
      f"hand-off timeouts re-run on the generic kernels: {timeouts}; decode paths {paths}")
