"""Soak of the multi-head one-launch kernels: N training steps (teacher-forcing rate 0.5: teacher-forced and free-running steps alternate at random)
of the paper-size model with two attention heads at B = 16 through solver.batch_iterator with the fused optimizer, a validation decode every
50 steps; counts hand-off timeouts (must stay 0 on an idle GPU) and records the decode paths taken.   python tools/soak_multihead.py [steps] [heads]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from las_pytorch_amd import dp, synth, _cabi
from las_pytorch_amd.optim import FusedClipAdam
from las_pytorch_amd.solver.solver import batch_iterator
from hip_util import build_las
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B, T, U = 32 // heads, 800, 128
c = synth.CONFIGS["P"]
sd = synth.make_state_dict(synth.config_shapes("P", multi_head=heads), seed=23, scale=0.1)
las = build_las(c, sd, max_label_len=U, multi_head=heads)
x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=23)).cuda()
idx, lens = synth.make_labels(B, U, c["V"], seed=23)
lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
red = dp.FlatGradAllReducer(las, direct=True)
opt = FusedClipAdam(red, lr=2e-4)
np.random.seed(0)
paths, losses = {}, []
t0 = time.perf_counter()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for s in range(N):
        loss, ler = batch_iterator(x, lab, las, opt, tf_rate=0.5, is_training=True, max_label_len=U, label_smoothing=0.1, use_gpu=True)
        key = (_cabi.last_path(_cabi.PATH_DECODE_FWD), _cabi.last_path(_cabi.PATH_DECODE_BWD))
        paths[key] = paths.get(key, 0) + 1
        losses.append(float(loss))
        if s % 50 == 49:
            batch_iterator(x, lab, las, opt, tf_rate=0.0, is_training=False, max_label_len=U, label_smoothing=0.1, use_gpu=True)
            vk = ("validation", _cabi.last_path(_cabi.PATH_DECODE_FWD))
            paths[vk] = paths.get(vk, 0) + 1
    timeouts = sum(1 for m in w if "hand-off timeout" in str(m.message))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert np.isfinite(losses).all()
print(f"heads={heads} B={B}: {N} solver steps (tf_rate 0.5) + {N // 50} validation decodes in {dt:.1f} s ({dt / N * 1e3:.2f} ms per step); loss {losses[0]:.4f} -> {losses[-1]:.4f}; "
      f"hand-off timeouts re-run on the generic kernels: {timeouts}; decode paths {paths}")
