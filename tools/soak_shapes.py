"""Soak of the round-6 mechanisms: solver steps that ALTERNATE between batches of different shapes — the headline batch (32, 800), the
long-utterance batch of BASELINE configs[4] (8, 3000) whose backward runs XCD-confined recurrences beside deferred, XCD-partitioned
weight-gradient groups on the library's side stream, and a (16, 1600) batch — so that the epoch-tagged hand-off scratch is re-used across
launches of different layouts, the deferred work is joined in front of every clip, and free-running steps / validation decodes are mixed in.
Counts hand-off timeouts (must be 0 on an otherwise idle GPU) and checks that DEFER_DW stayed on.   python tools/soak_shapes.py [rounds]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from las_pytorch_amd import dp, synth, _cabi
from las_pytorch_amd.optim import FusedClipAdam
from las_pytorch_amd.solver.solver import batch_iterator
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
U = 64
las, c, _ = bench.build_model("P", U, dev)
shapes = [(32, 800), (8, 3000), (16, 1600), (8, 3000), (4, 800)]
data = []
for i, (B, T) in enumerate(shapes):
    x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=17 + i)).to(dev)
    idx, lens = synth.make_labels(B, U, c["V"], seed=17 + i, ragged=True)
    data.append((x, torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).to(dev)))
red = dp.FlatGradAllReducer(las, direct=True, defer_dw=True)
opt = FusedClipAdam(red, lr=2e-4)
np.random.seed(0)
dw, losses = {}, []
t0 = time.perf_counter()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for r in range(N):
        for (B, T), (x, lab) in zip(shapes, data):
            loss, ler = batch_iterator(x, lab, las, opt, tf_rate=0.7, is_training=True, max_label_len=U, label_smoothing=0.1, use_gpu=True)
            key = (B, T, _cabi.last_path(_cabi.PATH_DW), _cabi.last_path(_cabi.PATH_REC_BWD))
            dw[key] = dw.get(key, 0) + 1
            losses.append(float(loss))
        if r % 20 == 19:
            x, lab = data[1]
            batch_iterator(x, lab, las, opt, tf_rate=0.0, is_training=False, max_label_len=U, label_smoothing=0.1, use_gpu=True)
    timeouts = sum(1 for m in w if "hand-off timeout" in str(m.message))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_cabi.check_device_errors()
assert np.isfinite(losses).all()
assert not red._deferred_keep
print(f"{N} rounds x {len(shapes)} shapes = {N * len(shapes)} solver steps in {dt:.1f} s; loss {losses[0]:.4f} -> {losses[-1]:.4f}; hand-off timeouts: {timeouts}; "
      f"DEFER_DW option at the end: {_cabi.get_option('DEFER_DW')}; (B, T, weight-gradient path, backward recurrence) counts: {dw}")
