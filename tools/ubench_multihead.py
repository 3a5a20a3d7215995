"""Multi-head attention at paper size: the Speller's teacher-forced decode, the whole training step, the greedy (validation) decode and a free-running training step with the one-launch PRE kernels
against the per-step kernels (SPELLER_PRE_MH=0), heads = 2 / 4, (B, T) = (32, 800), U = 128.   python tools/ubench_multihead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from las_pytorch_amd import synth, _cabi
from hip_util import build_las
from las_pytorch_amd.solver.solver import label_smoothing_loss_backward_device, stack_steps
B, T, U = int(os.environ.get("B", 32)), 800, 128
c = synth.CONFIGS["P"]
for heads in (2, 4):
    sd = synth.make_state_dict(synth.config_shapes("P", multi_head=heads), seed=23, scale=0.1)
    las = build_las(c, sd, max_label_len=U, multi_head=heads)
    x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=23)).cuda()
    idx, lens = synth.make_labels(B, U, c["V"], seed=23)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
    with torch.no_grad():
        feat = las.listener(x)
    for mh in (1, 0):
        _cabi.set_option("SPELLER_PRE_MH", mh)
        def fwd():
            with torch.no_grad():
                return las.speller(feat, ground_truth=lab, teacher_force_rate=1.0)
        def step():
            for p in las.parameters(): p.grad = None
            preds, _ = las(batch_data=x, batch_label=lab, teacher_force_rate=1.0, is_training=True)
            label_smoothing_loss_backward_device(stack_steps(preds), lab, 0.1)
        def greedy():
            with torch.no_grad():
                return las.speller(feat, ground_truth=None, teacher_force_rate=0.0)
        def free_step():
            for p in las.parameters(): p.grad = None
            preds, _ = las(batch_data=x, batch_label=lab, teacher_force_rate=0.0, is_training=True)
            label_smoothing_loss_backward_device(stack_steps(preds), lab, 0.1)
        res, paths = [], []
        for fn, n in ((fwd, 10), (step, 5), (greedy, 10), (free_step, 5)):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / n * 1e3)
            paths.append(f"{_cabi.last_path(_cabi.PATH_DECODE_FWD)} / {_cabi.last_path(_cabi.PATH_DECODE_BWD)}")
        print(f"heads={heads} B={B} SPELLER_PRE_MH={mh}: teacher-forced decode {res[0]:.2f} ms ({res[0] * 1e3 / U:.1f} us per step), training step (fwd + loss + bwd) {res[1]:.2f} ms "
              f"[{paths[1]}]; greedy decode {res[2]:.2f} ms ({res[2] * 1e3 / U:.1f} us per step), free-running training step {res[3]:.2f} ms [{paths[3]}]")
    _cabi.set_option("SPELLER_PRE_MH", 1)
    _cabi.check_device_errors()
