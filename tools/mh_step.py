"""N training steps (fwd + loss + bwd) of the paper-size model with HEADS attention heads at B = 32 // HEADS, for rocprofv3 --kernel-trace --stats.
   HEADS=2 python3 tools/mh_step.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from las_pytorch_amd import synth, _cabi
from las_pytorch_amd.solver.solver import label_smoothing_loss_backward_device, stack_steps
from hip_util import build_las
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
heads = int(os.environ.get("HEADS", 2))
B, T, U = 32 // heads, 800, 128
c = synth.CONFIGS["P"]
las = build_las(c, synth.make_state_dict(synth.config_shapes("P", multi_head=heads), seed=23, scale=0.1), max_label_len=U, multi_head=heads)
x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=23)).cuda()
idx, lens = synth.make_labels(B, U, c["V"], seed=23)
lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
for _ in range(N):
    for p in las.parameters(): p.grad = None
    preds, _ = las(batch_data=x, batch_label=lab, teacher_force_rate=1.0, is_training=True)
    label_smoothing_loss_backward_device(stack_steps(preds), lab, 0.1)
torch.cuda.synchronize()
_cabi.check_device_errors()
print("paths", _cabi.last_path(_cabi.PATH_DECODE_FWD), _cabi.last_path(_cabi.PATH_DECODE_BWD))
