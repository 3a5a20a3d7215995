"""Where a solver.batch_iterator step spends its time beyond the kernels: the headline batch through (a) bench.py's training step (no host read of
the loss), (b) batch_iterator as a train.py-style driver calls it (loss and LER read back every step), with the host time of each phase.
   python tools/solver_step_profile.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from las_pytorch_amd import dp, synth, _cabi
from las_pytorch_amd.optim import FusedClipAdam
from las_pytorch_amd.solver import solver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
las, c, _ = bench.build_model("P", 128, dev)
B, T, U = 32, 800, 128
x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=17)).to(dev)
idx, lens = synth.make_labels(B, U, c["V"], seed=17)
lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).to(dev)
red = dp.FlatGradAllReducer(las, direct=True)
opt = FusedClipAdam(red, lr=2e-4)
def run(fn, n):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
step = bench.make_train_step(las, x, lab, red, opt, tf_rate=1.0)
print(f"bench.py training step (no host read): {run(step, N):.3f} ms")
it = lambda: solver.batch_iterator(x, lab, las, opt, tf_rate=1.0, is_training=True, max_label_len=U, label_smoothing=0.1, use_gpu=True)
print(f"solver.batch_iterator (loss + LER read back each step): {run(it, N):.3f} ms")
# host time of the launches alone: how long Python needs to enqueue one step
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
host = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print(f"host time to enqueue one bench step: {host:.3f} ms")
