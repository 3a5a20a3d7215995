"""Times las_letter_error_rate (solver.py:11-24 on the device) on the headline batch and one solver.batch_iterator step around it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import synth
from las_pytorch_amd.solver.solver import LetterErrorRate_device, LetterErrorRate
B, U, V = int(os.environ.get("B", 32)), int(os.environ.get("U", 128)), 30
g = torch.Generator(device="cuda").manual_seed(0)
logp = torch.log_softmax(torch.randn(U, B, V, device="cuda", generator=g) * 3, -1).transpose(0, 1)      # (B,U,V) view of a (U,B,V) buffer
idx, lens = synth.make_labels(B, U, V, seed=3, ragged=True)
lab = torch.from_numpy(synth.onehot_labels(idx, lens, V)).cuda()
for _ in range(3): out = LetterErrorRate_device(logp, lab)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): out = LetterErrorRate_device(logp, lab)
e1.record(); torch.cuda.synchronize()
ref = LetterErrorRate(logp.argmax(-1).cpu().numpy(), lab.argmax(-1).cpu().numpy())
assert np.allclose(out.cpu().numpy(), np.array(ref), rtol=1e-6), (out.cpu().numpy()[:4], ref[:4])
print(f"las_letter_error_rate B={B} U={U}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call; matches the host form on {B} utterances")
