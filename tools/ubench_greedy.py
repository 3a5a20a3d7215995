"""Free-running (greedy, decode_mode 1) decode through the one-launch kernel: us per decode step (15.5 us at paper size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from las_pytorch_amd import Speller, synth
cfg = os.environ.get("CFG", "P"); B = int(os.environ.get("B", 32)); Tp = int(os.environ.get("TP", 100)); U = int(os.environ.get("U", 128))
c = synth.CONFIGS[cfg]
torch.manual_seed(0)
sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U, use_mlp_in_attention=True,
             mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu", listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
feat = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.3
with torch.no_grad():
    for _ in range(3): sp(feat, ground_truth=None, teacher_force_rate=0.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): sp(feat, ground_truth=None, teacher_force_rate=0.0)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"greedy decode {cfg} B={B} T'={Tp} U={U}: {ms:.3f} ms per call = {ms * 1e3 / U:.2f} us per step (whole Speller.forward)")
