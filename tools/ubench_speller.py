"""Times the Speller forward / backward alone (paper-size by default): ms per call and us per decode step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import Speller, synth
cfg = os.environ.get("CFG", "P"); B = int(os.environ.get("B", 32)); Tp = int(os.environ.get("TP", 100)); U = int(os.environ.get("U", 128))
c = synth.CONFIGS[cfg]
torch.manual_seed(0)
FREE = bool(os.environ.get("FREE"))      # free-running (greedy, decode_mode 1) instead of teacher forcing
sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U, use_mlp_in_attention=True,
             mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu", listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
feat = (torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.3).requires_grad_(True)
idx, lens = synth.make_labels(B, U, c["V"])
lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
def run(n, bwd):
    ef = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(n + 2):
        ef[0].record()
        preds, _ = sp(feat, ground_truth=None if FREE else lab, teacher_force_rate=0.0 if FREE else 1.0)
        loss = torch.stack(preds).square().mean()
        ef[1].record()
        if bwd: loss.backward()
        ef[2].record()
        torch.cuda.synchronize()
        if it >= 2: tf += ef[0].elapsed_time(ef[1]); tb += ef[1].elapsed_time(ef[2])
    return tf / n, tb / n
f, b = run(int(os.environ.get("N", 5)), not os.environ.get("NOBWD"))
print(f"speller {'free-running ' if FREE else ''}{cfg} B={B} Tp={Tp} U={U}: fwd {f:.3f} ms ({f*1e3/U:.2f} us/step)  bwd {b:.3f} ms ({b*1e3/U:.2f} us/step)")
