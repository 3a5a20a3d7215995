"""Soak of the one-launch decode kernels of the YAML sizes (speller_big.hip): N iterations of forward + backward on the one-launch and on the
per-step path against a per-step reference, reporting every tensor that leaves the tolerance (and which path it was).
   python tools/stress_big.py [iterations [U]]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from las_pytorch_amd import Speller, _cabi, synth
B, Tp, U = 16, 100, int(sys.argv[2]) if len(sys.argv) > 2 else 70
c = synth.CONFIGS["Y"]
torch.manual_seed(5)
sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U, use_mlp_in_attention=True,
             mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu", listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
feat0 = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5
idx, lens = synth.make_labels(B, U, c["V"], seed=11, ragged=True)
lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
w = torch.randn(U, B, c["V"], device="cuda")
def run(force):
    sp.force_generic = force
    sp.zero_grad(set_to_none=True)
    feat = feat0.clone().requires_grad_(True)
    preds, att = sp(feat, ground_truth=lab, teacher_force_rate=1.0)
    logp = torch.stack(preds)
    (logp * w).sum().backward()
    sp.force_generic = False
    return dict(logp=logp.detach().cpu().numpy(), dfeat=feat.grad.cpu().numpy(), **{"d" + n: p.grad.cpu().numpy() for n, p in sp.named_parameters()})
ref = run(True)
ref2 = run(True)
print("generic vs generic:", max(float(np.abs(ref[k] - ref2[k]).max()) for k in ref))
labs = idx  # (B, U)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for force in (False, True):
        r = run(force)
        worst = 0.0
        for k in ref:
            d = np.abs(r[k] - ref[k]); tol = 1e-3 * np.abs(ref[k]) + 1e-5 * max(1.0, float(np.abs(ref[k]).max()))
            bad = d > tol
            if bad.any():
                nz = np.nonzero(bad)
                print(f"iter {it} {'generic' if force else 'one-launch'}: {k}: {bad.sum()} bad, max err {d.max():.3e}; last-axis idx {np.unique(nz[-1])[:12]}, first-axis idx {np.unique(nz[0])[:12]} (n={len(np.unique(nz[0]))})")
            worst = max(worst, float(d.max()))
    torch.cuda.synchronize()
print("err word", int(_cabi.err_word(torch.device("cuda", 0))[0].item()))
print("labels with 25:", np.argwhere(idx == 25)[:10].tolist())
