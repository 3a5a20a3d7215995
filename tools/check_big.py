"""The one-launch decode forward of the reference's YAML sizes (speller_big.hip: Speller 1024x2, B <= 16) against the per-step
launch chain it replaces: outputs, every gradient (the per-step backward consumes the stash the kernel wrote), and the time
of a teacher-forced forward on both paths.   python tools/check_big.py [B Tp U]   (on the GPU box)"""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from las_pytorch_amd import Speller, _cabi, synth  # noqa: E402


def run(B, Tp, U, scale=None, trace_on=True):
    c = synth.CONFIGS["Y"]
    torch.manual_seed(5)
    sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U,
                 use_mlp_in_attention=True, mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu",
                 listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
    if scale is not None:
        with torch.no_grad():
            for p in sp.parameters():
                p.uniform_(-scale, scale)
    feat0 = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5
    idx, lens = synth.make_labels(B, U, c["V"], seed=11, ragged=True)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
    w = torch.randn(U, B, c["V"], device="cuda")
    L = _cabi.lib()
    L.las_debug_big_trace.argtypes = [ctypes.c_void_p]
    L.las_debug_big_trace.restype = None
    trace = torch.zeros(64 * 16 + 256 * 8, dtype=torch.int64, device="cuda")
    btrace = torch.zeros(4 * 64 * 16 + 256 * 8, dtype=torch.int64, device="cuda")
    res = []
    for force in (False, True):
        sp.force_generic = force
        L.las_debug_big_trace(trace.data_ptr() if (trace_on and not force) else None)
        L.las_debug_big_bwd_trace(btrace.data_ptr() if (trace_on and not force) else None)
        try:
            sp.zero_grad(set_to_none=True)
            feat = feat0.clone().requires_grad_(True)
            preds, att = sp(feat, ground_truth=lab, teacher_force_rate=1.0)
            logp = torch.stack(preds)
            (logp * w).sum().backward()
            res.append(dict(logp=logp.detach().cpu().numpy(), att=torch.stack([a[0] for a in att]).detach().cpu().numpy(),
                            dfeat=feat.grad.cpu().numpy(), **{"d" + n: p.grad.cpu().numpy() for n, p in sp.named_parameters()}))
        finally:
            sp.force_generic = False
            L.las_debug_big_trace(None)
            L.las_debug_big_bwd_trace(None)
    torch.cuda.synchronize()
    err = int(_cabi.err_word(torch.device("cuda", 0))[0].item())
    ran = int(trace.abs().sum().item()) != 0
    worst = 0.0
    for k in res[0]:
        a, b = res[0][k], res[1][k]
        scale_k = float(np.abs(b).max()) + 1e-30
        d = float(np.abs(a - b).max())
        tol = 1e-3 * scale_k + 1e-5 * max(1.0, scale_k)
        worst = max(worst, d / tol)
        flag = "" if d <= tol else "   <-- FAIL"
        print(f"  {k:28s} max|diff| {d:.3e}  scale {scale_k:.3e}{flag}")
    print(f"B={B} Tp={Tp} U={U}: kernel ran: {ran}, device error word {err:#x}, worst diff/tol {worst:.3f}")
    # forward time on both paths (no autograd)
    out = {}
    with torch.no_grad():
        for force in (False, True):
            sp.force_generic = force
            for _ in range(3):
                sp(feat0, ground_truth=lab, teacher_force_rate=1.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                sp(feat0, ground_truth=lab, teacher_force_rate=1.0)
            torch.cuda.synchronize()
            out[force] = (time.perf_counter() - t0) / n * 1e3
    tb = {}
    for force in (False, True):
        sp.force_generic = force
        def fb():
            sp.zero_grad(set_to_none=True)
            feat = feat0.clone().requires_grad_(True)
            preds, _ = sp(feat, ground_truth=lab, teacher_force_rate=1.0)
            (torch.stack(preds) * w).sum().backward()
        for _ in range(2):
            fb()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fb()
        torch.cuda.synchronize()
        tb[force] = (time.perf_counter() - t0) / 5 * 1e3
    sp.force_generic = False
    print(f"  forward + backward (with the deferred GEMMs): one launch each {tb[False]:.3f} ms, per-step launches {tb[True]:.3f} ms")
    ball = btrace.cpu().numpy().astype(np.float64) / 100.0
    bt = ball[:4096].reshape(4, 64, 16)
    bw = ball[4096:].reshape(256, 8)
    if U > 12 and bw[:, 1].max() > 0:
        base = bw[:, 0][bw[:, 0] > 0].min()
        for k, nm in enumerate(["carried dctx seen", "slice triple published", "dqpre published", "dG1 published", "dG1 flags seen", "dG0 published", "dG0 flags seen", "carried dctx published"]):
            v = bw[:, k][bw[:, k] > 0] - base
            print(f"    bwd step U-11, {nm}: n={len(v)} min {v.min():.2f} median {np.median(v):.2f} max {v.max():.2f} us")
    if bt[0, 2, 0] != 0 and U > 6:
        n = min(U, 64)
        step = np.diff(bt[0, 2:n, 0]).mean()
        print(f"  backward trace: {step:.2f} us per step; phases by matrix role (us from the step's start):")
        for role, nm in enumerate(["W_ih1", "W_hh1", "W_ctx", "W_hh0"]):
            rel = (bt[role, 2:n, :11] - bt[role, 2:n, 0:1]).mean(0)
            print(f"    {nm}: " + ", ".join(f"{k}:{v:.2f}" for k, v in enumerate(rel) if bt[role, 3, k] != 0))
    print(f"  forward: one launch {out[False]:.3f} ms, per-step launches {out[True]:.3f} ms  ({out[False] * 1e3 / U:.2f} / {out[True] * 1e3 / U:.2f} us per decode step)")
    if ran and trace_on:
        tall = trace.cpu().numpy()
        t = tall[:1024].reshape(64, 16)
        ws = tall[1024:].reshape(256, 8).astype(np.float64) / 100.0
        if U > 10:
            base = ws[:, 0].min()
            for k, nm in enumerate(["ctx flags seen", "h0 published", "h0 flags seen", "h1 + query parts published", "h1 flags seen", "ctx published", "W_hh1 product done"]):
                v = ws[:, k] - base
                print(f"    step 10, all workgroups, {nm}: min {v.min():.2f} median {np.median(v):.2f} max {v.max():.2f} us; by XCD (max): " + " ".join(f"{v[x::8].max():.2f}" for x in range(8)))
        n = min(U, 64)
        if n > 4:
            d = np.diff(t[2:n, :9].astype(np.float64), axis=1).mean(0) / 100.0     # 100 MHz shader clock counter -> us
            step = np.diff(t[2:n, 0].astype(np.float64)).mean() / 100.0
            names = ["ctx wait+mfma", "cell0", "h0 wait+mfma", "cell1+qpart+Whh0", "q wait+sum", "attention", "ctx publish", "Whh1"]
            print("  trace (workgroup 0, us): " + ", ".join(f"{nm} {v:.2f}" for nm, v in zip(names, d)) + f"; step {step:.2f}")
            print(f"    early-fetch fallbacks (all workgroups, whole launch): slabs {t[0, 14]}, query parts {t[0, 15]} of {256 * 8 * 4 * U} / {16 * B * 8 * U} wave loads")
            if t[2, 13] != 0:
                tt = t[2:n].astype(np.float64)
                print(f"    shader clock over the traced steps: {(tt[-1, 13] - tt[0, 13]) / (tt[-1, 0] - tt[0, 0]) * 100.0:.0f} MHz")
            if t[2, 12] != 0:      # sub-phases of the top cell's hand-off: stamps 2 -> 9 -> 10 -> 11 -> 12 -> 3
                tt = t[2:n].astype(np.float64)
                sub = [tt[:, 9] - tt[:, 2], tt[:, 10] - tt[:, 9], tt[:, 11] - tt[:, 10], tt[:, 12] - tt[:, 11], tt[:, 3] - tt[:, 12]]
                print("    h0 phase: " + ", ".join(f"{nm} {v.mean() / 100.0:.2f}" for nm, v in zip(["canary", "barrier", "load", "mfma issue", "red+barrier"], sub)))
    return worst, err, ran


def run_greedy(B, Tp, U, scale=None):
    """Free-running greedy decode (decode_mode 1) of the same kernel against the per-step chain: log-probs, arg-max sequences, attention,
    gradients (the hoisted backward runs the one-launch backward kernel on the greedy stash), time."""
    c = synth.CONFIGS["Y"]
    torch.manual_seed(6)
    sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U,
                 use_mlp_in_attention=True, mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu",
                 listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
    if scale is not None:
        with torch.no_grad():
            for p in sp.parameters():
                p.uniform_(-scale, scale)
    feat0 = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5
    w = torch.randn(U, B, c["V"], device="cuda")
    res, tms = [], {}
    for force in (False, True):
        sp.force_generic = force
        sp.zero_grad(set_to_none=True)
        feat = feat0.clone().requires_grad_(True)
        preds, att = sp(feat, ground_truth=None, teacher_force_rate=0.0)
        logp = torch.stack(preds)
        (logp * w).sum().backward()
        out = dict(logp=logp.detach().cpu().numpy(), att=torch.stack([a[0] for a in att]).detach().cpu().numpy(), dfeat=feat.grad.cpu().numpy())
        out.update({"d" + n: p.grad.cpu().numpy() for n, p in sp.named_parameters() if p.grad is not None})
        res.append(out)
        with torch.no_grad():
            for _ in range(2):
                sp(feat0, ground_truth=None, teacher_force_rate=0.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                sp(feat0, ground_truth=None, teacher_force_rate=0.0)
            torch.cuda.synchronize()
            tms[force] = (time.perf_counter() - t0) / 5 * 1e3
    sp.force_generic = False
    err = int(_cabi.err_word(torch.device("cuda", 0))[0].item())
    same = bool((res[0]["logp"].argmax(-1) == res[1]["logp"].argmax(-1)).all())
    nsym = len(np.unique(res[1]["logp"].argmax(-1)))
    worst = 0.0
    for k in res[0]:
        a, b = res[0][k], res[1][k]
        scale_k = float(np.abs(b).max()) + 1e-30
        d = float(np.abs(a - b).max())
        worst = max(worst, d / (1e-3 * scale_k + 1e-5 * max(1.0, scale_k)))
    print(f"greedy B={B} Tp={Tp} U={U}: arg-max sequences identical: {same} ({nsym} distinct symbols), worst diff/tol {worst:.3f}, error word {err:#x}; "
          f"forward {tms[False]:.3f} ms one launch, {tms[True]:.3f} ms per-step ({tms[False] * 1e3 / U:.2f} / {tms[True] * 1e3 / U:.2f} us per step)")
    return worst, err, same


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        cases = [tuple(int(v) for v in sys.argv[1:4])]
    else:
        cases = [(16, 100, 16), (3, 8, 6), (16, 200, 8), (5, 37, 7), (16, 100, 128)]
    bad = 0
    for B, Tp, U in cases:
        worst, err, ran = run(B, Tp, U)
        bad += (worst > 1.0) or err != 0 or not ran
    if len(sys.argv) < 4:
        for B, Tp, U, sc in [(16, 100, 24, None), (5, 37, 9, 0.08), (16, 100, 128, 0.05)]:
            worst, err, same = run_greedy(B, Tp, U, sc)
            bad += (worst > 1.0) or err != 0 or not same
    print("RESULT", "FAIL" if bad else "OK")
    sys.exit(1 if bad else 0)
