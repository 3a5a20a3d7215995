"""Generate golden vectors by running the UNMODIFIED reference (build container only).

    python tests/golden/make_golden.py

Imports ``/root/reference/model/las_model.py`` and ``solver/solver.py`` (stubbing the three
unused third-party imports they pull in at module top: pydub, python_speech_features,
editdistance), loads the deterministic weights of ``las_pytorch_amd.synth`` with
``load_state_dict(strict=True)`` (which also pins key names and shapes), runs the cases below
on CPU and stores inputs-by-recipe + expected outputs in ``tests/golden/*.npz``.

Only data is stored (seeds, shapes, expected outputs); no reference source travels.
The GPU box never runs this script (``/root/reference`` does not exist there).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def import_reference():
    def lev(a, b):
        a, b = list(a), list(b)
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]

    for name in ("pydub", "python_speech_features", "editdistance"):
        m = types.ModuleType(name)
        m.AudioSegment = object
        m.logfbank = None
        m.eval = lev
        sys.modules[name] = m
    sys.path.insert(0, REF)
    from model.las_model import LAS, Listener, Speller          # noqa: E402
    from solver.solver import batch_iterator, label_smoothing_loss  # noqa: E402
    return LAS, Listener, Speller, batch_iterator, label_smoothing_loss


from las_pytorch_amd import synth  # noqa: E402


def build_ref(LAS, Listener, Speller, c, sd_np, *, max_label_len, decode_mode=1, multi_head=1, use_mlp=True,
              activate="relu"):
    listener = Listener(input_feature_dim=c["F"], hidden_size=c["H"], num_layers=c["L"], rnn_unit="LSTM",
                        use_gpu=False)
    speller = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"],
                      max_label_len=max_label_len, use_mlp_in_attention=use_mlp, mlp_dim_in_attention=c["M"],
                      mlp_activate_in_attention=activate, listener_hidden_size=c["H"], multi_head=multi_head,
                      decode_mode=decode_mode, use_gpu=False)
    las = LAS(listener, speller)
    las.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    return las


def stack(lst):
    return torch.stack(lst, 0).detach().numpy()


def run_case(refmods, name, cfg_name, B, T, U, *, scale=None, seed=17, multi_head=1, use_mlp=True,
             activate="relu", free_len=None, full=True, with_grads=True, ragged=False, sub_t=1, sub_d=1, light=False, sub_u=1):
    LAS, Listener, Speller, batch_iterator, ls_loss = refmods
    c = synth.CONFIGS[cfg_name]
    shapes = synth.config_shapes(cfg_name, multi_head=multi_head, use_mlp=use_mlp)
    sd_np = synth.make_state_dict(shapes, seed=seed, scale=scale)
    x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=seed))
    idx, lens = synth.make_labels(B, U, c["V"], seed=seed, ragged=ragged)
    labels = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"]))
    free_len = free_len or U
    out = dict(meta=np.array([B, T, U, seed, multi_head, int(use_mlp), free_len, int(ragged), sub_t, sub_d], dtype=np.int64),
               scale=np.array([-1.0 if scale is None else scale], dtype=np.float64),
               cfg=np.array(cfg_name), activate=np.array(activate))

    las = build_ref(LAS, Listener, Speller, c, sd_np, max_label_len=free_len, decode_mode=1,
                    multi_head=multi_head, use_mlp=use_mlp, activate=activate)
    # G1: listener output per layer
    feats, h = [], x
    for l in range(c["L"]):
        h, _ = getattr(las.listener, f"pLSTM_layer{l}")(h)
        feats.append(h)
    for l, f in enumerate(feats):
        out[f"listener_l{l}"] = f.detach().numpy()[:, ::sub_t, ::sub_d].copy()
        out[f"listener_l{l}_sum"] = np.array([f.double().sum().item(), f.double().abs().sum().item()])

    # G2: teacher forced (tf_rate=1 → coin always true)
    preds, atts = las(batch_data=x, batch_label=labels, teacher_force_rate=1.0, is_training=True)
    out["tf_logp"] = stack(preds)                                   # (U,B,V)
    att = np.stack([stack(h_) for h_ in zip(*atts)], 0)             # (heads,U,B,T')
    out["tf_att"] = att[:, :, :, ::sub_t] if not full else att
    out["tf_att_sum"] = np.array([float(att.astype(np.float64).sum())])
    out["tf_argmax"] = out["tf_logp"].argmax(-1)
    top2 = np.sort(out["tf_logp"], axis=-1)[..., -2:]
    out["tf_margin"] = np.array([float((top2[..., 1] - top2[..., 0]).min())])

    # G3: greedy (is_training=False → free run for max_label_len steps, decode_mode 1)
    preds, _ = las(batch_data=x, batch_label=labels, teacher_force_rate=0.0, is_training=False)
    out["greedy_logp"] = stack(preds)
    out["greedy_argmax"] = out["greedy_logp"].argmax(-1)
    top2 = np.sort(out["greedy_logp"], axis=-1)[..., -2:]
    out["greedy_margin"] = np.array([float((top2[..., 1] - top2[..., 0]).min())])
    if sub_u > 1:      # long decodes: keep every sub_u-th step of the greedy log-probs (the arg-max sequence stays complete)
        out["greedy_logp"] = out["greedy_logp"][::sub_u].copy()
        out["sub_u"] = np.array([sub_u])

    # G4: decode_mode 0 free-run (feeds log-probs back); the headline-size ("light") cases keep the fixture small
    if not light:
        las.speller.decode_mode = 0
        preds, _ = las(batch_data=x, batch_label=labels, teacher_force_rate=0.0, is_training=False)
        out["mode0_logp"] = stack(preds)
        las.speller.decode_mode = 1

    # G5: losses, grads, clip, Adam step
    if with_grads:
        for kind, ls in ((("ls", 0.1),) if light else (("ls", 0.1), ("nll", 0.0))):
            las.zero_grad()
            preds, _ = las(batch_data=x, batch_label=labels, teacher_force_rate=1.0, is_training=True)
            pred_y = torch.cat([p.unsqueeze(1) for p in preds], 1)
            if ls > 0:
                loss = ls_loss(pred_y, labels.float(), label_smoothing=ls)
            else:
                loss = torch.nn.NLLLoss(ignore_index=0)(pred_y.permute(0, 2, 1), labels.argmax(2))
            loss.backward()
            out[f"loss_{kind}"] = np.array([loss.item()])
            gn = {k: p.grad.detach().numpy() for k, p in las.named_parameters()}
            out[f"gradnorm_{kind}"] = np.array([np.linalg.norm(g.astype(np.float64)) for g in gn.values()])
            out[f"gradtotal_{kind}"] = np.array([np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in gn.values()))])
            if kind == "ls":
                for k, g in gn.items():
                    out["grad/" + k] = g.copy() if full else g.reshape(-1)[:: max(1, g.size // 64)][:64].copy()
        # one full solver step through the reference's own batch_iterator (loss → bwd → clip 1.0 → Adam 2e-4)
        las2 = build_ref(LAS, Listener, Speller, c, sd_np, max_label_len=free_len, multi_head=multi_head,
                         use_mlp=use_mlp, activate=activate)
        opt = torch.optim.Adam(las2.parameters(), lr=2e-4)           # train.py:82, yaml lr
        np.random.seed(0)
        loss_np, ler = batch_iterator(x, labels, las2, opt, tf_rate=1.0, is_training=True, max_label_len=U,
                                      label_smoothing=0.1, use_gpu=False)
        out["step_loss"] = np.array([float(loss_np)])
        out["step_ler"] = np.array(ler, dtype=np.float64)
        out["step_param_sum"] = np.array([p.detach().double().sum().item() for p in las2.parameters()])
        out["step_param_delta"] = np.array([
            (p.detach().double() - torch.from_numpy(sd_np[k]).double()).abs().sum().item()
            for k, p in las2.named_parameters()])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    size = os.path.getsize(os.path.join(HERE, name + ".npz"))
    print(f"{name}: {size/1024:.1f} KB  tf_margin={out['tf_margin'][0]:.2e} greedy_margin={out['greedy_margin'][0]:.2e} "
          f"greedy={out['greedy_argmax'][:, 0].tolist()[:12]}")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    refmods = import_reference()
    # tiny config, stored in full (two weight scales; the 0.5 set saturates gates and activates the clip)
    run_case(refmods, "tiny_default", "tiny", B=2, T=16, U=5, free_len=7)
    run_case(refmods, "tiny_sat", "tiny", B=3, T=32, U=6, scale=0.5, free_len=9, ragged=True)
    run_case(refmods, "tiny_mh4", "tiny", B=2, T=16, U=5, multi_head=4, scale=0.3)
    run_case(refmods, "tiny_nomlp", "tiny", B=2, T=16, U=5, use_mlp=False, scale=0.3)
    run_case(refmods, "tiny_noact", "tiny", B=2, T=16, U=5, activate="None", scale=0.3)
    main_act(refmods)
    # S / P short utterances: full outputs, gradient slices
    run_case(refmods, "S_short", "S", B=4, T=64, U=8, full=False)
    run_case(refmods, "S_short_sat", "S", B=4, T=64, U=8, scale=0.2, full=False, ragged=True)
    run_case(refmods, "P_short", "P", B=4, T=64, U=8, full=False)
    run_case(refmods, "P_short_sat", "P", B=3, T=64, U=8, scale=0.15, full=False, ragged=True)
    # S / P LibriSpeech-shaped T=800: subsampled listener output, full log-probs
    run_case(refmods, "S_T800", "S", B=2, T=800, U=12, full=False, with_grads=False, sub_t=10, sub_d=8)
    run_case(refmods, "P_T800", "P", B=2, T=800, U=12, full=False, with_grads=False, sub_t=5, sub_d=16)
    main_more(refmods)
    main_big(refmods)


def main_act(refmods):
    """mlp_activate_in_attention other than relu (the reference resolves any torch.nn.functional name, las_model.py:270-273)."""
    run_case(refmods, "tiny_tanh", "tiny", B=2, T=16, U=5, activate="tanh", scale=0.3)
    run_case(refmods, "tiny_sigmoid", "tiny", B=3, T=16, U=4, activate="sigmoid", scale=0.3, multi_head=2)


def main_more(refmods):
    """The reference's shipped YAML sizes (Listener 512x3 / Speller 1024x2: 16 workgroups per sequence, per-step decode path) and a
    batch beyond one utterance per resident group at H=256 (B=40: the multi-utterance recurrence kernels), short utterances."""
    run_case(refmods, "Y_short", "Y", B=3, T=64, U=6, scale=0.08, full=False, light=True, sub_d=8)
    # seed chosen for a top-1 / top-2 log-prob margin well above fp32 noise (seed 17 gives 2.6e-6: a coin flip for any fp32 path)
    run_case(refmods, "P_B40_T64_U6", "P", B=40, T=64, U=6, full=False, light=True, ragged=True, sub_t=2, sub_d=8, seed=29, scale=0.1)


def main_big(refmods):
    """Headline-size cases (BASELINE.json configs[1], [2], [4]): the backward pass of the persistent kernels at the sizes
    the benchmark runs — 400-step BPTT chains on 256 resident workgroups, the attention backward split over 2 / 4 / 8
    workgroups per utterance — pinned to the reference's loss, per-parameter gradient norms and 64-element gradient
    slices, plus one whole solver step.  Stored subsampled ("light": no decode_mode-0 run, no NLL loss)."""
    big = dict(full=False, light=True, sub_t=10, sub_d=32)
    run_case(refmods, "P_B32_T800_U32", "P", B=32, T=800, U=32, **big)              # T'=100: 2 attention slices
    run_case(refmods, "P_B16_T1600_U8", "P", B=16, T=1600, U=8, ragged=True, **big)  # T'=200: 4 slices
    run_case(refmods, "P_B8_T3000_U16", "P", B=8, T=3000, U=16, **big)               # T'=375: 8 slices (configs[4])
    run_case(refmods, "S_B32_T800_U32", "S", B=32, T=800, U=32, ragged=True, **big)  # T'=200
    run_case(refmods, "S_B8_T3000_U8", "S", B=8, T=3000, U=8, **big)                 # T'=750
    main_big3(refmods)
    main_r6(refmods)


def main_big3(refmods):
    """Round 3: (a) the benchmark's EXACT shape (P, B=32, T=800, U=128: all 128 steps of the one-launch decode kernels, forward
    and backward); (b) headline-size cases with scaled weights, whose greedy arg-max sequences are not the constant
    [1,1,1,...] the default-initialised model emits (the printed greedy_margin / sequence shows it)."""
    big = dict(full=False, light=True, sub_t=10, sub_d=32)
    if os.environ.get("SKIP_U128") != "1":
        run_case(refmods, "P_B32_T800_U128", "P", B=32, T=800, U=128, sub_u=4, **big)
    run_case(refmods, "P_B32_T800_U32_s", "P", B=32, T=800, U=32, scale=0.2, seed=43, **big)      # greedy margin 7.5e-4, ~6 symbol changes per utterance; at scale >= 0.25 the saturated attention turns chaotic (oracle vs reference 2e-3)
    run_case(refmods, "P_B8_T3000_U16_s", "P", B=8, T=3000, U=16, scale=0.2, seed=43, **big)
    main_big4(refmods)


def main_big4(refmods):
    """Round 4: (a) 128 utterances per GPU at the benchmark's T (the matrix-pipe recurrences run 400 / 200 / 100 steps forward and
    backward, the decode loop takes more than 32 utterances); (b) the reference's shipped YAML sizes at T = 800, B = 16;
    (c) multi-head attention at paper size (heads = 4 and 2)."""
    big = dict(full=False, light=True, sub_t=10, sub_d=32)
    which = os.environ.get("BIG4", "abcd")
    if "a" in which:
        # default-scale weights: seed 17 has a top-1 / top-2 margin of 2.6e-3 over the 2048 greedy positions (most seeds: ~1e-6, a coin flip
        # for any fp32 path), its greedy sequence is one repeated symbol; the "_s" case (U(-0.2, 0.2) weights, seed 51: margin 1.4e-3)
        # decodes 24 distinct symbols
        big128 = dict(full=False, light=True, sub_t=20, sub_d=32)
        run_case(refmods, "P_B128_T800_U16", "P", B=128, T=800, U=16, ragged=True, seed=17, **big128)
        run_case(refmods, "P_B128_T800_U16_s", "P", B=128, T=800, U=16, ragged=True, seed=51, scale=0.2, **big128)
    if "b" in which:
        # seed chosen with tests/golden/seed_search.py: smallest top-1 / top-2 gap over the 256 greedy positions 8.8e-4 (two distinct symbols);
        # round 4's seed 37 had 7.2e-5, a thin margin for an arg-max identity test
        run_case(refmods, "Y_B16_T800_U16", "Y", B=16, T=800, U=16, scale=0.05, seed=69, **big)
    if "d" in which:
        # the YAML sizes with a 24-second utterance batch (T = 2400 -> T' = 300 > 256): the LONG instantiation of the one-launch decode kernels
        # (seed from tests/golden/seed_search.py: greedy / teacher-forced top-1 / top-2 gap 1.8e-2; round 4's seed 41 had 2.2e-4)
        run_case(refmods, "Y_B4_T2400_U8", "Y", B=4, T=2400, U=8, scale=0.05, seed=45, ragged=True, **big)
    if "c" in which:
        run_case(refmods, "P_short_mh4", "P", B=4, T=64, U=8, multi_head=4, scale=0.1, full=False, light=True, seed=23)
        run_case(refmods, "P_B32_T800_U16_mh2", "P", B=32, T=800, U=16, multi_head=2, scale=0.1, seed=23, **big)




def main_r6(refmods):
    """Round 6: (a) BASELINE configs[4] at the benchmark's own decode length (P, B = 8, T = 3000, U = 128: the hand-off rings of the
    16-workgroups-per-utterance decode kernels wrap 128 times; bench.py's secondary_long block asserts its first-step loss against
    this fixture's); (b) multi-head attention at real sizes: heads = 4 at (16, 800) and heads = 2 at T = 3000."""
    big = dict(full=False, light=True, sub_t=10, sub_d=32)
    which = os.environ.get("R6", "abc")
    if "a" in which:
        run_case(refmods, "P_B8_T3000_U128", "P", B=8, T=3000, U=128, sub_u=4, **big)
    if "b" in which:
        run_case(refmods, "P_B16_T800_U16_mh4", "P", B=16, T=800, U=16, multi_head=4, scale=0.1, seed=23, **big)
    if "c" in which:
        run_case(refmods, "P_B8_T3000_U8_mh2", "P", B=8, T=3000, U=8, multi_head=2, scale=0.1, seed=23, **big)


def make_mode2_golden(refmods):
    """decode_mode 2 (reference las_model.py:229-234, ``Categorical(raw_pred).sample()``): run the UNMODIFIED reference
    with a seeded torch generator and store its log-probs together with the Exp(1) draws torch.multinomial consumed
    (re-drawn from the same seed: one (B,V) ``exponential_`` per decode step is the loop's only use of the generator).
    A consumer that evaluates the sample as argmax_v p_v / q_v, p = logp / sum logp, must reproduce the stream."""
    LAS, Listener, Speller, _, _ = refmods
    for name, cfg_name, B, T, U, scale, seed in (("tiny_mode2", "tiny", 3, 16, 9, 0.3, 41), ("S_mode2", "S", 4, 64, 12, None, 43)):
        c = synth.CONFIGS[cfg_name]
        sd_np = synth.make_state_dict(synth.config_shapes(cfg_name), seed=17, scale=scale)
        x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=17))
        las = build_ref(LAS, Listener, Speller, c, sd_np, max_label_len=U, decode_mode=2)
        torch.manual_seed(seed)
        with torch.no_grad():
            preds, _ = las(batch_data=x, batch_label=None, teacher_force_rate=0.0, is_training=False)
        torch.manual_seed(seed)
        noise = torch.stack([torch.empty(B, c["V"]).exponential_(1) for _ in range(U)])
        logp = stack(preds)
        # (tests/test_oracle_golden.py::test_decode_mode2_sampling_semantics replays the samples from the stored draws and
        # must reproduce these log-probs at every step — that is what pins the stated semantics)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg=np.array(cfg_name), dims=np.array([B, T, U, seed]),
                            scale=np.array([-1.0 if scale is None else scale]), mode2_logp=logp, mode2_noise=noise.numpy())
        print(f"{name}: argmax path {logp.argmax(-1)[:, 0].tolist()}")


def make_trajectory_golden(refmods):
    """A short TRAINING RUN of the unmodified reference: 8 consecutive ``batch_iterator`` steps (teacher forcing, label
    smoothing 0.1, clip 1.0, Adam) on fixed data, S config.  Stored: the loss and the per-utterance letter error rates of
    every step and the parameters' checksums at the end — the drop-in must walk the same trajectory (BASELINE metric: LER parity)."""
    LAS, Listener, Speller, batch_iterator, _ = refmods
    c = synth.CONFIGS["S"]
    sd_np = synth.make_state_dict(synth.config_shapes("S"), seed=51, scale=0.08)
    B, T, U, steps, lr = 6, 64, 9, 8, 3e-3
    x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=51))
    idx, lens = synth.make_labels(B, U, c["V"], seed=51, ragged=True)
    labels = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"]))
    las = build_ref(LAS, Listener, Speller, c, sd_np, max_label_len=U)
    opt = torch.optim.Adam(las.parameters(), lr=lr)
    losses, lers = [], []
    np.random.seed(0)
    for _ in range(steps):
        loss, ler = batch_iterator(x, labels, las, opt, tf_rate=1.0, is_training=True, max_label_len=U, label_smoothing=0.1, use_gpu=False)
        losses.append(float(loss)); lers.append(ler)
    # validation call after training (free-running greedy decode, NLL loss)
    vloss, vler = batch_iterator(x, labels, las, opt, tf_rate=0.0, is_training=False, max_label_len=U, label_smoothing=0.1, use_gpu=False)
    np.savez_compressed(os.path.join(HERE, "S_trajectory.npz"), dims=np.array([B, T, U, steps, 51]), lr=np.array([lr]), scale=np.array([0.08]),
                        losses=np.array(losses), lers=np.array(lers, dtype=np.float64), val_loss=np.array([float(vloss)]),
                        val_ler=np.array(vler, dtype=np.float64),
                        param_sum=np.array([p.detach().double().sum().item() for p in las.parameters()]),
                        param_abs=np.array([p.detach().double().abs().sum().item() for p in las.parameters()]))
    print("S_trajectory: losses", [round(l, 5) for l in losses], "val", float(vloss), "ler", lers[-1])


def make_free_training_golden(refmods, which="tsp"):
    """Round 5: a TRAINING step of the unmodified reference that takes the free-running branch (teacher_force_rate 0 with is_training
    True: las_model.py:189 draws False, the loop runs max_label_len steps feeding back the one-hot arg-max, :205-227) followed by the
    label-smoothing loss and ``backward()``.  The fed-back symbols are not differentiable, so the gradient flows through the decoder state
    and the context only — what the drop-in's free-running forward + teacher-forced-style backward pair must reproduce.  Stored: log-probs,
    arg-max sequences and their top-1 / top-2 margin, the loss, per-parameter gradient norms and 64-element gradient slices."""
    LAS, Listener, Speller, batch_iterator, ls_loss = refmods
    cases = []
    if "t" in which: cases.append(("tiny_free_train", "tiny", 3, 32, 6, 0.3, 17))
    if "s" in which: cases.append(("S_free_train", "S", 4, 64, 8, 0.2, 43))
    if "p" in which: cases.append(("P_B32_T800_U16_free_train", "P", 32, 800, 16, 0.2, 43))      # weights / inputs of P_B32_T800_U32_s
    # round 6: the multi-head free-running training step at paper size (heads = 2: one slice of 16 utterances per launch)
    if "m" in which: cases.append(("P_B16_T800_U12_mh2_free_train", "P", 16, 800, 12, 0.1, 23, 2))
    # round 6: decode_mode 0 TRAINING steps (las_model.py:219-221: the raw log-probabilities are fed back, so — unlike mode 1 — the gradient
    # also flows through the fed-back input, :198-203 via the concatenation): what las_speller_bwd's feedback_mode0 path must reproduce
    if "0" in which:
        cases += [("tiny_mode0_train", "tiny", 3, 32, 6, 0.3, 17, 1, 0), ("S_mode0_train", "S", 4, 64, 8, 0.2, 43, 1, 0),
                  ("P_mode0_train", "P", 8, 128, 6, 0.15, 43, 1, 0)]
    for name, cfg_name, B, T, U, scale, seed, *rest in cases:
        heads = rest[0] if rest else 1
        dmode = rest[1] if len(rest) > 1 else 1
        c = synth.CONFIGS[cfg_name]
        sd_np = synth.make_state_dict(synth.config_shapes(cfg_name, multi_head=heads), seed=seed, scale=scale)
        x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=seed))
        idx, lens = synth.make_labels(B, U, c["V"], seed=seed, ragged=(cfg_name != "P"))
        labels = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"]))
        las = build_ref(LAS, Listener, Speller, c, sd_np, max_label_len=U, decode_mode=dmode, multi_head=heads)
        las.zero_grad()
        preds, _ = las(batch_data=x, batch_label=labels, teacher_force_rate=0.0, is_training=True)
        assert len(preds) == U
        pred_y = torch.cat([p.unsqueeze(1) for p in preds], 1)
        loss = ls_loss(pred_y, labels.float(), label_smoothing=0.1)
        loss.backward()
        logp = stack(preds)
        top2 = np.sort(logp, axis=-1)[..., -2:]
        gn = {k: p.grad.detach().numpy() for k, p in las.named_parameters()}
        out = dict(meta=np.array([B, T, U, seed], dtype=np.int64), scale=np.array([scale]), cfg=np.array(cfg_name), heads=np.array([heads]),
                   decode_mode=np.array([dmode]),
                   ragged=np.array([int(cfg_name != "P")]), free_logp=logp, free_argmax=logp.argmax(-1),
                   free_margin=np.array([float((top2[..., 1] - top2[..., 0]).min())]), loss_ls=np.array([loss.item()]),
                   gradnorm_ls=np.array([np.linalg.norm(g.astype(np.float64)) for g in gn.values()]),
                   gradtotal_ls=np.array([np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in gn.values()))]),
                   grad_keys=np.array(list(gn.keys())))
        for k, g in gn.items():
            out["grad/" + k] = g.reshape(-1)[:: max(1, g.size // 64)][:64].copy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(f"{name}: {os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024:.1f} KB margin={out['free_margin'][0]:.2e} loss={loss.item():.5f} "
              f"distinct symbols={len(np.unique(out['free_argmax']))} first utterance={out['free_argmax'][:, 0].tolist()}")


def make_collate_golden():
    """The reference's own ``collate_fn`` (utils/data.py:116-149) on a synthetic ragged batch.  ``utils/data.py`` imports
    torchaudio / enlighten / pydub at module top (unused by collate_fn): stubbed in sys.modules, nothing else is touched."""
    for name in ("pydub", "python_speech_features", "editdistance", "enlighten", "torchaudio"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    ta = sys.modules["torchaudio"]
    ta.compliance = types.SimpleNamespace(kaldi=types.SimpleNamespace(fbank=None))
    sys.modules["pydub"].AudioSegment = object
    sys.modules["python_speech_features"].logfbank = None
    sys.path.insert(0, REF)
    from utils.data import OneHotEncode, collate_fn          # noqa: E402
    rng = np.random.default_rng(7)
    lens = [(211, 17), (96, 30), (160, 1), (33, 8), (200, 22)]
    batch = []
    for i, (t, u) in enumerate(lens):
        feat = rng.standard_normal((t, 40)).astype(np.float32)
        idx = rng.integers(2, 30, size=u)
        targets = [OneHotEncode(j, 30) for j in idx]              # AudioDataset.__getitem__ (data.py:78-79)
        batch.append((f"utt{i}", feat, t, targets, [len(r) for r in targets]))
    ids, feature, label = collate_fn(batch)
    np.savez_compressed(os.path.join(HERE, "collate_case.npz"), seed=np.array([7]), lens=np.array(lens),
                        inputs_sum=np.array([feature["inputs"].double().sum().item(), feature["inputs"].double().abs().sum().item()]),
                        inputs_shape=np.array(feature["inputs"].shape), inputs_tail=feature["inputs"][:, -8:, :4].numpy(),
                        inputs_head=feature["inputs"][:, :4, :4].numpy(),
                        inputs_length=feature["inputs_length"].numpy(), targets=label["targets"].numpy(),
                        targets_length=label["targets_length"].numpy())
    print("collate_case written:", tuple(feature["inputs"].shape), tuple(label["targets"].shape), feature["inputs"].dtype, label["targets"].dtype)


def make_init_golden():
    """Seeded initialisation of the reference modules (torch.manual_seed(17), train.py:41): per-parameter checksums.
    The drop-in modules must consume the RNG identically (same registration order and init calls)."""
    LAS, Listener, Speller, _, _ = import_reference()
    out = {}
    for cfg_name in ("S", "P"):
        c = synth.CONFIGS[cfg_name]
        torch.manual_seed(17)
        listener = Listener(input_feature_dim=c["F"], hidden_size=c["H"], num_layers=c["L"], rnn_unit="LSTM", use_gpu=False)
        speller = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=8,
                          use_mlp_in_attention=True, mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu",
                          listener_hidden_size=c["H"], multi_head=1, decode_mode=1, use_gpu=False)
        las = LAS(listener, speller)
        out[f"{cfg_name}_sum"] = np.array([p.detach().double().sum().item() for p in las.parameters()])
        out[f"{cfg_name}_abs"] = np.array([p.detach().double().abs().sum().item() for p in las.parameters()])
        out[f"{cfg_name}_first"] = np.array([p.detach().reshape(-1)[0].item() for p in las.parameters()])
    np.savez_compressed(os.path.join(HERE, "init_seed17.npz"), **out)
    print("init_seed17 written")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "init":
    make_init_golden()

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "more":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main_more(import_reference())

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "traj":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    make_trajectory_golden(import_reference())

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "act":
    torch.manual_seed(0)
    main_act(import_reference())

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "big3":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main_big3(import_reference())

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "big4":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main_big4(import_reference())

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "extra":
    torch.set_num_threads(8)
    make_mode2_golden(import_reference())
    make_trajectory_golden(import_reference())
    make_collate_golden()

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "free":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    make_free_training_golden(import_reference(), sys.argv[2] if len(sys.argv) > 2 else "tsp")

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "r6":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main_r6(import_reference())

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "big":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main_big(import_reference())


if __name__ == "__main__" and len(sys.argv) == 1:
    main()
    make_init_golden()
    make_mode2_golden(import_reference())
    make_trajectory_golden(import_reference())
    make_collate_golden()
    make_free_training_golden(import_reference())


# tests/golden/ref_checkpoint_tiny.pth.tar: a checkpoint package written by the REFERENCE's own LAS.serialize
# (model/las_model.py:42-63) after one reference batch_iterator step on the tiny config (seed 29, scale 0.2); produced with
#   build_ref(...); batch_iterator(...); torch.save(las.serialize(opt, 3, 1.25, 2.5), "ref_checkpoint_tiny.pth.tar")
# and consumed by tests/test_cabi_and_host.py::test_reference_checkpoint_loads (checkpoint interop).
