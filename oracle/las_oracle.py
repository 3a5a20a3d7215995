"""CPU oracle for the LAS hot path — TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, in explicit per-time-step torch CPU ops, of the
arithmetic the reference delegates to ``torch.nn.LSTM`` / ``nn.Linear`` / ``torch.bmm``.
It is the *checker* for the HIP kernels: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``las_pytorch_amd``) never imports anything under ``oracle/``.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the unmodified reference
(``/root/reference/model/las_model.py``) in the build container, drives it with the
deterministic weights/inputs of ``las_pytorch_amd.synth`` and stores its outputs under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this oracle against those
vectors (<=1e-6 abs in fp32).  The reference itself has no tests or golden vectors
(SURVEY.md section 4).

Reference lines followed (all ``/root/reference/...``):
  * pyramid reshape + BiLSTM ............ model/las_model.py:81-91
  * listener stack ...................... model/las_model.py:129-134
  * speller step ........................ model/las_model.py:178-184
  * speller loop / init / feedback ...... model/las_model.py:186-238
  * attention (single + multi head) ..... model/las_model.py:275-318
  * TimeDistributed / CreateOnehot ...... utils/functions.py:54-63,72-77
  * label-smoothing loss / NLL / LER .... solver/solver.py:11-24,33-45,61-97
The LSTM cell equations are torch's (third-party, ``torch==1.5.0`` pinned by the
reference's requirements.txt:5): gates stacked i,f,g,o; two bias vectors; zero initial state.
"""
from __future__ import annotations

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# LSTM primitives
# --------------------------------------------------------------------------------------


def lstm_cell(x_t, h, c, w_ih, w_hh, b_ih, b_hh):
    """One LSTM cell step, PyTorch gate order i,f,g,o (torch.nn.LSTM semantics,
    call sites model/las_model.py:90,179)."""
    gates = x_t @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    H = h.shape[-1]
    i = torch.sigmoid(gates[..., 0 * H:1 * H])
    f = torch.sigmoid(gates[..., 1 * H:2 * H])
    g = torch.tanh(gates[..., 2 * H:3 * H])
    o = torch.sigmoid(gates[..., 3 * H:4 * H])
    c_new = f * c + i * g
    h_new = o * torch.tanh(c_new)
    return h_new, c_new


def lstm_direction(xr, w_ih, w_hh, b_ih, b_hh, reverse):
    """Unidirectional LSTM over (B,T,D) with zero initial state; ``reverse`` runs
    t = T-1..0 and stores h_t at position t (nn.LSTM bidirectional semantics)."""
    B, T, _ = xr.shape
    H = w_hh.shape[1]
    h = xr.new_zeros(B, H)
    c = xr.new_zeros(B, H)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        h, c = lstm_cell(xr[:, t, :], h, c, w_ih, w_hh, b_ih, b_hh)
        outs[t] = h
    return torch.stack(outs, dim=1)


def pblstm_layer(x, sd, prefix):
    """model/las_model.py:81-91 — halve time by concatenating frame pairs, then BiLSTM.
    ``sd`` maps reference state_dict keys to tensors; ``prefix`` e.g.
    ``listener.pLSTM_layer0.BLSTM.``."""
    B, T, D = x.shape
    if T % 2 != 0:
        raise RuntimeError("pBLSTM needs an even number of frames (las_model.py:86-87)")
    xr = x.contiguous().view(B, T // 2, 2 * D)
    fwd = lstm_direction(xr, sd[prefix + "weight_ih_l0"], sd[prefix + "weight_hh_l0"],
                         sd[prefix + "bias_ih_l0"], sd[prefix + "bias_hh_l0"], False)
    bwd = lstm_direction(xr, sd[prefix + "weight_ih_l0_reverse"], sd[prefix + "weight_hh_l0_reverse"],
                         sd[prefix + "bias_ih_l0_reverse"], sd[prefix + "bias_hh_l0_reverse"], True)
    return torch.cat([fwd, bwd], dim=-1)


def listener_forward(x, sd, num_layers, prefix="listener.", return_all=False):
    """model/las_model.py:129-134 — every layer (layer 0 included) halves time."""
    outs = []
    out = x
    for l in range(num_layers):
        out = pblstm_layer(out, sd, f"{prefix}pLSTM_layer{l}.BLSTM.")
        outs.append(out)
    return outs if return_all else out


# --------------------------------------------------------------------------------------
# Attention + Speller
# --------------------------------------------------------------------------------------


def _act(name):
    if name is None or name == "None":
        return None
    return getattr(torch.nn.functional, name)


def attention_forward(dec_state, feat, sd, prefix, use_mlp, activate, multi_head, keys=None):
    """model/las_model.py:275-318.  ``dec_state`` (B,Hs), ``feat`` (B,T',2H).
    Returns (list of per-head scores (B,T'), context (B,2H)).  ``keys`` may carry the
    loop-invariant psi(feat) (the reference recomputes it every step, :279; hoisting is
    result-identical)."""
    act = _act(activate) if use_mlp else None
    if use_mlp:
        q = dec_state @ sd[prefix + "phi.weight"].t() + sd[prefix + "phi.bias"]
        if keys is None:
            keys = attention_keys(feat, sd, prefix, use_mlp, activate)
        if act is not None:
            q = act(q)
    else:
        q = dec_state
        keys = feat
    if multi_head == 1:
        energy = torch.einsum("bm,btm->bt", q, keys)
        score = torch.softmax(energy, dim=-1)
        ctx = torch.einsum("bt,btd->bd", score, feat)
        return [score], ctx
    M = keys.shape[-1]
    scores, proj = [], []
    for hd in range(multi_head):
        e = torch.einsum("bm,btm->bt", q[:, hd * M:(hd + 1) * M], keys)
        s = torch.softmax(e, dim=-1)
        scores.append(s)
        proj.append(torch.einsum("bt,btd->bd", s, feat))
    ctx = torch.cat(proj, dim=-1) @ sd[prefix + "dim_reduce.weight"].t() + sd[prefix + "dim_reduce.bias"]
    return scores, ctx


def attention_keys(feat, sd, prefix, use_mlp, activate):
    """psi(feat) with activation — utils/functions.py:72-77 applied at las_model.py:279."""
    if not use_mlp:
        return feat
    B, T, D = feat.shape
    k = (feat.reshape(B * T, D) @ sd[prefix + "psi.weight"].t() + sd[prefix + "psi.bias"]).view(B, T, -1)
    act = _act(activate)
    return act(k) if act is not None else k


def speller_forward(feat, sd, *, num_layers, max_label_len, decode_mode, ground_truth=None,
                    teacher_force=False, use_mlp=True, activate="relu", multi_head=1,
                    prefix="speller.", sample_fn=None):
    """model/las_model.py:186-238.  ``teacher_force`` is the already-flipped coin
    (the reference draws ONE np.random.random_sample() per call, :189; callers of the
    oracle decide it explicitly).  ``ground_truth`` is the (B,U,V) one-hot label tensor.
    Returns (list[U] of (B,V) log-probs, list[U] of list[heads] of (B,T'))."""
    B = feat.shape[0]
    V = sd[prefix + "character_distribution.weight"].shape[0]
    Hs = sd[prefix + "rnn_layer.weight_hh_l0"].shape[1]
    if feat.shape[-1] != Hs:
        raise RuntimeError("Speller hidden_size must equal 2*listener_hidden_size (las_model.py:198)")
    if ground_truth is None:
        teacher_force = False
    y = feat.new_zeros(B, V)
    y[:, 0] = 1.0                                   # onehot(<sos>=0), las_model.py:193-195
    ctx = feat[:, 0, :]                             # las_model.py:198
    h = [feat.new_zeros(B, Hs) for _ in range(num_layers)]
    c = [feat.new_zeros(B, Hs) for _ in range(num_layers)]
    keys = attention_keys(feat, sd, prefix + "attention.", use_mlp, activate)
    steps = ground_truth.shape[1] if teacher_force else max_label_len   # :205-208
    preds, atts = [], []
    for s in range(steps):
        x = torch.cat([y, ctx], dim=-1)
        for l in range(num_layers):
            h[l], c[l] = lstm_cell(x, h[l], c[l],
                                   sd[f"{prefix}rnn_layer.weight_ih_l{l}"], sd[f"{prefix}rnn_layer.weight_hh_l{l}"],
                                   sd[f"{prefix}rnn_layer.bias_ih_l{l}"], sd[f"{prefix}rnn_layer.bias_hh_l{l}"])
            x = h[l]
        scores, ctx = attention_forward(x, feat, sd, prefix + "attention.", use_mlp, activate, multi_head, keys)
        logits = torch.cat([x, ctx], dim=-1) @ sd[prefix + "character_distribution.weight"].t() \
            + sd[prefix + "character_distribution.bias"]
        logp = torch.log_softmax(logits, dim=-1)
        preds.append(logp)
        atts.append(scores)
        if teacher_force:
            y = ground_truth[:, s, :].to(feat.dtype)            # :216-217
        elif decode_mode == 0:
            y = logp                                            # :220-221 (log-probs fed back)
        elif decode_mode == 1:
            y = torch.zeros_like(logp)                          # :223-227
            y[torch.arange(B), logp.argmax(dim=-1)] = 1.0
        else:
            if sample_fn is None:
                raise NotImplementedError("decode_mode 2 needs sample_fn (las_model.py:229-234)")
            idx = sample_fn(logp)
            y = torch.zeros_like(logp)
            y[torch.arange(B), idx] = 1.0
    return preds, atts


def speller_step(input_word, hidden, feat, sd, *, num_layers, use_mlp=True, activate="relu", multi_head=1, prefix="speller."):
    """model/las_model.py:178-184 (Speller.forward_step): input_word (B,V+Hs), hidden = (h,c) each (L,B,Hs) or None."""
    B = feat.shape[0]
    Hs = sd[prefix + "rnn_layer.weight_hh_l0"].shape[1]
    h = [feat.new_zeros(B, Hs) if hidden is None else hidden[0][l] for l in range(num_layers)]
    c = [feat.new_zeros(B, Hs) if hidden is None else hidden[1][l] for l in range(num_layers)]
    x = input_word
    for l in range(num_layers):
        h[l], c[l] = lstm_cell(x, h[l], c[l], sd[f"{prefix}rnn_layer.weight_ih_l{l}"], sd[f"{prefix}rnn_layer.weight_hh_l{l}"],
                               sd[f"{prefix}rnn_layer.bias_ih_l{l}"], sd[f"{prefix}rnn_layer.bias_hh_l{l}"])
        x = h[l]
    scores, ctx = attention_forward(x, feat, sd, prefix + "attention.", use_mlp, activate, multi_head)
    logits = torch.cat([x, ctx], dim=-1) @ sd[prefix + "character_distribution.weight"].t() \
        + sd[prefix + "character_distribution.bias"]
    return torch.log_softmax(logits, dim=-1), (torch.stack(h), torch.stack(c)), ctx, scores


def categorical_sampler(noise):
    """decode_mode 2 (model/las_model.py:229-234) as torch evaluates ``Categorical(raw_pred).sample()``: the log-probs
    passed as ``probs`` are renormalised (p = logp / sum logp, proportional to |log p|) and one sample is drawn as
    argmax_v p_v / q_v with q ~ Exp(1) (torch.multinomial's single-draw path).  ``noise`` (U,B,V) holds the q draws in
    the order the reference consumes them (one (B,V) draw per decode step); returns a ``sample_fn`` for
    ``speller_forward``."""
    it = iter(noise)

    def sample(logp):
        p = logp / logp.sum(dim=-1, keepdim=True)
        return (p / next(it)).argmax(dim=-1)
    return sample


def las_forward(x, labels, sd, cfg, *, teacher_force, is_training=True, sample_fn=None):
    """model/las_model.py:30-40.  ``cfg`` keys: listener_layers, speller_layers,
    max_label_len, decode_mode, use_mlp, activate, multi_head."""
    feat = listener_forward(x, sd, cfg["listener_layers"])
    gt = labels if is_training else None
    return speller_forward(feat, sd, num_layers=cfg["speller_layers"], max_label_len=cfg["max_label_len"],
                           decode_mode=cfg["decode_mode"], ground_truth=gt,
                           teacher_force=teacher_force and is_training,
                           use_mlp=cfg.get("use_mlp", True), activate=cfg.get("activate", "relu"),
                           multi_head=cfg.get("multi_head", 1), sample_fn=sample_fn)


# --------------------------------------------------------------------------------------
# Caller-side contract (solver/solver.py) — loss and LER
# --------------------------------------------------------------------------------------


def label_smoothing_loss(pred_y, true_y, label_smoothing=0.1):
    """solver/solver.py:33-45.  pred_y (B,U,V) log-probs, true_y (B,U,V) one-hot floats."""
    seq_len = true_y.sum(-1).sum(-1, keepdim=True)
    V = true_y.shape[-1]
    smooth = ((1.0 - label_smoothing) * true_y + label_smoothing / V) * true_y.sum(-1, keepdim=True)
    return -torch.mean(torch.sum(torch.sum(smooth * pred_y, dim=-1) / seq_len, dim=-1))


def nll_loss_ignore0(pred_y, true_idx):
    """solver/solver.py:62,70-74 — NLLLoss(ignore_index=0), mean over non-ignored tokens."""
    B, U, V = pred_y.shape
    lp = pred_y.reshape(B * U, V)
    idx = true_idx.reshape(B * U)
    mask = idx != 0
    picked = lp[torch.arange(B * U), idx]
    return -(picked * mask).sum() / mask.sum()


def edit_distance(a, b):
    """Levenshtein distance (the reference uses the ``editdistance`` C extension,
    solver/solver.py:5,23; restated here because it is not installed)."""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def letter_error_rate(pred_y, true_y):
    """solver/solver.py:11-24: drop 0s, stop prediction at first 1, truth drops 0 and 1,
    divide by stripped truth length."""
    out = []
    for p, t in zip(pred_y, true_y):
        ct = [int(w) for w in t if (w != 1 and w != 0)]
        cp = []
        for w in p:
            if w == 0:
                continue
            if w == 1:
                break
            cp.append(int(w))
        out.append(edit_distance(cp, ct) / len(ct))
    return out


def solver_step_loss(preds, labels_onehot, max_label_len, label_smoothing, is_training=True):
    """solver/solver.py:61-92 — stack, truncate, pick the loss kind; returns (loss, ler list)."""
    U = min(labels_onehot.shape[1], max_label_len)
    pred_y = torch.stack(preds, dim=1)[:, :U, :].contiguous()
    if label_smoothing == 0.0 or not is_training:
        true_idx = labels_onehot.argmax(dim=2)[:, :U]
        loss = nll_loss_ignore0(pred_y, true_idx)
        ler = letter_error_rate(pred_y.argmax(dim=2).numpy(), true_idx.numpy())
    else:
        true_y = labels_onehot[:, :U, :].to(pred_y.dtype)
        loss = label_smoothing_loss(pred_y, true_y, label_smoothing)
        ler = letter_error_rate(pred_y.argmax(dim=2).numpy(), true_y.argmax(dim=2).numpy())
    return loss, ler


def to_torch_sd(sd_np, dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in sd_np.items():
        t = torch.as_tensor(np.asarray(v)).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out
