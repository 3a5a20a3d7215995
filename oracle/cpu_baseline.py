"""CPU timing baseline of the LAS hot path — TEST/BENCH INFRASTRUCTURE ONLY (kind: "port").

Same op sequence as the reference's ``model/las_model.py`` built from the same ``torch.nn`` modules
(``nn.LSTM`` -> ATen/oneDNN ``mkldnn_rnn_layer`` on CPU, ``nn.Linear``, ``torch.bmm``, softmax, the
``(B,T',2H)`` repeat/mul/sum context, per-step recomputation of psi — las_model.py:81-91,178-184,275-297),
so that timing it on the GPU box's host cores is timing the reference's CPU path; the reference's own
Python files cannot travel to that box.  Equality with the imported reference is pinned by
``tests/test_oracle_golden.py::test_cpu_baseline_matches_reference`` (golden vectors).
Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import this module.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class RefListener(nn.Module):
    def __init__(self, F_in, H, L):
        super().__init__()
        self.L = L
        for l in range(L):
            d = F_in if l == 0 else 2 * H
            blk = nn.Module()
            blk.BLSTM = nn.LSTM(2 * d, H, 1, bidirectional=True, batch_first=True)
            setattr(self, f"pLSTM_layer{l}", blk)

    def forward(self, x):
        for l in range(self.L):
            B, T, D = x.shape
            x = x.contiguous().view(B, T // 2, 2 * D)                       # las_model.py:86-87
            x, _ = getattr(self, f"pLSTM_layer{l}").BLSTM(x)               # las_model.py:90
        return x


class RefSpeller(nn.Module):
    def __init__(self, V, Hs, Ls, M, H):
        super().__init__()
        self.V, self.Hs = V, Hs
        self.rnn_layer = nn.LSTM(V + Hs, Hs, num_layers=Ls, batch_first=True)
        att = nn.Module()
        att.phi = nn.Linear(2 * H, M)
        att.psi = nn.Linear(2 * H, M)
        self.attention = att
        self.character_distribution = nn.Linear(2 * Hs, V)

    def forward(self, feat, ground_truth, steps, teacher_force=True):
        B = feat.shape[0]
        y = torch.zeros(B, 1, self.V)
        y[:, 0, 0] = 1.0
        rnn_input = torch.cat([y, feat[:, 0:1, :]], dim=-1)                 # las_model.py:193-198
        hidden = None
        preds = []
        for s in range(steps):
            rnn_out, hidden = self.rnn_layer(rnn_input, hidden)             # :179
            q = F.relu(self.attention.phi(rnn_out))                         # :278
            k = F.relu(self.attention.psi(feat.contiguous().view(-1, feat.size(-1)))).view(B, feat.size(1), -1)  # :279 (every step)
            energy = torch.bmm(q, k.transpose(1, 2)).squeeze(dim=1)         # :289-291
            score = torch.softmax(energy, dim=-1)                           # :292
            ctx = torch.sum(feat * score.unsqueeze(2).repeat(1, 1, feat.size(2)), dim=1)   # :293-297
            cat = torch.cat([rnn_out.squeeze(dim=1), ctx], dim=-1)          # :181
            logp = torch.log_softmax(self.character_distribution(cat), dim=-1)   # :182
            preds.append(logp)
            if teacher_force:
                y = ground_truth[:, s:s + 1, :].float()                     # :216-217
            else:
                y = torch.zeros_like(logp)
                for i, j in enumerate(logp.topk(1)[1]):                     # :224-227
                    y[i, int(j)] = 1
                y = y.unsqueeze(1)
            rnn_input = torch.cat([y, ctx.unsqueeze(1)], dim=-1)            # :236
        return preds


class RefLAS(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.listener = RefListener(c["F"], c["H"], c["L"])
        self.speller = RefSpeller(c["V"], c["Hs"], c["Ls"], c["M"], c["H"])

    def forward(self, x, labels, steps, teacher_force=True):
        return self.speller(self.listener(x), labels, steps, teacher_force)


def label_smoothing_loss(pred_y, true_y, label_smoothing=0.1):
    """solver/solver.py:33-45."""
    seq_len = torch.sum(torch.sum(true_y, dim=-1), dim=-1, keepdim=True)
    V = true_y.size()[-1]
    smooth_y = ((1.0 - label_smoothing) * true_y + (label_smoothing / V)) * torch.sum(true_y, dim=-1, keepdim=True)
    return -torch.mean(torch.sum((torch.sum(smooth_y * pred_y, dim=-1) / seq_len), dim=-1))


def build(c, sd_np):
    m = RefLAS(c)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd_np.items()}, strict=True)
    return m


def best_threads(c, sd_np, x, onehot, *, train, candidates=(8, 16, 32)):
    """The decode loop is dispatch-bound (small ops): more threads than ~16-32 make it SLOWER (128 threads on the
    bench host: 16 s/step vs ~2 s).  One probe step per candidate; the fastest is the honest baseline."""
    best, best_t = None, None
    ncpu = torch.get_num_threads()
    for t in candidates:
        if t > max(ncpu, 1):
            continue
        r = time_cpu(c, sd_np, x, onehot, train=train, iters=1, warmup=0, threads=t)
        if best is None or r["ms_per_step"] < best:
            best, best_t = r["ms_per_step"], t
    return best_t or ncpu


def time_cpu(c, sd_np, x, onehot, *, train, iters=3, warmup=1, threads=None):
    """Times fwd (or fwd + label-smoothing loss + bwd + clip 1.0 + Adam 2e-4, solver.py:81-97) on the host.
    Returns dict(utt_per_s, ms_per_step, threads, iters)."""
    if threads:
        torch.set_num_threads(threads)
    m = build(c, sd_np)
    xt = torch.from_numpy(x)
    lab = torch.from_numpy(onehot)
    U = lab.shape[1]
    opt = torch.optim.Adam(m.parameters(), lr=2e-4)
    times = []
    for it in range(warmup + iters):
        t0 = time.perf_counter()
        if train:
            opt.zero_grad()
            preds = m(xt, lab, U)
            loss = label_smoothing_loss(torch.stack(preds, 1), lab.float(), 0.1)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1)
            opt.step()
        else:
            with torch.no_grad():
                m(xt, lab, U)
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    mean = float(np.mean(times))
    return dict(utt_per_s=x.shape[0] / mean, ms_per_step=mean * 1e3, threads=torch.get_num_threads(), iters=iters)
