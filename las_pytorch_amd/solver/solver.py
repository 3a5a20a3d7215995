"""Caller-side counterpart of the reference's ``solver/solver.py`` (the code that CALLS the hot path):
``batch_iterator`` (fwd -> loss -> bwd -> clip 1.0 -> optimizer step), ``label_smoothing_loss`` and
``LetterErrorRate``.  Same names, argument meaning and return values as the reference
(solver/solver.py:11-24,33-45,48-101), so a ``train.py``-style driver can import it unchanged.
The reference's ``editdistance`` C extension is replaced by a small Levenshtein (not installed here)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


def _edit_distance(a, b):
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def LetterErrorRate(pred_y, true_y):
    """solver/solver.py:11-24."""
    ed_accumalate = []
    for p, t in zip(pred_y, true_y):
        compressed_t = [w for w in t if (w != 1 and w != 0)]
        compressed_p = []
        for p_w in p:
            if p_w == 0:
                continue
            if p_w == 1:
                break
            compressed_p.append(p_w)
        ed_accumalate.append(_edit_distance(compressed_p, compressed_t) / len(compressed_t))
    return ed_accumalate


def label_smoothing_loss(pred_y, true_y, label_smoothing=0.1):
    """solver/solver.py:33-45 (pred_y log-probs (B,U,V); true_y one-hot floats padded with all-zero rows)."""
    assert pred_y.size() == true_y.size()
    seq_len = torch.sum(torch.sum(true_y, dim=-1), dim=-1, keepdim=True)
    class_dim = true_y.size()[-1]
    smooth_y = ((1.0 - label_smoothing) * true_y + (label_smoothing / class_dim)) * torch.sum(true_y, dim=-1, keepdim=True)
    loss = -torch.mean(torch.sum((torch.sum(smooth_y * pred_y, dim=-1) / seq_len), dim=-1))
    return loss


def batch_iterator(batch_data, batch_label, las_model, optimizer, tf_rate, is_training, max_label_len, label_smoothing,
                   use_gpu=True, vocab_dict=None, grad_hook=None):
    """solver/solver.py:48-101.  ``grad_hook`` (optional, not in the reference) runs between backward and the
    clip — the data-parallel driver all-reduces the flat gradient there."""
    max_label_len = min([batch_label.size()[1], max_label_len])
    criterion = nn.NLLLoss(ignore_index=0)
    optimizer.zero_grad()
    raw_pred_seq, _ = las_model(batch_data=batch_data, batch_label=batch_label, teacher_force_rate=tf_rate,
                                is_training=is_training)
    pred_y = (torch.cat([torch.unsqueeze(each_y, 1) for each_y in raw_pred_seq], 1)[:, :max_label_len, :]).contiguous()
    if label_smoothing == 0.0 or not (is_training):
        pred_y = pred_y.permute(0, 2, 1)
        true_y = torch.max(batch_label, dim=2)[1][:, :max_label_len].contiguous()
        loss = criterion(pred_y, true_y)
        batch_ler = LetterErrorRate(torch.max(pred_y.permute(0, 2, 1), dim=2)[1].cpu().numpy(), true_y.cpu().data.numpy())
    else:
        true_y = batch_label[:, :max_label_len, :].contiguous().type(torch.float32)
        loss = label_smoothing_loss(pred_y, true_y, label_smoothing=label_smoothing)
        batch_ler = LetterErrorRate(torch.max(pred_y, dim=2)[1].cpu().numpy(), torch.max(true_y, dim=2)[1].cpu().data.numpy())
    if is_training:
        loss.backward()
        if grad_hook is not None:
            grad_hook(las_model)
        torch.nn.utils.clip_grad_norm_(las_model.parameters(), 1)
        optimizer.step()
    batch_loss = loss.cpu().data.numpy()
    return batch_loss, batch_ler
