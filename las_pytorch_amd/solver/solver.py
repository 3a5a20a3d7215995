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


class _LSLossFn(torch.autograd.Function):
    """label_smoothing_loss and its gradient in one HIP kernel (las_ls_loss); pred_y (B,U,V) log-probs on the GPU,
    labels int64 one-hot (B,U_lab,V)."""

    @staticmethod
    def forward(ctx, pred_y, labels, smoothing):
        from .. import _cabi
        pred_y = pred_y.contiguous()
        B, U, V = pred_y.shape
        labels = labels.contiguous()
        L = _cabi.lib()
        loss = torch.empty(1, device=pred_y.device)
        scratch = torch.empty(B, device=pred_y.device)
        dlogp = torch.empty_like(pred_y) if ctx.needs_input_grad[0] else None
        _cabi.check(L.las_ls_loss(_cabi.ptr(pred_y), V, U * V, _cabi.ptr(labels), U, labels.shape[1], B, V, float(smoothing),
                                  _cabi.ptr(loss), _cabi.ptr(dlogp), V, U * V, _cabi.ptr(scratch), _cabi.stream_ptr()))
        ctx.dlogp = dlogp
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return ctx.dlogp * g, None, None


def label_smoothing_loss_device(pred_y, labels_onehot_int64, label_smoothing=0.1):
    """Fused on-device form of ``label_smoothing_loss`` (same value and gradient)."""
    return _LSLossFn.apply(pred_y, labels_onehot_int64, label_smoothing)


def LetterErrorRate_device(pred_y, labels_onehot_int64):
    """LetterErrorRate (solver/solver.py:11-24) computed on the GPU from log-probs (B,U,V) and int64 one-hot labels;
    returns a (B,) float tensor on the device (one host copy when the caller wants the list)."""
    from .. import _cabi
    pred_y = pred_y.detach().contiguous()
    B, U, V = pred_y.shape
    labels = labels_onehot_int64.contiguous()
    out = torch.empty(B, device=pred_y.device)
    work = torch.empty(4 * B * (U + 1), dtype=torch.int32, device=pred_y.device)
    _cabi.check(_cabi.lib().las_letter_error_rate(_cabi.ptr(pred_y), V, U * V, _cabi.ptr(labels), U, labels.shape[1], B, V,
                                                  _cabi.ptr(out), _cabi.ptr(work), _cabi.stream_ptr()))
    return out


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
    elif pred_y.is_cuda and batch_label.dtype == torch.int64:
        # device path: fused loss(+gradient) kernel and on-device LER: one host copy per batch instead of three
        loss = label_smoothing_loss_device(pred_y, batch_label, label_smoothing)
        batch_ler = LetterErrorRate_device(pred_y, batch_label).cpu().tolist()
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
