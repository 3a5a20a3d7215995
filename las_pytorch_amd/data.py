"""Input side of the hot path: the reference's collate contract (``utils/data.py:116-149``) with index labels and
device-side padding.

The reference pads on the host with NumPy and ships int64 one-hot labels ``(B,U,30)`` (8*V bytes per character); here
the ragged features are packed once, copied once, and padding / one-hot expansion happen on the device.  The OUTPUT
contract is unchanged — ``inputs (B,T,F)`` fp32 with T padded up to a multiple of ``2**listener_layers`` (the reference
uses the module-global ``listener_layers = 5`` -> 32, ``data.py:20,124-125``), ``targets (B,U,V)`` int64 one-hot whose
padding rows are ``onehot(PAD=0)`` (``data.py:133``), plus the two length vectors the model never reads
(``data.py:141-147``) — so ``solver.batch_iterator`` consumes it unchanged.
"""
from __future__ import annotations

import numpy as np
import torch

PAD = 0


def _pack(batch):
    """Host side of the collate: concatenate the ragged utterances once (one pinned staging buffer per kind)."""
    feat_len = [int(d[2]) for d in batch]
    idx = []
    for d in batch:
        t = np.asarray(d[3])
        idx.append(t.argmax(-1).astype(np.int64) if t.ndim == 2 else t.astype(np.int64).reshape(-1))
    tgt_len = [len(t) for t in idx]
    packed = np.concatenate([np.asarray(d[1], dtype=np.float32)[:n] for d, n in zip(batch, feat_len)], axis=0)
    labels = np.concatenate(idx) if sum(tgt_len) else np.zeros(0, np.int64)
    foff = np.concatenate(([0], np.cumsum(feat_len))).astype(np.int64)
    loff = np.concatenate(([0], np.cumsum(tgt_len))).astype(np.int64)
    return packed, foff, labels, loff, feat_len, tgt_len


def collate_fn_device(batch, device="cuda", listener_layers=5, vocab_size=30):
    """``batch``: list of ``(utt_id, feat (T_i,F) float array, feat_len, target, target_len)`` as the reference's
    ``AudioDataset.__getitem__`` yields them; ``target`` may be a list of one-hot rows (reference format) or a 1-D list
    / array of character indices.  Returns ``(utt_ids, {"inputs","inputs_length"}, {"targets","targets_length"})``.

    On a GPU the ragged frames and the character INDICES cross PCIe once (4 bytes per feature, 8 per character instead of
    the reference's 8*V per character) and ``las_collate_pad`` (csrc/misc.hip) writes the padded ``(B,T,F)`` features and
    the int64 one-hot ``(B,U,V)`` targets.  ``device="cpu"`` runs the same contract with torch index ops (host tests)."""
    utt_ids = [d[0] for d in batch]
    packed, foff, labels, loff, feat_len, tgt_len = _pack(batch)
    T = max(feat_len)
    mult = 2 ** listener_layers
    if T % mult != 0:
        T += mult - (T % mult)
    U = max(tgt_len)
    B, F = len(batch), packed.shape[1]
    feature_len = torch.tensor(feat_len, dtype=torch.int32)
    target_len = torch.tensor(tgt_len, dtype=torch.int32)
    if torch.device(device).type == "cuda":
        from . import _cabi
        dev = torch.device(device)
        stage = [torch.from_numpy(a).pin_memory().to(dev, non_blocking=True) for a in (packed, foff, labels, loff)]
        inputs = torch.empty(B, T, F, dtype=torch.float32, device=dev)
        targets = torch.empty(B, U, vocab_size, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _cabi.check(_cabi.lib().las_collate_pad(*[_cabi.ptr(t) for t in stage], B, T, F, U, vocab_size, _cabi.ptr(inputs),
                                                    _cabi.ptr(targets), _cabi.stream_ptr()))
        for t in stage:
            t.record_stream(torch.cuda.current_stream(dev))
    else:
        lens = torch.tensor(feat_len, dtype=torch.int64)
        rows = torch.repeat_interleave(torch.arange(B), lens)
        cols = torch.arange(int(lens.sum())) - torch.repeat_interleave(torch.from_numpy(foff[:-1]), lens)
        inputs = torch.zeros(B, T, F, dtype=torch.float32)
        inputs[rows, cols] = torch.from_numpy(packed)
        lab = torch.full((B, U), PAD, dtype=torch.int64)
        for b in range(B):
            lab[b, :tgt_len[b]] = torch.from_numpy(labels[loff[b]:loff[b + 1]])
        targets = torch.zeros(B, U, vocab_size, dtype=torch.int64)
        targets.scatter_(2, lab.unsqueeze(-1), 1)
    feature = {"inputs": inputs, "inputs_length": feature_len}
    label = {"targets": targets, "targets_length": target_len}
    return utt_ids, feature, label
