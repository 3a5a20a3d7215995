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


def collate_fn_device(batch, device="cuda", listener_layers=5, vocab_size=30):
    """``batch``: list of ``(utt_id, feat (T_i,F) float array, feat_len, target, target_len)`` as the reference's
    ``AudioDataset.__getitem__`` yields them; ``target`` may be a list of one-hot rows (reference format) or a 1-D list
    / array of character indices.  Returns ``(utt_ids, {"inputs","inputs_length"}, {"targets","targets_length"})``."""
    utt_ids = [d[0] for d in batch]
    feat_len = [int(d[2]) for d in batch]
    idx = []
    for d in batch:
        t = np.asarray(d[3])
        idx.append(t.argmax(-1).astype(np.int64) if t.ndim == 2 else t.astype(np.int64).reshape(-1))
    tgt_len = [len(t) for t in idx]
    T = max(feat_len)
    mult = 2 ** listener_layers
    if T % mult != 0:
        T += mult - (T % mult)
    U = max(tgt_len)
    B, F = len(batch), np.asarray(batch[0][1]).shape[1]
    # one packed host buffer, one copy, scatter into the zero-padded (B,T,F) tensor on the device
    packed = np.concatenate([np.asarray(d[1], dtype=np.float32)[:n] for d, n in zip(batch, feat_len)], axis=0)
    packed_t = torch.from_numpy(packed)
    if torch.device(device).type == "cuda":
        packed_t = packed_t.pin_memory()
    packed_d = packed_t.to(device, non_blocking=True)
    lens = torch.tensor(feat_len, dtype=torch.int64)
    rows = torch.repeat_interleave(torch.arange(B), lens)
    starts = torch.cumsum(lens, 0) - lens
    cols = torch.arange(int(lens.sum())) - torch.repeat_interleave(starts, lens)
    inputs = torch.zeros(B, T, F, dtype=torch.float32, device=device)
    inputs[rows.to(device), cols.to(device)] = packed_d
    # labels: indices -> one-hot on the device; padding rows are onehot(PAD) exactly as data.py:133
    lab = torch.full((B, U), PAD, dtype=torch.int64)
    for b, t in enumerate(idx):
        lab[b, :len(t)] = torch.from_numpy(t)
    targets = torch.zeros(B, U, vocab_size, dtype=torch.int64, device=device)
    targets.scatter_(2, lab.to(device).unsqueeze(-1), 1)
    feature = {"inputs": inputs, "inputs_length": torch.tensor(feat_len, dtype=torch.int32)}
    label = {"targets": targets, "targets_length": torch.tensor(tgt_len, dtype=torch.int32)}
    return utt_ids, feature, label
