"""Deterministic synthetic weights / inputs / labels for the LAS hot path.

Counter-based (NumPy Philox keyed by ``(seed, crc32(name))``) so the GPU box regenerates
bit-identical tensors from a few integers — nothing large has to travel.  Shapes and key
names follow the reference's ``state_dict`` (reference
``model/las_model.py:72-79,116-127,164-175,266-269``).
"""
from __future__ import annotations

import zlib

import numpy as np


def _gen(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def las_param_shapes(input_feature_dim, listener_hidden, listener_layers, speller_hidden, speller_layers,
                     vocab_size, mlp_dim=64, multi_head=1, use_mlp=True):
    """Ordered ``{state_dict key: shape}`` of LAS(Listener, Speller) in the reference's order."""
    H, Hs, V = listener_hidden, speller_hidden, vocab_size
    shapes = {}
    for l in range(listener_layers):
        d_in = 2 * input_feature_dim if l == 0 else 4 * H          # las_model.py:73,125 (x2 by the pyramid)
        p = f"listener.pLSTM_layer{l}.BLSTM."
        for suffix in ("", "_reverse"):
            shapes[p + "weight_ih_l0" + suffix] = (4 * H, d_in)
            shapes[p + "weight_hh_l0" + suffix] = (4 * H, H)
            shapes[p + "bias_ih_l0" + suffix] = (4 * H,)
            shapes[p + "bias_hh_l0" + suffix] = (4 * H,)
    for l in range(speller_layers):
        d_in = V + Hs if l == 0 else Hs                            # las_model.py:164-166
        p = "speller.rnn_layer."
        shapes[p + f"weight_ih_l{l}"] = (4 * Hs, d_in)
        shapes[p + f"weight_hh_l{l}"] = (4 * Hs, Hs)
        shapes[p + f"bias_ih_l{l}"] = (4 * Hs,)
        shapes[p + f"bias_hh_l{l}"] = (4 * Hs,)
    if use_mlp:
        shapes["speller.attention.phi.weight"] = (mlp_dim * multi_head, 2 * H)
        shapes["speller.attention.phi.bias"] = (mlp_dim * multi_head,)
        shapes["speller.attention.psi.weight"] = (mlp_dim, 2 * H)
        shapes["speller.attention.psi.bias"] = (mlp_dim,)
        if multi_head > 1:
            shapes["speller.attention.dim_reduce.weight"] = (2 * H, 2 * H * multi_head)
            shapes["speller.attention.dim_reduce.bias"] = (2 * H,)
    shapes["speller.character_distribution.weight"] = (V, 2 * Hs)
    shapes["speller.character_distribution.bias"] = (V,)
    return shapes


def make_state_dict(shapes, seed=17, scale=None):
    """fp32 numpy state_dict.  ``scale=None`` → PyTorch-default bound per tensor
    (LSTM: 1/sqrt(hidden); Linear: 1/sqrt(fan_in)); a float → U(-scale, scale) everywhere
    (a saturating set)."""
    sd = {}
    for name, shape in shapes.items():
        if scale is None:
            if "BLSTM" in name or "rnn_layer" in name:
                hidden = shape[0] // 4
                bound = 1.0 / np.sqrt(hidden)
            elif name.endswith("weight"):
                bound = 1.0 / np.sqrt(shape[1])
            else:  # Linear bias: bound by the fan_in of its weight
                bound = 1.0 / np.sqrt(shapes[name[:-4] + "weight"][1])
        else:
            bound = float(scale)
        u = _gen(seed, name).random(size=shape, dtype=np.float64)
        sd[name] = ((2.0 * u - 1.0) * bound).astype(np.float32)
    return sd


def make_inputs(B, T, F=80, seed=17, rank=0):
    """x ~ N(0,1) fp32 (B,T,F) — the reference seeds everything with 17 (train.py:41)."""
    return _gen(seed + 1000 * rank, f"x/{B}/{T}/{F}").standard_normal(size=(B, T, F), dtype=np.float64).astype(np.float32)


def make_labels(B, U, V=30, seed=17, rank=0, ragged=False):
    """Label indices (B,U) int64: symbols uniform in [2,V), last real symbol 1 (<eos>),
    0-padding after it when ``ragged`` (utils/data.py:133-143 pads one-hot labels with all-zero rows)."""
    g = _gen(seed + 1000 * rank, f"y/{B}/{U}/{V}")
    idx = g.integers(2, V, size=(B, U)).astype(np.int64)
    lens = np.full(B, U, dtype=np.int64)
    if ragged:
        lens = g.integers(max(2, U // 2), U + 1, size=B).astype(np.int64)
    for b in range(B):
        idx[b, lens[b] - 1] = 1
        idx[b, lens[b]:] = 0
    return idx, lens


def onehot_labels(idx, lens, V=30):
    """(B,U,V) int64 one-hot as utils/data.py:141-143 delivers them; padded rows are all-zero."""
    B, U = idx.shape
    oh = np.zeros((B, U, V), dtype=np.int64)
    for b in range(B):
        oh[b, np.arange(lens[b]), idx[b, :lens[b]]] = 1
    return oh


CONFIGS = {
    # name: (F, H, L, Hs, Ls, V, M)
    "tiny": dict(F=8, H=16, L=2, Hs=32, Ls=2, V=30, M=8),
    "S": dict(F=80, H=128, L=2, Hs=256, Ls=2, V=30, M=64),      # README.md:11 / BASELINE configs[0-1]
    "P": dict(F=80, H=256, L=3, Hs=512, Ls=2, V=30, M=64),      # paper-size, BASELINE configs[2-4]
    "Y": dict(F=40, H=512, L=3, Hs=1024, Ls=2, V=30, M=64),     # the reference's config/librispeech-config.yaml as shipped
}


def config_shapes(name, multi_head=1, use_mlp=True):
    c = CONFIGS[name]
    return las_param_shapes(c["F"], c["H"], c["L"], c["Hs"], c["Ls"], c["V"], c["M"], multi_head, use_mlp)
