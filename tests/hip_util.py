"""Helpers shared by the GPU parity tests."""
import json
import os

import numpy as np
import torch

from las_pytorch_amd import LAS, Listener, Speller


def build_las(c, sd_np, *, max_label_len, decode_mode=1, multi_head=1, use_mlp=True, activate="relu", device="cuda"):
    listener = Listener(input_feature_dim=c["F"], hidden_size=c["H"], num_layers=c["L"], rnn_unit="LSTM", use_gpu=True)
    speller = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"],
                      max_label_len=max_label_len, use_mlp_in_attention=use_mlp, mlp_dim_in_attention=c["M"],
                      mlp_activate_in_attention=activate, listener_hidden_size=c["H"], multi_head=multi_head,
                      decode_mode=decode_mode, use_gpu=True)
    las = LAS(listener, speller)
    las.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd_np.items()}, strict=True)
    return las.to(device)


_OBSERVED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_observed.jsonl")


def record(name, **kv):
    """Append an observed-error record (copied to profiles/ after a GPU run; ignored if the directory is read-only)."""
    try:
        os.makedirs(os.path.dirname(_OBSERVED), exist_ok=True)
        with open(_OBSERVED, "a") as f:
            f.write(json.dumps(dict(name=name, **kv)) + "\n")
    except OSError:
        pass


def grad_close(got, want, name, rtol=1e-3, floor=1e-5, global_scale=0.0, global_floor=5e-7):
    """Gradient comparison at the north-star tolerance, scaled to the tensor: |a-b| <= rtol*|b| + floor*max|b|
    (gradient tensors span 1e-2 .. 1e-8 in magnitude, so the absolute term of SURVEY.md section 8c's
    ``1e-3*|b| + 1e-5`` is taken relative to the tensor's largest element) + 5e-7 * ``global_scale`` (the largest
    per-parameter gradient norm of the model: a tensor whose true gradient is (nearly) zero — psi.bias under a
    softmax, which is shift invariant — is a cancellation residue and holds only rounding noise of the other tensors'
    magnitude, a few fp32 ulps (6e-8) of it.  Observed on psi.bias of the (32,800) S case: 0 with the fp32-MFMA GEMM,
    whose fmaf chains happen to round like the CPU's, 2.8e-8 absolute = 0.25e-6 of the global scale with the
    split-operand GEMM, whose sums are as accurate but round differently).  Records the observed worst ratio."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    scale = float(np.abs(want).max()) + 1e-30
    err = np.abs(got - want)
    atol = floor * scale + global_floor * global_scale
    ratio = float((err / (rtol * np.abs(want) + atol)).max())
    record(name, max_abs_err=float(err.max()), max_abs_want=scale, worst_ratio=ratio, rel_to_max=float(err.max() / scale))
    return assert_close(got, want, name, rtol=rtol, atol=atol)


def assert_close(got, want, name, rtol=1e-3, atol=1e-5):
    """North-star tolerance: |a-b| <= 1e-3*|b| + 1e-5."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    err = np.abs(got - want)
    tol = rtol * np.abs(want) + atol
    bad = err > tol
    if bad.any() or not np.isfinite(got).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{name}: {bad.sum()}/{bad.size} elements out of tolerance; worst at {i}: got {got[i]:.8g} "
                             f"want {want[i]:.8g} (abs err {err[i]:.3g}); max abs err {err.max():.3g}; finite={np.isfinite(got).all()}")
    return float(err.max())
