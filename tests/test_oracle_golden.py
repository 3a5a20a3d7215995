"""The CPU oracle (oracle/las_oracle.py) against the golden vectors captured from the reference
(tests/golden/make_golden.py).  Tolerance: <=1e-6 abs on log-probs / activations in fp32
(SURVEY.md section 8c); argmax sequences identical."""
import numpy as np
import pytest
import torch

from golden_util import (ALL_CASES, BIG_CASES, FREE_TRAIN_CASES, MODE0_TRAIN_CASES, free_train_decode_mode, load_case, load_free_train_case, oracle_cfg,
                         tf_argmax_mask)
from oracle import las_oracle as O

ATOL = 2e-6


def _atol(name):
    """The scaled-weight headline cases (U(-0.2, 0.2) weights on 160..1024-wide layers: saturated gates, peaked attention, ~6
    arg-max changes per greedy sequence) amplify the rounding-order difference between the oracle's explicit steps and the
    reference's oneDNN LSTM through 700 recurrent steps: observed 3.5e-5 abs on the log-probs (top-1 / top-2 margins are
    >= 5e-4, arg-max sequences identical).  At scale >= 0.25 the same comparison turns chaotic (2e-3), which is why the
    fixtures stop at 0.2."""
    return 1e-4 if name.endswith("_s") else ATOL


@pytest.mark.parametrize("name", ALL_CASES + BIG_CASES)
def test_forward_matches_reference(name):
    g, info, sd_np, x, idx, lens, onehot = load_case(name)
    sd = O.to_torch_sd(sd_np)
    xt = torch.from_numpy(x)
    lab = torch.from_numpy(onehot)
    c = info["cfg"]
    with torch.no_grad():
        feats = O.listener_forward(xt, sd, c["L"], return_all=True)
        for l, f in enumerate(feats):
            np.testing.assert_allclose(f.numpy()[:, ::info["sub_t"], ::info["sub_d"]], g[f"listener_l{l}"], atol=_atol(name), rtol=0)
            assert abs(f.double().sum().item() - g[f"listener_l{l}_sum"][0]) < 1e-3 * max(1.0, g[f"listener_l{l}_sum"][1] * 1e-3)
        preds, atts = O.las_forward(xt, lab, sd, oracle_cfg(info), teacher_force=True)
        logp = torch.stack(preds).numpy()
        np.testing.assert_allclose(logp, g["tf_logp"], atol=_atol(name), rtol=0)
        mask = tf_argmax_mask(g["tf_logp"])
        assert (logp.argmax(-1) == g["tf_argmax"])[mask].all() and mask.mean() > 0.99
        att = np.stack([torch.stack(h).numpy() for h in zip(*atts)], 0)
        if info["full"]:
            np.testing.assert_allclose(att, g["tf_att"], atol=_atol(name), rtol=0)
        else:
            np.testing.assert_allclose(att[:, :, :, ::info["sub_t"]], g["tf_att"], atol=_atol(name), rtol=0)
        preds, _ = O.las_forward(xt, lab, sd, oracle_cfg(info), teacher_force=False, is_training=False)
        logp = torch.stack(preds).numpy()
        assert (logp.argmax(-1) == g["greedy_argmax"]).all()
        logp = logp[::info["sub_u"]]
        assert logp.shape == g["greedy_logp"].shape
        np.testing.assert_allclose(logp, g["greedy_logp"], atol=_atol(name), rtol=0)
        if "mode0_logp" in g:
            preds, _ = O.las_forward(xt, lab, sd, oracle_cfg(info, decode_mode=0), teacher_force=False, is_training=False)
            np.testing.assert_allclose(torch.stack(preds).numpy(), g["mode0_logp"], atol=5e-6, rtol=0)


@pytest.mark.parametrize("name", [n for n in ALL_CASES + BIG_CASES if n not in ("S_T800", "P_T800")])
def test_loss_and_grads_match_reference(name):
    g, info, sd_np, x, idx, lens, onehot = load_case(name)
    xt = torch.from_numpy(x)
    lab = torch.from_numpy(onehot)
    for kind, ls in (("ls", 0.1), ("nll", 0.0)):
        if f"loss_{kind}" not in g:
            continue
        sd = O.to_torch_sd(sd_np, requires_grad=True)
        preds, _ = O.las_forward(xt, lab, sd, oracle_cfg(info), teacher_force=True)
        loss, _ = O.solver_step_loss(preds, lab, info["U"], ls)
        loss.backward()
        assert abs(loss.item() - g[f"loss_{kind}"][0]) < 2e-6 * max(1, abs(g[f"loss_{kind}"][0]))
        norms = np.array([sd[k].grad.double().norm().item() for k in sd_np])
        np.testing.assert_allclose(norms, g[f"gradnorm_{kind}"], rtol=2e-4, atol=1e-8)
        if kind == "ls":
            for k in sd_np:
                got = sd[k].grad.numpy()
                want = g["grad/" + k]
                if not info["full"]:
                    got = got.reshape(-1)[:: max(1, got.size // 64)][:64]
                np.testing.assert_allclose(got, want, rtol=1e-3, atol=2e-7)


@pytest.mark.parametrize("name", FREE_TRAIN_CASES[:2] + FREE_TRAIN_CASES[3:] + MODE0_TRAIN_CASES)
def test_free_running_training_step_matches_reference(name):
    """The oracle's free-running training step (decode_mode-1 feedback for max_label_len steps with autograd on, then the label-smoothing
    loss and backward; reference las_model.py:189,205-227 + solver.py:33-45,95) against the unmodified reference's: log-probs, arg-max
    sequence, loss, all gradient norms and slices.  (The paper-size fixture is checked on the GPU only: 40 s of CPU per run.)"""
    g, c, cfg_name, (B, T, U), sd_np, x, onehot, heads = load_free_train_case(name)
    sd = O.to_torch_sd(sd_np, requires_grad=True)
    lab = torch.from_numpy(onehot)
    cfg = dict(listener_layers=c["L"], speller_layers=c["Ls"], max_label_len=U, decode_mode=free_train_decode_mode(g), multi_head=heads)
    preds, _ = O.las_forward(torch.from_numpy(x), lab, sd, cfg, teacher_force=False)
    logp = torch.stack(preds).detach().numpy()
    assert (logp.argmax(-1) == g["free_argmax"]).all()
    np.testing.assert_allclose(logp, g["free_logp"], atol=5e-6 if free_train_decode_mode(g) == 1 else 2e-5, rtol=0)      # (mode 0 re-amplifies rounding every step)
    loss, _ = O.solver_step_loss(preds, lab, U, 0.1)
    loss.backward()
    assert abs(loss.item() - g["loss_ls"][0]) < 2e-6 * max(1, abs(g["loss_ls"][0]))
    assert list(sd_np.keys()) == [str(k) for k in g["grad_keys"]]
    norms = np.array([sd[k].grad.double().norm().item() for k in sd_np])
    np.testing.assert_allclose(norms, g["gradnorm_ls"], rtol=2e-4, atol=1e-8)
    for k in sd_np:
        got = sd[k].grad.numpy()
        np.testing.assert_allclose(got.reshape(-1)[:: max(1, got.size // 64)][:64], g["grad/" + k], rtol=1e-3, atol=2e-7)


@pytest.mark.parametrize("name", ["tiny_mode2", "S_mode2"])
def test_decode_mode2_sampling_semantics(name):
    """decode_mode 2 (reference las_model.py:229-234): the oracle's sampler (p = logp / sum logp, argmax p/q over the
    stored Exp(1) draws) replays the reference's sampled symbols, so its log-probs agree at EVERY step."""
    from golden_util import load_mode2_case
    g, c, sd_np, x = load_mode2_case(name)
    sd = O.to_torch_sd(sd_np)
    U = g["mode2_logp"].shape[0]
    with torch.no_grad():
        preds, _ = O.las_forward(torch.from_numpy(x), None, sd,
                                 dict(listener_layers=c["L"], speller_layers=c["Ls"], max_label_len=U, decode_mode=2),
                                 teacher_force=False, is_training=False,
                                 sample_fn=O.categorical_sampler(torch.from_numpy(g["mode2_noise"])))
    np.testing.assert_allclose(torch.stack(preds).numpy(), g["mode2_logp"], atol=ATOL, rtol=0)
    # and it is NOT the greedy path: at least one sampled symbol differs from the arg-max
    logp = torch.from_numpy(g["mode2_logp"])
    p = logp / logp.sum(-1, keepdim=True)
    assert ((p / torch.from_numpy(g["mode2_noise"])).argmax(-1) != logp.argmax(-1)).any()


def test_ler_handmade():
    # solver/solver.py:11-24 semantics: zeros skipped, prediction stops at first 1, truth strips 0 and 1
    pred = [[5, 0, 6, 7, 1, 9, 9], [2, 2, 2, 2, 2, 2, 2], [1, 5, 6, 7, 0, 0, 0], [3, 4, 5, 6, 1, 0, 0]]
    true = [[5, 6, 7, 1, 0, 0, 0], [2, 3, 1, 0, 0, 0, 0], [5, 6, 7, 1, 0, 0, 0], [3, 9, 5, 1, 0, 0, 0]]
    got = O.letter_error_rate(np.array(pred), np.array(true))
    assert got == [0.0, 6 / 2, 3 / 3, 2 / 3]
    assert O.edit_distance("kitten", "sitting") == 3
    with pytest.raises(ZeroDivisionError):
        O.letter_error_rate(np.array([[1, 0]]), np.array([[1, 0]]))


@pytest.mark.parametrize("name", ["S_short", "P_short_sat", "tiny_default"])
def test_cpu_baseline_matches_reference(name):
    """oracle/cpu_baseline.py (the nn.LSTM-module port that bench.py times on the host) == the reference."""
    from oracle import cpu_baseline as CB
    g, info, sd_np, x, idx, lens, onehot = load_case(name)
    m = CB.build(info["cfg"], sd_np)
    with torch.no_grad():
        preds = m(torch.from_numpy(x), torch.from_numpy(onehot), info["U"])
    np.testing.assert_allclose(torch.stack(preds).numpy(), g["tf_logp"], atol=ATOL, rtol=0)
    preds = m(torch.from_numpy(x), torch.from_numpy(onehot), info["U"])
    loss = CB.label_smoothing_loss(torch.stack(preds, 1), torch.from_numpy(onehot).float(), 0.1)
    assert abs(loss.item() - g["loss_ls"][0]) < 2e-6 * max(1, abs(g["loss_ls"][0]))
