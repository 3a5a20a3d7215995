"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors captured from the
reference and against the CPU oracle on fresh seeded inputs.  Tolerance |a-b| <= 1e-3|b| + 1e-5, argmax
character sequences identical (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from golden_util import (ALL_CASES, BIG_CASES, FREE_TRAIN_CASES, MODE0_TRAIN_CASES, free_train_decode_mode, load_case, load_free_train_case, oracle_cfg,
                         tf_argmax_mask)
from hip_util import assert_close, build_las, grad_close, record
from las_pytorch_amd import _cabi

pytestmark = pytest.mark.gpu

HIP_CASES = list(ALL_CASES) + list(BIG_CASES)


def _check_err():
    import las_pytorch_amd
    torch.cuda.synchronize()
    las_pytorch_amd.check_device_errors()


def _assert_path(name, info, **slots):
    """The fixture ran the kernel family it pins (golden_util.expected_paths): slots maps a phase key of info["paths"] to a
    las_debug_last_path slot.  LAS_PATH_PROBE=1 records the observed names instead of asserting (to rebuild the table)."""
    import os
    from las_pytorch_amd import _cabi
    torch.cuda.synchronize()
    for key, slot in slots.items():
        got = _cabi.last_path(slot)
        record(f"{name}/path/{key}", path=got, expected=info["paths"][key])
        if os.environ.get("LAS_PATH_PROBE") != "1":
            assert got == info["paths"][key], f"{name}: {key} ran on '{got}', the fixture pins '{info['paths'][key]}'"


@pytest.mark.parametrize("name", HIP_CASES)
def test_forward_golden(name):
    g, info, sd_np, x, idx, lens, onehot = load_case(name)
    c = info["cfg"]
    las = build_las(c, sd_np, max_label_len=info["free_len"], use_mlp=info["use_mlp"], activate=info["activate"],
                    multi_head=info["multi_head"])
    xt = torch.from_numpy(x).cuda()
    lab = torch.from_numpy(onehot).cuda()
    with torch.no_grad():
        h = xt
        for l in range(c["L"]):
            h, _ = getattr(las.listener, f"pLSTM_layer{l}")(h)
            assert_close(h.cpu().numpy()[:, ::info["sub_t"], ::info["sub_d"]], g[f"listener_l{l}"], f"{name}/listener_l{l}")
        preds, atts = las(batch_data=xt, batch_label=lab, teacher_force_rate=1.0, is_training=True)
        logp = torch.stack(preds).cpu().numpy()
        _assert_path(name, info, rec_fwd=_cabi.PATH_REC_FWD, tf=_cabi.PATH_DECODE_FWD)
        assert_close(logp, g["tf_logp"], f"{name}/tf_logp")
        mask = tf_argmax_mask(g["tf_logp"])
        assert (logp.argmax(-1) == g["tf_argmax"])[mask].all() and mask.mean() > 0.99, f"{name}: teacher-forced argmax differs"
        att = np.stack([torch.stack([a[hd] for a in atts]).cpu().numpy() for hd in range(info["multi_head"])], 0)
        want = g["tf_att"]
        assert_close(att if info["full"] else att[:, :, :, ::info["sub_t"]], want, f"{name}/tf_att", atol=1e-6)
        preds, _ = las(batch_data=xt, batch_label=lab, teacher_force_rate=0.0, is_training=False)
        logp = torch.stack(preds).cpu().numpy()
        _assert_path(name, info, greedy=_cabi.PATH_DECODE_FWD)
        assert (logp.argmax(-1) == g["greedy_argmax"]).all(), f"{name}: greedy argmax sequence differs"
        assert_close(logp[::info["sub_u"]], g["greedy_logp"], f"{name}/greedy_logp")
        if "mode0_logp" in g:
            # decode_mode 0 feeds the log-probs back as the next input (las_model.py:220-221): rounding differences are
            # re-amplified every step, so the deviation grows with the step index; observed worst is recorded
            las.speller.decode_mode = 0
            preds, _ = las(batch_data=xt, batch_label=lab, teacher_force_rate=0.0, is_training=False)
            got = torch.stack(preds).cpu().numpy()
            record(f"{name}/mode0_logp", max_abs_err=float(np.abs(got - g["mode0_logp"]).max()))
            assert_close(got, g["mode0_logp"], f"{name}/mode0_logp")
    _check_err()


# gradient tolerance: north-star 1e-3 relative + a floor of 1e-5 of the tensor's largest element (see hip_util.grad_close;
# observed worst ratios per case are written to gpurun_out/parity_observed.jsonl and kept under profiles/)
GRAD_RTOL = 1e-3
GRAD_FLOOR = 1e-5


def _loss_ls(preds, lab, U):
    from las_pytorch_amd.solver.solver import label_smoothing_loss
    pred_y = torch.cat([p.unsqueeze(1) for p in preds], 1)[:, :U, :].contiguous()
    return label_smoothing_loss(pred_y, lab[:, :U, :].float(), label_smoothing=0.1)


@pytest.fixture
def fp32_mfma_gemm():
    """The GEMMs on the fp32 matrix pipe (LAS_GEMM_ARITH=0) for one test; the default is the split-operand bf16-MFMA arithmetic."""
    from las_pytorch_amd import _cabi
    L = _cabi.lib()
    old = L.las_gemm_get_arith()
    L.las_gemm_set_arith(0)
    yield
    L.las_gemm_set_arith(old)


@pytest.mark.parametrize("name", ["P_B32_T800_U32", "S_B32_T800_U32", "tiny_mh4"])
def test_grads_golden_fp32_mfma_gemm(name, fp32_mfma_gemm):
    """The reference goldens at the benchmark's size with the GEMMs in their other arithmetic mode (both modes are shipped)."""
    test_grads_golden(name)


@pytest.mark.parametrize("name", [n for n in HIP_CASES if n not in ("S_T800", "P_T800")])
def test_grads_golden(name):
    g, info, sd_np, x, idx, lens, onehot = load_case(name)
    c = info["cfg"]
    las = build_las(c, sd_np, max_label_len=info["free_len"], use_mlp=info["use_mlp"], activate=info["activate"],
                    multi_head=info["multi_head"])
    xt = torch.from_numpy(x).cuda()
    lab = torch.from_numpy(onehot).cuda()
    preds, _ = las(batch_data=xt, batch_label=lab, teacher_force_rate=1.0, is_training=True)
    loss = _loss_ls(preds, lab, info["U"])
    loss.backward()
    _assert_path(name, info, rec_fwd=_cabi.PATH_REC_FWD, tf=_cabi.PATH_DECODE_FWD, rec_bwd=_cabi.PATH_REC_BWD, bwd=_cabi.PATH_DECODE_BWD)
    assert abs(loss.item() - g["loss_ls"][0]) <= 1e-4 * abs(g["loss_ls"][0]) + 1e-6
    names = [k for k, _ in las.named_parameters()]
    assert names == list(sd_np.keys())
    norms = np.array([p.grad.double().norm().item() for _, p in las.named_parameters()])
    total = np.sqrt((norms ** 2).sum())
    assert abs(total - g["gradtotal_ls"][0]) <= 2e-3 * g["gradtotal_ls"][0], (total, g["gradtotal_ls"][0])
    scale = float(g["gradnorm_ls"].max())
    np.testing.assert_allclose(norms, g["gradnorm_ls"], rtol=1e-3, atol=1e-6 * scale)
    for (k, p), want_norm in zip(las.named_parameters(), g["gradnorm_ls"]):
        got = p.grad.cpu().numpy()
        want = g["grad/" + k]
        if not info["full"]:
            got = got.reshape(-1)[:: max(1, got.size // 64)][:64]
        # the 64-element slice of a big tensor can miss its large entries: scale the floor by the tensor's RMS as well
        rms = want_norm / np.sqrt(max(1, p.numel()))
        grad_close(got, want, f"{name}/grad/{k}", rtol=GRAD_RTOL, floor=GRAD_FLOOR * max(1.0, rms / (np.abs(want).max() + 1e-30)),
                   global_scale=scale)
    _check_err()


@pytest.mark.parametrize("cfg_name,B,T,U,scale", [("S", 5, 96, 7, None), ("P", 6, 64, 6, 0.12), ("tiny", 3, 24, 4, 0.4),
                                                  ("Y", 3, 64, 5, 0.08),
                                                  ("S", 32, 800, 16, None), ("P", 32, 800, 16, None),
                                                  ("P", 32, 800, 128, None),
                                                  ("P", 72, 800, 8, None),
                                                  ("P", 8, 3000, 128, None), ("S", 8, 3000, 128, None)])
def test_forward_backward_vs_oracle(cfg_name, B, T, U, scale):
    """Fresh seeded inputs, sizes the goldens do not cover (odd batch, full LibriSpeech shape); the last two rows are BASELINE configs[4]
    at the benchmark's own decode length (B = 8, T = 3000, U = 128: the 16-workgroups-per-utterance decode kernels, whose per-step
    hand-off slabs are walked 128 times, forward and backward; paths asserted); ("P", 72, 800, 8) runs the matrix-pipe
    recurrences (B >= 64) with a partial last group of 8 utterances for 400 / 200 / 100 steps and the decode in slices of 32 + 32 + 8;
    the row before it is the benchmark's
    EXACT shape (P, B=32, T=800, U=128): log-probs, loss and every gradient of all 128 steps of the one-launch decode kernels
    against the CPU oracle (ragged label lengths here; tests/golden/P_B32_T800_U128.npz pins the same shape to the reference)."""
    from las_pytorch_amd import synth
    from oracle import las_oracle as O
    c = synth.CONFIGS[cfg_name]
    sd_np = synth.make_state_dict(synth.config_shapes(cfg_name), seed=5, scale=scale)
    x = synth.make_inputs(B, T, c["F"], seed=5)
    idx, lens = synth.make_labels(B, U, c["V"], seed=5, ragged=True)
    onehot = synth.onehot_labels(idx, lens, c["V"])
    big = False      # gradients are compared at every size, the full (32,800,80) LibriSpeech shape included
    # oracle (CPU)
    sd = O.to_torch_sd(sd_np, requires_grad=not big)
    xt = torch.from_numpy(x)
    lab = torch.from_numpy(onehot)
    cfg = dict(listener_layers=c["L"], speller_layers=c["Ls"], max_label_len=U, decode_mode=1)
    with torch.set_grad_enabled(not big):
        feat_o = O.listener_forward(xt, sd, c["L"])
        preds_o, atts_o = O.speller_forward(feat_o, sd, num_layers=c["Ls"], max_label_len=U, decode_mode=1,
                                            ground_truth=lab, teacher_force=True)
    las = build_las(c, sd_np, max_label_len=U)
    xg = torch.from_numpy(x).cuda()
    labg = lab.cuda()
    feat = las.listener(xg)
    assert_close(feat.detach().cpu().numpy(), feat_o.detach().numpy(), f"{cfg_name}/listener")
    preds, atts = las(batch_data=xg, batch_label=labg, teacher_force_rate=1.0, is_training=True)
    logp = torch.stack(preds)
    logp_o = torch.stack(preds_o).detach()
    assert_close(logp.detach().cpu().numpy(), logp_o.numpy(), f"{cfg_name}/logp")
    assert (logp.argmax(-1).cpu() == logp_o.argmax(-1)).all()
    if not big:
        loss_o, _ = O.solver_step_loss(preds_o, lab, U, 0.1)
        loss_o.backward()
        loss = _loss_ls(preds, labg, U)
        loss.backward()
        assert abs(loss.item() - loss_o.item()) <= 1e-4 * abs(loss_o.item()) + 1e-6
        gscale = max(float(sd[k].grad.norm()) for k in sd)
        for k, p in las.named_parameters():
            want = sd[k].grad.numpy()
            grad_close(p.grad.cpu().numpy(), want, f"oracle_{cfg_name}_B{B}_T{T}/grad/{k}", rtol=GRAD_RTOL, floor=GRAD_FLOOR,
                       global_scale=gscale)
        if T == 3000:      # configs[4]: both decode kernels in their one-launch (frame-split) form, not the per-step chains
            torch.cuda.synchronize()
            assert _cabi.last_path(_cabi.PATH_DECODE_FWD) == "persist_pre", _cabi.last_path(_cabi.PATH_DECODE_FWD)
            assert _cabi.last_path(_cabi.PATH_DECODE_BWD) == "persist_pre", _cabi.last_path(_cabi.PATH_DECODE_BWD)
    _check_err()


def test_free_running_training_step_vs_oracle():
    """A free-running TRAINING step (the coin of a teacher-forcing schedule below 1 came up 'free': decode_mode 1 feedback for max_label_len
    steps, then loss and backward; reference las_model.py:189,205-206,223-227) at the benchmark's T against the CPU oracle: log-probabilities,
    arg-max sequence, loss and all 38 gradients.  Forward on the free-running PRE kernel, backward on the teacher-forced PRE kernel over the
    emitted symbols (paths asserted).  Weights of the "_s" fixtures (U(-0.2, 0.2), seed 43: greedy margin 7.5e-4, ~6 symbol changes)."""
    from las_pytorch_amd import synth
    from oracle import las_oracle as O
    c = synth.CONFIGS["P"]
    B, T, U = 32, 800, 12
    sd_np = synth.make_state_dict(synth.config_shapes("P"), seed=43, scale=0.2)
    x = synth.make_inputs(B, T, c["F"], seed=43)
    idx, lens = synth.make_labels(B, U, c["V"], seed=43, ragged=True)
    onehot = synth.onehot_labels(idx, lens, c["V"])
    sd = O.to_torch_sd(sd_np, requires_grad=True)
    lab = torch.from_numpy(onehot)
    cfg = dict(listener_layers=c["L"], speller_layers=c["Ls"], max_label_len=U, decode_mode=1)
    preds_o, _ = O.las_forward(torch.from_numpy(x), lab, sd, cfg, teacher_force=False)
    loss_o, _ = O.solver_step_loss(preds_o, lab, U, 0.1)
    loss_o.backward()
    las = build_las(c, sd_np, max_label_len=U)
    xg, labg = torch.from_numpy(x).cuda(), lab.cuda()
    preds, _ = las(batch_data=xg, batch_label=labg, teacher_force_rate=0.0, is_training=True)
    assert _cabi.last_path(_cabi.PATH_DECODE_FWD) == "persist_pre_greedy", _cabi.last_path(_cabi.PATH_DECODE_FWD)
    logp, logp_o = torch.stack(preds), torch.stack(preds_o).detach()
    assert (logp.argmax(-1).cpu() == logp_o.argmax(-1)).all(), "free-running arg-max sequence differs from the oracle's"
    assert_close(logp.detach().cpu().numpy(), logp_o.numpy(), "free_P/logp")
    loss = _loss_ls(preds, labg, U)
    loss.backward()
    torch.cuda.synchronize()
    assert _cabi.last_path(_cabi.PATH_DECODE_BWD) == "persist_pre", _cabi.last_path(_cabi.PATH_DECODE_BWD)
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * abs(loss_o.item()) + 1e-6
    gscale = max(float(sd[k].grad.norm()) for k in sd)
    for k, p in las.named_parameters():
        grad_close(p.grad.cpu().numpy(), sd[k].grad.numpy(), f"oracle_free_P_B{B}_T{T}/grad/{k}", rtol=GRAD_RTOL, floor=GRAD_FLOOR, global_scale=gscale)
    _check_err()


@pytest.mark.parametrize("name", FREE_TRAIN_CASES + MODE0_TRAIN_CASES)
def test_free_running_training_step_golden(name):
    """A free-running training step against the UNMODIFIED reference's (fixtures of make_golden.py::make_free_training_golden): log-probs,
    arg-max sequences, the label-smoothing loss, all per-parameter gradient norms and 64-element slices.  The paper-size fixture (B = 32,
    T = 800, weights of the "_s" cases: 19 distinct symbols, margin 1.3e-3) must run the free-running PRE forward and the PRE backward."""
    g, c, cfg_name, (B, T, U), sd_np, x, onehot, heads = load_free_train_case(name)
    dmode = free_train_decode_mode(g)      # 1, or 0 for the round-6 fixtures whose fed-back log-probabilities carry gradient (las_model.py:219-221)
    las = build_las(c, sd_np, max_label_len=U, multi_head=heads, decode_mode=dmode)
    xg, labg = torch.from_numpy(x).cuda(), torch.from_numpy(onehot).cuda()
    preds, _ = las(batch_data=xg, batch_label=labg, teacher_force_rate=0.0, is_training=True)
    assert len(preds) == U
    mh = "_mh" if heads > 1 else ""
    if cfg_name == "P" and dmode == 1:
        assert _cabi.last_path(_cabi.PATH_DECODE_FWD) == f"persist_pre{mh}_greedy", _cabi.last_path(_cabi.PATH_DECODE_FWD)
    if cfg_name in ("S", "P") and dmode == 0:      # mode 0: the classic (feat-resident) one-launch kernel forward, its backward with the feedback term
        assert _cabi.last_path(_cabi.PATH_DECODE_FWD) == "persist", _cabi.last_path(_cabi.PATH_DECODE_FWD)
    logp = torch.stack(preds).detach().cpu().numpy()
    assert (logp.argmax(-1) == g["free_argmax"]).all(), "free-running arg-max sequence differs from the reference's"
    assert_close(logp, g["free_logp"], f"{name}/free_logp")
    loss = _loss_ls(preds, labg, U)
    loss.backward()
    torch.cuda.synchronize()
    if cfg_name == "P" and dmode == 1:
        assert _cabi.last_path(_cabi.PATH_DECODE_BWD) == f"persist_pre{mh}", _cabi.last_path(_cabi.PATH_DECODE_BWD)
    record(f"{name}/path/bwd", path=_cabi.last_path(_cabi.PATH_DECODE_BWD), expected="")
    assert abs(loss.item() - g["loss_ls"][0]) <= 1e-4 * abs(g["loss_ls"][0]) + 1e-6
    assert [k for k, _ in las.named_parameters()] == [str(k) for k in g["grad_keys"]]
    norms = np.array([p.grad.double().norm().item() for _, p in las.named_parameters()])
    scale = float(g["gradnorm_ls"].max())
    np.testing.assert_allclose(norms, g["gradnorm_ls"], rtol=1e-3, atol=1e-6 * scale)
    for (k, p), want_norm in zip(las.named_parameters(), g["gradnorm_ls"]):
        got = p.grad.cpu().numpy()
        got = got.reshape(-1)[:: max(1, got.size // 64)][:64]
        want = g["grad/" + k]
        rms = want_norm / np.sqrt(max(1, p.numel()))
        grad_close(got, want, f"{name}/grad/{k}", rtol=GRAD_RTOL, floor=GRAD_FLOOR * max(1.0, rms / (np.abs(want).max() + 1e-30)), global_scale=scale)
    _check_err()


def test_listener_batch320_vs_oracle():
    """First-hand check of the B > 256 matrix-pipe recurrence (``rec_fwd_mfma2``: two batches of 16 sequences per group, wave-specialised
    pipeline) and of ``rec_bwd_mfma`` at that batch: the whole paper-size Listener at (B, T) = (320, 800), forward and backward (a fixed
    random cotangent), against the CPU oracle — every layer output gradient path, dX and all 24 parameter gradients.  The launch path is
    asserted, so a residency fall-back cannot leave this green on the generic kernels."""
    from las_pytorch_amd import _cabi, synth
    from oracle import las_oracle as O
    c = synth.CONFIGS["P"]
    B, T = 320, 800
    sd_np = {k: v for k, v in synth.make_state_dict(synth.config_shapes("P"), seed=23, scale=0.1).items() if k.startswith("listener.")}
    x = synth.make_inputs(B, T, c["F"], seed=23)
    rng = np.random.default_rng(23)
    cot = rng.standard_normal((B, T // 8, 2 * c["H"])).astype(np.float32)
    sd = O.to_torch_sd(sd_np, requires_grad=True)
    xo = torch.from_numpy(x).requires_grad_(True)
    feat_o = O.listener_forward(xo, sd, c["L"])
    (feat_o * torch.from_numpy(cot)).sum().backward()
    from las_pytorch_amd import Listener
    lis = Listener(input_feature_dim=c["F"], hidden_size=c["H"], num_layers=c["L"], rnn_unit="LSTM", use_gpu=True)
    lis.load_state_dict({k[len("listener."):]: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    lis = lis.cuda()
    xg = torch.from_numpy(x).cuda().requires_grad_(True)
    feat = lis(xg)
    assert _cabi.last_path(_cabi.PATH_REC_FWD) == "rec_fwd_mfma2", _cabi.last_path(_cabi.PATH_REC_FWD)
    assert_close(feat.detach().cpu().numpy(), feat_o.detach().numpy(), "P_B320/listener")
    (feat * torch.from_numpy(cot).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _cabi.last_path(_cabi.PATH_REC_BWD) == "rec_bwd_mfma", _cabi.last_path(_cabi.PATH_REC_BWD)
    gscale = max(float(sd[k].grad.norm()) for k in sd)
    want_dx = xo.grad.numpy()
    grad_close(xg.grad.cpu().numpy(), want_dx, "oracle_P_B320_T800/grad/x", rtol=GRAD_RTOL, floor=GRAD_FLOOR, global_scale=0.0)
    for k, p in lis.named_parameters():
        grad_close(p.grad.cpu().numpy(), sd["listener." + k].grad.numpy(), f"oracle_P_B320_T800/grad/{k}", rtol=GRAD_RTOL, floor=GRAD_FLOOR,
                   global_scale=gscale)
    _check_err()


@pytest.mark.parametrize("cfg_name", ["S", "P"])
def test_generic_and_fast_recurrence_agree(cfg_name):
    from las_pytorch_amd import synth
    from las_pytorch_amd.model.las_model import set_force_generic
    c = synth.CONFIGS[cfg_name]
    sd_np = synth.make_state_dict(synth.config_shapes(cfg_name), seed=9)
    x = torch.from_numpy(synth.make_inputs(4, 64, c["F"], seed=9)).cuda()
    las = build_las(c, sd_np, max_label_len=4)
    outs = []
    for force in (False, True):
        set_force_generic(las, force)
        xg = x.clone().requires_grad_(True)
        feat = las.listener(xg)
        feat.square().sum().backward()
        outs.append((feat.detach().cpu().numpy(), xg.grad.cpu().numpy()))
    assert_close(outs[0][0], outs[1][0], "feat fast vs generic", rtol=1e-4, atol=1e-6)
    assert_close(outs[0][1], outs[1][1], "dx fast vs generic", rtol=1e-3, atol=1e-5 * float(np.abs(outs[1][1]).max()))
    _check_err()


def test_host_rng_and_surface():
    """Exactly one np.random.random_sample() per Speller.forward (reference las_model.py:189); list return types."""
    from las_pytorch_amd import synth
    c = synth.CONFIGS["tiny"]
    sd_np = synth.make_state_dict(synth.config_shapes("tiny"), seed=3)
    las = build_las(c, sd_np, max_label_len=9)
    x = torch.from_numpy(synth.make_inputs(2, 16, c["F"], seed=3)).cuda()
    idx, lens = synth.make_labels(2, 6, c["V"], seed=3)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
    np.random.seed(0)
    ref_stream = np.random.random_sample(8)
    np.random.seed(0)
    steps = []
    with torch.no_grad():
        for i in range(8):
            preds, atts = las(batch_data=x, batch_label=lab, teacher_force_rate=0.5, is_training=True)
            assert isinstance(preds, list) and isinstance(atts, list) and isinstance(atts[0], list)
            assert preds[0].shape == (2, c["V"]) and atts[0][0].shape == (2, 16 // 4)
            steps.append(len(preds))
    assert steps == [6 if r < 0.5 else 9 for r in ref_stream]
    with pytest.raises(RuntimeError):
        las.listener(torch.zeros(2, 6, c["F"], device="cuda")[:, :5])   # odd frame count after layer 0 (T=5)
    _check_err()


def test_forward_step_matches_oracle_and_loop():
    """Speller.forward_step (reference las_model.py:178-184) with caller-managed state: three chained steps equal the
    oracle's step function and the first steps of Speller.forward."""
    from las_pytorch_amd import synth
    from oracle import las_oracle as O
    c = synth.CONFIGS["S"]
    sd_np = synth.make_state_dict(synth.config_shapes("S"), seed=11, scale=0.15)
    las = build_las(c, sd_np, max_label_len=3)
    sd = O.to_torch_sd(sd_np)
    x = synth.make_inputs(3, 32, c["F"], seed=11)
    with torch.no_grad():
        feat_o = O.listener_forward(torch.from_numpy(x), sd, c["L"])
        feat = las.listener(torch.from_numpy(x).cuda())
        B, V = 3, c["V"]
        y = torch.zeros(B, V); y[:, 0] = 1.0
        inp_o = torch.cat([y, feat_o[:, 0, :]], -1)
        inp = torch.cat([y.cuda(), feat[:, 0, :]], -1).unsqueeze(1)
        hid_o, hid = None, None
        for step in range(3):
            lp_o, hid_o, ctx_o, sc_o = O.speller_step(inp_o, hid_o, feat_o, sd, num_layers=c["Ls"])
            lp, hid, ctx, sc = las.speller.forward_step(inp, hid, feat)
            assert_close(lp.cpu().numpy(), lp_o.numpy(), f"step{step}/logp")
            assert_close(ctx.cpu().numpy(), ctx_o.numpy(), f"step{step}/ctx")
            assert_close(sc[0].cpu().numpy(), sc_o[0].numpy(), f"step{step}/att", atol=1e-6)
            assert_close(hid[0].cpu().numpy(), hid_o[0].numpy(), f"step{step}/h")
            yo = torch.zeros(B, V); yo[torch.arange(B), lp_o.argmax(-1)] = 1.0
            inp_o = torch.cat([yo, ctx_o], -1)
            inp = torch.cat([yo.cuda(), ctx], -1).unsqueeze(1)
        # the loop API (greedy) walks through the same states
        preds, _ = las.speller(feat, ground_truth=None, teacher_force_rate=0)
        assert_close(preds[2].cpu().numpy(), lp.cpu().numpy(), "loop vs step", rtol=1e-5, atol=1e-6)
    _check_err()


def test_device_loss_and_ler_match_solver_counterpart():
    """las_ls_loss / las_letter_error_rate against the torch form of solver.py:33-45 and the
    Python LetterErrorRate, value and gradient, ragged labels included; and a full batch_iterator step on the device path."""
    from las_pytorch_amd import synth
    from las_pytorch_amd.solver import solver as S
    g = torch.Generator().manual_seed(4)
    B, U, V = 7, 19, 30
    idx, lens = synth.make_labels(B, U, V, seed=4, ragged=True)
    onehot = torch.from_numpy(synth.onehot_labels(idx, lens, V))
    logits = torch.randn(B, U, V, generator=g)
    ref_in = torch.log_softmax(logits, -1).requires_grad_(True)
    want = S.label_smoothing_loss(ref_in, onehot.float(), 0.1)
    want.backward()
    dev_in = torch.log_softmax(logits, -1).cuda().requires_grad_(True)
    got = S.label_smoothing_loss_device(dev_in, onehot.cuda(), 0.1)
    (got * 1.0).backward()
    assert abs(got.item() - want.item()) < 1e-6 * abs(want.item()) + 1e-7
    assert_close(dev_in.grad.cpu().numpy(), ref_in.grad.numpy(), "dlogp", rtol=1e-5, atol=1e-9)
    ler_dev = S.LetterErrorRate_device(dev_in, onehot.cuda()).cpu().numpy()
    ler_ref = S.LetterErrorRate(ref_in.detach().argmax(-1).numpy(), onehot.argmax(-1).numpy())
    np.testing.assert_allclose(ler_dev, np.array(ler_ref), rtol=1e-6)
    # whole caller step (fwd, fused loss, bwd, clip, Adam, LER) == the golden reference step
    gold, info, sd_np, x, _, _, oh = load_case("tiny_default")
    las = build_las(info["cfg"], sd_np, max_label_len=info["free_len"])
    opt = torch.optim.Adam(las.parameters(), lr=2e-4)
    np.random.seed(0)
    loss, ler = S.batch_iterator(torch.from_numpy(x).cuda(), torch.from_numpy(oh).cuda(), las, opt, tf_rate=1.0, is_training=True,
                                 max_label_len=info["U"], label_smoothing=0.1)
    assert abs(float(loss) - gold["step_loss"][0]) < 2e-5
    np.testing.assert_allclose(np.array(ler), gold["step_ler"], rtol=1e-6)
    sums = np.array([p.detach().double().sum().item() for p in las.parameters()])
    np.testing.assert_allclose(sums, gold["step_param_sum"], rtol=1e-4, atol=2e-4)
    _check_err()


@pytest.mark.parametrize("B,U,eos_rate", [(32, 128, 0.0), (5, 63, 0.05), (4, 64, 0.0), (3, 65, 0.2), (6, 300, 0.0), (3, 700, 0.01), (3, 1100, 0.0), (2, 2500, 0.0),
                                          (2, 4095, 0.0), (3, 2, 0.0), (4, 3, 0.5)])
def test_device_letter_error_rate_wave_kernel(B, U, eos_rate):
    """``las_letter_error_rate`` (one wave per utterance; anti-diagonal Levenshtein with 1 / 2 / 4 / 8 / 16 / 32 / 64 truth columns per lane) against
    the Python ``LetterErrorRate`` of the solver (reference solver/solver.py:11-24) on random predictions: <sos>/pad symbols inside the
    prediction (skipped), early <eos> (stops the prediction), ragged truths, every columns-per-lane instantiation, U = 2 and the 4095 limit;
    and on the (U,B,V) layout the decode kernels write (strided view)."""
    from las_pytorch_amd import synth
    from las_pytorch_amd.solver import solver as S
    V = 30
    rng = np.random.default_rng(1000 * B + U)
    idx, lens = synth.make_labels(B, U, V, seed=U, ragged=U > 2)
    onehot = torch.from_numpy(synth.onehot_labels(idx, lens, V))
    pred = rng.integers(2, V, size=(B, U))
    keep = rng.random((B, U)) < 0.6                     # 60 % of the steps repeat the truth: distances well below the maximum
    pred[keep] = np.maximum(idx, 2)[keep]
    pred[rng.random((B, U)) < 0.1] = 0                  # pad / <sos> inside the prediction
    pred[rng.random((B, U)) < eos_rate] = 1             # early <eos>
    logp = torch.full((U, B, V), -5.0)
    logp.scatter_(2, torch.from_numpy(pred.T.copy()).unsqueeze(-1), -0.1)
    want = np.array(S.LetterErrorRate(pred, onehot.argmax(-1).numpy()))
    dev = logp.cuda()
    got_ubv = S.LetterErrorRate_device(dev.transpose(0, 1), onehot.cuda()).cpu().numpy()       # strided (B,U,V) view of the (U,B,V) buffer
    got_buv = S.LetterErrorRate_device(dev.transpose(0, 1).contiguous(), onehot.cuda()).cpu().numpy()
    np.testing.assert_allclose(got_ubv, want, rtol=1e-6)
    np.testing.assert_allclose(got_buv, want, rtol=1e-6)
    _check_err()


@pytest.mark.parametrize("cfg_name,B,T,U", [("P", 1, 8, 1), ("S", 1, 4, 2), ("tiny", 1, 4, 1), ("P", 2, 16, 3)])
def test_edge_shapes_vs_oracle(cfg_name, B, T, U):
    """Smallest legal shapes: one utterance, one encoder frame after the pyramid (T = 2**L), a single decode step
    (no recurrent weight gradient), forward + backward against the oracle."""
    from las_pytorch_amd import synth
    from oracle import las_oracle as O
    c = synth.CONFIGS[cfg_name]
    sd_np = synth.make_state_dict(synth.config_shapes(cfg_name), seed=13, scale=0.1)
    x = synth.make_inputs(B, T, c["F"], seed=13)
    idx, lens = synth.make_labels(B, U, c["V"], seed=13)
    idx[:, -1] = 5                                   # keep at least one real symbol when U == 1
    onehot = synth.onehot_labels(idx, lens, c["V"])
    sd = O.to_torch_sd(sd_np, requires_grad=True)
    lab = torch.from_numpy(onehot)
    preds_o, _ = O.las_forward(torch.from_numpy(x), lab, sd, dict(listener_layers=c["L"], speller_layers=c["Ls"], max_label_len=U,
                                                                  decode_mode=1), teacher_force=True)
    loss_o, _ = O.solver_step_loss(preds_o, lab, U, 0.1)
    loss_o.backward()
    las = build_las(c, sd_np, max_label_len=U)
    preds, _ = las(batch_data=torch.from_numpy(x).cuda(), batch_label=lab.cuda(), teacher_force_rate=1.0, is_training=True)
    assert_close(torch.stack(preds).detach().cpu().numpy(), torch.stack(preds_o).detach().numpy(), "logp")
    loss = _loss_ls(preds, lab.cuda(), U)
    loss.backward()
    gscale = max(float(sd[k].grad.norm()) for k in sd)
    for k, p in las.named_parameters():
        want = sd[k].grad.numpy()
        grad_close(p.grad.cpu().numpy(), want, f"edge_{cfg_name}_B{B}_T{T}_U{U}/grad/{k}", rtol=GRAD_RTOL, floor=GRAD_FLOOR,
                   global_scale=gscale)
    _check_err()


@pytest.mark.parametrize("cfg_name,B,Tp,U,scale", [("P", 32, 100, 24, None), ("P", 5, 37, 7, 0.1), ("S", 17, 200, 9, None),
                                                    ("S", 32, 100, 12, 0.1), ("P", 1, 1, 3, None),
                                                    ("P", 8, 375, 5, None), ("P", 32, 200, 4, None), ("S", 8, 500, 4, None),
                                                    ("S", 16, 300, 4, None),
                                                    # batches beyond one launch: Speller._run slices them (32 + 32 + 6, 32 + 1)
                                                    ("P", 70, 50, 6, None), ("S", 33, 40, 5, None),
                                                    # Hs = 256 beyond T' = 448: the keys split by frames over 16 workgroups per utterance (round 5)
                                                    ("S", 8, 750, 6, None), ("S", 12, 896, 3, None), ("S", 3, 449, 4, 0.1), ("S", 1, 600, 3, None)])
def test_persistent_decode_kernel_matches_stepwise(cfg_name, B, Tp, U, scale):
    _persistent_vs_stepwise(cfg_name, B, Tp, U, scale, "relu")


@pytest.mark.parametrize("B,Tp", [(8, 750), (12, 896), (2, 449)])
def test_long_utterance_small_model_decode_takes_the_frame_split_kernel(B, Tp):
    """BASELINE configs[4] for the small model (Hs = 256, T' = 750): 192 KB of keys do not fit one workgroup's LDS — the teacher-forced forward
    keeps them split by frames over the 16 attention workgroups of an utterance, which exchange their energies every step
    (``AttnPreRole<256, 16>``); asserts that path (the per-step kernels took these shapes until round 5) and the PRE backward, whose frame table now reaches T' = 896 at Hs = 256."""
    from las_pytorch_amd import Speller, synth
    c = synth.CONFIGS["S"]
    U = 4
    torch.manual_seed(5)
    sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U, use_mlp_in_attention=True,
                 mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu", listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
    feat = (torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5).requires_grad_(True)
    idx, lens = synth.make_labels(B, U, c["V"], seed=11, ragged=True)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
    preds, _ = sp(feat, ground_truth=lab, teacher_force_rate=1.0)
    torch.stack(preds).sum().backward()
    torch.cuda.synchronize()
    assert (_cabi.last_path(_cabi.PATH_DECODE_FWD), _cabi.last_path(_cabi.PATH_DECODE_BWD)) == ("persist_pre", "persist_pre")
    _check_err()


@pytest.mark.parametrize("B,Tp,U,scale,activate", [(16, 100, 12, None, "relu"), (3, 8, 6, 0.08, "relu"), (16, 200, 5, None, "relu"),
                                                    (5, 37, 7, 0.08, "relu"), (1, 1, 3, None, "relu"), (16, 256, 3, None, "relu"),
                                                    (7, 130, 4, None, "None"), (16, 100, 72, None, "relu"),
                                                    (40, 50, 5, None, "relu"),       # beyond one launch: Speller._run slices 16 + 16 + 8
                                                    # T' > 256: the utterance's workgroups split the energies by frames and exchange them
                                                    (16, 300, 4, None, "relu"), (3, 437, 5, None, "relu"), (7, 480, 3, None, "relu")])
def test_one_launch_decode_of_the_yaml_sizes_matches_stepwise(B, Tp, U, scale, activate):
    """speller_big.hip (Speller 1024x2, attention MLP 64, B <= 16, T' <= 512 forward / 480 backward: the reference's config/librispeech-config.yaml), forward AND
    backward kernel, against the per-step launch chains: log-probs, attention weights and every gradient.  Cases: a full batch at T = 800, partial batches (rows beyond B are never stored), a single frame,
    the longest eligible encoder output, no attention activation, more than 64 steps (the trace buffer's depth; U = 72 and not 70: with
    this seed one query pre-activation of step 69 is within an ulp of 0, and the PER-STEP path's own run-to-run summation order — atomic
    split-K — flips its relu mask in about a third of the runs)."""
    _persistent_vs_stepwise("Y", B, Tp, U, scale, activate, big=True)


def test_persistent_decode_kernel_without_attention_activation():
    """mlp_activate_in_attention=None (reference las_model.py:262-264: no activation on phi/psi): the relu masks of the
    forward query and of its backward are switched off by a run-time flag in the persistent kernels."""
    _persistent_vs_stepwise("S", 9, 40, 6, None, "None")


@pytest.mark.parametrize("cfg_name,B,Tp,U,scale,activate,heads", [("P", 32, 100, 8, None, "tanh", 1), ("P", 11, 57, 5, 0.1, "sigmoid", 1),
                                                                   ("S", 16, 200, 6, None, "tanh", 1), ("P", 16, 100, 6, None, "sigmoid", 2),
                                                                   ("P", 8, 60, 5, 0.1, "tanh", 4), ("P", 8, 375, 4, None, "tanh", 1)])
def test_persistent_decode_kernels_with_tanh_and_sigmoid_attention(cfg_name, B, Tp, U, scale, activate, heads):
    """mlp_activate_in_attention = tanh / sigmoid (the reference resolves any torch.nn.functional name, las_model.py:270-273) on the one-launch
    decode kernels, single- and multi-head, forward and backward, against the per-step launch chains (which the reference-generated
    tiny_tanh / tiny_sigmoid goldens pin): the query activation and its derivative are a run-time switch (act_apply / act_grad) there."""
    _persistent_vs_stepwise(cfg_name, B, Tp, U, scale, activate, heads=heads)


@pytest.mark.parametrize("cfg_name,B,Tp,U", [("P", 32, 100, 16), ("P", 7, 57, 5), ("S", 20, 111, 6), ("S", 32, 200, 5), ("P", 16, 200, 4),
                                             ("P", 8, 375, 4), ("S", 16, 300, 4)])
def test_pre_multiplied_context_backward_matches_classic_persistent_backward(cfg_name, B, Tp, U):
    """las_speller_bwd with and without LAS_FLAG_TEACHER_FORCED on the SAME forward stash: with the flag the attention-backward
    workgroups contract the gate gradients with feat.W_ctx^T (speller_persist_bwd_pre_kernel), without it the classic
    persistent backward multiplies dG0 W_ctx on the chain.  Same gradients either way (fp32 re-association only)."""
    from las_pytorch_amd import Speller, synth
    from las_pytorch_amd.model import las_model as M
    c = synth.CONFIGS[cfg_name]
    torch.manual_seed(9)
    sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U,
                 use_mlp_in_attention=True, mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu",
                 listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
    feat0 = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5
    idx, lens = synth.make_labels(B, U, c["V"], seed=3, ragged=True)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
    w = torch.randn(U, B, c["V"], device="cuda")
    res = []
    saved = M.FLAG_TEACHER_FORCED
    for flag in (saved, 0):
        M.FLAG_TEACHER_FORCED = flag
        try:
            sp.zero_grad(set_to_none=True)
            feat = feat0.clone().requires_grad_(True)
            preds, _ = sp(feat, ground_truth=lab, teacher_force_rate=1.0)
            (torch.stack(preds) * w).sum().backward()
            res.append(dict(dfeat=feat.grad.cpu().numpy(), **{"d" + n: p.grad.cpu().numpy() for n, p in sp.named_parameters()}))
        finally:
            M.FLAG_TEACHER_FORCED = saved
    for k in res[0]:
        scale_k = float(np.abs(res[1][k]).max()) + 1e-30
        assert_close(res[0][k], res[1][k], f"pre vs classic persistent backward {k}", rtol=1e-3, atol=1e-5 * max(1.0, scale_k))
    _check_err()


@pytest.mark.parametrize("cfg_name,heads,B,Tp,U,scale,activate", [("P", 2, 16, 100, 12, None, "relu"), ("P", 2, 5, 37, 7, 0.1, "relu"), ("P", 4, 8, 100, 6, None, "relu"),
                                                                    ("P", 4, 3, 100, 4, 0.1, "relu"), ("P", 2, 10, 60, 5, None, "relu"), ("S", 2, 16, 200, 6, None, "relu"),
                                                                    ("S", 4, 8, 100, 5, 0.1, "None"), ("P", 2, 8, 224, 4, None, "relu"), ("P", 2, 1, 1, 3, None, "relu"),
                                                                    ("S", 2, 4, 400, 3, None, "relu"),
                                                                    # two slices of 16 / of 8 (Speller._run), then a ragged last slice
                                                                    ("P", 2, 32, 50, 5, None, "relu"), ("P", 4, 13, 40, 4, None, "relu"),
                                                                    # ragged remainder slices (16 + 8 at heads = 2, 8 + 4 at heads = 4): the tail slice takes the
                                                                    # one-launch kernel under the workgroup map of ITS size (path asserted on the tail)
                                                                    ("P", 2, 24, 50, 5, None, "relu"), ("P", 4, 12, 40, 4, None, "relu")])
def test_multi_head_one_launch_decode_matches_stepwise(cfg_name, heads, B, Tp, U, scale, activate):
    """Multi-head attention (reference las_model.py:298-314) on the one-launch decode kernels — one set of attention workgroups per (utterance,
    head), dim_reduce folded into the pre-multiplied context, the heads' sums exchanged before the bottom cell; backward: heads x frame slices —
    against the per-step launch chains: log-probs, every head's attention weights, dfeat and all parameter gradients (dim_reduce included)."""
    _persistent_vs_stepwise(cfg_name, B, Tp, U, scale, activate, heads=heads)
    assert _cabi.last_path(_cabi.PATH_DECODE_FWD) == "stepwise"        # (the second pass of the helper is the forced per-step one)


def _persistent_vs_stepwise(cfg_name, B, Tp, U, scale, activate, big=False, heads=1):
    """The one-launch teacher-forced decode loop (speller_persist.hip) against the per-step launch chain it replaces:
    outputs and every gradient (the backward pass consumes the stash the forward kernel wrote).  The larger-weight
    cases stay at U(-0.1,0.1): with the U(-0.5,0.5) set the attention softmax is an arg-max over energies of order 1e3
    and two correct fp32 summation orders legitimately pick different frames on near-ties (the saturating set is
    covered against the reference's own outputs by the *_sat golden cases, which run through this kernel too)."""
    import ctypes
    from las_pytorch_amd import Speller, _cabi, synth
    c = synth.CONFIGS[cfg_name]
    torch.manual_seed(5)
    sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U,
                 use_mlp_in_attention=True, mlp_dim_in_attention=c["M"], mlp_activate_in_attention=activate,
                 listener_hidden_size=c["H"], multi_head=heads, decode_mode=1).cuda()
    if scale is not None:
        with torch.no_grad():
            for p in sp.parameters():
                p.uniform_(-scale, scale)
    feat0 = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5
    idx, lens = synth.make_labels(B, U, c["V"], seed=11, ragged=True)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
    w = torch.randn(U, B, c["V"], device="cuda")
    L = _cabi.lib()
    L.las_debug_persist_trace.argtypes = [ctypes.c_void_p]
    L.las_debug_persist_trace.restype = None
    # las_debug_persist_trace: 3 roles x U steps x 8 stamps; las_debug_big_trace (speller_big.hip): 64 steps x 16 + 256 workgroups x 8
    trace = torch.zeros(64 * 16 + 256 * 8 if big else 3 * U * 8, dtype=torch.int64, device="cuda")
    btrace = torch.zeros(4 * 64 * 16 + 256 * 8, dtype=torch.int64, device="cuda")      # las_debug_big_bwd_trace: 4 roles x 64 steps x 16 + 256 x 8
    def set_trace(ptr):
        if big:
            L.las_debug_big_trace(ptr)
            L.las_debug_big_bwd_trace(btrace.data_ptr() if ptr else None)
        else:
            L.las_debug_persist_trace(ptr)
    res = []
    for force in (False, True):
        sp.force_generic = force
        set_trace(trace.data_ptr() if not force else None)
        try:
            sp.zero_grad(set_to_none=True)
            feat = feat0.clone().requires_grad_(True)
            preds, att = sp(feat, ground_truth=lab, teacher_force_rate=1.0)
            logp = torch.stack(preds)
            (logp * w).sum().backward()
            res.append(dict(logp=logp.detach().cpu().numpy(), att=torch.stack([torch.stack(list(a)) for a in att]).detach().cpu().numpy(),
                            dfeat=feat.grad.cpu().numpy(), **{"d" + n: p.grad.cpu().numpy() for n, p in sp.named_parameters()}))
            if not force:
                paths = (_cabi.last_path(_cabi.PATH_DECODE_FWD), _cabi.last_path(_cabi.PATH_DECODE_BWD))
        finally:
            sp.force_generic = False
            set_trace(None)
    torch.cuda.synchronize()
    if heads > 1:
        assert paths == ("persist_pre_mh", "persist_pre_mh"), paths      # (of the LAST slice: the ragged tail of a sliced batch included)
    assert int(trace.abs().sum().item()) != 0, "the persistent kernel did not run (shape not eligible?)"
    assert not big or int(btrace.abs().sum().item()) != 0, "the one-launch backward did not run"
    for k in res[0]:
        scale_k = float(np.abs(res[1][k]).max()) + 1e-30
        assert_close(res[0][k], res[1][k], f"persistent vs stepwise {k}", rtol=1e-3, atol=1e-5 * max(1.0, scale_k))
    _check_err()


@pytest.mark.parametrize("cfg_name,B", [("S", 4), ("P", 5), ("P", 40)])
def test_direct_gradient_write_matches_autograd_accumulation(cfg_name, B):
    """FlatGradAllReducer(direct=True): the backward kernels fill the flat gradient buffer themselves; the result must
    equal what autograd's per-parameter accumulation produces, also on a second step after zero()."""
    from las_pytorch_amd import dp, synth
    from las_pytorch_amd.model import las_model
    c = synth.CONFIGS[cfg_name]
    sd_np = synth.make_state_dict(synth.config_shapes(cfg_name), seed=4)
    x = torch.from_numpy(synth.make_inputs(B, 64, c["F"], seed=4)).cuda()
    idx, lens = synth.make_labels(B, 6, c["V"], seed=4)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
    flats = []
    other = build_las(c, sd_np, max_label_len=6)       # a second model in the process: never tagged, must keep accumulating
    for direct in (False, True):
        las = build_las(c, sd_np, max_label_len=6)
        red = dp.FlatGradAllReducer(las, direct=direct)
        assert (las_model._direct_targets(list(las.parameters())) is not None) == direct
        assert las_model._direct_targets(list(other.parameters())) is None
        for _ in range(2):
            red.zero()
            preds, _ = las(x, lab, 1.0, True)
            torch.stack(preds).square().mean().backward()
            red.check_views()
        flats.append(red.flat.detach().cpu().numpy().copy())
        if direct and B <= 32:      # (a sliced decode, B > 32, hands the Speller's gradients to autograd, which accumulates by definition)
            # a further backward WITHOUT zero(): the blocks were written in this zero-epoch, so LAS_FLAG_GRADS_ZEROED is not claimed, the
            # entry points fill their blocks themselves and the direct writes overwrite (they do not accumulate onto the stale values)
            assert not las_model._claim_prezeroed(list(las.parameters()))
            preds, _ = las(x, lab, 1.0, True)
            torch.stack(preds).square().mean().backward()
            assert_close(red.flat.detach().cpu().numpy(), flats[-1], "direct writes overwrite without zero()", rtol=1e-4,
                         atol=1e-6 * float(np.abs(flats[-1]).max()))
    # the untagged model accumulates over two backward passes (micro-batching) even while a direct reducer exists
    for _ in range(2):
        preds, _ = other(x, lab, 1.0, True)
        torch.stack(preds).square().mean().backward()
    acc = torch.cat([p.grad.reshape(-1) for p in other.parameters()]).cpu().numpy()
    assert_close(acc, 2 * flats[0], "accumulation of an untagged model", rtol=1e-4, atol=1e-6 * float(np.abs(flats[0]).max()))
    # zero_grad() with set_to_none drops the views: the next backward falls back to autograd accumulation and the
    # reducer refuses to all-reduce a stale buffer
    las.zero_grad(set_to_none=True)
    preds, _ = las(x, lab, 1.0, True)
    torch.stack(preds).square().mean().backward()
    with pytest.raises(RuntimeError):
        red.allreduce_mean()
    assert np.abs(flats[0]).max() > 0
    # split-K GEMMs accumulate with atomics, so two runs agree only to fp32 rounding of the sum order
    assert_close(flats[1], flats[0], "direct vs accumulated flat gradient", rtol=1e-4, atol=1e-6 * float(np.abs(flats[0]).max()))
    _check_err()


@pytest.mark.parametrize("cfg_name,B,Tp,U,decode_mode", [("P", 32, 100, 20, 1), ("P", 32, 100, 20, 0), ("S", 7, 150, 9, 1),
                                                          ("S", 32, 60, 6, 0), ("P", 3, 30, 5, 1), ("P", 40, 30, 5, 1),
                                                          # the YAML sizes (speller_big.hip, greedy only): full batch, partial batch, a sliced batch
                                                          ("Y", 16, 100, 20, 1), ("Y", 5, 37, 9, 1), ("Y", 24, 40, 5, 1), ("Y", 16, 400, 4, 1)])
def test_persistent_free_running_decode_matches_stepwise(cfg_name, B, Tp, U, decode_mode):
    """Greedy (decode_mode 1) and log-prob-feedback (decode_mode 0) decoding inside the one-launch kernel against the
    per-step launch chain: log-probabilities, attention, arg-max sequences, and (mode 1) the gradients."""
    big = cfg_name == "Y"
    import ctypes
    from las_pytorch_amd import Speller, _cabi, synth
    c = synth.CONFIGS[cfg_name]
    torch.manual_seed(6)
    sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U,
                 use_mlp_in_attention=True, mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu",
                 listener_hidden_size=c["H"], multi_head=1, decode_mode=decode_mode).cuda()
    feat0 = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5
    w = torch.randn(U, B, c["V"], device="cuda")
    L = _cabi.lib()
    L.las_debug_persist_trace.argtypes = [ctypes.c_void_p]
    L.las_debug_persist_trace.restype = None
    # las_debug_persist_trace: 3 roles x U steps x 8 stamps; las_debug_big_trace: 64 steps x 16 + 256 workgroups x 8
    trace = torch.zeros(64 * 16 + 256 * 8 if big else 3 * U * 8, dtype=torch.int64, device="cuda")
    set_trace = L.las_debug_big_trace if big else L.las_debug_persist_trace
    res = []
    for force in (False, True):
        sp.force_generic = force
        set_trace(trace.data_ptr() if not force else None)
        try:
            sp.zero_grad(set_to_none=True)
            feat = feat0.clone().requires_grad_(True)
            preds, att = sp(feat, ground_truth=None, teacher_force_rate=0.0)
            logp = torch.stack(preds)
            (logp * w).sum().backward()
            out = dict(logp=logp.detach().cpu().numpy(), att=torch.stack([a[0] for a in att]).detach().cpu().numpy(),
                       dfeat=feat.grad.cpu().numpy())
            out.update({"d" + n: p.grad.cpu().numpy() for n, p in sp.named_parameters() if p.grad is not None})
            res.append(out)
        finally:
            sp.force_generic = False
            set_trace(None)
    torch.cuda.synchronize()
    assert int(trace.abs().sum().item()) != 0, "the persistent kernel did not run (shape not eligible?)"
    assert (res[0]["logp"].argmax(-1) == res[1]["logp"].argmax(-1)).all(), "arg-max sequences differ"
    for k in res[0]:
        scale_k = float(np.abs(res[1][k]).max()) + 1e-30
        assert_close(res[0][k], res[1][k], f"persistent vs stepwise free-running {k}", rtol=1e-3, atol=1e-5 * max(1.0, scale_k))
    _check_err()


@pytest.mark.parametrize("cfg_name,heads,B,Tp,U", [("P", 2, 16, 100, 8), ("P", 4, 5, 37, 6), ("S", 2, 9, 120, 5), ("P", 2, 3, 20, 4)])
def test_multi_head_free_running_training_step_backward_on_the_one_launch_kernel(cfg_name, heads, B, Tp, U):
    """A free-running (decode_mode 1) TRAINING step with several attention heads: the forward is the free-running multi-head PRE kernel (the heads'
    shares of the character distribution meet inside the attention workgroups), the backward the teacher-forced multi-head PRE backward over the
    emitted symbols (it reads P and gx per head, which a stashing forward leaves whichever kernels it ran).  Against the per-step kernels both
    ways: log-probs, arg-max sequences, every gradient."""
    from las_pytorch_amd import Speller, synth
    c = synth.CONFIGS[cfg_name]
    torch.manual_seed(8)
    sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U,
                 use_mlp_in_attention=True, mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu",
                 listener_hidden_size=c["H"], multi_head=heads, decode_mode=1).cuda()
    feat0 = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5
    w = torch.randn(U, B, c["V"], device="cuda")
    res, paths = [], []
    for force in (False, True):
        sp.force_generic = force
        try:
            sp.zero_grad(set_to_none=True)
            feat = feat0.clone().requires_grad_(True)
            preds, att = sp(feat, ground_truth=None, teacher_force_rate=0.0)
            logp = torch.stack(preds)
            (logp * w).sum().backward()
            torch.cuda.synchronize()
            paths.append((_cabi.last_path(_cabi.PATH_DECODE_FWD), _cabi.last_path(_cabi.PATH_DECODE_BWD)))
            out = dict(logp=logp.detach().cpu().numpy(), dfeat=feat.grad.cpu().numpy())
            out.update({"d" + n: p.grad.cpu().numpy() for n, p in sp.named_parameters() if p.grad is not None})
            res.append(out)
        finally:
            sp.force_generic = False
    assert paths == [("persist_pre_mh_greedy", "persist_pre_mh"), ("stepwise", "stepwise")], paths
    assert (res[0]["logp"].argmax(-1) == res[1]["logp"].argmax(-1)).all()
    assert set(res[0]) == set(res[1])
    for k in res[0]:
        scale_k = float(np.abs(res[1][k]).max()) + 1e-30
        assert_close(res[0][k], res[1][k], f"multi-head free-running training {k}", rtol=1e-3, atol=1e-5 * max(1.0, scale_k))
    _check_err()


@pytest.mark.parametrize("cfg_name,heads,B,Tp,U", [("P", 2, 16, 100, 20), ("P", 4, 8, 100, 9), ("P", 2, 7, 57, 6), ("S", 2, 16, 200, 8), ("S", 4, 5, 90, 5),
                                                   ("P", 2, 8, 200, 4), ("P", 2, 32, 60, 5), ("P", 4, 11, 30, 4)])
def test_multi_head_free_running_decode_without_backward(cfg_name, heads, B, Tp, U):
    """Validation-style greedy decode (decode_mode 1 under torch.no_grad()) with several attention heads on the free-running multi-head PRE kernel
    against the per-step kernels: log-probabilities, arg-max sequences, every head's attention weights; partial batches, two slices (32 at
    heads = 2, 11 at heads = 4), 4 and 8 workgroups per (utterance, head)."""
    from las_pytorch_amd import Speller, synth
    c = synth.CONFIGS[cfg_name]
    torch.manual_seed(9)
    sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U,
                 use_mlp_in_attention=True, mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu",
                 listener_hidden_size=c["H"], multi_head=heads, decode_mode=1).cuda()
    feat = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5
    res, paths = [], []
    for force in (False, True):
        sp.force_generic = force
        try:
            with torch.no_grad():
                preds, att = sp(feat, ground_truth=None, teacher_force_rate=0.0)
            torch.cuda.synchronize()
            paths.append(_cabi.last_path(_cabi.PATH_DECODE_FWD))
            res.append((torch.stack(preds).cpu().numpy(), torch.stack([torch.stack(list(a)) for a in att]).cpu().numpy()))
        finally:
            sp.force_generic = False
    assert paths == ["persist_pre_mh_greedy", "stepwise"], paths
    assert (res[0][0].argmax(-1) == res[1][0].argmax(-1)).all(), "arg-max sequences differ"
    assert_close(res[0][0], res[1][0], "multi-head free-running logp")
    assert_close(res[0][1], res[1][1], "multi-head free-running attention", rtol=1e-3, atol=1e-6)
    _check_err()


@pytest.mark.parametrize("cfg_name,B,Tp,U", [("P", 32, 100, 24), ("P", 7, 57, 9), ("P", 16, 200, 6), ("S", 32, 200, 12), ("S", 20, 100, 5),
                                             ("P", 40, 100, 5), ("P", 32, 112, 3),
                                             # 16 workgroups per utterance, keys split by frames (BASELINE configs[4]: T' = 375 / 750), and the table's ends
                                             ("P", 8, 375, 6), ("S", 8, 750, 5), ("P", 3, 448, 3), ("S", 12, 896, 3), ("P", 5, 225, 4)])
def test_free_running_decode_without_backward_runs_the_pre_kernel(cfg_name, B, Tp, U):
    """Validation-style greedy decode (decode_mode 1 under torch.no_grad(): reference train.py:149-169, las_model.py:223-227) takes the
    free-running form of the pre-multiplied-context kernel — character distribution inside the attention workgroups, 4 and 8 of them per
    utterance, partial batches, a sliced batch (40), the largest T' of the 4-workgroup form (112) — and must give the per-step kernels'
    log-probabilities, attention weights and arg-max sequence.  With autograd on (a free-running TRAINING step) the same kernel runs and stashes
    for the teacher-forced-style backward (gradients: test_persistent_free_running_decode_matches_stepwise)."""
    from las_pytorch_amd import Speller, _cabi, synth
    c = synth.CONFIGS[cfg_name]
    torch.manual_seed(8)
    sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U,
                 use_mlp_in_attention=True, mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu",
                 listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
    feat = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5
    res = []
    for force in (False, True):
        sp.force_generic = force
        try:
            with torch.no_grad():
                preds, att = sp(feat, ground_truth=None, teacher_force_rate=0.0)
            torch.cuda.synchronize()
            if not force:
                assert _cabi.last_path(_cabi.PATH_DECODE_FWD) == "persist_pre_greedy", _cabi.last_path(_cabi.PATH_DECODE_FWD)
            res.append((torch.stack(preds).cpu().numpy(), torch.stack([a[0] for a in att]).cpu().numpy()))
        finally:
            sp.force_generic = False
    assert (res[0][0].argmax(-1) == res[1][0].argmax(-1)).all(), "arg-max sequences differ"
    assert_close(res[0][0], res[1][0], "pre free-running vs stepwise logp")
    assert_close(res[0][1], res[1][1], "pre free-running vs stepwise att", atol=1e-6)
    preds, _ = sp(feat.clone().requires_grad_(True), ground_truth=None, teacher_force_rate=0.0)      # autograd on: a backward may follow
    torch.cuda.synchronize()
    assert _cabi.last_path(_cabi.PATH_DECODE_FWD) == "persist_pre_greedy"
    assert_close(torch.stack(preds).detach().cpu().numpy(), res[1][0], "stashing free-running vs stepwise logp")
    _check_err()


def test_free_running_training_step_at_long_t_runs_the_one_launch_kernels_both_ways():
    """T' = 375 (BASELINE configs[4]) with decode_mode 1 and a backward pass: the free-running forward (the one-launch kernel with the keys
    split by frames over 16 workgroups per utterance) must leave
    P / gx in the reserve exactly as a teacher-forced forward does, because the backward takes the teacher-forced PRE kernel over the emitted
    symbols.  Against the all-generic path."""
    from las_pytorch_amd import Speller, _cabi, synth
    c = synth.CONFIGS["P"]
    B, Tp, U = 8, 375, 4
    torch.manual_seed(9)
    sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U, use_mlp_in_attention=True,
                 mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu", listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
    feat0 = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.5
    w = torch.randn(U, B, c["V"], device="cuda")
    res = []
    for force in (False, True):
        sp.force_generic = force
        try:
            sp.zero_grad(set_to_none=True)
            feat = feat0.clone().requires_grad_(True)
            preds, _ = sp(feat, ground_truth=None, teacher_force_rate=0.0)
            logp = torch.stack(preds)
            (logp * w).sum().backward()
            torch.cuda.synchronize()
            if not force:
                assert (_cabi.last_path(_cabi.PATH_DECODE_FWD), _cabi.last_path(_cabi.PATH_DECODE_BWD)) == ("persist_pre_greedy", "persist_pre")
            out = dict(logp=logp.detach().cpu().numpy(), dfeat=feat.grad.cpu().numpy())
            out.update({"d" + n: p.grad.cpu().numpy() for n, p in sp.named_parameters() if p.grad is not None})
            res.append(out)
        finally:
            sp.force_generic = False
    for k in res[0]:
        scale_k = float(np.abs(res[1][k]).max()) + 1e-30
        assert_close(res[0][k], res[1][k], f"long-T free-running {k}", rtol=1e-3, atol=1e-5 * max(1.0, scale_k))
    _check_err()


@pytest.mark.parametrize("name", ["tiny_mode2", "S_mode2"])
def test_decode_mode2_golden(name):
    """decode_mode 2 (reference las_model.py:229-234) on the device from the reference's own Exp(1) draws: the sampled
    symbols, hence the log-probs of every later step, equal the reference's."""
    from golden_util import load_mode2_case
    g, c, sd_np, x = load_mode2_case(name)
    U = g["mode2_logp"].shape[0]
    las = build_las(c, sd_np, max_label_len=U, decode_mode=2)
    xt = torch.from_numpy(x).cuda()
    las.speller.sample_noise = torch.from_numpy(g["mode2_noise"]).cuda()
    with torch.no_grad():
        preds, _ = las(batch_data=xt, batch_label=None, teacher_force_rate=0.0, is_training=False)
    assert_close(torch.stack(preds).cpu().numpy(), g["mode2_logp"], f"{name}/mode2_logp")
    # without caller-supplied draws the module draws its own (one (B,V) exponential_ per step on the device generator)
    las.speller.sample_noise = None
    torch.manual_seed(3)
    with torch.no_grad():
        a, _ = las(batch_data=xt, batch_label=None, teacher_force_rate=0.0, is_training=False)
        torch.manual_seed(3)
        b, _ = las(batch_data=xt, batch_label=None, teacher_force_rate=0.0, is_training=False)
    # same seed -> same draws -> same sampled symbols (the listener's split-K GEMMs accumulate with atomics, so the
    # log-probs of two runs agree to rounding, not bitwise)
    a, b = torch.stack(a), torch.stack(b)
    assert torch.isfinite(a).all() and torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    _check_err()


@pytest.mark.parametrize("cfg_name,heads,use_mlp,activate", [("S", 1, True, "relu"), ("tiny", 4, True, "relu"), ("tiny", 1, False, "None"),
                                                             ("P", 1, True, "None")])
def test_forward_step_is_differentiable_like_the_reference(cfg_name, heads, use_mlp, activate):
    """Speller.forward_step under autograd (reference las_model.py:178-184 is the autograd path there): three chained steps
    with caller-managed state, a loss on every output, against the oracle's step function differentiated by torch —
    gradients wrt the listener features, the initial input and every speller parameter."""
    from las_pytorch_amd import synth
    from oracle import las_oracle as O
    c = synth.CONFIGS[cfg_name]
    shapes = synth.config_shapes(cfg_name, multi_head=heads, use_mlp=use_mlp)
    sd_np = synth.make_state_dict(shapes, seed=31, scale=0.15)
    las = build_las(c, sd_np, max_label_len=3, multi_head=heads, use_mlp=use_mlp, activate=activate)
    sd = O.to_torch_sd(sd_np, requires_grad=True)
    B, Tp, V, Hs = 3, 9, c["V"], c["Hs"]
    g = torch.Generator().manual_seed(7)
    feat_o = (torch.randn(B, Tp, Hs, generator=g) * 0.5).requires_grad_(True)
    feat = feat_o.detach().clone().cuda().requires_grad_(True)
    x0_o = (torch.randn(B, V + Hs, generator=g) * 0.5).requires_grad_(True)
    x0 = x0_o.detach().clone().cuda().requires_grad_(True)
    w = [torch.randn(B, V, generator=g) for _ in range(3)]
    wc = torch.randn(B, Hs, generator=g)
    wh = torch.randn(c["Ls"], B, Hs, generator=g)
    kw = dict(num_layers=c["Ls"], use_mlp=use_mlp, activate=activate, multi_head=heads)
    # oracle chain
    inp, hid, loss_o = x0_o, None, 0.0
    for s in range(3):
        lp, hid, ctx, sc = O.speller_step(inp, hid, feat_o, sd, **kw)
        loss_o = loss_o + (lp * w[s]).sum()
        inp = torch.cat([lp.exp(), ctx], -1)                 # a differentiable feedback (gradient flows through y and ctx)
    loss_o = loss_o + (ctx * wc).sum() + (hid[0] * wh).sum() + (hid[1] * wh).sum() * 0.5
    loss_o.backward()
    # HIP chain
    inp, hid, loss = x0.unsqueeze(1), None, 0.0
    for s in range(3):
        lp, hid, ctx, sc = las.speller.forward_step(inp, hid, feat)
        loss = loss + (lp * w[s].cuda()).sum()
        inp = torch.cat([lp.exp(), ctx], -1).unsqueeze(1)
    loss = loss + (ctx * wc.cuda()).sum() + (hid[0] * wh.cuda()).sum() + (hid[1] * wh.cuda()).sum() * 0.5
    loss.backward()
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * abs(loss_o.item()) + 1e-5
    gs = max(float(t.grad.norm()) for k, t in sd.items() if k.startswith("speller.") and t.grad is not None)
    grad_close(feat.grad.cpu().numpy(), feat_o.grad.numpy(), f"step_{cfg_name}_h{heads}/dfeat", global_scale=gs)
    grad_close(x0.grad.cpu().numpy(), x0_o.grad.numpy(), f"step_{cfg_name}_h{heads}/dinput", global_scale=gs)
    for k, p in las.speller.named_parameters():
        want = sd["speller." + k].grad
        want = torch.zeros_like(sd["speller." + k]) if want is None else want
        grad_close(p.grad.cpu().numpy(), want.numpy(), f"step_{cfg_name}_h{heads}/grad/{k}", global_scale=gs)
    _check_err()


@pytest.mark.parametrize("heads,use_mlp,activate", [(1, True, "relu"), (4, True, "relu"), (1, False, "None"), (1, True, "None")])
def test_attention_module_forward_backward(heads, use_mlp, activate):
    """Attention.forward as a module of its own (reference las_model.py:275-318), value and gradients vs the oracle."""
    from las_pytorch_amd import synth
    from oracle import las_oracle as O
    c = synth.CONFIGS["tiny"]
    shapes = synth.config_shapes("tiny", multi_head=heads, use_mlp=use_mlp)
    sd_np = synth.make_state_dict(shapes, seed=33, scale=0.3)
    las = build_las(c, sd_np, max_label_len=3, multi_head=heads, use_mlp=use_mlp, activate=activate)
    sd = O.to_torch_sd(sd_np, requires_grad=True)
    B, Tp, Hs = 4, 11, c["Hs"]
    g = torch.Generator().manual_seed(9)
    feat_o = torch.randn(B, Tp, Hs, generator=g).requires_grad_(True)
    st_o = torch.randn(B, Hs, generator=g).requires_grad_(True)
    wc = torch.randn(B, Hs, generator=g)
    sc_o, ctx_o = O.attention_forward(st_o, feat_o, sd, "speller.attention.", use_mlp, activate, heads)
    (ctx_o * wc).sum().backward()
    feat = feat_o.detach().clone().cuda().requires_grad_(True)
    st = st_o.detach().clone().cuda().requires_grad_(True)
    sc, ctx = las.speller.attention(st.unsqueeze(1), feat)
    assert isinstance(sc, list) and len(sc) == heads and sc[0].shape == (B, Tp)
    (ctx * wc.cuda()).sum().backward()
    assert_close(ctx.detach().cpu().numpy(), ctx_o.detach().numpy(), "attention context")
    for h in range(heads):
        assert_close(sc[h].cpu().numpy(), sc_o[h].detach().numpy(), f"attention score head {h}", atol=1e-6)
    gs = float(feat_o.grad.norm())
    grad_close(feat.grad.cpu().numpy(), feat_o.grad.numpy(), f"attn_h{heads}_{use_mlp}/dfeat", global_scale=gs)
    grad_close(st.grad.cpu().numpy(), st_o.grad.numpy(), f"attn_h{heads}_{use_mlp}/dstate", global_scale=gs)
    for k, p in las.speller.attention.named_parameters():
        # psi.bias has an identically zero gradient under the shift-invariant softmax when there is no activation: what is
        # compared there is the cancellation noise of a sum of O(1) terms, hence the floor relative to the other gradients
        grad_close(p.grad.cpu().numpy(), sd["speller.attention." + k].grad.numpy(), f"attn_h{heads}_{use_mlp}/grad/{k}", global_scale=gs,
                   global_floor=1e-6)
    _check_err()
