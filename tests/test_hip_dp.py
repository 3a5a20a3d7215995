"""GPU tests of the data-parallel plumbing around the HIP path: the flat-gradient reducer with RCCL (1-rank group: the
collective really runs through librccl), the C-ABI communicator (las_comm_* / las_allreduce_f32), and the caller-side
solver step (zeroing, all-reduce, clip, error-word check) over two consecutive steps."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from hip_util import assert_close, build_las

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _case(B=4, T=64, U=6, cfg="S", seed=4):
    from las_pytorch_amd import synth
    c = synth.CONFIGS[cfg]
    sd_np = synth.make_state_dict(synth.config_shapes(cfg), seed=seed)
    x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=seed)).cuda()
    idx, lens = synth.make_labels(B, U, c["V"], seed=seed)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
    return c, sd_np, x, lab, U


def _grad_once(las, red, x, lab):
    from las_pytorch_amd.solver.solver import label_smoothing_loss_device
    red.zero()
    preds, _ = las(batch_data=x, batch_label=lab, teacher_force_rate=1.0, is_training=True)
    label_smoothing_loss_device(torch.stack(preds, 1), lab, 0.1).backward()
    red.allreduce_mean()
    return red.flat.detach().cpu().numpy().copy()


@pytest.fixture
def one_rank_nccl():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    yield
    if created:
        dist.destroy_process_group()


def test_flat_reducer_over_rccl_one_rank_equals_local_gradient(one_rank_nccl):
    """HIP modules + FlatGradAllReducer(direct=True) with the all-reduce FORCED through a 1-rank nccl (= RCCL) group:
    the averaged flat gradient equals the gradient of a run without torch.distributed."""
    from las_pytorch_amd import dp
    import las_pytorch_amd
    c, sd_np, x, lab, U = _case()
    ref = _grad_once(*(lambda m: (m, dp.FlatGradAllReducer(m, direct=False)))(build_las(c, sd_np, max_label_len=U)), x, lab)
    las = build_las(c, sd_np, max_label_len=U)
    red = dp.FlatGradAllReducer(las, force=True, direct=True)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    got = _grad_once(las, red, x, lab)
    got2 = _grad_once(las, red, x, lab)                      # second step: views still alive, buffer re-zeroed
    scale = float(np.abs(ref).max())
    assert scale > 0
    assert_close(got, ref, "rccl(1 rank) flat gradient", rtol=1e-4, atol=1e-6 * scale)
    assert_close(got2, ref, "rccl(1 rank) flat gradient, step 2", rtol=1e-4, atol=1e-6 * scale)
    torch.cuda.synchronize()
    las_pytorch_amd.check_device_errors()


def test_cabi_communicator_allreduce():
    """las_comm_uid / las_comm_init / las_allreduce_f32 (include/las_hip.h) on a 1-rank communicator: sum and average are
    the identity, the call is stream-ordered, and the reducer can be pointed at it instead of torch.distributed."""
    from las_pytorch_amd import dp
    uid = dp.CabiComm.new_uid()
    assert len(uid) == 128 and any(uid)
    comm = dp.CabiComm(0, 1, uid)
    try:
        t = torch.randn(1 << 20, device="cuda")
        want = t.clone()
        comm.allreduce_(t, average=False)
        comm.allreduce_(t, average=True)
        torch.cuda.synchronize()
        assert torch.equal(t, want)
        c, sd_np, x, lab, U = _case()
        ref = _grad_once(*(lambda m: (m, dp.FlatGradAllReducer(m)))(build_las(c, sd_np, max_label_len=U)), x, lab)
        las = build_las(c, sd_np, max_label_len=U)
        red = dp.FlatGradAllReducer(las, force=True, direct=True, comm=comm)
        got = _grad_once(las, red, x, lab)
        assert_close(got, ref, "C-ABI communicator flat gradient", rtol=1e-4, atol=1e-6 * float(np.abs(ref).max()))
        with pytest.raises(RuntimeError):
            comm.allreduce_(torch.zeros(4, device="cuda", dtype=torch.float64))
    finally:
        comm.destroy()
    with pytest.raises(RuntimeError, match="las_comm_init"):
        comm.allreduce_(torch.zeros(4, device="cuda"))


def test_batch_iterator_two_steps_with_reducer_matches_plain_optimizer_path():
    """solver.batch_iterator with a FlatGradAllReducer attached (zero / all-reduce / clip on the flat buffer, direct
    gradient writes) walks through the same parameters as the plain path (optimizer.zero_grad + clip_grad_norm_) over two
    Adam steps, and still equals the reference's golden solver step after the first."""
    from golden_util import load_case
    from las_pytorch_amd import dp
    from las_pytorch_amd.solver import solver as S
    gold, info, sd_np, x, _, _, oh = load_case("tiny_sat")          # the clip is active in this case
    xg, lab = torch.from_numpy(x).cuda(), torch.from_numpy(oh).cuda()
    params = []
    for use_reducer in (False, True):
        las = build_las(info["cfg"], sd_np, max_label_len=info["free_len"])
        if use_reducer:
            dp.FlatGradAllReducer(las, direct=True)
        opt = torch.optim.Adam(las.parameters(), lr=2e-4)
        for step in range(2):
            np.random.seed(0)
            loss, ler = S.batch_iterator(xg, lab, las, opt, tf_rate=1.0, is_training=True, max_label_len=info["U"], label_smoothing=0.1)
            if step == 0:
                assert abs(float(loss) - gold["step_loss"][0]) < 2e-5
                np.testing.assert_allclose(np.array(ler), gold["step_ler"], rtol=1e-6)
                sums = np.array([p.detach().double().sum().item() for p in las.parameters()])
                np.testing.assert_allclose(sums, gold["step_param_sum"], rtol=1e-4, atol=2e-4)
        params.append(torch.cat([p.detach().reshape(-1) for p in las.parameters()]).cpu().numpy())
    assert_close(params[1], params[0], "parameters after two steps, reducer vs plain", rtol=1e-5, atol=1e-6)
    # validation call (is_training=False): NLL loss + device LER, no parameter change
    before = params[1].copy()
    loss, ler = S.batch_iterator(xg, lab, las, opt, tf_rate=0.0, is_training=False, max_label_len=info["free_len"], label_smoothing=0.1)
    assert np.isfinite(float(loss)) and len(ler) == xg.shape[0]
    after = torch.cat([p.detach().reshape(-1) for p in las.parameters()]).cpu().numpy()
    assert np.array_equal(before, after)


def test_device_error_word_is_reported_and_cleared():
    """A nonzero device error word (what a persistent kernel leaves after a hand-off timeout) raises at the step's
    synchronisation point and through the non-blocking poll, and is cleared so that the next call runs normally."""
    import las_pytorch_amd
    from las_pytorch_amd import _cabi
    c, sd_np, x, lab, U = _case()
    las = build_las(c, sd_np, max_label_len=U)
    with torch.no_grad():
        las(batch_data=x, batch_label=lab, teacher_force_rate=1.0, is_training=True)
    torch.cuda.synchronize()
    las_pytorch_amd.check_device_errors()
    w = _cabi.err_word("cuda")
    w[0] = 0xDEAD0001 - (1 << 32)
    with pytest.raises(RuntimeError, match="hand-off"):
        las_pytorch_amd.check_device_errors()
    assert int(w[0].item()) == 0
    w[0] = 0xDEAD0001 - (1 << 32)
    with pytest.raises(RuntimeError, match="hand-off"):
        for _ in range(4):                                   # the poll sees the snapshot taken behind the previous call
            with torch.no_grad():
                las(batch_data=x, batch_label=lab, teacher_force_rate=1.0, is_training=True)
            torch.cuda.synchronize()
    assert int(w[0].item()) == 0
    with torch.no_grad():
        preds, _ = las(batch_data=x, batch_label=lab, teacher_force_rate=1.0, is_training=True)
    assert torch.isfinite(torch.stack(preds)).all()
    torch.cuda.synchronize()
    las_pytorch_amd.check_device_errors()


def test_training_trajectory_matches_reference():
    """LER parity over a training run: eight solver steps of the HIP modules (fused loss, direct gradient writes into the flat
    buffer, clip, Adam) against the trajectory of the unmodified reference on the same data (tests/golden/S_trajectory.npz):
    loss of every step, letter error rates of every step, the validation call after training, final parameter checksums."""
    from golden_util import load_trajectory_case
    from hip_util import record
    from las_pytorch_amd import dp
    from las_pytorch_amd.solver import solver as S
    import las_pytorch_amd
    g, c, sd_np, x, onehot, U, steps, lr = load_trajectory_case()
    for use_reducer in (True, False):
        las = build_las(c, sd_np, max_label_len=U)
        if use_reducer:
            dp.FlatGradAllReducer(las, direct=True)
        opt = torch.optim.Adam(las.parameters(), lr=lr)
        xg, lab = torch.from_numpy(x).cuda(), torch.from_numpy(onehot).cuda()
        np.random.seed(0)
        worst = 0.0
        for s in range(steps):
            loss, ler = S.batch_iterator(xg, lab, las, opt, tf_rate=1.0, is_training=True, max_label_len=U, label_smoothing=0.1)
            worst = max(worst, abs(float(loss) - g["losses"][s]) / abs(g["losses"][s]))
            assert abs(float(loss) - g["losses"][s]) < 1e-3 * abs(g["losses"][s]), (s, float(loss), g["losses"][s])
            np.testing.assert_allclose(np.array(ler), g["lers"][s], rtol=1e-6)
        vloss, vler = S.batch_iterator(xg, lab, las, opt, tf_rate=0.0, is_training=False, max_label_len=U, label_smoothing=0.1)
        assert abs(float(vloss) - g["val_loss"][0]) < 1e-3 * abs(g["val_loss"][0])
        np.testing.assert_allclose(np.array(vler), g["val_ler"], rtol=1e-6)
        sums = np.array([p.detach().double().sum().item() for p in las.parameters()])
        np.testing.assert_allclose(sums, g["param_sum"], rtol=1e-3, atol=1e-3 * float(np.abs(g["param_abs"]).max()) * 1e-3)
        record(f"S_trajectory/reducer={use_reducer}", max_abs_err=worst, worst_ratio=worst / 1e-3)
    torch.cuda.synchronize()
    las_pytorch_amd.check_device_errors()


def test_bench_line_contract():
    """`python bench.py` (short run, one GPU) prints ONE JSON line with the fields the driver's contract names, the hot-path
    `roofline` block, the GEMM `roofline_mfma` block, the measured GEMM accuracy of both arithmetic modes and the same step timed
    with the GEMMs on the fp32 matrix pipe; the CPU baseline leg is exercised by the full default run only (it takes ~30 s)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-sweep"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert j["metric"] == base["metric"] and j["unit"] == "utt/s" and j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 2
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32"
    assert j["value"] > 0 and abs(j["value"] - 32 / (j["ms_per_step"] * 1e-3)) < 0.01 * j["value"]
    assert "workload" in j["config"] and "gemm_arith" in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    m = j["roofline_mfma"]
    assert m["bound"] == "mfma" and abs(m["frac"] - m["achieved"] / m["peak"]) < 2e-3 and len(m["launches"]) >= 15
    acc = j["gemm_accuracy"]
    assert acc["split_bf16_err_ulp"] <= 1.25 * acc["mfma_f32_err_ulp"] + 0.5, acc
    assert j["gemm_arith_variant"]["ms_per_step"] > 0
    # round 3: the two decode kernels that dominate the step are priced too (HIP events around the single launch), and the line
    # checks its own first-step loss against the reference's
    for key in ("roofline_speller_fwd", "roofline_speller_bwd"):
        q = j[key]
        assert q["bound"] == "hbm" and q["kernel_ms"] > 0 and abs(q["frac"] - q["achieved"] / q["peak"]) < 1e-4
        assert abs(q["us_per_decode_step"] - q["kernel_ms"] * 1e3 / 128) < 1e-2
    assert 3 * j["roofline_speller_fwd"]["algorithmic_bytes"] == j["roofline_speller_bwd"]["algorithmic_bytes"]
    cfgj = j["config"]
    assert abs(cfgj["first_step_loss"] - cfgj["first_step_loss_reference"]) <= 1e-4 * abs(cfgj["first_step_loss_reference"])
    assert j["allreduce_ms"] is None and j["rccl_ranks"] == 0


def test_bench_rccl_path_on_one_gpu_and_strong_scaling_flag():
    """The N > 1 code path end to end on one GPU: LAS_FORCE_DIST=1 initialises RCCL, every step all-reduces the flat gradient +
    error flag, the collective is timed alone (allreduce_ms), and --scaling strong splits --global-batch over the ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LAS_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MASTER_PORT", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-sweep",
                        "--no-mfma", "--no-secondary", "--no-roofline", "--scaling", "strong", "--global-batch", "16"],
                       capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert j["rccl_ranks"] == 1 and j["scaling"] == "strong" and j["config"]["per_gpu_batch"] == 16 and j["config"]["global_batch"] == 16
    assert j["allreduce_ms"] > 0 and j["allreduce_bytes"] > 39e6
    # a launcher / --gpus mismatch fails loudly instead of reporting a 1-GPU number as N-GPU
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, cwd=root, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
