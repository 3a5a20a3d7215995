"""Data-parallel path on CPU: two gloo processes, equal utterance shards, ONE all-reduce of the flat gradient;
the averaged gradient must equal the single-process gradient of the whole batch (the loss is a batch mean,
reference solver/solver.py:43).  The model here is the CPU oracle wrapped in nn.Parameters (the HIP modules need
a GPU); the code under test is las_pytorch_amd.dp."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from las_pytorch_amd import dp, synth
from oracle import las_oracle as O


class OracleLAS(torch.nn.Module):
    def __init__(self, sd_np):
        super().__init__()
        self.keys = list(sd_np.keys())
        self.params = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(v.copy())) for v in sd_np.values()])

    def forward(self, x, onehot, U):
        sd = dict(zip(self.keys, self.params))
        preds, _ = O.las_forward(x, onehot, sd, dict(listener_layers=2, speller_layers=2, max_label_len=U, decode_mode=1),
                                 teacher_force=True)
        loss, _ = O.solver_step_loss(preds, onehot, U, 0.1)
        return loss


def _data(B=4, T=16, U=5):
    c = synth.CONFIGS["tiny"]
    sd_np = synth.make_state_dict(synth.config_shapes("tiny"), seed=23, scale=0.3)
    x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=23))
    idx, lens = synth.make_labels(B, U, c["V"], seed=23)
    onehot = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"]))
    return sd_np, x, onehot, U


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd_np, x, onehot, U = _data()
        model = OracleLAS(sd_np)
        red = dp.FlatGradAllReducer(model)
        sl = dp.shard_batch(x.shape[0], rank, world)
        for it in range(2):                      # second iteration checks that zero() keeps the views alive
            red.zero()
            model(x[sl], onehot[sl], U).backward()
            red.check_views()
            red.allreduce_mean()
        total = red.clip_(1.0)
        dp.sync_coin()
        coin = np.random.random_sample()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), flat=red.flat.numpy(), total=float(total), coin=coin)
    finally:
        dist.destroy_process_group()


def test_two_rank_allreduce_equals_full_batch_gradient(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    np.testing.assert_array_equal(r0["flat"], r1["flat"])          # replicas stay identical
    assert r0["coin"] == r1["coin"]                                  # one shared teacher-forcing coin per step
    # single process, whole batch
    sd_np, x, onehot, U = _data()
    model = OracleLAS(sd_np)
    red = dp.FlatGradAllReducer(model)
    model(x, onehot, U).backward()
    want_total = float(red.clip_(1.0))
    np.testing.assert_allclose(r0["flat"], red.flat.numpy(), rtol=2e-4, atol=2e-7)
    assert abs(r0["total"] - want_total) < 1e-5 * want_total
    # clip semantics == torch.nn.utils.clip_grad_norm_(params, 1) (solver/solver.py:96)
    model2 = OracleLAS(sd_np)
    model2(x, onehot, U).backward()
    t = torch.nn.utils.clip_grad_norm_(model2.parameters(), 1)
    flat2 = torch.cat([p.grad.reshape(-1) for p in model2.parameters()]).numpy()
    np.testing.assert_allclose(red.flat.numpy(), flat2, rtol=1e-5, atol=1e-8)
    assert abs(float(t) - want_total) < 1e-5 * want_total


class OracleLASModel(torch.nn.Module):
    """The oracle behind the reference's LAS call signature (batch_data=, batch_label=, teacher_force_rate=, is_training=),
    drawing the teacher-forcing coin from NumPy's global RNG exactly once per call like Speller.forward (las_model.py:189)."""

    def __init__(self, sd_np, max_label_len):
        super().__init__()
        self.keys = list(sd_np.keys())
        self.params = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(v.copy())) for v in sd_np.values()])
        self.max_label_len = max_label_len
        self.coins = []

    def forward(self, batch_data, batch_label, teacher_force_rate, is_training=True):
        sd = dict(zip(self.keys, self.params))
        coin = bool(np.random.random_sample() < teacher_force_rate)
        self.coins.append(coin)
        return O.las_forward(batch_data, batch_label, sd, dict(listener_layers=2, speller_layers=2, max_label_len=self.max_label_len,
                                                               decode_mode=1), teacher_force=coin, is_training=is_training)


def _solver_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from las_pytorch_amd.solver import solver as S
        sd_np, x, onehot, U = _data()
        model = OracleLASModel(sd_np, max_label_len=U)
        dp.FlatGradAllReducer(model)                         # attaches itself: batch_iterator zeroes / all-reduces / clips it
        opt = torch.optim.Adam(model.parameters(), lr=2e-4)
        sl = dp.shard_batch(x.shape[0], rank, world)
        np.random.seed(100 + rank)                           # different host RNG streams: only sync_coin keeps the coins equal
        losses = []
        for step in range(3):
            loss, ler = S.batch_iterator(x[sl], onehot[sl], model, opt, tf_rate=0.5, is_training=True, max_label_len=U,
                                         label_smoothing=0.1, use_gpu=False)
            losses.append(float(loss))
        # the documented foot-gun: zero_grad() with set_to_none drops the views -> the reducer refuses instead of reducing zeros
        opt.zero_grad()
        refused = False
        try:
            model._las_flat_reducer.allreduce_mean()
        except RuntimeError:
            refused = True
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy()
        np.savez(os.path.join(out_dir, f"s{rank}.npz"), params=flat, coins=np.array(model.coins), losses=np.array(losses), refused=refused)
    finally:
        dist.destroy_process_group()


def test_batch_iterator_two_ranks_keeps_replicas_identical(tmp_path):
    """solver.batch_iterator on two gloo ranks for three Adam steps with tf_rate=0.5: the teacher-forcing coin is shared,
    the flat gradient is averaged before the clip, and the replicas end bit-identical (the zero_grad / stale
    flat buffer divergence)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.start_processes(_solver_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "s0.npz"), np.load(tmp_path / "s1.npz")
    assert r0["coins"].tolist() == r1["coins"].tolist() and len(r0["coins"]) == 3
    np.testing.assert_array_equal(r0["params"], r1["params"])
    assert bool(r0["refused"]) and bool(r1["refused"])
    sd_np, _, _, _ = _data()
    start = np.concatenate([v.reshape(-1) for v in sd_np.values()])
    assert np.abs(r0["params"] - start).max() > 1e-5          # the optimizer really stepped


def _flag_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from las_pytorch_amd.solver import solver as S
        sd_np, x, onehot, U = _data()
        model = OracleLASModel(sd_np, max_label_len=U)
        red = dp.FlatGradAllReducer(model)
        opt = torch.optim.Adam(model.parameters(), lr=2e-4)
        sl = dp.shard_batch(x.shape[0], rank, world)
        outcome = []
        for step in range(3):
            if step == 1 and rank == 1:
                # rank 1's "device" reports a hand-off timeout in step 1: zero() clears the flag at the top of batch_iterator, so the
                # injection goes in through the gradient hook, i.e. after the backward and before the collective — where the GPU
                # path copies its error word into the flag element
                hook = lambda m: red.inject_error(-3.0)
            else:
                hook = None
            try:
                S.batch_iterator(x[sl], onehot[sl], model, opt, tf_rate=1.0, is_training=True, max_label_len=U, label_smoothing=0.1,
                                 use_gpu=False, grad_hook=hook)
                outcome.append(0)
            except RuntimeError as e:
                assert "peer rank" in str(e)
                outcome.append(1)
        np.savez(os.path.join(out_dir, f"f{rank}.npz"), outcome=np.array(outcome), flag=float(red.flag[0]))
    finally:
        dist.destroy_process_group()


def test_error_flag_reaches_every_rank_through_the_gradient_allreduce(tmp_path):
    """The device-error flag rides in the gradient all-reduce (dp.FlatGradAllReducer.flag): a timeout on ONE rank must stop (or, with
    the fused optimizer on a GPU, re-run) the step on EVERY rank, or the replicas diverge.  Two gloo ranks, rank 1 injects a flag in
    step 1: both ranks see a nonzero flag in exactly that step and raise; steps 0 and 2 run normally on both."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.start_processes(_flag_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "f0.npz"), np.load(tmp_path / "f1.npz")
    assert r0["outcome"].tolist() == [0, 1, 0] and r1["outcome"].tolist() == [0, 1, 0]
    assert float(r0["flag"]) == 0.0 and float(r1["flag"]) == 0.0          # the next step's zero() cleared it


def _eight_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from las_pytorch_amd.solver import solver as S
        sd_np, x, onehot, U = _data(B=16)
        # (a) the averaged shard gradient is the full-batch gradient
        model = OracleLAS(sd_np)
        red = dp.FlatGradAllReducer(model)
        sl = dp.shard_batch(x.shape[0], rank, world)
        red.zero()
        model(x[sl], onehot[sl], U).backward()
        red.allreduce_mean()
        grad = red.flat.numpy().copy()
        # (b) three solver steps (shared coin, clip after the all-reduce, Adam) keep the replicas identical; (c) rank 5 injects the
        # device-error flag in step 1: every rank must refuse that step
        model = OracleLASModel(sd_np, max_label_len=U)
        red = dp.FlatGradAllReducer(model)
        opt = torch.optim.Adam(model.parameters(), lr=2e-4)
        np.random.seed(100 + rank)
        outcome = []
        for step in range(4):
            hook = (lambda m: red.inject_error(-3.0)) if (step == 1 and rank == 5) else None
            try:
                S.batch_iterator(x[sl], onehot[sl], model, opt, tf_rate=0.5, is_training=True, max_label_len=U, label_smoothing=0.1,
                                 use_gpu=False, grad_hook=hook)
                outcome.append(0)
            except RuntimeError as e:
                assert "peer rank" in str(e)
                outcome.append(1)
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy()
        np.savez(os.path.join(out_dir, f"e{rank}.npz"), grad=grad, params=flat, coins=np.array(model.coins), outcome=np.array(outcome))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_gradient_replicas_and_error_flag(tmp_path):
    """BASELINE configs[3]'s world size without the hardware: eight gloo ranks, two utterances each.  (a) one all-reduce of the flat
    gradient gives every rank the full-batch gradient; (b) solver.batch_iterator for four steps with tf_rate = 0.5 and different host RNG
    streams: the teacher-forcing coin is shared and the replicas end bit-identical; (c) a device-error flag injected on ONE rank (5) in
    step 1 stops that step on ALL eight.  Replaces the reference's nn.DataParallel (train.py:76-78)."""
    world = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.start_processes(_eight_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    rs = [np.load(tmp_path / f"e{r}.npz") for r in range(world)]
    sd_np, x, onehot, U = _data(B=16)
    model = OracleLAS(sd_np)
    red = dp.FlatGradAllReducer(model)
    model(x, onehot, U).backward()
    for r in rs:
        np.testing.assert_array_equal(r["grad"], rs[0]["grad"])
        np.testing.assert_array_equal(r["params"], rs[0]["params"])
        assert r["coins"].tolist() == rs[0]["coins"].tolist()
        assert r["outcome"].tolist() == [0, 1, 0, 0]
    np.testing.assert_allclose(rs[0]["grad"], red.flat.numpy(), rtol=5e-4, atol=5e-7)
    start = np.concatenate([v.reshape(-1) for v in sd_np.values()])
    assert np.abs(rs[0]["params"] - start).max() > 1e-5


def test_oracle_training_trajectory_matches_reference():
    """Eight consecutive solver steps (this build's batch_iterator: loss, clip, Adam) on the CPU oracle walk the trajectory the
    unmodified reference walked (tests/golden/S_trajectory.npz): per-step loss, per-utterance LER, validation call."""
    from golden_util import load_trajectory_case
    from las_pytorch_amd.solver import solver as S
    g, c, sd_np, x, onehot, U, steps, lr = load_trajectory_case()
    model = OracleLASModel(sd_np, max_label_len=U)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    xt, lab = torch.from_numpy(x), torch.from_numpy(onehot)
    np.random.seed(0)
    for s in range(steps):
        loss, ler = S.batch_iterator(xt, lab, model, opt, tf_rate=1.0, is_training=True, max_label_len=U, label_smoothing=0.1, use_gpu=False)
        assert abs(float(loss) - g["losses"][s]) < 2e-4 * abs(g["losses"][s]), (s, float(loss), g["losses"][s])
        np.testing.assert_allclose(np.array(ler), g["lers"][s], rtol=1e-6)
    vloss, vler = S.batch_iterator(xt, lab, model, opt, tf_rate=0.0, is_training=False, max_label_len=U, label_smoothing=0.1, use_gpu=False)
    assert abs(float(vloss) - g["val_loss"][0]) < 5e-4 * abs(g["val_loss"][0])
    np.testing.assert_allclose(np.array(vler), g["val_ler"], rtol=1e-6)


def test_shard_batch():
    assert dp.shard_batch(32, 3, 8) == slice(12, 16)
    try:
        dp.shard_batch(10, 0, 4)
        raise SystemExit("expected an assertion")
    except AssertionError:
        pass
