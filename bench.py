#!/usr/bin/env python3
"""Headline benchmark of the LAS hot path on MI355X (BASELINE.json: utterances/sec, B=32, T=800, 80-mel, fwd+bwd).

    python bench.py --gpus N --steps K --warmup W [--workload P_train|S_train|P_fwd|S_fwd|P_long] [--batch 32]

A "step" is one pass of the hot path over one synthetic batch per GPU: Listener (pyramidal BiLSTM) + Speller
(teacher-forced decode, U=128) forward, the reference's label-smoothing loss (solver/solver.py:33-45), backward
through every HIP kernel, ONE flat gradient all-reduce (N>1), global-norm clip at 1.0 (solver.py:96) and an
Adam step (lr 2e-4, train.py:82) — i.e. everything solver.batch_iterator does per batch except the host-side
LER bookkeeping.  Inputs are resident in HBM before the timed region.  One process per GPU over RCCL / xGMI: under
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`` every rank runs this file; started bare
with ``--gpus N`` (N > 1) it launches those N ranks itself as child processes before touching the GPU and relays rank 0's
JSON line.

Rank 0 prints ONE JSON line with ``roofline`` (dominant kernel: the layer-0 pBLSTM forward recurrence, timed with
HIP events on the launch stream), ``roofline_mfma`` (the step's MFMA GEMMs against the 157.3 TF fp32-MFMA peak),
``sweep`` (the recurrence at larger per-GPU batches) and ``cpu_baseline`` (oracle/cpu_baseline.py — the
nn.LSTM-module port of the reference's CPU path — timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC for RCCL; must be set before HIP initialises

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (config, T, U, train)
    "P_train": ("P", 800, 128, True),     # BASELINE configs[2]/[3]: paper-size, fwd+bwd  (the metric's "fwd+bwd")
    "S_train": ("S", 800, 128, True),
    "P_fwd": ("P", 800, 128, False),
    "S_fwd": ("S", 800, 128, False),       # BASELINE configs[1]
    "P_long": ("P", 3000, 128, True),      # BASELINE configs[4] (use --batch 8)
    "S_long": ("S", 3000, 128, True),      # ... with the small model: T' = 750, keys 192 KB (do not fit the LDS: per-step decode forward)
    "Y_train": ("Y", 800, 128, True),      # config/librispeech-config.yaml sizes (512x3 / 1024x2, 40-mel; yaml batch 16)
}
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3  # dense fp32-input MFMA peak (same guide)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (same guide); the split-operand GEMM spends six bf16 MFMA products per fp32 product


def gemm_arith():
    """(mode, description, MFMA peak in fp32-equivalent TFLOP/s) of the GEMM arithmetic the library is running in."""
    from las_pytorch_amd import _cabi
    mode = _cabi.lib().las_gemm_get_arith()
    if mode == 1:
        return mode, ("fp32 operands split exactly into three bf16 terms in registers, six partial products on "
                      "v_mfma_f32_32x32x16_bf16, fp32 accumulation (dropped terms < 2^-26 |a*b|: error vs float64 not above the fp32 MFMA's, "
                      "see gemm_accuracy)"), MFMA_BF16_PEAK_TF / 6.0
    return mode, "fp32 operands on v_mfma_f32_32x32x2_f32", MFMA_F32_PEAK_TF


def gemm_accuracy():
    """One GEMM of the step (L2 input projection, 3200x1024x1024) in both arithmetic modes against float64, measured in this run:
    max |C - AB| / (|A||B|) in units of 2^-24 (the scale of an fp32 dot-product error bound)."""
    from las_pytorch_amd import _cabi
    L = _cabi.lib()
    M, N, K = 3200, 1024, 1024
    g = torch.Generator(device="cuda").manual_seed(17)
    A = torch.randn(M, K, device="cuda", generator=g); Bm = torch.randn(N, K, device="cuda", generator=g)
    ref = A.double() @ Bm.double().t(); mag = A.double().abs() @ Bm.double().abs().t()
    old, out = L.las_gemm_get_arith(), {}
    for mode, key in ((0, "mfma_f32_err_ulp"), (1, "split_bf16_err_ulp")):
        L.las_gemm_set_arith(mode)
        C = torch.empty(M, N, device="cuda")
        _cabi.check(L.las_gemm_f32(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), None, None, M, N, K, K, K, N, 1, 1, 1, 0, 0, 0, 1, 0, 0,
                                   _cabi.stream_ptr()))
        out[key] = round(float(((C.double() - ref).abs() / mag).max()) / 2.0 ** -24, 3)
    L.las_gemm_set_arith(old)
    out["shape"] = f"{M}x{N}x{K}, N(0,1) operands; torch CPU/GPU float64 reference"
    return out


def _sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def csrc_sha16():
    """One hash over everything the kernels are built from: every file of las_pytorch_amd/csrc (sources, headers, the Makefile
    with its CXXFLAGS) and include/las_hip.h.  The committed PMC measurements carry it; a stale one is refused, not quoted."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "las_pytorch_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "las_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(tag, shape):
    """HBM bytes per launch from a committed counter run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected as
    guides/MI355X_MICROARCH.md prescribes; tools/make_pmc_profiles.py writes the JSON): the newest profiles/*<tag>*.json whose shape and
    source hash match this build.  Returns (bytes | None, source string | None)."""
    want = csrc_sha16()
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not (name.endswith(".json") and tag in name):
            continue
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
        if j.get("shape") != shape:
            continue
        if j.get("csrc_sha16") == want:
            return int(j["traffic_bytes"]), "profiles/" + name
        return None, f"profiles/{name} is stale (kernel sources / flags changed since it was measured): not quoted"
    return None, None


def live_pmc_traffic(B, T_l, H, timeout=150):
    """The dominant kernel's HBM traffic RE-MEASURED in this run: two child processes `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --
    python3 tools/ubench_rec.py` (separate passes, counters only, as guides/MI355X_MICROARCH.md prescribes; a counter run cannot nest inside
    this process, and a child is the allowed way to start another GPU program), summarised by tools/make_pmc_profiles.py into
    profiles/<round>_pmc_rec_fwd_live.json.  Returns (bytes, source) or (None, reason): the caller then quotes the committed file."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="las_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", B=str(B), T=str(T_l), H=str(H))
    dbs = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--pmc", counter, "-d", out, "-o", "x", "--", sys.executable, os.path.join(ROOT, "tools", "ubench_rec.py")],
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
            found = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not found:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            dbs[counter] = found[0]
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        os.environ["LAS_PROFILE_SUFFIX"] = "_live"
        import importlib
        mk = importlib.import_module("make_pmc_profiles")
        mk.SUFFIX = "_live"
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            mk.rec(dbs["FETCH_SIZE"], dbs["WRITE_SIZE"], B, T_l, H)
        name = f"{mk.ROUND}_pmc_rec_fwd_live.json"
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
        return int(j["traffic_bytes"]), f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two child passes over tools/ubench_rec.py in this run) -> profiles/{name}"
    except Exception as e:      # (an optional re-measurement must never cost the driver line)
        return None, f"live counter run failed: {type(e).__name__}: {e}"[:200]
    finally:
        os.environ.pop("LAS_PROFILE_SUFFIX", None)
        shutil.rmtree(tmp, ignore_errors=True)


def build_model(cfg_name, U, device):
    from las_pytorch_amd import LAS, Listener, Speller, synth
    c = synth.CONFIGS[cfg_name]
    sd_np = synth.make_state_dict(synth.config_shapes(cfg_name), seed=17)
    listener = Listener(input_feature_dim=c["F"], hidden_size=c["H"], num_layers=c["L"], rnn_unit="LSTM", use_gpu=True)
    speller = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U,
                      use_mlp_in_attention=True, mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu",
                      listener_hidden_size=c["H"], multi_head=1, decode_mode=1, use_gpu=True)
    las = LAS(listener, speller)
    las.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    return las.to(device), c, sd_np


def batch128_block(las, c, device, T, U, reducer, opt, B=128, iters=5):
    """The headline model at 128 utterances per GPU (four times the headline batch): what a per-GPU batch beyond the decode kernels' 32
    utterances per launch costs — the decode loop runs as four serial 32-utterance slices (one launch each way per slice), the recurrences
    and GEMMs take the whole batch.  VERDICT round 5, item 4: the weight-stationary throughput decode for B >= 64 is NOT built; this block
    keeps the cost of that decision in the driver line.  Never the metric."""
    from las_pytorch_amd import _cabi, synth
    x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=17)).to(device)
    idx, lens = synth.make_labels(B, U, c["V"], seed=17)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).to(device)
    step = make_train_step(las, x, lab, reducer, opt)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    import las_pytorch_amd
    las_pytorch_amd.check_device_errors()
    slice_b = 32
    return {"workload": f"P_train at (B={B},T={T}) per GPU, teacher-forced U={U}, the same full training step",
            "value": round(B / dt, 1), "unit": "utt/s", "ms_per_step": round(dt * 1e3, 3), "steps": iters,
            "decode_slices": (B + slice_b - 1) // slice_b,
            "decode_paths": [_cabi.last_path(_cabi.PATH_DECODE_FWD), _cabi.last_path(_cabi.PATH_DECODE_BWD)],
            "rec_paths": [_cabi.last_path(_cabi.PATH_REC_FWD), _cabi.last_path(_cabi.PATH_REC_BWD)],
            "note": "decode loop = serial 32-utterance slices of the one-launch kernels (no throughput decode for B >= 64 exists: declined, see DESIGN.md); "
                    "ratio to the headline = value / headline value"}


def multi_head_block(device, T, U, heads=2, B=16, iters=5):
    """The same training step (fwd + label-smoothing loss + bwd; no optimizer) with TWO attention heads (reference las_model.py:298-314;
    multi_head is 1 in the shipped YAMLs) at 16 utterances — what one launch of the multi-head decode kernels holds — on the one-launch
    kernels and, for comparison, on the per-step launch chains (option SPELLER_PRE_MH = 0).  Never the metric; reported beside it."""
    from las_pytorch_amd import LAS, Listener, Speller, _cabi, synth
    from las_pytorch_amd.solver.solver import label_smoothing_loss_backward_device, stack_steps
    c = synth.CONFIGS["P"]
    sd_np = synth.make_state_dict(synth.config_shapes("P", multi_head=heads), seed=23, scale=0.1)
    listener = Listener(input_feature_dim=c["F"], hidden_size=c["H"], num_layers=c["L"], rnn_unit="LSTM", use_gpu=True)
    speller = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U, use_mlp_in_attention=True,
                      mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu", listener_hidden_size=c["H"], multi_head=heads, decode_mode=1, use_gpu=True)
    las = LAS(listener, speller)
    las.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    las = las.to(device)
    x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=23)).to(device)
    idx, lens = synth.make_labels(B, U, c["V"], seed=23)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).to(device)

    def step():
        for p in las.parameters():
            p.grad = None
        preds, _ = las(batch_data=x, batch_label=lab, teacher_force_rate=1.0, is_training=True)
        label_smoothing_loss_backward_device(stack_steps(preds), lab, 0.1)

    out = {"workload": f"P sizes with multi_head = {heads}, (B={B},T={T}), teacher-forced U={U}: fwd + loss + bwd"}
    old = _cabi.get_option("SPELLER_PRE_MH")
    try:
        for key, mh in (("one_launch", 1), ("per_step", 0)):
            _cabi.set_option("SPELLER_PRE_MH", mh)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                step()
            torch.cuda.synchronize()
            out[key] = {"ms_per_step": round((time.perf_counter() - t0) / iters * 1e3, 3),
                        "decode_paths": [_cabi.last_path(_cabi.PATH_DECODE_FWD), _cabi.last_path(_cabi.PATH_DECODE_BWD)]}
    finally:
        _cabi.set_option("SPELLER_PRE_MH", old)
    _cabi.check_device_errors()
    return out


def roofline_rec_fwd(c, B, T, iters=20, with_traffic=True):
    """Times the dominant kernel (layer-0 forward recurrence, rec_fwd_fast<H>) alone with HIP events on the stream it
    is launched on, on real pre-activations, and prices it against the HBM roofline with the ALGORITHMIC bytes of the
    pBLSTM layer it belongs to: B*4*T_l*(D_l+2H) + weights."""
    from las_pytorch_amd import _cabi, synth
    L = _cabi.lib()
    H, F = c["H"], c["F"]
    T_l, D_l = T // 2, 2 * F
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.from_numpy(synth.make_inputs(B, T, F, seed=17)).cuda()
    bound = 1.0 / np.sqrt(H)
    ws = []
    for _ in range(2):
        ws += [(torch.rand(4 * H, D_l, generator=g) * 2 - 1) * bound, (torch.rand(4 * H, H, generator=g) * 2 - 1) * bound,
               (torch.rand(4 * H, generator=g) * 2 - 1) * bound, (torch.rand(4 * H, generator=g) * 2 - 1) * bound]
    ws = [w.cuda() for w in ws]
    flags = _cabi.FLAG_STASH
    out = torch.empty(B, T_l, 2 * H, device="cuda")
    err = _cabi.err_word("cuda")
    n_g = 2 * B * T_l * 4 * H
    gates = torch.empty(n_g, device="cuda")
    pre = torch.empty(n_g, device="cuda")
    # pre-activations: X W_ih^T + b_ih + b_hh per direction via the exported GEMM
    xv = x.view(B * T_l, D_l)
    for d in range(2):
        _cabi.check(L.las_gemm_f32(_cabi.ptr(xv), _cabi.ptr(ws[4 * d]), pre.data_ptr() + d * (n_g // 2) * 4, _cabi.ptr(ws[4 * d + 2]),
                                   _cabi.ptr(ws[4 * d + 3]), B * T_l, 4 * H, D_l, D_l, D_l, 4 * H, 1, 1, 1, 0, 0, 0, 1, 0, 0,
                                   _cabi.stream_ptr()))
    cbuf = torch.empty(2 * B * T_l * H, device="cuda")
    hprev = torch.empty(2 * B * T_l * H, device="cuda")
    xbuf = torch.empty(L.las_rec_xbuf_bytes(B, H) // 4 + 4, device="cuda")
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for it in range(iters + 3):
        gates.copy_(pre)
        if it >= 3:
            evs[it - 3][0].record()
        _cabi.check(L.las_pblstm_rec_fwd(_cabi.ptr(gates), _cabi.ptr(ws[1]), _cabi.ptr(ws[5]), _cabi.ptr(out), _cabi.ptr(cbuf),
                                         _cabi.ptr(hprev), B, T_l, H, _cabi.ptr(xbuf), _cabi.ptr(err), flags, _cabi.stream_ptr()))
        if it >= 3:
            evs[it - 3][1].record()
    torch.cuda.synchronize()
    _cabi.check_device_errors()
    ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    w_bytes = 4 * 2 * (4 * H * D_l + 4 * H * H + 8 * H)
    alg_bytes = B * 4 * T_l * (D_l + 2 * H) + w_bytes
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes): measured
    # offline on this kernel and shape and committed under profiles/ (a counter run cannot nest inside this process)
    # The JSON names the kernel source it was measured on (sha256 of pblstm_rec.hip): a stale file is refused, not quoted.
    traffic, traffic_src = pmc_traffic("pmc_rec_fwd", dict(B=B, T_l=T_l, H=H)) if with_traffic else (None, None)
    return dict(bound="hbm", kernel=f"rec_fwd_fast<{H}> layer0 (B={B},T_l={T_l})", achieved=round(achieved, 2), peak=HBM_PEAK_GBS,
                unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_src,
                kernel_ms=round(ms, 4), us_per_step=round(ms * 1e3 / T_l, 3), algorithmic_bytes=alg_bytes)


def step_gemm_launches(c, B, T, U):
    """The MFMA GEMM LAUNCHES of one training step exactly as las_capi.hip issues them: each entry is
    (name, kind, problems) with kind "single" (las_gemm_f32: batch / bias-capable path) or "group" (las_gemm_f32_group: one
    launch for several weight-gradient GEMMs, or one dual-K-source GEMM).  problem = (M, N, K, a_kc, b_kc, batch, splitk, K1)."""
    H, F, L, Hs, V, M = c["H"], c["F"], c["L"], c["Hs"], c["V"], c["M"]
    Tp, UB = T >> L, U * B
    out = []
    for l in range(L):
        BT, D = B * (T >> (l + 1)), (2 * F if l == 0 else 4 * H)
        out.append((f"L{l} X W_ih^T (2 dirs batched)", "single", [(BT, 4 * H, D, 1, 1, 2, 1, 0)]))
    out.append(("keys psi", "single", [(B * Tp, M, 2 * H, 1, 1, 1, max(1, min(2 * H // 64, 256 // max(1, -(-B * Tp // 128)))), 0)]))
    # pre-multiplied-context decode kernels (speller_persist*.hip: 4 / 8 / 16 attention workgroups per utterance) and their GEMMs
    pre = B <= 32 and Hs in (256, 512) and any(1024 * ws // Hs <= 32 and Tp <= 14 * (1024 * ws // Hs) and Hs // 4 + ws * B <= 256
                                               for ws in (4, 8, 16))
    if pre:
        out.append(("P = feat W_ctx^T (decode chain pre-product)", "single", [(B * Tp, 4 * Hs, Hs, 1, 1, 1, 1, 0)]))
        out.append(("ctx = att feat (per utterance)", "single", [(U, Hs, Tp, 1, 0, B, 1, 0)]))
    out.append(("logits [h|ctx] W_c^T (dual K)", "group", [(UB, V, 2 * Hs, 1, 1, 1, 0, Hs)]))
    out.append(("dcat = dz W_c", "single", [(UB, 2 * Hs, V, 1, 0, 1, 1, 0)]))
    if pre:
        out.append(("e0 = dcat_ctx feat^T (per utterance)", "single", [(U, Tp, Hs, 1, 1, B, 1, 0)]))
        out.append(("dctx = dG0 W_ctx", "single", [(UB, Hs, 4 * Hs, 1, 0, 1, 1, 0)]))
    out.append(("dfeat (per utterance)", "single", [(Tp, Hs, U, 0, 0, B, 1, 0)]))
    out.append(("dK (per utterance)", "single", [(Tp, M, U, 0, 0, B, 1, 0)]))
    out.append(("dW_psi", "single", [(M, Hs, B * Tp, 0, 0, 1, 0, 0)]))
    out.append(("dfeat += dK W_psi", "single", [(B * Tp, Hs, M, 1, 0, 1, 1, 0)]))
    out.append(("speller dW group (8 GEMMs, one launch)", "group",
                [(V, Hs, UB, 0, 0, 1, 0, 0), (V, Hs, UB, 0, 0, 1, 0, 0), (M, Hs, UB, 0, 0, 1, 0, 0), (4 * Hs, V, UB, 0, 0, 1, 0, 0),
                 (4 * Hs, Hs, UB, 0, 0, 1, 0, 0), (4 * Hs, Hs, UB - B, 0, 0, 1, 0, 0), (4 * Hs, Hs, UB, 0, 0, 1, 0, 0),
                 (4 * Hs, Hs, UB - B, 0, 0, 1, 0, 0)]))
    for l in reversed(range(L)):
        BT, D = B * (T >> (l + 1)), (2 * F if l == 0 else 4 * H)
        out.append((f"L{l} dW group (4 GEMMs, one launch)", "group",
                    [(4 * H, D, BT, 0, 0, 1, 0, 0), (4 * H, H, BT, 0, 0, 1, 0, 0), (4 * H, D, BT, 0, 0, 1, 0, 0), (4 * H, H, BT, 0, 0, 1, 0, 0)]))
        if l > 0:
            out.append((f"L{l} dX (both directions, dual K)", "group", [(BT, D, 8 * H, 1, 0, 1, 0, 4 * H)]))
    return out


def roofline_mfma(c, B, T, U, reps=10):
    """Every MFMA GEMM launch of one training step, issued through the same entry points and in the same grouping as
    las_capi.hip does, HIP-event timed on the launch stream: flops / time against the dense fp32-MFMA peak.  (2*M*N*K flops
    are ALGORITHMIC: stream-K atomics, bias and activation epilogues are not counted.)"""
    import ctypes
    from las_pytorch_amd import _cabi
    L = _cabi.lib()
    tot_us, tot_fl, rows = 0.0, 0.0, []
    for name, kind, probs in step_gemm_launches(c, B, T, U):
        bufs, descs = [], (_cabi.GemmDescC * len(probs))()
        for i, (M, N, K, a_kc, b_kc, batch, splitk, K1) in enumerate(probs):
            A = torch.randn(batch * M * K, device="cuda"); Bm = torch.randn(batch * N * K, device="cuda")
            C = torch.zeros(batch * M * N, device="cuda")
            bufs.append((A, Bm, C))
            Ka = K1 if K1 else K                          # dual K: the two sources are K1 and K-K1 wide with the same leading dimension
            lda = Ka if a_kc else M; ldb = Ka if b_kc else N
            d = descs[i]
            d.A, d.B, d.C = A.data_ptr(), Bm.data_ptr(), C.data_ptr()
            d.A2 = A.data_ptr() + 4 * (M * K1 if a_kc else K1 * M) if K1 else None
            d.B2 = Bm.data_ptr() + 4 * (N * K1 if b_kc else K1 * N) if K1 else None
            d.M, d.N, d.K, d.K1, d.lda, d.ldb, d.ldc = M, N, K, K1, lda, ldb, N
            d.a_kc, d.b_kc, d.accumulate, d.c_zeroed = a_kc, b_kc, 0, int(kind == "group" and not K1)

        def call():
            if kind == "group":
                _cabi.check(L.las_gemm_f32_group(descs, len(probs), _cabi.stream_ptr()))
            else:
                M, N, K, a_kc, b_kc, batch, splitk, _ = probs[0]
                A, Bm, C = bufs[0]
                _cabi.check(L.las_gemm_f32(A.data_ptr(), Bm.data_ptr(), C.data_ptr(), None, None, M, N, K, K if a_kc else M, K if b_kc else N, N,
                                           a_kc, b_kc, batch, M * K, N * K, M * N, splitk, 0, 0, _cabi.stream_ptr()))
        for _ in range(2):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        fl = sum(2.0 * p[5] * p[0] * p[1] * p[2] for p in probs)
        tot_us += us; tot_fl += fl
        rows.append({"launch": name, "gemms": len(probs), "us": round(us, 1), "tflops": round(fl / us / 1e6, 1)})
        del bufs
    tf = tot_fl / tot_us / 1e6
    mode, desc, peak = gemm_arith()
    return dict(bound="mfma", kernel="gemm_f32_kernel / gemm_group_kernel: every MFMA GEMM launch of one training step; " + desc,
                achieved=round(tf, 1), peak=round(peak, 1), unit="TFLOP/s (fp32 flops of the GEMMs: 2MNK)", frac=round(tf / peak, 4),
                peak_note=("dense bf16 MFMA peak 2500 TF / 6 partial products per fp32 product" if mode == 1 else "dense fp32-input MFMA peak"),
                frac_of_fp32_mfma_peak=round(tf / MFMA_F32_PEAK_TF, 4),
                flops_per_step=int(tot_fl), gemm_ms_per_step=round(tot_us / 1e3, 3), launches=rows)


def sweep_rec(c, T, batches=(32, 128, 512, 2048)):
    """The layer-0 forward recurrence alone at growing per-GPU batch (SURVEY.md section 8d asks for the batch sweep next to
    the B=32 headline: at B=32 the kernel is a latency-bound 400-step chain, larger batches amortise the hand-off)."""
    out = []
    for Bs in batches:
        r = roofline_rec_fwd(c, Bs, T, iters=5, with_traffic=False)
        out.append({"B": Bs, "kernel_ms": r["kernel_ms"], "achieved_GBs": r["achieved"], "frac": r["frac"],
                    "us_per_step": r["us_per_step"], "us_per_utterance_step": round(r["us_per_step"] / Bs, 4)})
        torch.cuda.empty_cache()
    return out


def speller_param_bytes(c):
    Hs, V, M = c["Hs"], c["V"], c["M"]
    n = 4 * Hs * (V + Hs) + 4 * Hs * Hs + 2 * 4 * Hs * Hs + 4 * 4 * Hs + 2 * (M * Hs + M) + V * 2 * Hs + V
    return 4 * n


def roofline_speller(step, c, B, T, U, iters=10):
    """The two kernels that dominate the step by time — the one-launch decode loop forward and backward — each timed ALONE with
    HIP events on its launch stream (library option TIME_KERNELS: events recorded around that one launch inside liblas_hip.so,
    read back with las_debug_kernel_ms) while the real training step runs.  Algorithmic bytes per SURVEY.md section 8d: per utterance
    feat 4*T'*2H + log-probs 4*U*V + attention 4*U*T', plus the Speller's weights once per launch; backward = 3x (every operand read
    again, gradients written).  Both are chains of U dependent steps: the fraction is the honest distance from the HBM roofline,
    us_per_decode_step is what the design actually works on."""
    import ctypes
    from las_pytorch_amd import _cabi
    Tp = T >> c["L"]
    per_utt = 4 * Tp * c["Hs"] + 4 * U * c["V"] + 4 * U * Tp
    alg_fwd = B * per_utt + speller_param_bytes(c)
    _cabi.set_option("TIME_KERNELS", 1)
    ms = {0: [], 1: []}
    try:
        for _ in range(iters):
            step()
            for which in (0, 1):
                v = ctypes.c_float(0.0)
                if _cabi.lib().las_debug_kernel_ms(which, ctypes.byref(v)) == 0:
                    ms[which].append(float(v.value))
    finally:
        _cabi.set_option("TIME_KERNELS", 0)
    out = {}
    for which, key, alg, tag in ((0, "roofline_speller_fwd", alg_fwd, "pmc_speller_fwd"), (1, "roofline_speller_bwd", 3 * alg_fwd, "pmc_speller_bwd")):
        if not ms[which]:
            continue           # the per-step kernels ran (shape outside the one-launch kernels)
        t = float(np.mean(ms[which]))
        ach = alg / (t * 1e-3) / 1e9
        traffic, src = pmc_traffic(tag, dict(B=B, Tp=Tp, U=U, Hs=c["Hs"]))
        out[key] = dict(bound="hbm", kernel=("speller_persist_fwd_pre_kernel" if which == 0 else "speller_persist_bwd_pre_kernel") + f"<{c['Hs']}> (B={B},T'={Tp},U={U})",
                        achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5), traffic=traffic,
                        traffic_source=src, kernel_ms=round(t, 4), us_per_decode_step=round(t * 1e3 / U, 3), algorithmic_bytes=int(alg))
    return out


def greedy_decode_block(las, x, U, iters=10):
    """The validation path (reference train.py:149-169: free-running decode_mode 1 for max_label_len steps, no backward): the Speller's greedy
    decode of the headline batch alone, under torch.no_grad(), whole ``Speller.forward`` calls including their pre-products."""
    from las_pytorch_amd import _cabi
    with torch.no_grad():
        feat = las.listener(x)
        for _ in range(3):
            las.speller(feat, ground_truth=None, teacher_force_rate=0.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            las.speller(feat, ground_truth=None, teacher_force_rate=0.0)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return {"workload": f"greedy (decode_mode 1) decode of the headline batch, {U} steps, no backward", "ms_per_call": round(ms, 3),
            "us_per_decode_step": round(ms * 1e3 / U, 2), "kernel_path": _cabi.last_path(_cabi.PATH_DECODE_FWD)}


def secondary_long(device, U, B=8, T=3000, steps=10, warmup=3, with_roofline=True):
    """BASELINE.json configs[4] beside the headline: the paper-size model on a batch of 8 thirty-second utterances (T = 3000 frames,
    T' = 375), the same full training step.  The Listener recurrences are 2 625 dependent steps each way on 64 of the 256 CUs
    (8 utterances x 2 directions x 4 CUs: a longer sequence offers no more parallelism), the decode loop keeps P[b] = feat[b] W_ctx^T
    resident in 8 workgroups per utterance (T' = 375 exceeds what 4 hold).  Returns the block with its own roofline entries: the
    layer-0 recurrence at T_l = 1500 and the two decode kernels at T' = 375, event-timed like the headline's."""
    from las_pytorch_amd import dp, synth
    from las_pytorch_amd.optim import FusedClipAdam
    las, c, _ = build_model("P", U, device)
    x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=17)).to(device)
    idx, lens = synth.make_labels(B, U, c["V"], seed=17)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).to(device)
    reducer = dp.FlatGradAllReducer(las, direct=True)
    opt = FusedClipAdam(reducer, lr=2e-4)
    step = make_train_step(las, x, lab, reducer, opt)
    # parity hook, as for the headline: the loss at the initial weights equals what the UNMODIFIED reference computes on the same inputs
    # (tests/golden/P_B8_T3000_U128.npz: same seeds, same shapes, U = 128)
    first_loss = float(step().item())
    ref_loss = None
    gpath = os.path.join(ROOT, "tests", "golden", f"P_B{B}_T{T}_U{U}.npz")
    if os.path.exists(gpath):
        ref_loss = float(np.load(gpath)["loss_ls"][0])
        assert abs(first_loss - ref_loss) <= 1e-4 * abs(ref_loss), f"P_long first-step loss {first_loss} != reference {ref_loss}"
    for _ in range(max(0, warmup - 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    import las_pytorch_amd
    las_pytorch_amd.check_device_errors()
    assert np.isfinite(float(last.item()))
    out = {"workload": f"P_long (BASELINE configs[4]): Listener 256x3 / Speller 512x2, (B={B},T={T},F=80), teacher-forced U={U}, the same full "
                       "training step (fwd + loss + bwd + clip + Adam)",
           "value": round(B / dt, 1), "unit": "utt/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
           "first_step_loss": first_loss, "first_step_loss_reference": ref_loss,
           "serial_recurrence_steps_each_way": sum(T >> (l + 1) for l in range(c["L"])), "recurrence_cus": 2 * B * (c["H"] * c["H"] // 16384),
           "residency": "P keys 96 KB in LDS + P[b] column slices in registers: 16 attention workgroups per utterance (persist_pre both ways); greedy decode at this T' "
                        "keeps Q^T (57 KB) in LDS and splits the keys by frames over those 16 workgroups (persist_pre_greedy)"}
    if with_roofline:
        r = roofline_rec_fwd(c, B, T, iters=5, with_traffic=True)
        out["roofline"] = r
        out.update(roofline_speller(step, c, B, T, U, iters=5))
    del las, reducer, opt, step
    torch.cuda.empty_cache()
    try:      # the small model at the same T (T' = 750: 192 KB of keys per utterance exceed one workgroup's LDS; split by frames over 16 workgroups)
        las_s, c_s, _ = build_model("S", U, device)
        xs = torch.from_numpy(synth.make_inputs(B, T, c_s["F"], seed=17)).to(device)
        red_s = dp.FlatGradAllReducer(las_s, direct=True)
        opt_s = FusedClipAdam(red_s, lr=2e-4)
        step_s = make_train_step(las_s, xs, lab, red_s, opt_s)
        for _ in range(warmup):
            step_s()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_s()
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t0) / steps
        from las_pytorch_amd import _cabi
        las_pytorch_amd.check_device_errors()
        out["small_model"] = {"workload": f"S_long: Listener 128x2 / Speller 256x2, (B={B},T={T}), the same training step", "value": round(B / dts, 1), "unit": "utt/s",
                              "ms_per_step": round(dts * 1e3, 3), "decode_paths": [_cabi.last_path(_cabi.PATH_DECODE_FWD), _cabi.last_path(_cabi.PATH_DECODE_BWD)]}
        del las_s, red_s, opt_s, step_s
        torch.cuda.empty_cache()
    except Exception as e:      # (an optional side figure must never cost the driver line)
        out["small_model"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def allreduce_alone_ms(reducer, iters=10):
    """The step's ONE collective (flat fp32 gradient + the error flag) timed alone on the step's stream; meaningful for N > 1
    ranks, and with LAS_FORCE_DIST=1 it shows the 1-rank RCCL launch floor."""
    if not reducer._collective():
        return None
    for _ in range(2):
        reducer.allreduce_mean()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        reducer.allreduce_mean()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def make_train_step(las, x, lab, reducer, opt, graph=False, tf_rate=1.0):
    """One training step of the benchmark as a callable returning the loss tensor: zero the flat gradient, forward (teacher
    forced), fused label-smoothing loss + gradient, backward through every HIP kernel, ONE gradient all-reduce (N > 1), global-norm
    clip at 1.0 + Adam as the fused launch pair.  graph=True captures zero + forward + loss + backward into one HIP graph
    (all-reduce / clip / Adam stay eager behind the replay)."""
    from las_pytorch_amd.solver.solver import label_smoothing_loss_backward_device, label_smoothing_loss_device, stack_steps
    seeded = os.environ.get("LAS_BENCH_LOSS_BACKWARD", "seed") == "seed"      # "autograd": loss.backward() from the scalar (A/B)

    def fwd_bwd():
        reducer.zero()
        preds, _ = las(batch_data=x, batch_label=lab, teacher_force_rate=tf_rate, is_training=True)
        # fused loss + gradient kernel, no copies; the gradient seeds the backward directly (what solver.batch_iterator does)
        with reducer.deferring():      # as in solver.batch_iterator; a no-op unless the reducer was built with defer_dw / LAS_DEFER_DW=1 (measured not to pay)
            if seeded:
                return label_smoothing_loss_backward_device(stack_steps(preds), lab, 0.1)
            loss = label_smoothing_loss_device(stack_steps(preds), lab, 0.1)
            loss.backward()
        return loss

    def tail():
        reducer.allreduce_mean()
        opt.step_clipped(1.0)

    if not graph:
        def step():
            loss = fwd_bwd()
            tail()
            return loss
        return step
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = fwd_bwd()

    def step():  # noqa: F811
        g.replay()
        tail()
        return static_loss
    step.graph = g
    return step


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg_name, B, T, U, train):
    from las_pytorch_amd import synth
    from oracle import cpu_baseline as CB
    c = synth.CONFIGS[cfg_name]
    sd_np = synth.make_state_dict(synth.config_shapes(cfg_name), seed=17)
    x = synth.make_inputs(B, T, c["F"], seed=17)
    idx, lens = synth.make_labels(B, U, c["V"], seed=17)
    onehot = synth.onehot_labels(idx, lens, c["V"])
    host_threads = torch.get_num_threads()
    threads = CB.best_threads(c, sd_np, x, onehot, train=train)
    r = CB.time_cpu(c, sd_np, x, onehot, train=train, iters=3, warmup=1, threads=threads)
    # single thread (SURVEY.md section 8d asks for it: the decode loop is dispatch-bound), on the full batch (~8 s per step at paper size)
    Bs = B
    r1 = CB.time_cpu(c, sd_np, x[:Bs], onehot[:Bs], train=train, iters=1, warmup=1, threads=1)
    torch.set_num_threads(host_threads)
    return dict(value=round(r["utt_per_s"], 3), unit="utt/s", cores=r["threads"], kind="port",
                sample=f"{r['iters']} steps after 1 warm-up of the same workload (B={B},T={T},U={U}, {'fwd+loss+bwd+clip+Adam' if train else 'fwd'}) "
                       f"on the host CPU with the fastest of 8/16/32 torch threads (host offers {host_threads}; the dispatch-bound decode "
                       f"loop slows down beyond that), torch {torch.__version__} oneDNN LSTM path, {r['ms_per_step']:.0f} ms/step",
                ms_per_step=round(r["ms_per_step"], 1), cpu_model=cpu_model_name(), host_threads=host_threads,
                single_thread={"value": round(r1["utt_per_s"], 3), "unit": "utt/s", "cores": 1, "ms_per_step": round(r1["ms_per_step"], 1),
                               "sample": f"1 step after 1 warm-up of the same workload (B={Bs}, the full batch), one torch thread"})


def launch_ranks(n):
    """Bare ``python bench.py --gpus N``: start the N ranks as children of this process (one per GPU, torch.distributed.run
    with a loopback rendezvous) BEFORE anything here initialises the GPU, let them write to our stdout/stderr, and return
    their exit status.  ``device_count()`` does not create a HIP context on this image."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible on this node", file=sys.stderr)
        return 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd, env = rank_launch_command(n, port, sys.argv[1:], os.environ)
    return subprocess.call(cmd, env=env)


def is_rank_process(environ):
    """True inside a rank started by torch.distributed.run (it exports WORLD_SIZE / RANK / LOCAL_RANK); bare ``python bench.py --gpus N`` is
    the launcher."""
    return "WORLD_SIZE" in environ


def rank_launch_command(n, port, argv, environ):
    """Command line and environment of the N-rank launch (pure function: tests/test_cabi_and_host.py checks it on CPU so that the first real
    ``--gpus 8`` run cannot die on plumbing): the driver's own form — ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py <same flags>`` — with the loopback rendezvous (the container hostname may not
    resolve) and dmabuf IPC (RCCL between processes fails with the legacy IPC mode on this driver)."""
    env = dict(environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return cmd, env


def preflight_problems(table):
    """The checks of ``dist_preflight`` on the gathered per-rank rows (pure function: unit-tested on CPU)."""
    problems = []
    seen = {}
    for row in table:
        key = (row["host"], row["id"] if row["id"] not in ("None", "?:?") else row["device"])
        if key in seen:
            problems.append(f"ranks {seen[key]} and {row['rank']} share one GPU ({key})")
        seen[key] = row["rank"]
        if row["cus"] != 256:
            problems.append(f"rank {row['rank']}: {row['name']} reports {row['cus']} CUs (the one-launch kernels are sized for 256; they fall back to "
                            "the per-step kernels)")
    return problems


def dist_preflight(device, rank, world):
    """First-run diagnostics of a multi-GPU launch, before any timing: every rank reports (host, GPU identity, CU count); the ranks
    of one host must sit on DISTINCT devices (two ranks on one GPU make every persistent kernel wait for the other's CUs — the
    symptom would be hand-off timeouts, not an error message) and every device must offer the 256 CUs the one-launch kernels are
    sized for.  Raises on every rank with the full table; returns the table for the JSON line."""
    import socket
    props = torch.cuda.get_device_properties(device)
    ident = getattr(props, "uuid", None)
    ident = str(ident) if ident is not None else f"{getattr(props, 'pci_bus_id', '?')}:{getattr(props, 'pci_device_id', '?')}"
    mine = dict(rank=rank, host=socket.gethostname(), device=int(device.index), id=ident, name=props.name,
                cus=int(props.multi_processor_count), visible=torch.cuda.device_count())
    table = [None] * world
    dist.all_gather_object(table, mine)
    problems = preflight_problems(table)
    fatal = [p for p in problems if "share one GPU" in p]
    if rank == 0 and problems:
        print("bench.py preflight: " + "; ".join(problems), file=sys.stderr, flush=True)
    if fatal:
        raise SystemExit("bench.py preflight failed: " + "; ".join(fatal))
    return dict(ranks=table, warnings=problems)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="P_train", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (weak scaling: fixed per-GPU batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not re-measure the roofline kernel's HBM traffic with child rocprofv3 passes")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-mfma", action="store_true")
    ap.add_argument("--graph", type=int, default=0, help="1: replay fwd+loss+bwd as one captured HIP graph per step")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch utterances per GPU (default); strong: --global-batch utterances split over the GPUs")
    ap.add_argument("--global-batch", type=int, default=256, help="total utterances per step with --scaling strong (SURVEY 8d config 3: 256)")
    args = ap.parse_args()

    if not is_rank_process(os.environ) and args.gpus > 1:
        return launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("LAS_FORCE_DIST") == "1"      # the latter: exercise the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:          # single process (LAS_FORCE_DIST=1): any free port; launchers set their own
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1]); sk.close()
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the RCCL group has {dist.get_world_size()} ranks")
        preflight = dist_preflight(device, rank, world)

    import las_pytorch_amd
    from las_pytorch_amd import dp, synth
    from las_pytorch_amd.optim import FusedClipAdam

    cfg_name, T, U, train = WORKLOADS[args.workload]
    B = args.batch
    if args.scaling == "strong":
        if args.global_batch % world != 0:
            raise SystemExit(f"bench.py: --global-batch {args.global_batch} does not divide over {world} ranks")
        B = args.global_batch // world
    las, c, _ = build_model(cfg_name, U, device)
    x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=17, rank=rank)).to(device)
    idx, lens = synth.make_labels(B, U, c["V"], seed=17, rank=rank)
    lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).to(device)
    labf = lab.float()

    first_loss = None
    if train:
        reducer = dp.FlatGradAllReducer(las, force=os.environ.get("LAS_FORCE_DIST") == "1", direct=True)
        opt = FusedClipAdam(reducer, lr=2e-4)          # clip_grad_norm_(params, 1) + Adam(lr 2e-4) as two HIP launches
        eager_step = make_train_step(las, x, lab, reducer, opt)
        first_loss = float(eager_step().item())         # the loss at the initial weights: checked against the reference below
        step = make_train_step(las, x, lab, reducer, opt, graph=True) if args.graph else eager_step
    else:
        def step():
            with torch.no_grad():
                preds, _ = las(batch_data=x, batch_label=lab, teacher_force_rate=1.0, is_training=True)
            return preds[-1]

    for _ in range(args.warmup):
        step()
    if train:
        reducer.check_views()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    rank_ms = None
    if use_dist:
        # the line reports the MAX over ranks (the contract); the per-rank values (time from the common barrier to each rank's own
        # synchronize, and to the closing barrier) say which rank, if any, the others wait for
        own = torch.tensor([dt_own, dt], device=device, dtype=torch.float64)
        allt = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(allt, own)
        rank_ms = [round(float(v[0]) / args.steps * 1e3, 4) for v in allt]
        dt = max(float(v[1]) for v in allt)
    las_pytorch_amd.check_device_errors()
    final = float(last.float().mean().item())
    assert np.isfinite(final), "non-finite result in the timed region"
    # parity hook inside the benchmark itself: rank 0's loss at the initial weights on the default workload equals the loss the
    # UNMODIFIED reference computes on the same inputs (tests/golden/P_B32_T800_U128.npz: same seeds, same shapes)
    ref_loss = None
    gpath = os.path.join(ROOT, "tests", "golden", "P_B32_T800_U128.npz")
    if train and rank == 0 and args.workload == "P_train" and B == 32 and os.path.exists(gpath):
        ref_loss = float(np.load(gpath)["loss_ls"][0])
        assert abs(first_loss - ref_loss) <= 1e-4 * abs(ref_loss), f"first-step loss {first_loss} != reference {ref_loss}"

    # every rank takes part in what contains a collective: the step's all-reduce timed alone, and the training steps under which the
    # two decode kernels are event-timed
    ar_ms = allreduce_alone_ms(reducer) if train else None
    # (the library's kernel timer records nothing while a HIP graph is replayed: time the two decode kernels under the EAGER step)
    speller_roof = roofline_speller(eager_step, c, B, T, U) if (train and not args.no_roofline) else {}
    if rank == 0:
        ms = dt / args.steps * 1e3
        try:
            metric_name = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]     # the reference's headline metric, verbatim
        except Exception:
            metric_name = "utterances/sec (B=32, T~800, 80-mel) fwd+bwd"
        res = {
            "metric": metric_name, "value": round(world * B * args.steps / dt, 2), "unit": "utt/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "rccl_ranks": dist.get_world_size() if use_dist else 0,
            **({"ms_per_step_by_rank": rank_ms, "ms_per_step_min": min(rank_ms), "ms_per_step_max": max(rank_ms),
                "preflight": preflight} if use_dist else {}),
            "config": {"workload": f"{args.workload}: Listener {c['H']}x{c['L']} / Speller {c['Hs']}x{c['Ls']}, "
                                   f"(B={B},T={T},F={c['F']}) log-mel per GPU, teacher-forced U={U}, "
                                   + ("fwd + label-smoothing loss + bwd + grad all-reduce + clip(1.0) + Adam" if train else "fwd only"),
                       "per_gpu_batch": B, "global_batch": B * world, "frames": T, "decode_steps": U,
                       "parallelism": f"dp{world}", "final_loss_or_logp": round(final, 6),
                       "first_step_loss": first_loss, "first_step_loss_reference": ref_loss,
                       "optimizer": "clip(1.0) + Adam(2e-4) as las_clip_adam (two HIP launches on the flat gradient)" if train else None},
        }
        res["config"]["gemm_arith"] = gemm_arith()[1]
        if world == 1 and train and not args.no_mfma:
            # the same step with the GEMMs on the fp32 matrix pipe (LAS_GEMM_ARITH=0), timed right after the headline loop, and the
            # measured accuracy of both GEMM arithmetics against float64
            from las_pytorch_amd import _cabi
            mode0 = _cabi.lib().las_gemm_get_arith()
            _cabi.lib().las_gemm_set_arith(1 - mode0)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nalt = max(5, args.steps // 2)
            for _ in range(nalt):
                step()
            torch.cuda.synchronize()
            dta = (time.perf_counter() - t1) / nalt
            _cabi.lib().las_gemm_set_arith(mode0)
            res["gemm_arith_variant"] = {"arith": "fp32 operands on v_mfma_f32_32x32x2_f32 (LAS_GEMM_ARITH=0)" if mode0 == 1 else "split-operand bf16 MFMA (LAS_GEMM_ARITH=1)",
                                         "value": round(B / dta, 2), "unit": "utt/s", "ms_per_step": round(dta * 1e3, 3), "steps": nalt}
            res["gemm_accuracy"] = gemm_accuracy()
            # the same step when the teacher-forcing coin comes up "free" (reference las_model.py:189,205-206: the YAML's schedule makes that
            # 10-50 % of the training steps): decode_mode 1 feedback for max_label_len steps, loss, backward — PRE kernels both ways
            free_step = make_train_step(las, x, lab, reducer, opt, tf_rate=0.0)
            for _ in range(3):
                free_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nalt):
                free_step()
            torch.cuda.synchronize()
            dtf = (time.perf_counter() - t1) / nalt
            from las_pytorch_amd import _cabi as _c
            res["free_running_step_variant"] = {"workload": "the same training step with the decode free-running (decode_mode 1, teacher_force_rate 0)", "value": round(B / dtf, 2),
                                                "unit": "utt/s", "ms_per_step": round(dtf * 1e3, 3), "steps": nalt,
                                                "decode_paths": [_c.last_path(_c.PATH_DECODE_FWD), _c.last_path(_c.PATH_DECODE_BWD)]}
            # the step as a train.py-style driver runs it: solver.batch_iterator (the same launches + the device letter error rate + the loss,
            # the rates and the device error word read back to the host every step, reference solver/solver.py:48-101) — NOT the metric (the
            # headline step never reads the loss), reported beside it
            from las_pytorch_amd.solver.solver import batch_iterator
            np.random.seed(0)
            it = lambda: batch_iterator(x, lab, las, opt, tf_rate=1.0, is_training=True, max_label_len=U, label_smoothing=0.1, use_gpu=True)
            nsol = max(50, args.steps)
            for _ in range(10):
                it()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nsol):
                it()
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t1) / nsol
            res["caller_step_variant"] = {"workload": "solver.batch_iterator on the headline batch: the same training step + las_letter_error_rate + one host read of "
                                                      "(loss, B letter error rates, device error word) per step", "value": round(B / dts, 2), "unit": "utt/s",
                                          "ms_per_step": round(dts * 1e3, 3), "steps": nsol}
        if not args.no_roofline:
            res["roofline"] = roofline_rec_fwd(c, B, T)
            if world == 1 and not args.no_live_pmc:
                # re-observe the dominant kernel's HBM traffic in THIS run (two child rocprofv3 counter passes, ~20 s); the committed
                # counter file stays as the fall-back and as the cross-check
                live, live_src = live_pmc_traffic(B, T // 2, c["H"])
                res["roofline"]["traffic_committed"] = res["roofline"]["traffic"]
                res["roofline"]["traffic_committed_source"] = res["roofline"]["traffic_source"]
                if live is not None:
                    res["roofline"]["traffic"], res["roofline"]["traffic_source"] = live, live_src
                else:
                    res["roofline"]["traffic_live_error"] = live_src
            res.update(speller_roof)
        if train:
            res["allreduce_ms"] = None if ar_ms is None else round(ar_ms, 4)
            res["allreduce_bytes"] = 4 * reducer.flat_ext.numel()
        if world == 1 and train and not args.no_mfma:
            res["roofline_mfma"] = roofline_mfma(c, B, T, U)
        if world == 1 and not args.no_sweep and not args.no_roofline:
            res["sweep"] = {"kernel": f"layer-0 forward recurrence, H={c['H']}, T_l={T // 2}: rec_fwd_fast (B=32), "
                                      "rec_fwd_mfma (64 <= B <= 256: 16 utterances per group on the bf16 matrix pipe), rec_fwd_mfma2 (B > 256: the same as a wave-specialised "
                                      "pipeline, two or three batches of 16 per group in flight)", "points": sweep_rec(c, T)}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg_name, B, T, U, train)
        if world == 1 and args.workload == "P_train" and not args.no_secondary:
            res["config"]["greedy_decode"] = greedy_decode_block(las, x, U)
            if train and B == 32:
                try:
                    res["batch128"] = batch128_block(las, c, device, T, U, reducer, opt)
                    res["batch128"]["ratio_to_headline"] = round(res["batch128"]["value"] / res["value"], 3)
                except Exception as e:      # (an optional side figure must never cost the driver line)
                    res["batch128"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            try:
                res["multi_head_variant"] = multi_head_block(device, T, U)
            except Exception as e:      # (an optional side figure must never cost the driver line)
                res["multi_head_variant"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            # BASELINE.json configs[1] (small 128/256 model, forward only) measured beside the headline for reference
            del las
            torch.cuda.empty_cache()
            las_s, c_s, _ = build_model("S", U, device)
            xs = torch.from_numpy(synth.make_inputs(B, T, c_s["F"], seed=17)).to(device)
            with torch.no_grad():
                for _ in range(3):
                    las_s(batch_data=xs, batch_label=lab, teacher_force_rate=1.0, is_training=True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(10):
                    las_s(batch_data=xs, batch_label=lab, teacher_force_rate=1.0, is_training=True)
                torch.cuda.synchronize()
                dts = (time.perf_counter() - t1) / 10
            res["config"]["secondary"] = {"workload": "S_fwd (BASELINE configs[1]): Listener 128x2 / Speller 256x2, forward only, same inputs",
                                          "value": round(B / dts, 1), "unit": "utt/s", "ms_per_step": round(dts * 1e3, 3)}
            del las_s, xs
            torch.cuda.empty_cache()
            res["config"]["secondary_long"] = secondary_long(device, U, with_roofline=not args.no_roofline)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
