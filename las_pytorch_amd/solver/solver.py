"""The code that CALLS the hot path: this build's counterpart of the reference's ``solver/solver.py``.

Only the public names, argument meaning and return values follow the reference (``LetterErrorRate`` :11-24,
``label_smoothing_loss`` :33-45, ``batch_iterator`` :48-101) so that a ``train.py``-style driver imports it unchanged;
the bodies are organised around the device path:

* the loss of a training step and its gradient come from ONE HIP kernel (``las_ls_loss``) reading the ``(U,B,V)``
  log-prob buffer the decode kernel wrote and the int64 one-hot labels as the collate function delivers them,
* the letter error rate is computed on the GPU (``las_letter_error_rate``), so a step costs one small device-to-host
  copy instead of the reference's three ``.cpu().numpy()`` round trips,
* with a ``FlatGradAllReducer`` attached to the model (data parallel, or just to get the flat gradient buffer) the
  gradients are zeroed / all-reduced / clipped on that one buffer,
* the device error word of the persistent kernels is read where the host synchronises anyway (the loss copy).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

CLIP_NORM = 1.0     # the reference hard-codes clip_grad_norm_(..., 1) (solver/solver.py:96; the YAML's max_norm is unused)


# --------------------------------------------------------------------------------------------------
# letter error rate
# --------------------------------------------------------------------------------------------------
def _levenshtein(a: np.ndarray, b: np.ndarray) -> int:
    """Edit distance between two symbol arrays, one DP row at a time, vectorised over the row (the reference calls
    the ``editdistance`` C extension, which is not installed here)."""
    if len(a) < len(b):
        a, b = b, a
    if len(b) == 0:
        return int(len(a))
    row = np.arange(len(b) + 1)
    for sym in a:
        diag = row[:-1] + (b != sym)                    # substitution / match
        up = row[1:] + 1                                # deletion
        best = np.minimum(diag, up)
        new = np.empty_like(row)
        new[0] = row[0] + 1
        # insertion is a running "previous + 1" minimum: new[j] = min(best[j-1], new[j-1] + 1)
        shifted = np.concatenate(([new[0]], best)) - np.arange(len(b) + 1)
        new = np.minimum.accumulate(shifted) + np.arange(len(b) + 1)
        row = new
    return int(row[-1])


def _strip_prediction(seq: np.ndarray) -> np.ndarray:
    """Symbols of a decoded row up to (excluding) the first <eos>=1, with the <sos>/pad symbol 0 removed."""
    stop = np.flatnonzero(seq == 1)
    if stop.size:
        seq = seq[:stop[0]]
    return seq[seq != 0]


def LetterErrorRate(pred_y, true_y):
    """Per-utterance edit distance between the decoded and the true character sequence, divided by the true length
    (reference solver/solver.py:11-24: 0 and 1 are dropped from the truth; the prediction skips 0 and stops at the
    first 1).  ``pred_y`` / ``true_y``: integer arrays (B, U).  Returns a list of B floats; an utterance whose label
    holds no symbol >= 2 raises ZeroDivisionError, as in the reference."""
    rates = []
    for hyp, ref in zip(np.asarray(pred_y), np.asarray(true_y)):
        ref = ref[ref > 1]
        if ref.size == 0:
            raise ZeroDivisionError("LetterErrorRate: a label without any character (solver/solver.py:23)")
        rates.append(_levenshtein(_strip_prediction(hyp), ref) / int(ref.size))
    return rates


DEVICE_LER_MAX_STEPS = 4095


def LetterErrorRate_device(pred_y, labels_onehot_int64, out=None):
    """The same quantity from log-probs (B,U,V) and int64 one-hot labels without leaving the GPU; returns a (B,)
    float tensor on the device (``out`` when given)."""
    from .. import _cabi
    pred_y = pred_y.detach()
    if pred_y.dtype != torch.float32:
        raise RuntimeError("LetterErrorRate_device needs fp32 log-probs")
    B, U, V = pred_y.shape
    labels = labels_onehot_int64.contiguous()
    if max(U, labels.shape[1]) > DEVICE_LER_MAX_STEPS:
        # the wave kernel keeps a row of the edit-distance table in registers (<= 4095 steps); the reference has no length limit:
        # longer decodes take the host form (one copy of the arg-max sequences)
        rates = LetterErrorRate(pred_y.argmax(-1).cpu().numpy(), labels.argmax(-1).cpu().numpy())
        res = torch.tensor(rates, dtype=torch.float32, device=pred_y.device)
        if out is not None:
            out.copy_(res)
            return out
        return res
    if out is None:
        out = torch.empty(B, device=pred_y.device)
    _cabi.check(_cabi.lib().las_letter_error_rate(_cabi.ptr_strided(pred_y), pred_y.stride(1), pred_y.stride(0), _cabi.ptr(labels), U,
                                                  labels.shape[1], B, V, _cabi.ptr(out), None, _cabi.stream_ptr()))
    return out


# --------------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------------
def label_smoothing_loss(pred_y, true_y, label_smoothing=0.1):
    """Label-smoothed cross entropy of the reference (solver/solver.py:33-45) for log-probs ``pred_y`` (B,U,V) and
    one-hot float labels ``true_y`` (B,U,V) whose padding rows are all zero:

        target[b,u,:] = ((1-eps) * y + eps/V) on labelled steps, 0 on padded ones
        loss = - mean_b  sum_u <target[b,u], logp[b,u]> / (number of labelled steps of b)

    Plain torch (any device); the training step uses the fused HIP form below."""
    if pred_y.shape != true_y.shape:
        raise AssertionError(f"label_smoothing_loss: {tuple(pred_y.shape)} vs {tuple(true_y.shape)}")
    V = true_y.shape[-1]
    labelled = true_y.sum(dim=-1, keepdim=True)                        # (B,U,1): 1 on real steps, 0 on padding
    target = labelled * ((1.0 - label_smoothing) * true_y + label_smoothing / V)
    per_step = (target * pred_y).sum(dim=-1)                            # (B,U)
    per_utt = per_step.sum(dim=-1, keepdim=True) / labelled.sum(dim=(1, 2)).unsqueeze(-1)
    return -per_utt.sum(dim=-1).mean()


class _FusedSmoothedLoss(torch.autograd.Function):
    """``las_ls_loss``: value and d(loss)/d(logp) in one launch.  ``logp`` may be any (B,U,V) view with unit stride
    along V — in particular the transposed view of the decode kernel's (U,B,V) buffer — so no copy is made."""

    @staticmethod
    def forward(ctx, logp, labels, smoothing, loss_out=None):
        from .. import _cabi
        if logp.dtype != torch.float32 or logp.stride(2) != 1:
            logp = logp.float().contiguous()
        B, U, V = logp.shape
        labels = labels.contiguous()
        dev = logp.device
        loss = torch.empty(1, device=dev) if loss_out is None else loss_out
        scratch = torch.empty(B, device=dev)
        # the gradient is laid out like its input (for the (U,B,V)-based view: step-major), so that it reaches the decode
        # kernel's backward as a contiguous (U,B,V) tensor without a transposing copy
        grad = torch.empty_like(logp) if ctx.needs_input_grad[0] else None
        if grad is not None and grad.stride(2) != 1:
            grad = torch.empty(B, U, V, device=dev)
        gs = (grad.stride(1), grad.stride(0)) if grad is not None else (V, U * V)
        _cabi.check(_cabi.lib().las_ls_loss(_cabi.ptr_strided(logp), logp.stride(1), logp.stride(0), _cabi.ptr(labels), U,
                                            labels.shape[1], B, V, float(smoothing), _cabi.ptr(loss), _cabi.ptr_strided(grad),
                                            gs[0], gs[1], _cabi.ptr(scratch), _cabi.stream_ptr()))
        ctx.grad = grad
        return loss[0]

    @staticmethod
    def backward(ctx, upstream):
        return ctx.grad * upstream, None, None, None


def label_smoothing_loss_device(pred_y, labels_onehot_int64, label_smoothing=0.1):
    """Fused on-device form of ``label_smoothing_loss`` (same value and gradient)."""
    return _FusedSmoothedLoss.apply(pred_y, labels_onehot_int64, label_smoothing)


def label_smoothing_loss_backward_device(pred_y, labels_onehot_int64, label_smoothing=0.1, loss_out=None):
    """``loss = label_smoothing_loss_device(...); loss.backward()`` without the two launches autograd adds for a scalar root (the
    ``ones_like(loss)`` fill and ``grad * upstream``): ``las_ls_loss`` produces d(loss)/d(logp) in the same launch as the loss, and
    that gradient is handed to autograd as the seed of ``pred_y`` directly.  Returns the (detached) loss tensor (``loss_out[0]`` when a
    1-element fp32 device tensor is given)."""
    with torch.no_grad():
        fn = _FusedSmoothedLoss
        class _Ctx:                      # the Function's forward, outside the graph
            needs_input_grad = (True, False, False)
        ctx = _Ctx()
        loss = fn.forward(ctx, pred_y.detach(), labels_onehot_int64, label_smoothing, loss_out)
    pred_y.backward(gradient=ctx.grad)
    return loss


# --------------------------------------------------------------------------------------------------
# one step
# --------------------------------------------------------------------------------------------------
def stack_steps(step_logp, steps=None):
    """The list of per-step ``(B,V)`` log-prob tensors as one ``(B,steps,V)`` tensor (what the reference builds with
    ``torch.cat([unsqueeze(each, 1) ...])``, solver/solver.py:68).  When the steps are the ``unbind`` views of one
    ``(U,B,V)`` buffer — the HIP Speller returns exactly that — the result is a transposed VIEW of the buffer: no copy
    forward, and the gradient flows back into the buffer's own layout."""
    steps = len(step_logp) if steps is None else steps
    first = step_logp[0]
    base = getattr(first, "_base", None)
    if (base is not None and base.dim() == 3 and base.shape[0] >= steps and base.is_contiguous()
            and all(t._base is base and t.data_ptr() == base.data_ptr() + i * base.stride(0) * base.element_size()
                    for i, t in enumerate(step_logp[:steps]))):
        return (base if steps == base.shape[0] else base[:steps]).transpose(0, 1)
    return torch.stack(list(step_logp[:steps]), dim=1)


def _on_hip_path(logp, labels):
    return logp.is_cuda and labels.dtype == torch.int64 and logp.dtype == torch.float32


def _attached_reducer(model):
    return getattr(model, "_las_flat_reducer", None)


def _loss_and_ler(logp, labels, steps, smoothed, label_smoothing):
    """``logp`` (B,steps,V) log-probs, ``labels`` (B,U_lab,V) one-hot (int64 from the collate function, or float)."""
    if _on_hip_path(logp, labels):
        ler = LetterErrorRate_device(logp, labels)
        if smoothed:
            return label_smoothing_loss_device(logp, labels, label_smoothing), ler
        target_idx = labels[:, :steps].argmax(dim=-1)
        return F.nll_loss(logp.transpose(1, 2), target_idx, ignore_index=0), ler
    # host / float-label path: the torch forms (used by the CPU tests of this module against the golden vectors)
    onehot = labels[:, :steps].to(torch.float32)
    target_idx = onehot.argmax(dim=-1)
    ler = LetterErrorRate(logp.argmax(dim=-1).cpu().numpy(), target_idx.cpu().numpy())
    if smoothed:
        return label_smoothing_loss(logp, onehot, label_smoothing=label_smoothing), ler
    return F.nll_loss(logp.transpose(1, 2), target_idx, ignore_index=0), ler


def batch_iterator(batch_data, batch_label, las_model, optimizer, tf_rate, is_training, max_label_len, label_smoothing,
                   use_gpu=True, vocab_dict=None, grad_hook=None, _retry=False):
    """One batch through the model: forward, loss, and when ``is_training`` backward, clip at 1.0 and the optimizer
    step (reference solver/solver.py:48-101; same arguments, returns ``(loss as a NumPy scalar, list of per-utterance
    letter error rates)``).  ``use_gpu`` / ``vocab_dict`` are accepted for signature parity; tensors stay on the device
    they arrive on.

    Data parallel: attach a ``las_pytorch_amd.dp.FlatGradAllReducer`` to the model (its constructor does) and this
    function zeroes, all-reduces and clips the flat gradient buffer; the teacher-forcing coin is synchronised across
    ranks first.  With a ``las_pytorch_amd.optim.FusedClipAdam`` optimizer the clip and the Adam step are two HIP launches.
    ``grad_hook(model)`` (optional) runs between backward and the clip for callers that exchange gradients themselves.

    A hand-off timeout of a persistent kernel (another kernel was resident on the GPU) invalidates the step.  When nothing
    irreversible happened — validation, or training with ``FusedClipAdam``, whose update kernel skips itself when the error
    word (on any rank) is set — the step is re-run ONCE on the generic kernels with a warning; otherwise it raises."""
    from .. import _cabi, dp
    from ..optim import FusedClipAdam
    steps = min(int(batch_label.shape[1]), int(max_label_len))
    reducer = _attached_reducer(las_model)
    fused = isinstance(optimizer, FusedClipAdam)
    if fused and reducer is not optimizer.reducer:
        raise RuntimeError("FusedClipAdam was built on a different FlatGradAllReducer than the one attached to the model")
    if is_training and not fused and batch_data.is_cuda and _cabi.handoff_spin_log2() < 21:
        # this step cannot be rolled back after a hand-off timeout (no fused update to skip itself): give the persistent kernels' bounded
        # waits the long budget (~340 ms instead of ~42 ms) before a shared GPU / debugger / pre-empted queue costs the training run
        _cabi.set_handoff_spin_log2(21)
    if reducer is not None:
        reducer.zero()                        # keeps every p.grad a view of the flat buffer
    else:
        optimizer.zero_grad()
    # data parallel: all ranks must take the same forced / free-running branch.  Only then does the coin matter; no-op otherwise
    coin_state = np.random.get_state() if (is_training and not _retry) else None
    coin_token = dp.sync_coin() if (is_training and 0.0 < float(tf_rate) < 1.0) else None

    # validation (is_training False: the reference's train.py:149-169) has no backward: run it without a graph, so that the kernels keep no
    # stash and the free-running decode takes its forward-only form (the reference keeps autograd on there, to no effect on the results)
    with _cabi.polls_deferred(), torch.set_grad_enabled(bool(is_training) and torch.is_grad_enabled()):      # a hand-off timeout is dealt with below, for the step as a whole
        step_logp, _ = las_model(batch_data=batch_data, batch_label=batch_label, teacher_force_rate=tf_rate, is_training=is_training)
    dp.restore_coin(coin_token)
    if len(step_logp) < steps:
        raise RuntimeError(f"the model decoded {len(step_logp)} steps but {steps} are scored")
    logp = stack_steps(step_logp, steps)                     # (B,steps,V); a strided view when the steps share one buffer
    smoothed = bool(is_training) and label_smoothing != 0.0
    fused_bwd = bool(is_training) and smoothed and _on_hip_path(logp, batch_label) and logp.requires_grad
    packed = None       # single-rank device path: loss and letter error rates land behind the device error word, ONE copy fetches all three
    if fused_bwd:       # loss, its gradient and the backward seed from one launch (no ones_like fill, no grad * upstream)
        if not (reducer is not None and reducer._collective()):
            packed = _cabi.step_readback(logp.device, logp.shape[0])
        ler = LetterErrorRate_device(logp, batch_label, out=None if packed is None else packed[1])
        if reducer is not None:      # (weight-gradient GEMM groups may run beside the next layer's recurrence: dp.FlatGradAllReducer.deferring)
            with reducer.deferring():
                loss = label_smoothing_loss_backward_device(logp, batch_label, label_smoothing, loss_out=None if packed is None else packed[0])
        else:
            loss = label_smoothing_loss_backward_device(logp, batch_label, label_smoothing, loss_out=None if packed is None else packed[0])
    else:
        loss, ler = _loss_and_ler(logp, batch_label, steps, smoothed, label_smoothing)

    if is_training:
        if not fused_bwd:
            if reducer is not None:
                with reducer.deferring():
                    loss.backward()
            else:
                loss.backward()
        if grad_hook is not None:
            grad_hook(las_model)
        if reducer is not None:
            reducer.allreduce_mean()
            if fused:
                optimizer.step_clipped(CLIP_NORM)
            else:
                reducer.clip_(CLIP_NORM)
                optimizer.step()
        else:
            torch.nn.utils.clip_grad_norm_(las_model.parameters(), CLIP_NORM)
            optimizer.step()

    # the step's ONE host synchronisation point: the loss and, in a multi-rank run, the all-reduced device-error flag travel together
    peer_flag = 0.0
    own_words = None
    if packed is not None:
        word, batch_loss, ler = _cabi.read_step(logp.device, logp.shape[0])      # (batch_loss: a 0-d float32 ndarray, as the other branches)
        own_words = {_cabi._dev_index(logp.device): word}
    elif is_training and reducer is not None and reducer._collective() and not loss.is_cuda:
        batch_loss = loss.detach().numpy()
        peer_flag = float(reducer.flag[0])
    elif is_training and reducer is not None and reducer._collective():
        both = torch.stack([loss.detach().reshape(()).float(), reducer.flag[0]]).cpu()
        batch_loss = both[0].numpy().astype(np.float32, copy=False)
        if str(loss.dtype) != "torch.float32":
            batch_loss = loss.detach().cpu().numpy()
        peer_flag = float(both[1])
    else:
        batch_loss = loss.detach().cpu().numpy()
    if torch.is_tensor(ler):
        ler = ler.cpu().tolist()
    if not logp.is_cuda and peer_flag != 0.0:
        raise RuntimeError("a peer rank reported a device-side hand-off timeout (all-reduced error flag "
                           f"{peer_flag:g}); this step's averaged gradient is invalid on every rank")
    if logp.is_cuda:
        failed = None
        try:
            _cabi.check_device_errors(own_words)       # a hand-off timeout in a persistent kernel invalidates this step
        except _cabi.DeviceHandoffError as e:
            failed = e
        peer_failed = peer_flag != 0.0
        if failed is not None or peer_failed:
            recoverable = (not is_training) or fused          # the fused update skipped itself on every rank (all-reduced flag)
            if _retry or not recoverable:
                raise failed if failed is not None else RuntimeError("a peer rank reported a device-side hand-off timeout")
            import warnings
            from ..model.las_model import set_force_generic
            warnings.warn(f"liblas_hip: persistent-kernel hand-off timeout ({failed or 'reported by a peer rank'}); re-running this "
                          "step once on the generic kernels (another kernel was resident on the GPU?)")
            if is_training and reducer is not None and reducer.direct and _cabi.last_path(_cabi.PATH_DW) in ("deferred", "joined"):
                # (the XCD-confined recurrences number their workgroups by the XCC id they find themselves on: should a dispatcher ever place
                # blocks differently, a role is missing and the spin timeout lands here — leave that mode for the rest of the process)
                _cabi.set_option("DEFER_DW", 0)
            if fused and is_training:
                optimizer.rollback_step()
            if coin_state is not None:
                np.random.set_state(coin_state)               # the re-run must draw the same teacher-forcing coin
            before = {m: m.force_generic for m in las_model.modules() if hasattr(m, "force_generic")}
            set_force_generic(las_model, True)
            try:
                return batch_iterator(batch_data, batch_label, las_model, optimizer, tf_rate, is_training, max_label_len,
                                      label_smoothing, use_gpu, vocab_dict, grad_hook, _retry=True)
            finally:
                for m, v in before.items():
                    m.force_generic = v
    return batch_loss, ler
