"""Drop-in ``Listener`` / ``Speller`` / ``LAS`` modules backed by liblas_hip.so.

Mirrors the public surface of the reference's ``model/las_model.py`` (constructor signatures, attribute
names, ``state_dict`` keys, return types — SURVEY.md section 8b) so ``train.py`` / ``solver.batch_iterator``
style callers work unchanged, while every FLOP of forward and backward runs in the HIP kernels:

  reference                                   | here
  --------------------------------------------+---------------------------------------------------------
  pBLSTMLayer.forward  (las_model.py:81-91)   | las_pblstm_fwd / las_pblstm_bwd   (MFMA GEMM + persistent RNN)
  Listener.forward     (las_model.py:129-134) | stack of the above
  Speller.forward      (las_model.py:186-238) | las_attn_keys_fwd + las_speller_fwd / las_speller_bwd
  Attention.forward    (las_model.py:275-297) | fused into the speller step kernels (psi hoisted out of the loop)
  LAS.forward/serialize(las_model.py:24-63)   | same logic

There is no CPU or eager fallback: CPU tensors raise.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from .. import _cabi
from .._cabi import (FLAG_DEFER_DW, FLAG_FORCE_GENERIC, FLAG_GRADS_ZEROED, FLAG_STASH, FLAG_TEACHER_FORCED, SpellerDesc, SpellerGrads, check, lib,
                     ptr, stream_ptr)

def set_force_generic(module, flag=True):
    """A/B switch for tests and profiling: every pBLSTM layer / Speller under ``module`` uses the generic kernels
    (L2-streaming recurrence, per-step decode launches; ``LAS_FLAG_FORCE_GENERIC``) instead of the persistent ones.
    Per-module state: two models in one process do not influence each other."""
    for m in module.modules():
        if isinstance(m, (pBLSTMLayer, Speller)):
            m.force_generic = bool(flag)


def _direct_targets(params):
    """``las_pytorch_amd.dp.FlatGradAllReducer(direct=True)`` tags the parameters it owns with the address of its flat
    buffer; for those (and only those) the backward kernels write the gradient straight into ``p.grad`` (a view of that
    buffer) and autograd is handed ``None``, instead of fresh tensors that AccumulateGrad adds into ``p.grad`` with one
    small kernel per parameter (36 launches per step at the reference's sizes).  Valid when every parameter is used
    once per backward and the buffer is re-zeroed every step — the loop of solver/solver.py:95-97.  Any parameter
    whose ``.grad`` is no longer that view (``zero_grad(set_to_none=True)``, a foreign module) switches the whole call
    back to ordinary autograd accumulation."""
    out = []
    for p in params:
        base = getattr(p, "_las_direct_base", None)
        g = getattr(p, "grad", None)
        if (base is None or g is None or not p.requires_grad or g.dtype != torch.float32 or not g.is_contiguous()
                or g.shape != p.shape or g.device != p.device or g.untyped_storage().data_ptr() != base):
            return None
        out.append(g)
    return out


def _claim_prezeroed(params):
    """True when every one of these direct-write parameters' gradient views is still as ``FlatGradAllReducer.zero()`` left it (zeroed this
    step, not written since): the backward entry point may then skip its own fill of the block (``LAS_FLAG_GRADS_ZEROED``).  Marks them
    written."""
    ok = True
    capturing = any(p.is_cuda for p in params) and torch.cuda.is_current_stream_capturing()
    for p in params:
        owner = getattr(p, "_las_direct_owner", None)
        if owner is None or getattr(p, "_las_written_epoch", None) == owner.zero_epoch or owner.zero_epoch == 0:
            ok = False
        elif capturing and not getattr(owner, "zero_in_capture", False):
            ok = False      # a replay would skip the fill without the zero() that justified it (dp.FlatGradAllReducer.zero)
    for p in params:
        owner = getattr(p, "_las_direct_owner", None)
        if owner is not None:
            p._las_written_epoch = owner.zero_epoch
    return ok


def _defer_owner(direct_params):
    """The flat-gradient reducer that owns these direct-write parameters, when the caller opened ``reducer.deferring()`` (the weight-gradient
    GEMM group of this backward call may then stay on the library's side stream: LAS_FLAG_DEFER_DW); None otherwise."""
    if not direct_params:
        return None
    owner = getattr(direct_params[0], "_las_direct_owner", None)
    if owner is None or not getattr(owner, "_defer_active", False) or torch.cuda.is_current_stream_capturing():
        return None
    return owner


def _flags(stash, force_generic=False):
    return (FLAG_STASH if stash else 0) | (FLAG_FORCE_GENERIC if force_generic else 0)


def _f32c(t):
    if not t.is_cuda:
        raise RuntimeError("the LAS HIP path needs CUDA(ROCm) tensors; there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError("the LAS HIP path computes in fp32; got " + str(t.dtype))
    return t.contiguous()


class _LSTMParams(nn.Module):
    """Parameter container with torch.nn.LSTM's names, shapes, registration order and init
    (U(-1/sqrt(H), 1/sqrt(H)) in registration order), so checkpoints and seeded initialisation match the
    reference's ``nn.LSTM`` (las_model.py:72-79,164-166).  It owns no forward: the kernels do."""

    def __init__(self, input_size, hidden_size, num_layers=1, bidirectional=False):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.bidirectional = input_size, hidden_size, num_layers, bidirectional
        dirs = 2 if bidirectional else 1
        for layer in range(num_layers):
            for d in range(dirs):
                in_size = input_size if layer == 0 else hidden_size * dirs
                sfx = "_reverse" if d == 1 else ""
                self.register_parameter(f"weight_ih_l{layer}{sfx}", nn.Parameter(torch.empty(4 * hidden_size, in_size)))
                self.register_parameter(f"weight_hh_l{layer}{sfx}", nn.Parameter(torch.empty(4 * hidden_size, hidden_size)))
                self.register_parameter(f"bias_ih_l{layer}{sfx}", nn.Parameter(torch.empty(4 * hidden_size)))
                self.register_parameter(f"bias_hh_l{layer}{sfx}", nn.Parameter(torch.empty(4 * hidden_size)))
        stdv = 1.0 / math.sqrt(hidden_size)
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)

    def extra_repr(self):
        return f"{self.input_size}, {self.hidden_size}, num_layers={self.num_layers}, bidirectional={self.bidirectional}"


def _require_lstm(rnn_unit):
    if str(rnn_unit).upper() != "LSTM":
        raise NotImplementedError(f"rnn_unit={rnn_unit!r}: only LSTM is implemented by the HIP path "
                                  "(the reference's configs use LSTM only, config/librispeech-config.yaml:20,26)")


# --------------------------------------------------------------------------------------------------
# pBLSTM layer
# --------------------------------------------------------------------------------------------------
class _PBLSTMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mode, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        force_generic, grad_on = mode      # grad_on: the CALLER's grad mode (inside Function.forward it is always off)
        x = _f32c(x)
        B, T_in, D_in = x.shape
        H = w_hh_f.shape[1]
        if T_in % 2 != 0:
            raise RuntimeError(f"pBLSTM needs an even number of frames, got {T_in} (reference las_model.py:86-87)")
        ws = [_f32c(w) for w in (w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r)]
        # needs_input_grad stays True under torch.no_grad() (it mirrors requires_grad): no backward can follow then, so nothing is stashed
        stash = grad_on and any(ctx.needs_input_grad)
        flags = _flags(stash, force_generic)
        L = lib()
        out = torch.empty(B, T_in // 2, 2 * H, device=x.device, dtype=torch.float32)
        reserve = torch.empty(L.las_pblstm_reserve_floats(B, T_in, H, flags), device=x.device, dtype=torch.float32)
        check(L.las_pblstm_fwd(ptr(x), B, T_in, D_in, H, *[ptr(w) for w in ws], ptr(out), ptr(reserve),
                               ptr(_cabi.err_word(x.device)), flags, stream_ptr()))
        if stash:
            ctx.save_for_backward(x, ws[0], ws[1], ws[4], ws[5], reserve)
            ctx.direct = _direct_targets((w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r))
            ctx.direct_params = (w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r) if ctx.direct else None
            ctx.dims = (B, T_in, D_in, H)
            ctx.need_dx = ctx.needs_input_grad[1]
            ctx.flags = flags
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w_ih_f, w_hh_f, w_ih_r, w_hh_r, reserve = ctx.saved_tensors
        B, T_in, D_in, H = ctx.dims
        dout = _f32c(dout)
        L = lib()
        dev = x.device
        work = torch.empty(L.las_pblstm_bwd_workspace_floats(B, T_in, H), device=dev, dtype=torch.float32)
        dx = torch.empty_like(x) if ctx.need_dx else None
        g = ctx.direct or [torch.empty_like(w_ih_f), torch.empty_like(w_hh_f), torch.empty(4 * H, device=dev),
                           torch.empty(4 * H, device=dev), torch.empty_like(w_ih_r), torch.empty_like(w_hh_r),
                           torch.empty(4 * H, device=dev), torch.empty(4 * H, device=dev)]
        zeroed = FLAG_GRADS_ZEROED if (ctx.direct and _claim_prezeroed(ctx.direct_params)) else 0
        owner = _defer_owner(ctx.direct_params) if ctx.direct else None
        check(L.las_pblstm_bwd(ptr(x), ptr(dout), B, T_in, D_in, H, ptr(w_ih_f), ptr(w_hh_f), ptr(w_ih_r), ptr(w_hh_r),
                               ptr(reserve), ptr(work), ptr(dx), *[ptr(t) for t in g], ptr(_cabi.err_word(dev)),
                               ctx.flags | zeroed | (FLAG_DEFER_DW if owner is not None else 0), stream_ptr()))
        if owner is not None:      # what the deferred GEMM group reads must outlive this call (autograd frees the saved tensors on return)
            owner._deferred_keep.append((x, dout, reserve, work))
        return (None, dx, *([None] * 8 if ctx.direct else g))


class pBLSTMLayer(nn.Module):
    """Reference model/las_model.py:66-91: halve the time resolution by concatenating frame pairs, then a
    bidirectional LSTM.  ``forward`` returns ``(output, hidden)`` like the reference; ``hidden`` (the final
    (h_n, c_n), discarded by every reference caller, las_model.py:130,132) is returned as ``None``."""

    def __init__(self, input_feature_dim, hidden_dim, rnn_unit="LSTM", dropout_rate=0.0):
        super().__init__()
        _require_lstm(rnn_unit)
        self.rnn_unit = nn.LSTM          # attribute kept for parity with the reference (class object, :69)
        self.BLSTM = _LSTMParams(input_feature_dim * 2, hidden_dim, 1, bidirectional=True)
        self.dropout_rate = dropout_rate  # dropout on a 1-layer LSTM is a no-op in torch as well
        self.force_generic = False        # see set_force_generic()

    def forward(self, input_x):
        p = self.BLSTM
        if input_x.is_cuda:
            _cabi.poll_device_errors(input_x.device)
        out = _PBLSTMFn.apply((self.force_generic, torch.is_grad_enabled()), input_x, p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0,
                              p.weight_ih_l0_reverse, p.weight_hh_l0_reverse, p.bias_ih_l0_reverse, p.bias_hh_l0_reverse)
        return out, None


class Listener(nn.Module):
    """Reference model/las_model.py:96-134.  Same constructor; ``use_gpu`` accepted and ignored (as there)."""

    def __init__(self, input_feature_dim, hidden_size, num_layers, rnn_unit, use_gpu, dropout_rate=0.0, **kwargs):
        super().__init__()
        self.input_feature_dim = input_feature_dim
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.rnn_unit = rnn_unit
        self.dropout_rate = dropout_rate
        assert num_layers >= 1, "Listener should have at least 1 layer"
        # attribute names pLSTM_layer{l} are the checkpoint keys (reference las_model.py:116-127)
        for l in range(num_layers):
            in_dim = input_feature_dim if l == 0 else 2 * hidden_size
            self.add_module(f"pLSTM_layer{l}", pBLSTMLayer(in_dim, hidden_size, rnn_unit=rnn_unit, dropout_rate=dropout_rate))

    def _layers(self):
        return [getattr(self, f"pLSTM_layer{l}") for l in range(self.num_layers)]

    def forward(self, input_x):
        feat = input_x
        for layer in self._layers():
            feat, _hidden = layer(feat)          # the final (h_n, c_n) is dropped, as in the reference (:130,132)
        return feat


# --------------------------------------------------------------------------------------------------
# Speller
# --------------------------------------------------------------------------------------------------
class Attention(nn.Module):
    """Parameter container + configuration of the reference's Attention (las_model.py:249-273).  Its
    arithmetic (las_model.py:275-297) is fused into the speller step kernels."""

    def __init__(self, mlp_preprocess_input, preprocess_mlp_dim, activate, mode="dot", input_feature_dim=512, multi_head=1):
        super().__init__()
        self.mode = mode.lower()
        self.mlp_preprocess_input = mlp_preprocess_input
        self.multi_head = multi_head
        self.input_feature_dim = input_feature_dim
        if self.mode != "dot":
            raise NotImplementedError("only dot attention exists in the reference (las_model.py:315-317)")
        if multi_head > 1 and not mlp_preprocess_input:
            raise NotImplementedError("multi-head attention needs use_mlp_in_attention (phi/dim_reduce only exist with the MLP, "
                                      "reference las_model.py:264-269)")
        self.activate = None
        if mlp_preprocess_input:
            self.preprocess_mlp_dim = preprocess_mlp_dim
            self.phi = nn.Linear(input_feature_dim, preprocess_mlp_dim * multi_head)
            self.psi = nn.Linear(input_feature_dim, preprocess_mlp_dim)
            if self.multi_head > 1:
                self.dim_reduce = nn.Linear(input_feature_dim * multi_head, input_feature_dim)
            if activate != "None":
                if activate not in _ACT_CODES:
                    raise NotImplementedError(f"mlp_activate_in_attention={activate!r}: the HIP path implements "
                                              f"{sorted(k for k in _ACT_CODES if k)} and 'None'")
                self.activate = activate

    def _params(self):
        ps = []
        if self.mlp_preprocess_input:
            ps += [self.phi.weight, self.phi.bias, self.psi.weight, self.psi.bias]
            if self.multi_head > 1:
                ps += [self.dim_reduce.weight, self.dim_reduce.bias]
        return ps

    def forward(self, decoder_state, listener_feature):
        """Reference las_model.py:275-318: ``decoder_state`` (B,1,2H) [or (B,2H)], ``listener_feature`` (B,T',2H) ->
        ``([attention_score (B,T') per head], context (B,2H))``.  Differentiable wrt the decoder state, the listener
        features and phi / psi / dim_reduce (``las_attention_fwd`` / ``las_attention_bwd``); the returned scores are
        for inspection and carry no gradient.  Inside ``Speller.forward`` the same arithmetic runs fused in the decode
        kernels; this entry point serves callers that use the module on its own."""
        use_mlp = bool(self.mlp_preprocess_input)
        cfg = (use_mlp, _ACT_CODES[self.activate], int(self.preprocess_mlp_dim) if use_mlp else 0, int(self.multi_head))
        ctx, att = _AttentionFn.apply(cfg, decoder_state.reshape(decoder_state.shape[0], -1), listener_feature, *self._params())
        return list(att.unbind(0)), ctx


# attention activation name (mlp_activate_in_attention, reference las_model.py:270-273) -> las_speller_desc::relu code
_ACT_CODES = {None: 0, "relu": 1, "tanh": 2, "sigmoid": 3}


def _speller_desc(B, Tp, D, Hs, V, M, L, use_mlp, relu, lstm, rest, heads=1):
    """Fill the C descriptor from (already contiguous fp32) parameter tensors; returns it (pointers only — the
    caller keeps the tensors alive).  ``rest`` = [phi.w, phi.b, psi.w, psi.b, (dim_reduce.w, dim_reduce.b),] c.w, c.b"""
    d = SpellerDesc()
    d.B, d.Tp, d.D, d.Hs, d.V, d.M, d.L = B, Tp, D, Hs, V, M, L
    d.use_mlp, d.relu, d.multi_head = int(use_mlp), int(relu), int(heads)
    for l in range(L):
        d.w_ih[l], d.w_hh[l], d.b_ih[l], d.b_hh[l] = (ptr(lstm[4 * l + i]) for i in range(4))
    rest = list(rest)
    if use_mlp:
        d.w_phi, d.b_phi, d.w_psi, d.b_psi = (ptr(t) for t in rest[:4])
        rest = rest[4:]
        if heads > 1:
            d.w_dr, d.b_dr = ptr(rest[0]), ptr(rest[1])
            rest = rest[2:]
    d.w_c, d.b_c = ptr(rest[0]), ptr(rest[1])
    return d



def _attn_desc(B, Tp, D, use_mlp, relu, M, heads, params):
    """Descriptor for the attention-only entry points (no LSTM / character-distribution weights)."""
    d = SpellerDesc()
    d.B, d.Tp, d.D, d.Hs, d.V, d.M, d.L = B, Tp, D, D, 1, M, 1
    d.use_mlp, d.relu, d.multi_head = int(use_mlp), int(relu), int(heads)
    if use_mlp:
        d.w_phi, d.b_phi, d.w_psi, d.b_psi = (ptr(t) for t in params[:4])
        if heads > 1:
            d.w_dr, d.b_dr = ptr(params[4]), ptr(params[5])
    return d


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, state, feat, *params):
        use_mlp, relu, M, heads = cfg
        state, feat = _f32c(state), _f32c(feat)
        B, Tp, D = feat.shape
        if state.shape != (B, D):
            raise RuntimeError(f"decoder_state must be (B,1,{D}) / (B,{D}), got {tuple(state.shape)}")
        # never direct gradient writes here: a stand-alone attention call may share phi / psi with other uses in the same
        # backward (Speller.forward, further Attention.forward calls), and every las_attention_bwd OVERWRITES its outputs —
        # fresh tensors + autograd's AccumulateGrad sum them correctly
        direct = None
        params = [_f32c(p) for p in params]
        dev = feat.device
        Lh, stream = lib(), stream_ptr()
        d = _attn_desc(B, Tp, D, use_mlp, relu, M, heads, params)
        keys = None
        if use_mlp:
            keys = torch.empty(B, Tp, M, device=dev, dtype=torch.float32)
            check(Lh.las_attn_keys_fwd(d, ptr(feat), ptr(keys), stream))
        att = torch.empty(heads, B, Tp, device=dev, dtype=torch.float32)
        context = torch.empty(B, D, device=dev, dtype=torch.float32)
        reserve = torch.empty(max(4, Lh.las_attention_reserve_floats(d)), device=dev, dtype=torch.float32)
        check(Lh.las_attention_fwd(d, ptr(feat), ptr(keys), ptr(state), ptr(att), ptr(context), ptr(reserve), stream))
        ctx.mark_non_differentiable(att)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(state, feat, keys, att, reserve, *params)
            ctx.cfg, ctx.direct = cfg, direct
        return context, att

    @staticmethod
    def backward(ctx, dctx, _datt):
        state, feat, keys, att, reserve, *params = ctx.saved_tensors
        use_mlp, relu, M, heads = ctx.cfg
        B, Tp, D = feat.shape
        dev = feat.device
        Lh = lib()
        d = _attn_desc(B, Tp, D, use_mlp, relu, M, heads, params)
        grads = ctx.direct or [torch.empty_like(p) for p in params]
        g = SpellerGrads()
        if use_mlp:
            g.dw_phi, g.db_phi, g.dw_psi, g.db_psi = (ptr(t) for t in grads[:4])
            if heads > 1:
                g.dw_dr, g.db_dr = ptr(grads[4]), ptr(grads[5])
        dfeat = torch.empty_like(feat)
        dstate = torch.empty_like(state)
        g.dfeat = ptr(dfeat)
        work = torch.empty(Lh.las_attention_bwd_workspace_floats(d), device=dev, dtype=torch.float32)
        check(Lh.las_attention_bwd(d, ptr(feat), ptr(keys), ptr(state), ptr(att), ptr(reserve), ptr(_f32c(dctx)), ptr(dstate), g,
                                   ptr(work), stream_ptr()))
        return (None, dstate, dfeat, *([None] * len(grads) if ctx.direct else grads))


class _SpellerStepFn(torch.autograd.Function):
    """One decode step with caller-managed state and its backward (``las_speller_step_fwd`` / ``las_speller_step_bwd``)."""

    @staticmethod
    def forward(ctx, cfg, feat, x, h_in, c_in, *params):
        (L, use_mlp, relu, M, V, heads) = cfg
        feat, x = _f32c(feat), _f32c(x)
        B, Tp, D = feat.shape
        # forward_step is by nature called once per decode step, so every parameter is used U times per backward and each
        # las_speller_step_bwd call overwrites its gradient outputs: no direct writes into p.grad here, autograd sums the steps
        direct = None
        params = [_f32c(p) for p in params]
        lstm, rest = params[:4 * L], params[4 * L:]
        Hs = lstm[1].shape[1]
        if D != Hs:
            raise RuntimeError(f"Speller hidden_size ({Hs}) must equal 2*listener_hidden_size ({D}) (reference las_model.py:198)")
        if x.shape != (B, V + Hs):
            raise RuntimeError(f"input_word must be (B,1,{V + Hs}), got {tuple(x.shape)}")
        dev = feat.device
        Lh, stream = lib(), stream_ptr()
        d = _speller_desc(B, Tp, D, Hs, V, M, L, use_mlp, relu, lstm, rest, heads)
        keys = None
        if use_mlp:
            keys = torch.empty(B, Tp, M, device=dev, dtype=torch.float32)
            check(Lh.las_attn_keys_fwd(d, ptr(feat), ptr(keys), stream))
        if h_in is not None:
            h_in, c_in = _f32c(h_in), _f32c(c_in)
        train = any(ctx.needs_input_grad)
        logp = torch.empty(B, V, device=dev); context = torch.empty(B, D, device=dev); att = torch.empty(heads, B, Tp, device=dev)
        h_out = torch.empty(L, B, Hs, device=dev); c_out = torch.empty(L, B, Hs, device=dev)
        work = torch.empty(Lh.las_speller_step_workspace_floats(d), device=dev, dtype=torch.float32)
        reserve = torch.empty(max(4, Lh.las_speller_step_reserve_floats(d)), device=dev, dtype=torch.float32) if train else None
        check(Lh.las_speller_step_fwd(d, ptr(feat), ptr(keys), ptr(x), ptr(h_in), ptr(c_in), ptr(logp), ptr(h_out), ptr(c_out),
                                      ptr(context), ptr(att), ptr(work), ptr(reserve), stream))
        ctx.mark_non_differentiable(att)
        if train:
            ctx.has_state = h_in is not None
            saved = [feat, keys, x, logp, h_out, c_out, context, att, reserve] + ([h_in, c_in] if ctx.has_state else []) + params
            ctx.save_for_backward(*saved)
            ctx.cfg, ctx.direct = cfg, direct
        return logp, h_out, c_out, context, att

    @staticmethod
    def backward(ctx, dlogp, dh_out, dc_out, dctx, _datt):
        feat, keys, x, logp, h_out, c_out, context, att, reserve, *rest_saved = ctx.saved_tensors
        (L, use_mlp, relu, M, V, heads) = ctx.cfg
        h_in = c_in = None
        if ctx.has_state:
            h_in, c_in, *params = rest_saved
        else:
            params = rest_saved
        B, Tp, D = feat.shape
        Hs = params[1].shape[1]
        dev = feat.device
        Lh = lib()
        lstm, rest = params[:4 * L], params[4 * L:]
        d = _speller_desc(B, Tp, D, Hs, V, M, L, use_mlp, relu, lstm, rest, heads)
        grads = ctx.direct or [torch.empty_like(p) for p in params]
        g = _speller_grads(grads, L, use_mlp, heads)
        dfeat = torch.empty_like(feat)
        g.dfeat = ptr(dfeat)
        dx = torch.empty_like(x)
        dh_in = torch.empty(L, B, Hs, device=dev); dc_in = torch.empty(L, B, Hs, device=dev)
        work = torch.empty(Lh.las_speller_step_bwd_workspace_floats(d), device=dev, dtype=torch.float32)
        opt = lambda t: ptr(_f32c(t)) if t is not None else None
        keep = [t if t is None else _f32c(t) for t in (dlogp, dh_out, dc_out, dctx)]      # keep the contiguous copies alive
        check(Lh.las_speller_step_bwd(d, ptr(feat), ptr(keys), ptr(x), ptr(h_in), ptr(c_in), ptr(logp), ptr(h_out), ptr(c_out),
                                      ptr(context), ptr(att), ptr(reserve), *[ptr(t) for t in keep], ptr(dx), ptr(dh_in), ptr(dc_in),
                                      g, ptr(work), stream_ptr()))
        return (None, dfeat, dx, dh_in if ctx.has_state else None, dc_in if ctx.has_state else None,
                *([None] * len(grads) if ctx.direct else grads))


def _speller_grads(grads, L, use_mlp, heads):
    g = SpellerGrads()
    for l in range(L):
        g.dw_ih[l], g.dw_hh[l], g.db_ih[l], g.db_hh[l] = (ptr(grads[4 * l + i]) for i in range(4))
    rg = grads[4 * L:]
    if use_mlp:
        g.dw_phi, g.db_phi, g.dw_psi, g.db_psi = (ptr(t) for t in rg[:4])
        rg = rg[4:]
        if heads > 1:
            g.dw_dr, g.db_dr = ptr(rg[0]), ptr(rg[1])
            rg = rg[2:]
    g.dw_c, g.db_c = ptr(rg[0]), ptr(rg[1])
    return g


class _SpellerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, feat, labels, noise, *params):
        (U, teacher_forced, decode_mode, L, use_mlp, relu, M, V, heads, force_generic, allow_direct, grad_on) = cfg
        feat = _f32c(feat)
        B, Tp, D = feat.shape
        direct = _direct_targets(params) if allow_direct else None      # a sliced batch uses every parameter once per slice
        ctx.direct_params = tuple(params) if direct else None
        params = [_f32c(p) for p in params]
        lstm, rest = params[:4 * L], params[4 * L:]
        Hs = lstm[1].shape[1]
        if D != Hs:
            raise RuntimeError(f"Speller hidden_size ({Hs}) must equal 2*listener_hidden_size ({D}) (reference las_model.py:198)")
        dev = feat.device
        Lh = lib()
        d = _speller_desc(B, Tp, D, Hs, V, M, L, use_mlp, relu, lstm, rest, heads)
        stream = stream_ptr()
        keys = None
        if use_mlp:
            keys = torch.empty(B, Tp, M, device=dev, dtype=torch.float32)
            check(Lh.las_attn_keys_fwd(d, ptr(feat), ptr(keys), stream))
        u_lab = 0
        if labels is not None:
            if labels.dtype != torch.int64:
                labels = labels.to(torch.int64)
            labels = labels.contiguous()
            u_lab = labels.shape[1]
        logp = torch.empty(U, B, V, device=dev, dtype=torch.float32)
        att = torch.empty(U, heads, B, Tp, device=dev, dtype=torch.float32)
        reserve = torch.empty(Lh.las_speller_reserve_floats(d, U), device=dev, dtype=torch.float32)
        # no backward will follow otherwise (validation / inference under torch.no_grad(): needs_input_grad alone stays True there): skip the
        # backward-only products — and a free-running decode then takes the forward-only kernel (speller_persist_pre_greedy_eligible)
        stash = grad_on and any(ctx.needs_input_grad)
        if noise is not None:
            noise = _f32c(noise)
            if tuple(noise.shape) != (U, B, V):
                raise RuntimeError(f"decode_mode 2 needs Exp(1) draws of shape {(U, B, V)}, got {tuple(noise.shape)}")
        ctx.set_materialize_grads(False)       # (the attention output takes no gradient: do not let autograd fill a zero tensor for it)
        check(Lh.las_speller_fwd(d, ptr(feat), ptr(keys), ptr(labels) if teacher_forced else None, u_lab, U,
                                 int(teacher_forced), decode_mode, ptr(noise), ptr(logp), ptr(att), None, ptr(reserve),
                                 ptr(_cabi.err_word(dev)), _flags(stash, force_generic), stream))
        ctx.mark_non_differentiable(att)
        if stash:
            ctx.save_for_backward(feat, keys, logp, att, reserve, *params)
            ctx.direct = direct
            ctx.cfg = cfg
            ctx.dims = (B, Tp, D, Hs)
        return logp, att

    @staticmethod
    def backward(ctx, dlogp, _datt):
        feat, keys, logp, att, reserve, *params = ctx.saved_tensors
        (U, teacher_forced, decode_mode, L, use_mlp, relu, M, V, heads, force_generic, _allow_direct, _grad_on) = ctx.cfg
        B, Tp, D, Hs = ctx.dims
        dev = feat.device
        if dlogp is None:                      # (set_materialize_grads(False): only the attention output was used downstream)
            dlogp = torch.zeros_like(logp)
        dlogp = _f32c(dlogp)
        lstm, rest = params[:4 * L], params[4 * L:]
        Lh = lib()
        d = _speller_desc(B, Tp, D, Hs, V, M, L, use_mlp, relu, lstm, rest, heads)
        grads = ctx.direct or [torch.empty_like(p) for p in params]
        dfeat = torch.empty_like(feat)
        g = SpellerGrads()
        for l in range(L):
            g.dw_ih[l], g.dw_hh[l], g.db_ih[l], g.db_hh[l] = (ptr(grads[4 * l + i]) for i in range(4))
        rg = grads[4 * L:]
        if use_mlp:
            g.dw_phi, g.db_phi, g.dw_psi, g.db_psi = (ptr(t) for t in rg[:4])
            rg = rg[4:]
            if heads > 1:
                g.dw_dr, g.db_dr = ptr(rg[0]), ptr(rg[1])
                rg = rg[2:]
        g.dw_c, g.db_c = ptr(rg[0]), ptr(rg[1])
        g.dfeat = ptr(dfeat)
        work = torch.empty(Lh.las_speller_bwd_workspace_floats(d, U), device=dev, dtype=torch.float32)
        mode0 = int((not teacher_forced) and decode_mode == 0)
        zeroed = FLAG_GRADS_ZEROED if (ctx.direct and _claim_prezeroed(ctx.direct_params)) else 0
        owner = _defer_owner(ctx.direct_params) if ctx.direct else None
        check(Lh.las_speller_bwd(d, ptr(feat), ptr(keys), ptr(logp), ptr(att), ptr(dlogp), U, mode0, ptr(reserve),
                                 ptr(work), g, ptr(_cabi.err_word(dev)),
                                 _flags(True, force_generic) | (FLAG_TEACHER_FORCED if (teacher_forced or decode_mode == 1) else 0) | zeroed |
                                 (FLAG_DEFER_DW if owner is not None else 0), stream_ptr()))
        if owner is not None:
            owner._deferred_keep.append((feat, keys, logp, att, dlogp, reserve, work))
        return (None, dfeat, None, None, *([None] * len(grads) if ctx.direct else grads))


class Speller(nn.Module):
    """Reference model/las_model.py:138-238.  Same constructor, attributes and return types."""

    def __init__(self, vocab_size, hidden_size, rnn_unit, num_layers, max_label_len, use_mlp_in_attention,
                 mlp_dim_in_attention, mlp_activate_in_attention, listener_hidden_size, multi_head, decode_mode,
                 use_gpu=True, **kwargs):
        super().__init__()
        _require_lstm(rnn_unit)
        self.rnn_unit = nn.LSTM           # the reference stores the class (las_model.py:156)
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.max_label_len = max_label_len
        self.decode_mode = decode_mode
        self.use_gpu = use_gpu
        self.float_type = torch.cuda.FloatTensor if use_gpu else torch.FloatTensor
        self.label_dim = vocab_size
        if num_layers > _cabi.MAX_L:
            raise NotImplementedError(f"at most {_cabi.MAX_L} speller layers")
        self.rnn_layer = _LSTMParams(vocab_size + hidden_size, hidden_size, num_layers=num_layers)
        self.attention = Attention(mlp_preprocess_input=use_mlp_in_attention, preprocess_mlp_dim=mlp_dim_in_attention,
                                   activate=mlp_activate_in_attention, input_feature_dim=2 * listener_hidden_size,
                                   multi_head=multi_head)
        self.character_distribution = nn.Linear(hidden_size * 2, vocab_size)
        self.softmax = nn.LogSoftmax(dim=-1)   # attribute parity only; the log-softmax is fused in the step kernel
        self.force_generic = False             # see set_force_generic()
        self.sample_noise = None               # decode_mode 2: optional caller-supplied Exp(1) draws (U,B,V)

    def _params(self):
        ps = []
        for l in range(self.num_layers):
            ps += [getattr(self.rnn_layer, f"weight_ih_l{l}"), getattr(self.rnn_layer, f"weight_hh_l{l}"),
                   getattr(self.rnn_layer, f"bias_ih_l{l}"), getattr(self.rnn_layer, f"bias_hh_l{l}")]
        a = self.attention
        if a.mlp_preprocess_input:
            ps += [a.phi.weight, a.phi.bias, a.psi.weight, a.psi.bias]
            if a.multi_head > 1:
                ps += [a.dim_reduce.weight, a.dim_reduce.bias]
        ps += [self.character_distribution.weight, self.character_distribution.bias]
        return ps

    def _run(self, listener_feature, ground_truth, teacher_force, steps):
        a = self.attention
        use_mlp = bool(a.mlp_preprocess_input)
        cfg = (int(steps), bool(teacher_force), int(self.decode_mode), int(self.num_layers), use_mlp,
               _ACT_CODES[a.activate], int(a.preprocess_mlp_dim) if use_mlp else 0, int(self.label_dim), int(a.multi_head),
               bool(self.force_generic))
        noise = None
        if not teacher_force and self.decode_mode not in (0, 1):
            # decode_mode 2 (reference las_model.py:229-234): one Categorical(raw_pred).sample() per step, which torch
            # evaluates as argmax_v p_v / q_v with q ~ Exp(1) drawn from the device's generator, one (B,V) draw per step.
            # ``sample_noise`` (U,B,V) lets a caller (or a test pinning the reference's stream) supply the draws.
            noise = self.sample_noise
            if noise is None:
                B = listener_feature.shape[0]
                noise = torch.stack([torch.empty(B, self.label_dim, device=listener_feature.device).exponential_(1)
                                     for _ in range(int(steps))])
        params = self._params()
        grad_on = torch.is_grad_enabled()
        labels = ground_truth if teacher_force else None
        B = listener_feature.shape[0]
        nb = self._decode_slice(listener_feature, params, cfg) if listener_feature.is_cuda else 0
        if nb <= 0 or B <= nb:
            return _SpellerFn.apply(cfg + (True, grad_on), listener_feature, labels, noise, *params)
        # Batches beyond what one launch of the decode kernels takes: slices of nb utterances, each decoded in ONE launch (the per-step
        # kernels would need U launch chains for the whole batch: P at B=128 26 ms -> see DESIGN.md 3.3).  The utterances of a batch
        # are independent in the Speller (reference las_model.py:205-236), so this is the same arithmetic per utterance; parameter
        # gradients of the slices are summed by autograd (no direct writes: every parameter is used once per slice).
        feats = listener_feature.split(nb, 0)
        labs = labels.split(nb, 0) if labels is not None else [None] * len(feats)
        noises = noise.split(nb, 1) if noise is not None else [None] * len(feats)
        outs = [_SpellerFn.apply(cfg + (False, grad_on), f, l, n, *params) for f, l, n in zip(feats, labs, noises)]
        return torch.cat([o[0] for o in outs], 1), torch.cat([o[1] for o in outs], 2)

    def _decode_slice(self, feat, params, cfg):
        (_U, teacher_forced, decode_mode, L, use_mlp, relu, M, V, heads, force_generic) = cfg
        if force_generic or feat.dim() != 3:
            return 0
        B, Tp, D = feat.shape
        lstm, rest = params[:4 * L], params[4 * L:]
        d = _speller_desc(B, Tp, D, lstm[1].shape[1], V, M, L, use_mlp, relu, [_f32c(p) for p in lstm], [_f32c(p) for p in rest], heads)
        return int(lib().las_speller_decode_batch(d, int(teacher_forced), int(decode_mode)))

    def forward(self, listener_feature, ground_truth=None, teacher_force_rate=0.9):
        if ground_truth is None:
            teacher_force_rate = 0
        # exactly one draw from the global NumPy RNG per call, like the reference (las_model.py:189)
        teacher_force = True if np.random.random_sample() < teacher_force_rate else False
        if (ground_truth is None) or (not teacher_force):
            max_step = self.max_label_len
        else:
            max_step = ground_truth.size()[1]
        if listener_feature.is_cuda:
            _cabi.poll_device_errors(listener_feature.device)
        logp, att = self._run(listener_feature, ground_truth, teacher_force, max_step)
        raw_pred_seq = list(logp.unbind(0))                 # list[U] of (B,V)   (callers cat them, solver.py:68)
        attention_record = [list(a.unbind(0)) for a in att.unbind(0)]     # list[U] of list[heads] of (B,T')
        return raw_pred_seq, attention_record

    def forward_step(self, input_word, last_hidden_state, listener_feature):
        """One decode step with caller-managed state (reference las_model.py:178-184): ``input_word`` (B,1,V+Hs),
        ``last_hidden_state`` = (h, c) each (L,B,Hs) or None, returns (raw_pred (B,V), (h, c), context (B,2H),
        [attention_score (B,T')]).  Differentiable like the reference's (autograd flows into ``input_word``, the state,
        the listener features and every parameter through ``las_speller_step_bwd``); under ``torch.no_grad()`` no stash
        is kept.  ``Speller.forward`` does not go through here: it runs the whole loop in the decode kernels."""
        a = self.attention
        use_mlp = bool(a.mlp_preprocess_input)
        if listener_feature.is_cuda:
            _cabi.poll_device_errors(listener_feature.device)
        B = listener_feature.shape[0]
        cfg = (int(self.num_layers), use_mlp, _ACT_CODES[a.activate], int(a.preprocess_mlp_dim) if use_mlp else 0, int(self.label_dim),
               int(a.multi_head))
        h_in = c_in = None
        if last_hidden_state is not None:
            h_in, c_in = last_hidden_state
        logp, h_out, c_out, ctx, att = _SpellerStepFn.apply(cfg, listener_feature, input_word.reshape(B, -1), h_in, c_in, *self._params())
        return logp, (h_out, c_out), ctx, list(att.unbind(0))


class LAS(nn.Module):
    """Reference model/las_model.py:24-63."""

    def __init__(self, listener, speller):
        super().__init__()
        self.listener = listener
        self.speller = speller

    def forward(self, batch_data, batch_label, teacher_force_rate, is_training=True):
        listener_feature = self.listener(batch_data)
        if is_training:
            raw_pred_seq, attention_record = self.speller(listener_feature, ground_truth=batch_label,
                                                          teacher_force_rate=teacher_force_rate)
        else:
            raw_pred_seq, attention_record = self.speller(listener_feature, ground_truth=None, teacher_force_rate=0)
        return raw_pred_seq, attention_record

    def serialize(self, optimizer, epoch, tr_loss, val_loss):
        enc, dec = self.listener, self.speller
        package = dict(einput=enc.input_feature_dim, ehidden=enc.hidden_size, elayer=enc.num_layers, edropout=enc.dropout_rate,
                       dvocab_size=dec.label_dim, dhidden=dec.hidden_size, dlayer=dec.num_layers,
                       state_dict=self.state_dict(), optim_dict=optimizer.state_dict(), epoch=epoch)
        # the reference's dict literal writes "etype" twice (listener's string, then the speller's class, :50,54): the
        # second assignment wins, so a package carries the Speller's rnn_unit class under "etype"
        package["etype"] = dec.rnn_unit
        if tr_loss is not None:
            package.update(tr_loss=tr_loss, val_loss=val_loss)
        return package
