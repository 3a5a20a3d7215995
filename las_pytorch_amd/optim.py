"""``FusedClipAdam``: ``clip_grad_norm_(params, max_norm)`` + ``torch.optim.Adam.step()`` as TWO HIP launches over the
flat gradient buffer (``las_clip_adam``, include/las_hip.h) instead of ~12 elementwise ATen kernels.

Reference counterpart: ``torch.nn.utils.clip_grad_norm_(las_model.parameters(), 1)`` + ``optimizer.step()`` in
solver/solver.py:96-97 with ``torch.optim.Adam(las.parameters(), lr=...)`` from train.py:82.  Same update rule (amsgrad off,
weight decay 0), same ``state_dict`` layout as ``torch.optim.Adam`` (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), so
optimizer checkpoints move between the two.

It needs every gradient in one flat fp32 buffer: pass the model's ``FlatGradAllReducer`` (``las_pytorch_amd.dp``), which
keeps ``p.grad`` as views of that buffer.  ``solver.batch_iterator`` recognises the class and calls ``step_clipped`` in place
of ``reducer.clip_() ; optimizer.step()``.

Device-side safety: the update kernel reads the error word of the step's persistent kernels and leaves parameters and
moments untouched when it is set (a hand-off timeout invalidates the step's gradients); ``rollback_step`` then undoes the host's
step count and the caller re-runs the step on the generic kernels.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _cabi


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0):
        params = list(reducer.params)
        if not params or not params[0].is_cuda:
            raise RuntimeError("FusedClipAdam runs on the GPU (there is no CPU fallback)")
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("FusedClipAdam needs contiguous fp32 parameters")
        # the param-group keys of torch.optim.Adam ride along at their (only supported) defaults, so that this optimizer's
        # state_dict loads into a torch.optim.Adam — which adopts the checkpoint's groups as they are
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, max_norm=max_norm, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False))
        self.reducer = reducer
        dev = reducer.flat.device
        self.exp_avg = torch.zeros_like(reducer.flat)
        self.exp_avg_sq = torch.zeros_like(reducer.flat)
        self.total_norm = torch.zeros(1, device=dev)
        self._work = torch.empty(_cabi.lib().las_clip_adam_workspace_floats(), device=dev)
        self._steps = 0
        self._step_ts = []                        # one step tensor per parameter, as torch.optim.Adam keeps them
        offs, off = [0], 0
        for p in params:
            off += p.numel()
            offs.append(off)
        self._n = len(params)
        self._offsets = (C.c_int64 * (self._n + 1))(*offs)
        self._ptrs = (C.c_void_p * self._n)()
        self._link_state()

    def _link_state(self):
        """``self.state[p]`` in torch.optim.Adam's layout, the moments being views of the flat buffers."""
        off = 0
        # every parameter gets its OWN step tensor: torch.optim.Adam increments ``state[p]["step"]`` once per parameter, so a
        # checkpoint whose parameters share one tensor would advance by N per step after being loaded there
        self._step_ts = [torch.tensor(float(self._steps)) for _ in self.reducer.params]
        for p, st in zip(self.reducer.params, self._step_ts):
            n = p.numel()
            self.state[p] = dict(step=st, exp_avg=self.exp_avg[off:off + n].view_as(p),
                                 exp_avg_sq=self.exp_avg_sq[off:off + n].view_as(p))
            off += n

    def _set_steps(self, steps):
        self._steps = steps
        for st in self._step_ts:
            st.fill_(float(steps))

    def load_state_dict(self, state_dict):
        # las_clip_adam implements plain Adam only: refuse BEFORE anything is replaced (a caller that catches the error keeps an optimizer
        # whose state still aliases the flat moment buffers)
        for g in state_dict.get("param_groups", []):
            bad = [k for k in ("weight_decay", "amsgrad", "maximize") if g.get(k)]
            if bad:
                raise RuntimeError(f"FusedClipAdam: the checkpoint's param group sets {bad}, which las_clip_adam does not implement")
        super().load_state_dict(state_dict)       # replaces the state tensors by copies: pour them back into the flat buffers
        for g in self.param_groups:               # a torch.optim.Adam checkpoint has no clip threshold
            g.setdefault("max_norm", self.defaults["max_norm"])
        off, steps = 0, 0
        for p in self.reducer.params:
            st, n = self.state[p], p.numel()
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps = int(st["step"])
            off += n
        self._steps = steps
        self._link_state()

    @torch.no_grad()
    def step_clipped(self, max_norm=None):
        """Clip the flat gradient to ``max_norm`` (default: the constructor's) by its global norm and apply one Adam step.
        Returns the device tensor holding the gradient norm before clipping (no host synchronisation)."""
        g = self.param_groups[0]
        self.reducer.check_views()
        self._steps += 1
        for i, p in enumerate(self.reducer.params):
            self._ptrs[i] = p.data_ptr()
        dev = self.reducer.flat.device
        with torch.cuda.device(dev):
            _cabi.check(_cabi.lib().las_clip_adam(
                self._ptrs, self._offsets, self._n, self.reducer.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                float(g["max_norm"] if max_norm is None else max_norm), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                float(g["eps"]), self._steps, self.total_norm.data_ptr(), self._work.data_ptr(), self.reducer.error_flag_ptr(),
                _cabi.stream_ptr()))
        self._set_steps(self._steps)
        return self.total_norm

    def step(self, closure=None):
        """``optimizer.step()`` semantics (no clipping) for callers that clip themselves."""
        loss = closure() if closure is not None else None
        self.step_clipped(max_norm=0.0)
        return loss

    def rollback_step(self):
        """The last ``step_clipped`` was skipped on the device (error word set): take back the host's step count."""
        self._set_steps(max(0, self._steps - 1))

    def zero_grad(self, set_to_none=False):
        self.reducer.zero()      # keeps every p.grad a view of the flat buffer
