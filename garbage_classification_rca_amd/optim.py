"""Fused optimizers over the engine's flat arenas (torch.optim.SGD / AdamW as built at main_both.py:544-549).

They subclass ``torch.optim.Optimizer`` so that ``param_groups[...]['lr']`` edits (main_both.py:700-701) and
``ReduceLROnPlateau`` (main_both.py:560, 769) keep working, but ``step()`` is ONE kernel launch per trainable span that
also refreshes the bf16 working copy, and ``zero_grad()`` is one memset that keeps the ``.grad`` views alive.
Parameters outside the arena (the reference's unused head layers) never receive gradients and are skipped, exactly as
torch skips ``grad is None``.
"""
import torch

from . import lib as L


class _FlatOptimizer(torch.optim.Optimizer):
    def __init__(self, model, defaults):
        self.model = model
        self.engine = model.engine
        super().__init__([p for p in model.parameters()], defaults)
        self._step = 0

    def _spans(self):
        tt, ti = self.model._train_flags()
        return self.engine.trainable_spans(tt, ti)

    def _named_spans(self):
        """(name, lo, hi) of the trainable arena spans, never merged: text encoder | image encoder | head."""
        tt, ti = self.model._train_flags()
        i0, h0 = self.engine.groups["image_emb"][0], self.engine.groups["head"][0]
        out = []
        if tt:
            out.append(("text", 0, i0))
        if ti:
            out.append(("image", i0, h0))
        out.append(("head", h0, self.engine.arena.total))
        return out

    @torch.no_grad()
    def zero_grad(self, set_to_none: bool = False):
        self.engine.arena.g.zero_()
        if any(p.grad is None for p in self.model._arena_params.values()):
            self.model._attach_grads()


class FlatSGD(_FlatOptimizer):
    def __init__(self, model, lr=1e-3, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        gp = self.param_groups[0]
        ar = self.engine.arena
        for a, b in self._spans():
            L.sgd_step(ar.p[a:b], ar.g[a:b], None if ar.lp is None else ar.lp[a:b], b - a, float(gp["lr"]),
                       float(gp["weight_decay"]), grad_scale, lp_lo=(None if ar.lp_lo is None else ar.lp_lo[a:b]))
        if ar.lp is not None:
            # the kernel rewrote the bf16 copy of every stepped span; un-stepped spans did not change since the
            # forward that validated the copy
            ar.lp_valid = True


class FlatAdamW(_FlatOptimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(model, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        ar = self.engine.arena
        self.m = torch.zeros_like(ar.p)
        self.v = torch.zeros_like(ar.p)
        # torch.optim.AdamW keeps a step count PER PARAMETER that starts when the parameter first receives a gradient
        # (main_both.py:544-549 builds it over all parameters; frozen ones have grad None and are skipped).  The encoders
        # are frozen for --epochs and unfrozen afterwards (main_both.py:690-697): their bias correction must start at
        # t = 1 then, not at the head's count.  One counter per span that freezes / unfreezes as a unit.
        self._span_steps = {"text": 0, "image": 0, "head": 0}

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        gp = self.param_groups[0]
        ar = self.engine.arena
        self._step += 1
        for name, a, b in self._named_spans():
            self._span_steps[name] += 1
            L.adamw_step(ar.p[a:b], ar.g[a:b], self.m[a:b], self.v[a:b], None if ar.lp is None else ar.lp[a:b], b - a,
                         float(gp["lr"]), float(gp["betas"][0]), float(gp["betas"][1]), float(gp["eps"]),
                         float(gp["weight_decay"]), self._span_steps[name], grad_scale,
                         lp_lo=(None if ar.lp_lo is None else ar.lp_lo[a:b]))
        if ar.lp is not None:
            ar.lp_valid = True
