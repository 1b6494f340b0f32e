"""Iteration -> render mode / densification switches of the fitting loop.

Same schedule as reference utils/train_util.py:8-92 driven by the same OptimizationParams fields; restated
as pure functions of the iteration plus a thin stateful wrapper with the reference's property names.
"""
from __future__ import annotations

from .generate import GenerateMode


def render_mode_at(it: int, opt):
    a = opt.full_precision_training_total
    b = a + opt.quantized_training_total
    c = b + opt.entropy_constrained_train_total
    d = c + opt.ste_entropy_constrained_train_total
    if it <= a:
        return GenerateMode.TRAINING_FULL_PRECISION
    if it <= b:
        return GenerateMode.TRAINING_QUANTIZED
    if it <= c:
        return GenerateMode.TRAINING_ENTROPY
    if it <= d:
        return GenerateMode.TRAININ_STE_ENTROPY
    return None


def gaussian_statis_at(it: int, opt) -> bool:
    fp = opt.full_precision_training_total
    if fp <= it < fp + opt.pause_densification:
        return False
    return opt.update_until > it > opt.start_stat


def adjust_anchor_at(it: int, opt) -> bool:
    fp = opt.full_precision_training_total
    if it >= opt.update_until:
        return False
    if fp <= it <= fp + opt.pause_densification:
        return False
    return it > opt.update_from and it % opt.update_interval == 0


class TrainingController:
    def __init__(self, opt_params):
        self.current_iteration = 0
        self.opt_params = opt_params
        self._entropy_constrained = False

    @property
    def render_mode(self):
        mode = render_mode_at(self.current_iteration, self.opt_params)
        if mode in (GenerateMode.TRAINING_ENTROPY, GenerateMode.TRAININ_STE_ENTROPY):
            self._entropy_constrained = True
        return mode

    @property
    def entropy_constrained(self):
        return self._entropy_constrained

    @property
    def gaussian_statis(self):
        return gaussian_statis_at(self.current_iteration, self.opt_params)

    @property
    def gaussian_adjust_anchor(self):
        return adjust_anchor_at(self.current_iteration, self.opt_params)

    @property
    def clean_denorm(self):
        return self.current_iteration == self.opt_params.update_until

    def step(self):
        self.current_iteration += 1
