"""Synthetic workload of the benchmark: Gaussian images shaped like the dataset contract of
train.py:195-200 (``x`` fp32 [N,3,S,S] ImageNet-normalised) and one filled, rotated square ("diamond" indentation) per mask with
1-12 % foreground like data/masks.  Host-side, seeded; the values are part of the parity contract: the oracle's generator draws
the same numbers (tests/test_host_cpu.py::test_synthetic_matches_oracle)."""
from __future__ import annotations

import math
import random

import numpy as np
import torch


def seed_everything(seed: int = 42) -> None:
    """The three seeds the reference sets before it builds the model (train.py:207-226, 560); the CUDA generator is seeded by
    ``torch.manual_seed`` as well."""
    for f in (random.seed, np.random.seed, torch.manual_seed):
        f(seed)


def synthetic_batch(n: int, size: int, seed: int = 1234):
    """(x [n,3,S,S] ~ N(0,1), y [n,1,S,S] in {0,1}) on the host.  Draw order: all of x, then one row per image of
    (centre x, centre y, half diagonal, angle) uniforms."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, size, size, generator=gen)
    draws = torch.rand(n, 4, generator=gen)
    col = torch.arange(size, dtype=torch.float32).view(1, 1, size)
    row = torch.arange(size, dtype=torch.float32).view(1, size, 1)
    per_image = lambda v: v.view(n, 1, 1)
    cx = per_image(0.3 + 0.4 * draws[:, 0]) * size           # centre in the middle 40 % of the frame
    cy = per_image(0.3 + 0.4 * draws[:, 1]) * size
    half = per_image(0.08 + 0.17 * draws[:, 2]) * size        # half diagonal 8-25 % of the side
    theta = per_image(draws[:, 3] * (math.pi / 2))
    c, s = torch.cos(theta), torch.sin(theta)
    u, v = col - cx, row - cy
    a = u * c + v * s                                         # coordinates in the square's own frame
    b = -u * s + v * c
    inside = (a.abs() + b.abs()) <= half                      # L1 ball = square standing on a corner
    return x, inside.float().unsqueeze(1)
