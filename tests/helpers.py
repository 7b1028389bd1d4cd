"""Shared test inputs: small seeded scenes the CPU oracle finishes in seconds."""
import math

import torch

from mi3dgs import scenes


def small_scene(n=400, seed=0, width=64, height=48, n_views=2, fx=60.0, big=False):
    """Random blob seen by a ring of cameras; scales large enough that splats cover many tiles."""
    g = torch.Generator().manual_seed(seed)
    means = torch.rand(n, 3, generator=g) * 2.0 - 1.0
    lo, hi = (0.05, 0.4) if big else (0.03, 0.2)
    ls = torch.rand(n, 3, generator=g) * (math.log(hi) - math.log(lo)) + math.log(lo)
    P = scenes._params(means, ls, g)
    P["shN"] = P["shN"] * 3.0
    vms, ks = scenes.ring_cameras(n_views, 4.0, 0.5, 2.0, fx, width, height, g)
    return scenes.Scene("test", P, vms, ks, width, height)


def activated(P, dtype=torch.float64):
    return dict(means=P["means"].to(dtype), quats=P["quats"].to(dtype), scales=P["scales"].to(dtype).exp(),
                opacities=torch.sigmoid(P["opacities"].to(dtype)),
                sh=torch.cat([P["sh0"], P["shN"]], dim=1).to(dtype))


def rel_err(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))
