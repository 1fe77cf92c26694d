"""Deterministic synthetic input of SURVEY.md 8(d): a virtual pointer on a Lissajous path, one splat
per frame with force = (pos_f - pos_{f-1}) * forceMultiplier (the rule of FluidSolver.swift:417) and
colour = rainbow HSB((f mod 360)/360, 1, 1) * colorMultiplier (FluidSolver.swift:422-424 with the wall
clock replaced by the frame number).  No RNG.  Values are computed in fp64 and rounded to fp32 once,
so every consumer (HIP path, CPU oracle) receives identical float arguments.
"""
import math

import numpy as np

from .presets import hsb_to_rgb

F = np.float32


def pointer_pos(W, H, f, pointer=0, n_pointers=1):
    """Grid-unit position of pointer `pointer` at frame f.  Extra pointers (config 4: one per row slab)
    are phase-shifted and centred on their own horizontal band so every slab receives splats."""
    ph = 2.0 * math.pi * pointer / max(1, n_pointers)
    x = W * (0.5 + 0.35 * math.sin(2.0 * math.pi * 3.0 * f / 360.0 + ph))
    if n_pointers == 1:
        y = H * (0.5 + 0.35 * math.sin(2.0 * math.pi * 2.0 * f / 360.0 + math.pi / 2.0))
    else:
        band = H / n_pointers
        y = band * (pointer + 0.5 + 0.35 * math.sin(2.0 * math.pi * 2.0 * f / 360.0 + math.pi / 2.0 + ph))
    return F(x), F(y)


def splat(W, H, f, pointer=0, n_pointers=1, forceMultiplier=24.0, colorMultiplier=100.0):
    """(pos[2], force[2], color[3]) as fp32 for frame f >= 0; the first frame has zero force
    (first touch, FluidSolver.swift:415-419)."""
    px, py = pointer_pos(W, H, f, pointer, n_pointers)
    if f == 0:
        force = (F(0), F(0))
    else:
        qx, qy = pointer_pos(W, H, f - 1, pointer, n_pointers)
        force = ((px - qx) * F(forceMultiplier), (py - qy) * F(forceMultiplier))
    rgb = hsb_to_rgb(F(f % 360) / F(360.0), 1.0, 1.0)[:3] * F(colorMultiplier)
    return (px, py), force, (rgb[0], rgb[1], rgb[2])
