"""Helpers for the GPU parity tests: pitched device planes via torch (device-memory plumbing only)."""
import ctypes as C

import numpy as np
import torch

from fluidsolver_amd import lib as L

F = np.float32


def dev_planes(arr, pitch=None, pad=float("nan")):
    """numpy [P,rows,W] (or [rows,W]) -> cuda tensor [P,rows,pitch]; the padding columns are NaN so a
    kernel that let padding leak into the domain would be caught."""
    a = np.asarray(arr, dtype=F)
    if a.ndim == 2:
        a = a[None]
    P, rows, W = a.shape
    pitch = pitch or L.pitch_for_width(W)
    t = torch.full((P, rows, pitch), pad, dtype=torch.float32)
    t[:, :, :W] = torch.from_numpy(a)
    return t.cuda()


def host_planes(t, W):
    return t[:, :, :W].cpu().numpy()


def ptr(t, plane=0):
    return C.c_void_p(t[plane].data_ptr())


def stride(t):
    return t.stride(0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def grid(W, H, pitch=None, y0=0, rows=None, yb=None, ye=None):
    rows = H if rows is None else rows
    yb = y0 if yb is None else yb
    ye = y0 + rows if ye is None else ye
    return L.Grid(W, H, pitch or L.pitch_for_width(W), y0, rows, yb, ye)


def sync():
    torch.cuda.synchronize()
