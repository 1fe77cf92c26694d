"""How much of the dye advection (DRAM-bound) hides behind the diffuse loop (VALU-bound, cache-resident) when both
run at once?  Raw fs_k_* calls on three streams at 3072^2: {diffuse u || diffuse v} then advect(dye), against
{diffuse u || diffuse v || advect(dye)}.  A measurement for the 'dye advection of frame f under the velocity chain
of frame f+1' idea; nothing in the library uses it."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsolver_amd import lib as L
lib = L.load()
W = H = 3072
g = L.Grid(W, H, W, 0, H, 0, H)
gen = torch.Generator(device="cuda").manual_seed(1)
vel = (torch.rand((2, H, W), generator=gen, device="cuda") * 2 - 1) * 0.01
vel_t = torch.zeros_like(vel)
vadv = (torch.rand((2, H, W), generator=gen, device="cuda") * 2 - 1) * 20.0     # the velocity the dye is traced through
dye = torch.rand((3, H, W), generator=gen, device="cuda")
dye_t = torch.zeros_like(dye)
err = torch.zeros(1, dtype=torch.int32, device="cuda")
s = [torch.cuda.Stream() for _ in range(3)]
sp = [C.c_void_p(x.cuda_stream) for x in s]
plane = W * H
P = lambda t, i=0: C.c_void_p(t[i].data_ptr())
ib = C.c_int()

def diffuse(i, st):
    L.check(lib.fs_k_diffuse_n(C.byref(g), P(vel, i), P(vel_t, i), 1, plane, 1.5e-4, 0.5, 40, C.byref(ib), st))
def advect(st):
    L.check(lib.fs_k_advect_fade(C.byref(g), P(vadv, 0), P(vadv, 1), P(dye), P(dye_t), 3, plane, 0.5, 0.002, C.c_void_p(err.data_ptr()), st))

def run(mode, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(s[0])
    for _ in range(reps):
        f = torch.cuda.Event(); f.record(s[0]); s[1].wait_event(f); s[2].wait_event(f)
        diffuse(0, sp[0]); diffuse(1, sp[1])
        if mode == "together":
            advect(sp[2])
        j1, j2 = torch.cuda.Event(), torch.cuda.Event()
        j1.record(s[1]); j2.record(s[2]); s[0].wait_event(j1); s[0].wait_event(j2)
        if mode == "sequential":
            advect(sp[0])
        if mode == "diffuse only":
            pass
    e1.record(s[0])
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for rep in range(3):
    for mode in ("diffuse only", "sequential", "together"):
        t = run(mode)
        if rep: print("%-14s %7.1f us" % (mode, t), flush=True)
