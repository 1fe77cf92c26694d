"""What the slab frame's CFL bound costs the projection: fs_k_project_n against fs_k_project_n_cfl on the same planes,
microseconds per loop (GPU events).  usage: probe_cfl_cost.py [W] [H] [sweeps] [reps]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidsolver_amd import lib as L
W = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
N = int(sys.argv[3]) if len(sys.argv) > 3 else 40
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
lib = L.load()
pitch = (W + 63) // 64 * 64
g = L.Grid(W, H, pitch, 0, H, 0, H)
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(3)
def plane(): return torch.rand((H, pitch), device=dev, generator=gen) * 2 - 1
u0, v0 = plane(), plane()
u, v, div, pa, pb = (torch.empty_like(u0) for _ in range(5))
cfl = torch.zeros(1, device=dev)
flag = C.c_int()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def loop(with_cfl):
    u.copy_(u0); v.copy_(v0)
    if with_cfl:
        L.check(lib.fs_k_project_n_cfl(C.byref(g), u.data_ptr(), v.data_ptr(), div.data_ptr(), pa.data_ptr(), pb.data_ptr(), N,
                                       cfl.data_ptr(), C.byref(flag), st))
    else:
        L.check(lib.fs_k_project_n(C.byref(g), u.data_ptr(), v.data_ptr(), div.data_ptr(), pa.data_ptr(), pb.data_ptr(), N, C.byref(flag), st))
for which in (False, True, False, True):
    for _ in range(10): loop(which)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): loop(which)
    b.record(); torch.cuda.synchronize()
    print("%dx%d N=%d  cfl=%d: %.1f us per loop (two plane copies included)" % (W, H, N, which, a.elapsed_time(b) / reps * 1e3), flush=True)
print("bound", float(cfl.item()), "max|v|", float(v[:, :W].abs().max().item()))
