"""Config 3 (config 2 + 1,048,576 tracers in the frame's graph), GPU time per frame.  usage: time_config3.py [frames] [KEY=VALUE options ...]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fluidsolver_amd import lib as L
a = [x for x in sys.argv[1:] if "=" not in x]
n = int(a[0]) if a else 300
wl = bench.WORKLOADS["config3"]
s, ps = bench.make_single(3072, 3072, wl["params"], wl["particles"])
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        s.set_option(getattr(L, "OPT_" + k), int(v))
for f in range(40):
    bench.frame_single(s, ps, f, 3072, 3072)
s.sync()
ms = C.c_float()
lib = L.load(); L.check(lib.fs_timer_begin(s.ctx), s.ctx)
for f in range(40, 40 + n):
    bench.frame_single(s, ps, f, 3072, 3072)
L.check(lib.fs_timer_end(s.ctx, C.byref(ms)), s.ctx)
s.sync()
print("config 3 %s: %.1f us/frame" % (" ".join(sys.argv[1:]), ms.value / n * 1e3), flush=True)
s.close()
