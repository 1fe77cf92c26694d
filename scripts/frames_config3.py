"""N frames of config 3 (config 2 + 1,048,576 tracers in the frame's graph): the program for kernel traces.  usage: frames_config3.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
wl = bench.WORKLOADS["config3"]
s, ps = bench.make_single(3072, 3072, wl["params"], wl["particles"])
for f in range(n):
    bench.frame_single(s, ps, f, 3072, 3072)
s.sync(); s.close()
