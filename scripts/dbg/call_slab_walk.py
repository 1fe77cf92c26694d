"""Walk `seed` of tests/test_gpu_native_slabs.py::test_random_walks_on_slabs_equal_single_gpu outside pytest, with torch's HIP
context brought up first as tests/conftest.py does, and a native backtrace on SIGSEGV (build scripts/dbg/segv_bt.c first:
gcc -O1 -g -shared -fPIC -o scripts/dbg/libsegv_bt.so scripts/dbg/segv_bt.c).  Walk 69 is the one that ends in the HIP runtime's
graph launch (profiles/r03_y_graph_launch_stream_pool.log).  usage: python scripts/dbg/call_slab_walk.py [seed] [notorch]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 69
if "notorch" not in sys.argv:
    import torch
    torch.cuda.init()
    torch.zeros(1, device="cuda")
bt = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsegv_bt.so")
if os.path.exists(bt):
    C.CDLL(bt).install_segv_bt()
os.environ["FS_FUZZ_SEEDS"] = str(seed)
import test_gpu_native_slabs as T
T.test_random_walks_on_slabs_equal_single_gpu(seed)
print("walk", seed, "survived")
