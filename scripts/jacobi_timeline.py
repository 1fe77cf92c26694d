"""Per-workgroup timeline of ONE k_jacobi_block launch (probe build: FS_PROBE_LIB=1), from the 100 MHz clock stamps
the kernel leaves in g_timeline: entry / tile loaded / sweeps done / exit, plus HW_ID and XCC_ID.  Answers how many
workgroups a CU really runs at a time, how long the load / sweep / store phases of a workgroup last and how much of
the launch a CU sits empty.
usage: FS_PROBE_LIB=1 python3 scripts/jacobi_timeline.py [--size 3072] [--op pressure|diffuse] [--depth N]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=3072)
    ap.add_argument("--op", default="pressure")
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--variant", type=int, default=-1)
    a = ap.parse_args()
    assert os.environ.get("FS_PROBE_LIB") == "1", "needs the probe build: FS_PROBE_LIB=1"
    import bench
    from fluidsolver_amd import lib as L
    lib = L.load()
    lib.fs_probe_timeline.restype = C.c_int
    lib.fs_probe_timeline.argtypes = [C.c_void_p, C.c_long]
    W = H = a.size
    s, _ = bench.make_single(W, H, (1.5e-4, 0.0, 0.5, 10.0, 0.002, 40), 0)
    key = L.OPT_BLOCK_DEPTH_PRESSURE if a.op == "pressure" else L.OPT_BLOCK_DEPTH_DIFFUSE
    if a.depth:
        s.set_option(key, a.depth)
    if a.variant >= 0:
        s.set_option(L.OPT_TILE_VARIANT, a.variant)
    depth = a.depth or lib.fs_config_get(key)
    if a.op == "pressure":
        div = np.empty((1, H, W), np.float32)
        lib.fs_harness_fill_uniform(div.ctypes.data_as(C.c_void_p), div.size, -1.0, 1.0, 1234)
        s.write_field(L.FIELD_DIVERGENCE, div)
        stage = L.STAGE_PRESSURE_CLEARED
    else:
        vel = np.empty((2, H, W), np.float32)
        lib.fs_harness_fill_uniform(vel.ctypes.data_as(C.c_void_p), vel.size, -1.0, 1.0, 4321)
        s.write_field(L.FIELD_VELOCITY, vel)
        stage = L.STAGE_DIFFUSE_VELOCITY
    # 3 launches back to back (the first of a pressure loop reads no p): the timeline kept is the LAST one's
    for _ in range(3):
        s.stage(stage, 3 * depth)
    s.sync()
    n = 8192
    tl = np.zeros((n, 8), np.uint64)
    L.check(lib.fs_probe_timeline(tl.ctypes.data_as(C.c_void_p), tl.size))
    tl = tl[tl[:, 0] > 0]
    t0 = tl[:, 0].min()
    t = (tl[:, :4].astype(np.int64) - int(t0)) * 0.01        # us
    hw, xcc = tl[:, 4].astype(np.int64), tl[:, 5].astype(np.int64) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print("workgroups %d, distinct CUs %d, launch span %.2f us (first entry -> last exit)" % (len(t), len(set(cuid.tolist())), t[:, 3].max()))
    life, load, sweep, store = t[:, 3] - t[:, 0], t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    for name, v in (("lifetime", life), ("load", load), ("sweeps", sweep), ("store", store)):
        print("  %-9s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (name, v.mean(), *np.percentile(v, [10, 50, 90]), v.max()))
    print("  sweeps phase per sweep: %.3f us (depth %d)" % (sweep.mean() / depth, depth))
    # entry times: how staggered is the dispatch?
    order = np.sort(t[:, 0])
    print("  entry times: first 512 workgroups entered within %.2f us; workgroup #513 at %.2f us, last at %.2f us"
          % (order[min(511, len(order) - 1)], order[min(512, len(order) - 1)], order[-1]))
    # concurrency per CU over the launch
    span = t[:, 3].max()
    grid_t = np.arange(0.0, span, 0.05)
    per_cu = {}
    for i in range(len(t)):
        per_cu.setdefault(int(cuid[i]), []).append((t[i, 0], t[i, 3]))
    conc = np.zeros((len(per_cu), len(grid_t)))
    for k, (c, iv) in enumerate(sorted(per_cu.items())):
        for (b, e) in iv:
            conc[k] += (grid_t >= b) & (grid_t < e)
    print("  workgroups resident per CU, averaged over the span: %.2f (max seen %d); share of CU-time with 0 / 1 / 2 resident: %.2f / %.2f / %.2f"
          % (conc.mean(), conc.max(), (conc == 0).mean(), (conc == 1).mean(), (conc >= 2).mean()))
    wg_per_cu = np.array([len(v) for v in per_cu.values()])
    print("  workgroups per CU: min %d max %d  histogram %s" % (wg_per_cu.min(), wg_per_cu.max(), np.bincount(wg_per_cu).tolist()))
    last = np.array([max(e for _, e in v) for v in per_cu.values()])
    first_free = np.array([sorted(e for _, e in v)[-2] if len(v) > 1 else 0.0 for v in per_cu.values()])
    print("  last exit per CU: p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us; the CU's other slot fell idle at p50 %.2f us"
          % (*np.percentile(last, [10, 50, 90]), last.max(), np.percentile(first_free, 50)))
    print("  by XCC: " + "  ".join("x%d: %d wg, last %.1f" % (x, (xcc == x).sum(), t[xcc == x, 3].max()) for x in sorted(set(xcc.tolist()))))
    late = sorted(per_cu, key=lambda c: -max(e for _, e in per_cu[c]))[:4]
    for c in late:
        print("  late CU %4d: " % c + "  ".join("[%.2f %.2f]" % iv for iv in sorted(per_cu[c])))
    for lo, hi in ((0, 6), (6, 14), (14, 22), (22, 99)):
        m = (t[:, 0] >= lo) & (t[:, 0] < hi)
        if m.any():
            print("  workgroups entering in [%d, %d) us: %4d, load mean %.2f  sweeps mean %.2f  lifetime mean %.2f" % (lo, hi, m.sum(), load[m].mean(), sweep[m].mean(), life[m].mean()))
    edge = tl[:, 6].astype(bool)
    tx, ty = tl[:, 7].astype(np.int64) & 0xFFFF, tl[:, 7].astype(np.int64) >> 16
    print("  edge tiles %d of %d: lifetime mean %.2f (edge) vs %.2f (interior); sweeps %.2f vs %.2f; load %.2f vs %.2f"
          % (edge.sum(), len(edge), life[edge].mean(), life[~edge].mean(), sweep[edge].mean(), sweep[~edge].mean(), load[edge].mean(), load[~edge].mean()))
    for c in late:
        idx = [i for i in range(len(t)) if int(cuid[i]) == c]
        print("  late CU %4d tiles: " % c + "  ".join("(%d,%d)%s" % (tx[i], ty[i], "E" if edge[i] else "") for i in sorted(idx, key=lambda i: t[i, 0])))
    # a few CUs in full
    for c in sorted(per_cu)[:3]:
        print("  CU %4d: " % c + "  ".join("[%.2f %.2f]" % iv for iv in sorted(per_cu[c])))
    s.close()
