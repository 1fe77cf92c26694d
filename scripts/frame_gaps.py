"""From a rocprofv3 kernel trace of bench.py: per frame (splat to splat), busy time (union of kernel intervals),
idle time, and the time during which two kernels overlap.  usage: frame_gaps.py <bench_kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_add_forces" in r[2]]
frames = []
for a, b in zip(starts[:-1], starts[1:]):
    ks = rows[a:b]
    if len(ks) < 10 or len(ks) > 40:
        continue
    t0, t1 = ks[0][0], rows[b][0]
    busy, overlap, cur_end = 0, 0, t0
    for s, e, _ in ks:
        if s >= cur_end:
            busy += e - s
        else:
            overlap += min(e, cur_end) - s
            busy += max(0, e - cur_end)
        cur_end = max(cur_end, e)
    frames.append((t1 - t0, busy, t1 - t0 - busy, overlap, len(ks), sum(e - s for s, e, _ in ks)))
frames = frames[len(frames) // 2:]          # steady state (graph replays)
n = len(frames)
print("%d frames: wall %.1f us, busy %.1f us, idle %.1f us, two kernels at once %.1f us, kernels/frame %.1f, sum of kernel times %.1f us"
      % (n, *(sum(f[i] for f in frames) / n / 1e3 for i in range(4)), sum(f[4] for f in frames) / n, sum(f[5] for f in frames) / n / 1e3))
