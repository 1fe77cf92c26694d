"""From a rocprofv3 kernel trace of bench.py: the timeline of ONE steady-state frame (splat to splat) -- per kernel its
start offset, duration and the gap since everything before it had finished.  usage: frame_timeline.py <kernel_trace.csv> [frame]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_add_forces" in r[2]]
pairs = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if 10 <= b - a <= 60]
a, b = pairs[int(sys.argv[2]) if len(sys.argv) > 2 else (3 * len(pairs)) // 4]
t0, cur_end, idle = rows[a][0], rows[a][0], 0
for s, e, name in rows[a:b + 1]:
    gap = s - cur_end
    short = name.replace("void ", "").split("(")[0][:44]
    print("%8.1f us  +%6.1f us  gap %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3 if gap > 0 else 0.0, short))
    if gap > 0:
        idle += gap
    cur_end = max(cur_end, e)
print("frame %.1f us, idle %.1f us" % ((rows[b][0] - t0) / 1e3, idle / 1e3))
periods = sorted((rows[y][0] - rows[x][0]) / 1e3 for x, y in pairs[len(pairs) // 3:])
print("frame period over the last %d frames of the trace: median %.1f us, mean %.1f us, min %.1f, max %.1f"
      % (len(periods), periods[len(periods) // 2], sum(periods) / len(periods), periods[0], periods[-1]))
