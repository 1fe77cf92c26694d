"""Mean duration and calls per frame of every kernel over the steady frames (splat to splat) of a rocprofv3 kernel trace.
usage: kernel_means.py <kernel_trace.csv>"""
import csv, sys, collections
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
starts = [i for i, r in enumerate(rows) if "k_add_forces" in r[2]]
starts = starts[len(starts) // 3:]
acc = collections.defaultdict(lambda: [0, 0.0])
for s, e, name in rows[starts[0]:starts[-1]]:
    k = name.replace("void ", "").split("(")[0][:60]
    acc[k][0] += 1
    acc[k][1] += (e - s) / 1e3
frames = len(starts) - 1
busy = 0.0
for k, (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %5.2f per frame  mean %7.2f us  %7.1f us per frame" % (k, n / frames, us / n, us / frames))
    busy += us / frames
print("%d frames, period %.1f us, sum of kernel time %.1f us per frame" % (frames, (rows[starts[-1]][0] - rows[starts[0]][0]) / 1e3 / frames, busy))
