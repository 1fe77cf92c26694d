"""Print the kernels of the second-to-last frame of a rocprofv3 kernel trace (csv) of scripts/frames_only.py."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'add_forces' in r['Kernel_Name']]
i0, i1 = idx[-3], idx[-2]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    n = r['Kernel_Name'].replace('void fs::', '').replace('fs::', '')[:52]
    print("%8.1f %8.1f  q%s  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r.get('Queue_Id', '?'), n))
