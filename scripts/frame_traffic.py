"""HBM-side bytes of ONE config-2 frame from two PMC passes (FETCH_SIZE, WRITE_SIZE) over scripts/frames_only.py:
    rocprofv3 --pmc FETCH_SIZE -f csv -d <dirF> -o f -- python3 scripts/frames_only.py 30
    rocprofv3 --pmc WRITE_SIZE -f csv -d <dirW> -o f -- python3 scripts/frames_only.py 30
    python3 scripts/frame_traffic.py <dirF>/f_counter_collection.csv <dirW>/f_counter_collection.csv 30 profiles/frame_traffic.json
Read = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM),
write = WRITE_SIZE x 1024.  The context's set-up fills (memset kernels of the runtime) are left out."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd.build import source_sha_all


def total(path, counter):
    per = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and "__amd_rocclr" not in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
    return per


fetch, write, frames, out = total(sys.argv[1], "FETCH_SIZE"), total(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3]), sys.argv[4]
kern = {k: {"read_bytes_per_frame": 2048.0 * fetch.get(k, 0.0) / frames, "write_bytes_per_frame": 1024.0 * write.get(k, 0.0) / frames}
        for k in sorted(set(fetch) | set(write))}
tot = sum(v["read_bytes_per_frame"] + v["write_bytes_per_frame"] for v in kern.values())
json.dump({"source_sha_all": source_sha_all(), "frames": frames, "workload": "config 2 (scripts/frames_only.py), graph replays, two-band projection",
           "bytes_per_frame": tot, "kernels": kern}, open(out, "w"), indent=1)
print("%.1f MB per frame" % (tot / 1e6))
for k, v in kern.items():
    print("  %-60s read %8.1f MB  write %8.1f MB" % (k[:60], v["read_bytes_per_frame"] / 1e6, v["write_bytes_per_frame"] / 1e6))
