"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half the bytes of a wide
coalesced streaming read (MI355X_MICROARCH.md, section HBM), so the read side is doubled;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
usage: python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


if __name__ == "__main__":
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes per launch; read = 2 x FETCH_SIZE x 1024 (gfx950 correction), write = WRITE_SIZE x 1024",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fr = fetch.get(k, [0.0])
        wr = write.get(k, [0.0])
        rd = 2.0 * 1024.0 * sum(fr) / len(fr)
        wb = 1024.0 * sum(wr) / len(wr)
        out["kernels"][short(k)] = {"launches": len(fr), "read_bytes": rd, "write_bytes": wb, "total_bytes": rd + wb}
    key = [k for k in out["kernels"] if k.startswith("fs::k_jacobi_block<0")]
    if key:
        out["k_jacobi_block_pressure_bytes_per_launch"] = out["kernels"][key[0]]["total_bytes"]
    js = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(js + "\n")
    for k, v in out["kernels"].items():
        print("%-40s n=%4d read %8.1f MB write %8.1f MB total %8.1f MB" % (k[:40], v["launches"], v["read_bytes"] / 1e6,
                                                                        v["write_bytes"] / 1e6, v["total_bytes"] / 1e6))
