"""Per-kernel means of rocprofv3 --pmc counters (one or more counter_collection.csv files, one pass each).
usage: python scripts/pmc_summary.py out.json <counter_collection.csv>...   (prints a table, writes the JSON)"""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


if __name__ == "__main__":
    acc = defaultdict(lambda: defaultdict(list))
    for path in sys.argv[2:]:
        with open(path) as f:
            for row in csv.DictReader(f):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {"units": "mean per launch (rocprofv3 sums over XCDs / SEs)", "kernels": {}}
    for k in sorted(acc):
        out["kernels"][k] = {c: sum(v) / len(v) for c, v in sorted(acc[k].items())}
        out["kernels"][k]["launches"] = max(len(v) for v in acc[k].values())
    open(sys.argv[1], "w").write(json.dumps(out, indent=1) + "\n")
    for k, v in out["kernels"].items():
        print(k[:70])
        for c, x in v.items():
            print("    %-28s %16.1f" % (c, x))
