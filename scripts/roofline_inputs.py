"""Merge the outputs of scripts/profile_jacobi.sh (PMC means + kernel-trace stats, one set per grid size) into ONE file
bench.py reads: HBM bytes per launch, traced duration and the SQ counters of the two Jacobi kernels, stamped with the
sha of the kernel sources they were measured on (fluidsolver_amd.build.source_sha).
usage (on the GPU box, after profile_jacobi.sh <tag> 3072 and ... 16384):
    python3 scripts/roofline_inputs.py <tag> gpurun_out/<out>.json"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsolver_amd.build import source_sha   # noqa: E402

CLOCK_GHZ = 2.4   # nominal; the chip holds ~2.3 GHz under an all-VALU load (profiles/r02_valu_issue.log)
SIMDS = 256 * 4


def kernel_rows(path):
    with open(path) as f:
        return {r["Name"]: r for r in csv.DictReader(f)}


if __name__ == "__main__":
    tag, out_path = sys.argv[1], sys.argv[2]
    out = {"source_sha": source_sha(), "collected_by": "scripts/profile_jacobi.sh %s <size>; scripts/roofline_inputs.py" % tag,
           "units": "bytes and counters are means per launch; hbm_bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (gfx950: "
                    "FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM); SQ_* cycle counters "
                    "are quad-cycles summed over all waves / SIMDs", "kernels": {}}
    for size in (3072, 16384):
        base = os.path.join(ROOT, "gpurun_out", "jac_%s_%d" % (tag, size))
        if not os.path.exists(base + "_pmc.json"):
            continue
        pmc = json.load(open(base + "_pmc.json"))["kernels"]
        stats = kernel_rows(base + "_kernel_stats.csv")
        for op, label in ((0, "pressure"), (1, "diffuse")):
            names = [k for k in pmc if k.startswith("fs::k_jacobi") and ("<%d," % op) in k]
            if not names:
                continue
            k = max(names, key=lambda n: pmc[n]["launches"])
            c = pmc[k]
            traced = [r for n, r in stats.items() if k in n]
            ns = float(traced[0]["AverageNs"]) if traced else None
            hbm = 2.0 * 1024.0 * c["FETCH_SIZE"] + 1024.0 * c["WRITE_SIZE"]
            e = {"kernel": k, "grid": size, "launches_profiled": c["launches"], "fetch_bytes": 2.0 * 1024.0 * c["FETCH_SIZE"],
                 "write_bytes": 1024.0 * c["WRITE_SIZE"], "hbm_bytes": hbm, "traced_ns": ns,
                 "sq": {n: v for n, v in c.items() if n.startswith("SQ_") or n.startswith("GRBM_")}}
            if ns:
                e["hbm_frac_traced"] = hbm / (ns * 1e-9) / 8e12
                # VALU pipe occupancy: busy quad-cycles x 4 / (SIMDs x cycles of the launch)
                e["valu_busy_frac_traced"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * ns * CLOCK_GHZ)
                e["valu_cycles_per_inst"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
            out["kernels"]["%s_%d" % (label, size)] = e
    json.dump(out, open(out_path, "w"), indent=1)
    for n, e in out["kernels"].items():
        print("%-16s %8.1f us  %8.1f MB  hbm_frac %.3f  valu_busy %.3f" % (n, (e["traced_ns"] or 0) / 1e3, e["hbm_bytes"] / 1e6,
                                                                         e.get("hbm_frac_traced", 0), e.get("valu_busy_frac_traced", 0)))
