"""Builds libfluidsolver_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m fluidsolver_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfluidsolver_hip.so")
LIB_PROBE = os.path.join(HERE, "libfluidsolver_hip_probe.so")   # -DFS_PROBE: timing probes compiled in (scripts/ only)
SOURCES = ["passes.hip", "jacobi.hip", "context.hip", "slab.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "ctx.h"), os.path.join(ROOT, "include", "fluidsolver.h")]

# -ffp-contract=off: no FMA is formed, so every fp32 op rounds exactly like the CPU oracle.
# hipcc's defaults already give correctly rounded fp32 divide/sqrt and keep denormals.
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def source_sha():
    """sha256 over what the Jacobi kernels are compiled from (jacobi.hip, common.h, the build flags): what the PMC
    profile under profiles/roofline_inputs.json was measured on.  bench.py prints PMC-derived figures only when this
    matches the tree it runs from."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted([os.path.join(CSRC, "jacobi.hip"), os.path.join(CSRC, "common.h")]):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS[:6]).encode())
    return h.hexdigest()[:16]


def source_sha_all():
    """sha256 over every kernel source, the private headers and the build flags: what a whole-frame profile was measured on."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted([os.path.join(CSRC, s) for s in SOURCES] + HEADERS[:2]):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS[:6]).encode())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, probe=False):
    if not probe and not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + (["-DFS_PROBE"] + os.environ.get("FS_PROBE_DEFINES", "").split() if probe else []) + ["-o", LIB_PROBE if probe else LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PROBE if probe else LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, probe="--probe" in sys.argv)
