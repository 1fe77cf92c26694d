"""bench.py's own plumbing, without a GPU: the watchdog that turns a run which stops making progress into ONE JSON
line naming the phase it stopped in (a hung halo exchange on the driver's 8-GPU node must not be a silent hang)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_watchdog_reports_phase_on_one_json_line():
    env = dict(os.environ, FS_BENCH_WATCHDOG_S="0.01")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    # `import torch` alone outlives 10 ms: the watchdog fires in the first phase, before anything needs a GPU
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == 1
    assert "watchdog" in out["error"] and "phase 'start'" in out["error"]
