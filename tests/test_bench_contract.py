"""bench.py's launch contract without a GPU: `python bench.py --gpus N` (N > 1) must start its own ranks -- round 1 died on
an assert when WORLD_SIZE was unset -- and relay their exit code; a hang must end with "hang": true and a non-zero exit
code (round 1's watchdog printed the headline and exited 0)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_gpus_n_spawns_ranks_and_relays_their_failure_without_a_gpu():
    env = dict(os.environ, PMG_BENCH_SPAWN_TIMEOUT="240")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-mgmc", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300)
    out = r.stdout + r.stderr
    assert "AssertionError" not in out
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if not have_gpu:
        # two child ranks were started by torch.distributed.run and each refused to run without a GPU
        assert r.returncode != 0
        assert out.count("bench.py needs a GPU") >= 2, out[-2000:]


def test_watchdog_reports_a_hang_with_a_nonzero_exit_code():
    code = (
        "import sys, time; sys.path.insert(0, %r); import bench\n"
        "dog = bench.Watchdog(0, 4, 20, 5)\n"
        "dog.out = {'metric': bench.METRIC, 'value': 123.0}\n"
        "dog.arm(0.2, 'secondary lines')\n"
        "time.sleep(30)\n" % str(ROOT)
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["hang"] is True and line["hang_stage"] == "secondary lines" and line["value"] == 123.0
    # before any headline exists the line still says so
    code2 = code.replace("dog.out = {'metric': bench.METRIC, 'value': 123.0}\n", "")
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=60)
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 3 and line["hang"] is True and line["value"] is None and line["n_gpus"] == 4
