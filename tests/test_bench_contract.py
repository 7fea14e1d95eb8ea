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
        # child ranks were started by torch.distributed.run and refused to run without a GPU (the launcher ends the
        # sibling as soon as the first rank has failed, so the second refusal is not always printed)
        assert r.returncode != 0
        assert out.count("bench.py needs a GPU") >= 1, out[-2000:]


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


import pytest


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields():
    """one small end-to-end run of bench.py on the GPU: ONE JSON line with the driver's fields, `roofline` and `cpu_baseline`"""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--grid-n", "64", "--steps", "4", "--warmup", "2", "--no-mgmc", "--cpu-n", "32"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1) < 1e-9 and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["algorithmic_bytes_per_launch"] == 12 * 64 ** 3
    sc = rf["stream_ceiling"]  # the measured ceiling of the access mix beside the vendor peak
    assert sc["unit"] == "GB/s" and sc["checked"] is True and 0 < sc["achieved"] < rf["peak"] and abs(sc["sweep_over_ceiling"] - rf["achieved"] / sc["achieved"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb and cb["unit"] == "samples/s"
    assert d["finite"] is True and d["clock_settle"]["settle_launches"] > 0


def test_multi_gpu_record_is_self_describing_by_construction():
    """round 4: for N > 1 the line carries, per rank, the device index, its PCI bus id, peer access to the
    z-neighbours' devices, the rank's own timed-region seconds and its halo-wait polls, plus ranks_seen (torch's world size,
    ncclCommCount on the rccl transport) and min / median / max of the per-rank times.  Without a GPU: the C-ABI entry point
    the record is read from exists with the struct layout the Python side assumes, and bench.py assembles every field."""
    import ctypes as C

    from parmgmc_amd import capi

    assert "pmg_dist_describe" in capi.declared_symbols() and hasattr(capi.lib, "pmg_dist_describe")
    assert C.sizeof(capi.DistDescription) == 8 * 4 + 8 + 32 + 8  # pmg_dist_description in include/parmgmc_hip.h
    hdr = (ROOT / "include" / "parmgmc_hip.h").read_text()
    assert "int32_t  rank, nranks, device, neighbour[2], peer_access[2], rccl_comm_count;" in hdr and "char     pci_bus_id[32], transport[8];" in hdr
    keys = set(capi.DistDescription().as_dict())
    assert {"rank", "device", "pci_bus_id", "peer_access_lo_hi", "neighbour_ranks", "transport", "rccl_comm_count", "halo_wait_polls"} <= keys
    src = (ROOT / "bench.py").read_text()
    for field in ('"ranks_seen"', '"torch_world_size"', '"rccl_comm_count"', '"timed_region_s"', '"min"', '"median"', '"max"', '"halo_wait_polls_total"', '"ranks"', '"local_rank"', "all_gather_object"):
        assert field in src, field
    cdrv = (ROOT / "examples" / "pmg_bench.c").read_text()
    assert "pmg_dist_describe" in cdrv and "pci_bus_id" in cdrv and "halo_wait_polls" in cdrv


@pytest.mark.gpu
def test_two_ranks_sharing_the_gpu_print_the_self_describing_record():
    """the one-GPU rehearsal of `bench.py --gpus 2` (tools/bench_shared_gpu.sh): the N > 1 fields are there and sane"""
    env = dict(os.environ, PMG_BENCH_SHARE_DEVICE="1", PMG_BENCH_SPAWN_TIMEOUT="500")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--grid-n", "64", "--steps", "6", "--warmup", "2", "--no-mgmc", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["halo_check"].startswith("bit-identical")
    assert d["ranks_seen"]["torch_world_size"] == 2 and d["ranks_seen"]["records"] == 2 and d["ranks_seen"]["distinct_devices"] == 1  # shared device
    assert d["timed_region_s"]["min"] <= d["timed_region_s"]["median"] <= d["timed_region_s"]["max"]
    assert d["halo_wait_polls_total"] >= 0
    rk = sorted(d["ranks"], key=lambda x: x["rank"])
    assert [x["rank"] for x in rk] == [0, 1] and all(x["transport"] == "ipc" and x["device"] == 0 and len(x["pci_bus_id"]) >= 7 for x in rk)
    assert rk[0]["neighbour_ranks"] == [-1, 1] and rk[1]["neighbour_ranks"] == [0, -1]
    assert rk[0]["peer_access_lo_hi"] == [-1, 1] and rk[1]["peer_access_lo_hi"] == [1, -1]
