"""ISA check of the in-kernel halo hand-shake ("ipc" transport, kernels_grid.hip).

Every wavefront of a face block stores its results into the z-neighbour's receive block over xGMI (system-scope
write-through stores, `sc0 sc1`) and the block then reports with ONE counter increment behind a workgroup barrier.
The neighbour's flag word may only be raised after ALL of those stores have completed, so each wavefront must drain
its own stores (`s_waitcnt vmcnt(0)`) BEFORE the barrier: on gfx950 a workgroup-scope fence + s_barrier waits for
lgkmcnt only.  hipcc cross-compiles without a GPU, so this runs in the CPU suite.

(replaces the per-colour VecScatterBegin/End of reference src/mc_sor.c:318-319)
"""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "parmgmc_amd" / "csrc" / "kernels_grid.hip"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "kernels_grid.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S", str(SRC), "-o", str(out)], stderr=subprocess.DEVNULL)
    return out.read_text()


def _functions(text):
    """split the assembly into (symbol, body lines) per kernel"""
    cur, body, res = None, [], []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            if cur:
                res.append((cur, body))
            cur, body = m.group(1), []
        elif cur is not None:
            body.append(line.strip())
            if line.strip().startswith("s_endpgm") and False:
                pass
    if cur:
        res.append((cur, body))
    return res


def test_face_wavefronts_drain_peer_stores_before_the_block_reports(isa):
    checked = 0
    for name, body in _functions(isa):
        if "grid_color_sweep_kernel" not in name:
            continue
        bars = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
        peer = [i for i, l in enumerate(body) if l.startswith("global_store_dwordx2") and "sc0 sc1" in l]
        if not bars:
            # no barrier = not a HALO instantiation: it must not contain peer plane stores either
            assert not peer, f"{name}: peer stores without the reporting barrier"
            continue
        assert peer, f"{name}: reporting barrier but no system-scope plane stores"
        for b in bars:
            before = [i for i in peer if i < b]
            if not before:
                continue
            window = body[before[-1] + 1 : b]
            waits = [l for l in window if l.startswith("s_waitcnt") and "vmcnt(0)" in l]
            assert waits, f"{name}: no `s_waitcnt vmcnt(0)` between the last sc0 sc1 plane store and s_barrier:\n" + "\n".join(window)
            checked += 1
    # NOISY x OMEGA1 x {plain, PACKED} HALO instantiations
    assert checked >= 8, f"only {checked} HALO sweep kernels found in the ISA"


def test_flag_store_is_system_scope_and_after_the_counter(isa):
    """the flag words are raised with system-scope stores issued after the atomic counter returned (its s_waitcnt)"""
    for name, body in _functions(isa):
        if "grid_color_sweep_kernel" not in name or not any(l.startswith("s_barrier") for l in body):
            continue
        b = max(i for i, l in enumerate(body) if l.startswith("s_barrier"))
        tail = body[b:]
        at = [i for i, l in enumerate(tail) if l.startswith("global_atomic_add")]
        assert at, f"{name}: no counter increment after the barrier"
        flags = [i for i, l in enumerate(tail) if l.startswith("global_store_dwordx2") and "sc0 sc1" in l and i > at[0]]
        assert flags, f"{name}: no system-scope flag store after the counter"
        between = tail[at[0] : flags[0]]
        assert any(l.startswith("s_waitcnt") and "vmcnt(0)" in l for l in between), f"{name}: flag store not ordered behind the counter's return"
