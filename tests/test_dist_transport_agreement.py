"""world_size-2/3 `gloo` test (CPU) of the transport agreement of parmgmc_amd.dist: a transport whose constructor fails on
ONE rank alone (the ipc self-test can: IpcSlabDriver._selftest) must be torn down with the SAME collectives on every
rank -- the rank without a driver included -- before the next candidate's constructor starts its own collectives.
Round 2's DistAIJMGMC called the collective destroy() only where a driver existed: the failing rank went on into the next
candidate's broadcast while its peers sat in a barrier (dist.py:770).  Drivers are fakes that record their
calls and use the collectives the real constructors use (broadcast_object_list / barrier); nothing here touches a GPU."""
import os
import socket

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, q):
    import torch.distributed as dist

    from parmgmc_amd import dist as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []

    class Fake:
        """what the real drivers do at construction: ipc = all-gather + barrier, then a LOCAL self-test; rccl = a broadcast"""

        def __init__(self, name):
            self.name, self._h = name, 1
            if name == "ipc":
                blobs = [None] * world
                dist.all_gather_object(blobs, f"blob{rank}")
                dist.barrier()
                self.selftest_error = None
                if mode == "selftest_reports" and rank == 1:
                    self.selftest_error = RuntimeError("ipc halo self-test: wrong data from the low neighbour")
                if mode == "ctor_raises" and rank == 1:
                    raise RuntimeError("peer access denied")
            else:
                payload = [f"uid-from-{rank}" if rank == 0 else None]
                dist.broadcast_object_list(payload, src=0)
                assert payload[0] == "uid-from-0"  # a rank that skipped a barrier would receive something else or hang

        def disconnect(self):
            log.append((self.name, "disconnect"))

        def free(self):
            log.append((self.name, "free"))
            self._h = 0

    drv, name = D._agree_on_driver(["ipc", "rccl"], Fake, rank, world, None, "test")
    q.put((rank, name, drv.name if drv else None, log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["selftest_reports", "ctor_raises", "all_fine"])
def test_a_transport_failing_on_one_rank_is_dropped_in_step(world, mode):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)  # a desynchronised collective would hang here
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        _, name, dname, log = res[r]
        if mode == "all_fine":
            assert (name, dname, log) == ("ipc", "ipc", [])
            continue
        assert name == dname == "rccl"  # every rank ends on the same transport
        has_ipc_object = not (mode == "ctor_raises" and r == 1)  # a constructor that raised leaves no object behind
        assert log == ([("ipc", "disconnect"), ("ipc", "free")] if has_ipc_object else [])
