"""Multi-GPU drivers: one process per GPU, torch.distributed for the bootstrap (and for the fall-back transport).

Replaces the reference's MPI design -- row-block ownership of a MATMPIAIJ with one ghost VecScatter per colour
(reference src/mc_sor.c:317-340, scatter plan :152-214):

* ``DistGridSampler`` -- sorgibbs / mcgibbs on the DMDA operator, z-slabs (`slab_cuts`, PETSc's ownership rule).  The
  sample loop is the C loop of pmg_dist.c over the "ipc" transport (peer stores + flag words, hand-shake inside the
  sweep kernel) or RCCL send/recv (``IpcSlabDriver`` / ``RcclSlabDriver``); ``run_samples`` + ``SlabHalo`` is the same
  schedule over torch.distributed point-to-point ops (RCCL tensors, or gloo through host memory in the CPU tests).
* ``DistMGMC`` -- the Multigrid Monte Carlo V-cycle on z-slabs (pmg_mgmc_create_dmda_slab), incl. low-rank updates.
* ``DistMCSOR`` -- general MATAIJ matrices by row blocks: per-colour device sweeps, ghost updates between them.

Because noise is a function of global indices only, every chain is the single-device chain for any number of ranks.
torch is plumbing here: the sweeps are the HIP kernels behind the C-ABI.
"""
from __future__ import annotations

from .capi import SOR_BACKWARD_SWEEP, SOR_FORWARD_SWEEP, SOR_SYMMETRIC_SWEEP
from .slab import slab_cuts


class SlabHalo:
    """Ghost-plane exchange of one colour of a flat colour-partitioned vector with the z-neighbours.

    `planes[color][side] = (owned_offset, ghost_offset, count)` as returned by ``pmg_grid_halo_plane``."""

    def __init__(self, rank: int, world: int, planes, group=None):
        self.rank, self.world, self.planes, self.group = rank, world, planes, group

    def start(self, y, color: int):
        import torch
        import torch.distributed as dist

        # RCCL ("nccl") moves device memory directly over xGMI.  The gloo backend (CPU tests, and the two-ranks-
        # on-one-GPU test where RCCL cannot be used) needs host buffers: stage device planes through the host.
        staged = y.is_cuda and dist.get_backend(self.group) == "gloo"
        ops, post = [], []
        for side, peer in ((0, self.rank - 1), (1, self.rank + 1)):
            if peer < 0 or peer >= self.world:
                continue
            own, ghost, n = self.planes[color][side]
            if staged:
                snd, rcv = y[own:own + n].cpu(), torch.empty(n, dtype=y.dtype)
                post.append((ghost, n, rcv))
            else:
                snd, rcv = y[own:own + n], y[ghost:ghost + n]
            ops.append(dist.P2POp(dist.isend, snd, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, rcv, peer, self.group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return (reqs, post, y) if staged else reqs

    @staticmethod
    def finish(reqs):
        if isinstance(reqs, tuple):
            works, post, y = reqs
            for r in works:
                r.wait()
            for ghost, n, rcv in post:
                y[ghost:ghost + n].copy_(rcv)
            return
        for r in reqs:
            r.wait()

    def exchange(self, y, color: int):
        self.finish(self.start(y, color))


def color_schedule(sweep_type: int):
    """Order of (direction, colours) of one sample: forward = colours 0,1; backward = 1,0 (reference
    src/mc_sor.c:257,274); symmetric = forward then backward with a fresh draw each (src/pc_mcgibbs.c:172-181)."""
    if sweep_type == SOR_SYMMETRIC_SWEEP:
        return [(0, 1), (1, 0)]
    return [(0, 1)] if sweep_type == SOR_FORWARD_SWEEP else [(1, 0)]


def run_samples(sweep_planes, halo: SlabHalo, nz: int, b, y, its: int, sweep_type: int, counter0: int) -> int:
    """The distributed sample loop with communication/computation overlap.

    sweep_planes(color, kbegin, kcount, b, y, counter) sweeps the local points of one colour on owned planes
    [kbegin, kbegin+kcount) (HIP kernel in production, an injected CPU kernel in the gloo tests).

    Invariant: the ghost planes of colour c are current once the exchange started after the last sweep of colour
    c has completed.  Per colour c: wait for the exchange of colour 1-c; sweep the two boundary planes (the only
    ones that read ghost planes); start the exchange of colour c; sweep the interior planes while it is in
    flight -- the overlap PCPARSOR gets by starting `botsct` before its INT1 rows (reference
    src/pc_parsor.c:739-745), here per colour instead of the reference's blocking per-colour VecScatter
    (src/mc_sor.c:318-319)."""
    pending = {0: halo.start(y, 0), 1: halo.start(y, 1)}  # the caller's y has no ghost values yet
    edge = sorted({0, nz - 1})
    ctr = counter0
    for _ in range(its):
        for order in color_schedule(sweep_type):
            for c in order:
                halo.finish(pending[1 - c])  # colour c reads colour 1-c across the slab faces
                pending[1 - c] = []
                for k in edge:
                    sweep_planes(c, k, 1, b, y, ctr)
                halo.finish(pending[c])  # an older exchange of colour c must not land after the new one starts
                pending[c] = halo.start(y, c)
                if nz > 2:
                    sweep_planes(c, 1, nz - 2, b, y, ctr)
            ctr += 1
    halo.finish(pending[0])
    halo.finish(pending[1])
    return ctr


def _all_ok(err, group, what: str) -> None:
    """Agreement point: raises on EVERY rank if `err` is set on any rank (all-reduce MIN of an ok flag)."""
    import torch
    import torch.distributed as dist

    flag = torch.tensor([0 if err else 1], device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) != 1:
        raise RuntimeError(f"{what}: {err or 'failed on another rank'}")


def _teardown(drv, world: int, group) -> None:
    """Orderly end of a transport, COLLECTIVE over all ranks of `group` -- entered by every rank whether or not it holds
    a driver (drv None: its constructor failed on this rank alone): unmap the peers' blocks, barrier, free the own block.
    A rank that frees a block a peer still has mapped and exports a fresh one at once gets "invalid argument" from
    hipIpcGetMemHandle; a rank that skipped the barrier would meet its peers in a different collective."""
    if drv is not None and getattr(drv, "_h", None):
        drv.disconnect()
    if world > 1:
        import torch
        import torch.distributed as dist

        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dist.barrier(group=group)
    if drv is not None:
        drv.free()


def _agree_on_driver(candidates, make, rank: int, world: int, group, what: str):
    """Try the transports in order; every rank must end up on the same one.  make(name) builds a driver (collective
    internally; may fail on one rank alone, by raising or by reporting `selftest_error`); success is agreed with an
    all-reduce, and after a failed agreement EVERY rank -- with or without a driver of its own -- runs the same collective
    tear-down before the next candidate's constructor starts its own collectives.  Returns (driver, name) or (None, None)."""
    import torch
    import torch.distributed as dist

    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    for cand in candidates:
        drv, err = None, None
        try:
            drv = make(cand)
            err = getattr(drv, "selftest_error", None)
        except Exception as e:  # noqa: BLE001
            err = e
        flag = torch.tensor([0 if err else 1], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 1:
            return drv, cand
        _teardown(drv, world, group)
        if rank == 0:
            print(f"[parmgmc_amd] {what}: transport '{cand}' unavailable ({err if err else 'on another rank'}); trying the next one", flush=True)
    return None, None


class RcclSlabDriver:
    """The C sample loop of pmg_dist.c: RCCL send/recv called straight from the host library, comm stream + events,
    no Python in the loop.  The 128-byte ncclUniqueId of rank 0 is broadcast through torch.distributed."""

    def __init__(self, grid, rank: int, world: int, loopback: bool = False, group=None):
        import ctypes as C

        from . import capi
        from .capi import check, lib

        self._h = C.c_void_p()
        self._world, self._group = world, group
        path = capi.torch_rccl_path()
        pbytes = path.encode() if path else None
        uid = None
        if world > 1 or loopback:
            buf = C.create_string_buffer(128)
            err = None
            if rank == 0:
                try:
                    check(lib.pmg_dist_get_unique_id(pbytes, buf))
                except Exception as e:  # noqa: BLE001 -- the other ranks are about to enter the broadcast: tell them
                    err = f"{type(e).__name__}: {e}"
            payload = [None if err else bytes(buf.raw)]
            if world > 1:
                import torch.distributed as dist

                dist.broadcast_object_list(payload, src=0, group=group)
            if payload[0] is None:
                raise RuntimeError(err or "rank 0 could not create the RCCL unique id")
            uid = C.create_string_buffer(payload[0], 128)
        check(lib.pmg_dist_create(grid._h if grid is not None else None, rank, world, uid, pbytes, int(loopback), C.byref(self._h)))  # grid None: a transport without a slab
        self._grid = grid  # keep alive

    def sample_cvec(self, b, y, its, scaled, sweep_type, seed, counter0):
        import ctypes as C

        from .capi import check, lib
        from .wrappers import _ptr, _stream

        out = C.c_uint64()
        check(lib.pmg_dist_sample_cvec(self._h, _ptr(b), _ptr(y), its, int(scaled), sweep_type, seed, counter0, C.byref(out), _stream()))
        return out.value

    def check(self):
        """after torch.cuda.synchronize(): raises if a device-side wait for a halo flag gave up"""
        from .capi import check, lib

        check(lib.pmg_dist_check(self._h))

    def describe(self, lo_device: int = -1, hi_device: int = -1) -> dict:
        """pmg_dist_describe: device, PCI bus id, peer access to the z-neighbours' devices, transport, ncclCommCount, halo
        wait polls -- this rank's entry of the multi-GPU bench record"""
        import ctypes as C

        from .capi import DistDescription, check, lib

        d = DistDescription()
        check(lib.pmg_dist_describe(self._h, lo_device, hi_device, C.byref(d)))
        return d.as_dict()

    def disconnect(self):
        """local: unmap the peers' receive blocks"""
        from .capi import lib

        if self._h:
            lib.pmg_dist_ipc_disconnect(self._h)

    def free(self):
        """local: free the handle and the own receive block (after every peer has unmapped it)"""
        import ctypes as C

        from .capi import lib

        if self._h:
            lib.pmg_dist_destroy(C.byref(self._h))

    def destroy(self):
        """Collective, orderly tear-down: unmap the peers' blocks, barrier, free the own block."""
        if not self._h:
            return
        _teardown(self, getattr(self, "_world", 1), getattr(self, "_group", None))

    def __del__(self):
        try:
            import ctypes as C

            from .capi import lib

            if self._h:
                lib.pmg_dist_destroy(C.byref(self._h))
        except Exception:
            pass


class IpcSlabDriver(RcclSlabDriver):
    """The C sample loop of pmg_dist.c with the "ipc" transport: boundary planes are stored device-to-device straight
    into the neighbour's receive block (hipIpc memory) and announced by flag words; torch.distributed only carries
    the bootstrap (all-gather of the handle blobs, one barrier)."""

    def __init__(self, grid, rank: int, world: int, group=None, loopback: bool = False):
        import ctypes as C
        import os

        from .capi import check, lib

        self._h = C.c_void_p()
        self._grid = grid
        self._world, self._group = (1 if loopback else world), group
        if loopback:
            check(lib.pmg_dist_create_ipc(grid._h if grid is not None else None, 0, 1, C.byref(self._h)))
            check(lib.pmg_dist_ipc_connect_loopback(self._h))
            return
        import torch.distributed as dist

        # every step that can fail on ONE rank alone (allocation, export, opening a peer's handle) is followed by an
        # agreement, so that no rank is left waiting in a collective the failing rank never enters
        err, blob, nb = None, None, C.c_int32()
        try:  # local phase: allocate the receive block, export its handle
            check(lib.pmg_dist_create_ipc(grid._h if grid is not None else None, rank, world, C.byref(self._h)))
            check(lib.pmg_dist_ipc_blob_bytes(C.byref(nb)))
            blob = C.create_string_buffer(nb.value)
            check(lib.pmg_dist_ipc_export(self._h, blob))
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        def agree(err, what):  # raises on every rank together: all of them tear the half-built transport down in step
            try:
                _all_ok(err, group, what)
            except Exception:
                _teardown(self, world, group)
                raise

        agree(err, "ipc: allocating / exporting the receive block")
        blobs = [None] * world
        dist.all_gather_object(blobs, bytes(blob.raw), group=group)
        try:  # local phase: map the neighbours' (all peers') blocks
            lo = C.create_string_buffer(blobs[rank - 1], nb.value) if rank > 0 else None
            hi = C.create_string_buffer(blobs[rank + 1], nb.value) if rank < world - 1 else None
            check(lib.pmg_dist_ipc_connect(self._h, lo, hi))
            if world > 2:  # all-peer mappings: the all-gather of the distributed V-cycle becomes one step
                bufs = [C.create_string_buffer(b_, nb.value) for b_ in blobs]
                arr = (C.c_void_p * world)(*[C.cast(b_, C.c_void_p) for b_ in bufs])
                check(lib.pmg_dist_ipc_connect_all(self._h, arr))
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        agree(err, "ipc: opening the peers' memory handles")
        dist.barrier(group=group)
        # the self-test can fail on ONE rank alone and no collective follows inside this constructor: the failure is kept,
        # not raised, so that the caller (_agree_on_driver) still holds the object for the collective tear-down
        self.selftest_error = None
        try:
            self._selftest(rank, world)
        except Exception as e:  # noqa: BLE001
            self.selftest_error = e

    def _selftest(self, rank, world):
        """One round trip of a known pattern with both z-neighbours (peer copies into their blocks, flag words, copy
        out): if peer access or the flags do not work on this machine the constructor fails -- on every rank, the
        wait gives up after about a minute -- and DistGridSampler moves on to the next transport."""
        import ctypes as C

        import torch

        from .capi import check, lib
        from .wrappers import _ptr, _stream

        n = 4096
        send = torch.full((2, n), float(rank + 1), dtype=torch.float64, device="cuda") + torch.arange(n, dtype=torch.float64, device="cuda") / n
        recv = torch.zeros((2, n), dtype=torch.float64, device="cuda")
        P, I64 = C.c_void_p * 1, C.c_int64 * 1
        nn = I64(n)
        check(lib.pmg_dist_exchange(self._h, 1, P(_ptr(send[0])), nn, P(_ptr(recv[0])), nn, P(_ptr(send[1])), nn, P(_ptr(recv[1])), nn, _stream()))
        torch.cuda.synchronize()
        frac = torch.arange(n, dtype=torch.float64, device="cuda") / n
        if rank > 0 and not torch.equal(recv[0], float(rank) + frac):
            raise RuntimeError("ipc halo self-test: wrong data from the low neighbour")
        if rank < world - 1 and not torch.equal(recv[1], float(rank + 2) + frac):
            raise RuntimeError("ipc halo self-test: wrong data from the high neighbour")
        # all-gather (all-peer mappings for more than two ranks): rank r contributes 64 + r values
        counts = [64 + r for r in range(world)]
        offs = [sum(counts[:r]) for r in range(world)]
        buf = torch.zeros(sum(counts), dtype=torch.float64, device="cuda")
        buf[offs[rank]:offs[rank] + counts[rank]] = float(rank + 1)
        A64 = C.c_int64 * world
        check(lib.pmg_dist_allgather(self._h, _ptr(buf), A64(*offs), A64(*counts), _stream()))
        torch.cuda.synchronize()
        want = torch.cat([torch.full((counts[r],), float(r + 1), dtype=torch.float64, device="cuda") for r in range(world)])
        if not torch.equal(buf, want):
            raise RuntimeError("ipc halo self-test: all-gather returned wrong data")


class DistGridSampler:
    """sorgibbs/mcgibbs sampler for the grid operator on `world` GPUs (this process = one slab).

    transport "rccl" (default when torch.distributed runs on the nccl backend): the C loop of pmg_dist.c;
    transport "torch": the Python loop `run_samples` over torch.distributed P2P (gloo in the CPU tests; also the
    automatic fall-back if RCCL cannot be initialised)."""

    def __init__(self, nx, ny, nz, kappa, rank, world, omega=1.0, sweep_type=SOR_FORWARD_SWEEP, scaled=True, group=None, transport=None):
        import os

        from .wrappers import GridMCSOR

        cuts = slab_cuts(nz, world)
        self.rank, self.world = rank, world
        self.grid = GridMCSOR(nx, ny, nz, kappa, kz0=cuts[rank], nz_owned=cuts[rank + 1] - cuts[rank])
        self.grid.set_omega(omega)
        self.sweep_type, self.scaled = sweep_type, scaled
        planes = [[self.grid.halo_plane(c, s) for s in (0, 1)] for c in (0, 1)]
        self.halo = SlabHalo(rank, world, planes, group)
        self.rccl = None
        self.transport = "none" if world == 1 else "torch"
        want = transport or os.environ.get("PMG_DIST_TRANSPORT")
        if world > 1 and want != "torch":
            import torch
            import torch.distributed as dist

            on_gpu = dist.get_backend(group) == "nccl"
            # preference: ipc (peer copies, lowest latency) -> rccl (ncclSend/ncclRecv) -> torch P2P; every rank must
            # take the same path, so success is agreed with an all-reduce after each attempt
            candidates = [want] if want else (["ipc", "rccl"] if on_gpu else [])
            drv, name = _agree_on_driver(candidates, lambda cand: IpcSlabDriver(self.grid, rank, world, group=group) if cand == "ipc" else RcclSlabDriver(self.grid, rank, world, group=group), rank, world, group, "halo exchange")
            if drv is not None:
                self.rccl, self.transport = drv, name

    def check(self):
        """after torch.cuda.synchronize(): raises if the halo transport lost a neighbour"""
        if self.rccl is not None:
            self.rccl.check()

    def describe(self, lo_device: int = -1, hi_device: int = -1) -> dict:
        """this rank's entry of a multi-GPU bench record (see RcclSlabDriver.describe); the torch P2P loop has no C object"""
        if self.rccl is not None:
            return self.rccl.describe(lo_device, hi_device)
        import torch

        dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
        pci = "unknown"
        try:
            p = torch.cuda.get_device_properties(dev)
            pci = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except Exception:  # noqa: BLE001
            pass
        return {"rank": self.rank, "nranks": self.world, "device": dev, "pci_bus_id": pci, "neighbour_ranks": [self.rank - 1 if self.rank > 0 else -1, self.rank + 1 if self.rank < self.world - 1 else -1], "peer_access_lo_hi": [-1, -1], "transport": self.transport, "rccl_comm_count": 0, "halo_wait_polls": 0}

    def destroy(self):
        """collective: orderly tear-down of the halo transport (call on every rank before building another one)"""
        if self.rccl is not None:
            self.rccl.destroy()
            self.rccl = None

    def sample_cvec(self, b, y, its: int, seed: int, counter0: int = 0) -> int:
        if self.world == 1:
            self.grid.set_sweep_type(self.sweep_type)
            return self.grid.sample_cvec(b, y, its, seed, counter0, self.scaled)
        if self.rccl:
            return self.rccl.sample_cvec(b, y, its, self.scaled, self.sweep_type, seed, counter0)
        g = self.grid
        return run_samples(lambda c, k0, nk, bb, yy, ctr: g.sweep_color_planes_cvec(c, k0, nk, bb, yy, True, self.scaled, seed, ctr), self.halo, g.nz, b, y, its, self.sweep_type, counter0)


class DistMGMC:
    """Multigrid Monte Carlo on `world` GPUs (this process = one z-slab of every distributed level): the C V-cycle of
    pmg_mgmc.c on top of the halo transport of pmg_dist.c ("ipc" or "rccl"; reference PCGAMGMC over MPI ranks,
    src/pc_gamgmc.c + src/mc_sor.c:298-381).  b and y of `sample` are this rank's planes in natural order."""

    def __init__(self, nx, ny, nz, kappa, levels, rank, world, group=None, transport=None):
        import ctypes as C

        import numpy as np

        from .capi import check, lib
        from .wrappers import MGMC

        self.rank, self.world = rank, world
        self.cuts = slab_cuts(nz, world)
        if world == 1:
            self.grid_sampler, self.transport = None, "none"
            self.mg = MGMC(nx, ny, nz, kappa, levels)
        else:
            self.grid_sampler = DistGridSampler(nx, ny, nz, kappa, rank, world, group=group, transport=transport)
            self.transport = self.grid_sampler.transport
            if self.grid_sampler.rccl is None:
                raise RuntimeError("the distributed V-cycle needs the ipc or rccl halo transport (torch P2P carries only the stand-alone sweeps)")
            cuts = np.asarray(self.cuts, np.int32)
            mg = MGMC.__new__(MGMC)
            mg.nx, mg.ny, mg.nz, mg.levels = nx, ny, nz, levels
            mg._h = C.c_void_p()
            check(lib.pmg_mgmc_create_dmda_slab(nx, ny, nz, kappa, levels, self.grid_sampler.grid._h, self.grid_sampler.rccl._h, cuts.ctypes.data, C.byref(mg._h)))
            mg.n = nx * ny * (self.cuts[rank + 1] - self.cuts[rank])
            self.mg = mg
        self.n_local = nx * ny * (self.cuts[rank + 1] - self.cuts[rank])
        self.plane_range = (self.cuts[rank], self.cuts[rank + 1])

    def __getattr__(self, name):  # set_smoother, set_coarse, set_correction_form, setup, sample, ...
        return getattr(self.mg, name)

    def destroy(self):
        self.mg.destroy()  # before the transport and the grid it borrows
        if self.grid_sampler is not None:
            self.grid_sampler.destroy()
        self.grid_sampler = None


class DistMCSOR:
    """mcgibbs / sorgibbs sampler for an assembled MATAIJ matrix distributed by ROW BLOCKS over `world` ranks
    (reference MCSORApply_MPIAIJ, src/mc_sor.c:298-381: for every colour, update the ghost values, then sweep the
    colour's rows; scatter plan as MatCreateScatters :152-214, de-duplicated per ghost column).

    This rank owns rows [row0, row1) of the global matrix: rowptr / colidx / vals are its rows with GLOBAL column
    indices, colors its rows' entries of a globally valid distance-1 colouring (ncolors colours).  The off-process
    columns become ghost rows (identity rows in an extra colour that is never swept) of a local sliced-ELL operator;
    the per-colour sweeps run on the device (pmg_mcsor_sweep_color_layout), the ghost values travel through
    torch.distributed point-to-point messages between them.  Noise is keyed on the global row, entries keep the order
    of the global CSR row: the chain is the single-process chain bit for bit."""

    def __init__(self, rowptr, colidx, vals, row0, row1, n_global, colors, ncolors, rank, world, omega=1.0, sweep_type=SOR_FORWARD_SWEEP, scaled=True, group=None, transport=None):
        import os

        import numpy as np
        import torch
        import torch.distributed as dist

        from .wrappers import MCSOR

        self.rank, self.world, self.group = rank, world, group
        self.row0, self.row1, self.nloc, self.ncolors = row0, row1, row1 - row0, ncolors
        self.sweep_type, self.scaled = sweep_type, scaled
        rp = np.asarray(rowptr, np.int64)
        ci = np.asarray(colidx, np.int64)
        v = np.asarray(vals, np.float64)
        assert len(rp) == self.nloc + 1 and len(colors) == self.nloc
        is_ghost = (ci < row0) | (ci >= row1)
        self.ghosts = np.unique(ci[is_ghost])  # sorted global ids of the ghost columns
        ng = len(self.ghosts)
        loc = np.where(is_ghost, self.nloc + np.searchsorted(self.ghosts, ci), ci - row0)
        # local operator: my rows (entries in the order of the global CSR row) + one identity row per ghost column
        rp2 = np.concatenate([rp, rp[-1] + 1 + np.arange(ng)]).astype(np.int32)
        ci2 = np.concatenate([loc, self.nloc + np.arange(ng)]).astype(np.int32)
        v2 = np.concatenate([v, np.ones(ng)])
        col2 = np.concatenate([np.asarray(colors, np.int32), np.full(ng, ncolors, np.int32)])
        self.mc = MCSOR(rp2, ci2, v2, user_colors=col2)
        self.mc.set_omega(omega)
        self.mc.setup()
        self.mc.set_noise_row_offset(row0)
        pos = self.mc.get_layout().astype(np.int64)
        self.ld = self.mc.layout_len()
        self.pos_local = torch.as_tensor(pos[: self.nloc], device="cuda")
        # --- scatter plan: who owns my ghosts, what do the others need from me, in which colour does it change ---
        ranges = [None] * world
        dist.all_gather_object(ranges, (row0, row1), group=group)
        starts = np.array([r[0] for r in ranges] + [n_global])
        owner = np.searchsorted(starts, self.ghosts, side="right") - 1
        want = [self.ghosts[owner == p] for p in range(world)]  # global rows I need from rank p
        asked = [None] * world
        dist.all_gather_object(asked, want, group=group)  # asked[q][p] = rows rank q needs from rank p
        mycol = np.asarray(colors, np.int64)
        reply = [mycol[np.asarray(asked[q][rank], np.int64) - row0] if q != rank else None for q in range(world)]
        replies = [None] * world
        dist.all_gather_object(replies, reply, group=group)  # replies[p][q] = colours of the rows q asked from p
        self.send, self.recv = [], []  # per colour: list of (peer, positions in my layout vector)
        for c in range(ncolors):
            snd, rcv = [], []
            for p in range(world):
                if p == rank:
                    continue
                rows_p = np.asarray(asked[p][rank], np.int64)  # what p needs from me
                sel = rows_p[mycol[rows_p - row0] == c] if len(rows_p) else rows_p
                if len(sel):
                    snd.append((p, torch.as_tensor(pos[sel - row0], device="cuda")))
                g = want[p]
                selg = g[np.asarray(replies[p][rank]) == c] if len(g) else g
                if len(selg):
                    rcv.append((p, torch.as_tensor(pos[self.nloc + np.searchsorted(self.ghosts, selg)], device="cuda")))
            self.send.append(snd)
            self.recv.append(rcv)
        self._staged = dist.get_backend(group) != "nccl"  # gloo: through host memory
        # --- the C driver (pmg_distmcsor.c): colour sweeps and ghost updates in one C loop on the stream, the updates as
        # one all-gather of the colour's boundary values over the "ipc" or "rccl" transport of pmg_dist.c.  transport
        # "torch" (or none available) keeps the Python loop over torch.distributed point-to-point messages below.
        self._c, self._drv, self.transport = None, None, "none" if world == 1 else "torch"
        want_tr = transport or os.environ.get("PMG_DIST_TRANSPORT")
        if world > 1 and want_tr != "torch":
            on_gpu = dist.get_backend(group) == "nccl"
            drv, name = _agree_on_driver([want_tr] if want_tr else (["ipc", "rccl"] if on_gpu else []), lambda cand: IpcSlabDriver(None, rank, world, group=group) if cand == "ipc" else RcclSlabDriver(None, rank, world, group=group), rank, world, group, "row-block sampler")
            if drv is not None:
                self._drv, self.transport = drv, name
        if self._drv is not None:
            import ctypes as C

            from .capi import check, lib

            others = [np.asarray(asked[p][rank], np.int64) for p in range(world) if p != rank]
            needed = np.unique(np.concatenate(others)) if others else np.zeros(0, np.int64)  # my rows that another rank reads
            send_lists = [needed[mycol[needed - row0] == c] for c in range(ncolors)]  # sorted global rows, by colour
            all_lists = [None] * world
            dist.all_gather_object(all_lists, send_lists, group=group)
            counts = np.array([[len(all_lists[r][c]) for r in range(world)] for c in range(ncolors)], np.int64)
            offs = np.concatenate([np.zeros((ncolors, 1), np.int64), np.cumsum(counts, axis=1)[:, :-1]], axis=1)
            send_ptr = np.concatenate([[0], np.cumsum([len(l) for l in send_lists])]).astype(np.int64)
            send_pos = (np.concatenate([pos[l - row0] for l in send_lists]) if send_ptr[-1] else np.zeros(0)).astype(np.int32)
            gid = np.concatenate([want[p] for p in range(world)]) if ng else np.zeros(0, np.int64)  # my ghosts, by owner
            gown = np.concatenate([np.full(len(want[p]), p) for p in range(world)]) if ng else np.zeros(0, np.int64)
            gcol = np.concatenate([np.asarray(replies[p][rank] if p != rank else [], np.int64) for p in range(world)]) if ng else np.zeros(0, np.int64)
            rsrc, rpos, rptr = [], [], [0]
            for c in range(ncolors):
                sel = np.nonzero(gcol == c)[0]
                for q in sel:
                    pown = int(gown[q])
                    rsrc.append(int(offs[c, pown] + np.searchsorted(all_lists[pown][c], gid[q])))
                    rpos.append(int(pos[self.nloc + np.searchsorted(self.ghosts, gid[q])]))
                rptr.append(len(rsrc))
            recv_ptr, recv_src, recv_pos = np.asarray(rptr, np.int64), np.asarray(rsrc, np.int32), np.asarray(rpos, np.int32)
            self._c = C.c_void_p()
            check(lib.pmg_distmcsor_create(self.mc._h, self._drv._h, ncolors, send_ptr.ctypes.data, send_pos.ctypes.data, np.ascontiguousarray(counts).ctypes.data, recv_ptr.ctypes.data, recv_src.ctypes.data, recv_pos.ctypes.data, C.byref(self._c)))

    def set_lowrank(self, B_owned, S):
        """MATLRC operator A + B diag(S) B^T (reference MCSORSetUp's LRC branch, src/mc_sor.c:572-595): B_owned = this
        rank's rows of B (nloc x k).  C driver only (transport "ipc" / "rccl"); collective."""
        import numpy as np

        from .capi import check, lib

        if self._c is None:
            raise RuntimeError("the low-rank update of the row-block sampler needs the C driver (transport 'ipc' or 'rccl')")
        B = np.asarray(B_owned, np.float64)
        S = np.ascontiguousarray(S, np.float64)
        assert B.shape == (self.nloc, len(S))
        nl = self.nloc + len(self.ghosts)
        Bl = np.zeros((nl, len(S)), order="F")
        Bl[: self.nloc] = B
        check(lib.pmg_distmcsor_set_lowrank(self._c, len(S), nl, self.nloc, Bl.ctypes.data, S.ctypes.data))

    def new_layout(self):
        import torch

        return torch.zeros(self.ld, dtype=torch.float64, device="cuda")

    def to_layout(self, nat_local, out=None):
        out = self.new_layout() if out is None else out
        out[self.pos_local] = nat_local
        return out

    def from_layout(self, lay):
        return lay[self.pos_local]

    def exchange(self, y, color: int):
        """ghost update for the rows of one colour (VecScatterBegin/End of src/mc_sor.c:318-319)"""
        import torch
        import torch.distributed as dist

        ops, bufs = [], []
        for p, idx in self.send[color]:
            t = y[idx]
            t = t.cpu() if self._staged else t
            ops.append(dist.P2POp(dist.isend, t, p, self.group))
        for p, idx in self.recv[color]:
            t = torch.empty(len(idx), dtype=torch.float64, device="cpu" if self._staged else "cuda")
            bufs.append((idx, t))
            ops.append(dist.P2POp(dist.irecv, t, p, self.group))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        for idx, t in bufs:
            y[idx] = t.cuda() if self._staged else t

    def sample_layout(self, b, y, its: int, seed: int, counter0: int = 0) -> int:
        """`its` samples on layout vectors (b: my rows filled, y: my rows filled; ghost entries are refreshed here)"""
        if self._c is not None:  # the C loop: no Python between colours
            import ctypes as C

            from .capi import check, lib
            from .wrappers import _ptr, _stream

            out = C.c_uint64()
            check(lib.pmg_distmcsor_sample_layout(self._c, _ptr(b), _ptr(y), its, int(self.scaled), self.sweep_type, seed, counter0, C.byref(out), _stream()))
            return out.value
        for c in range(self.ncolors):
            self.exchange(y, c)
        ctr = counter0
        for _ in range(its):
            for direction in ((SOR_FORWARD_SWEEP, SOR_BACKWARD_SWEEP) if self.sweep_type == SOR_SYMMETRIC_SWEEP else (self.sweep_type,)):
                order = range(self.ncolors) if direction == SOR_FORWARD_SWEEP else range(self.ncolors - 1, -1, -1)
                for c in order:
                    self.mc.sweep_color_layout(c, b, y, True, self.scaled, seed, ctr)
                    self.exchange(y, c)
                ctr += 1
        return ctr

    def destroy(self):
        if self._c is not None:
            import ctypes as C

            from .capi import lib

            lib.pmg_distmcsor_destroy(C.byref(self._c))
            self._c = None
        if self._drv is not None:
            self._drv.destroy()
        self._drv = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def rowblock_plan(ci_global, row0: int, row1: int, n_global: int, colors_owned, ncolors: int, extra_ghosts, rank: int, world: int, group=None):
    """Ghost set and per-colour ghost-update plan of one row block (what MatCreateScatters builds per colour in the
    reference, src/mc_sor.c:152-214; here de-duplicated per ghost row and laid out for ONE all-gather per colour).

    ci_global: the global column indices of this rank's rows; extra_ghosts: further global rows of other ranks whose
    values this rank reads (restriction / interpolation columns).  Returns (ghosts, plan) with ghosts = sorted global
    ids (local row nloc + q) and plan = dict(send_ptr, send_rows, counts, recv_ptr, recv_src, recv_rows): local OWNED rows
    this rank contributes per colour (by ascending global row), the ranks' block lengths, and for every ghost row the
    index of its value in the colour's gather buffer (blocks in rank order).  Collective over `group` (host objects)."""
    import numpy as np
    import torch.distributed as dist

    ci = np.asarray(ci_global, np.int64)
    nloc = row1 - row0
    off = ci[(ci < row0) | (ci >= row1)]
    extra = np.asarray(extra_ghosts, np.int64)
    extra = extra[(extra < row0) | (extra >= row1)]
    ghosts = np.unique(np.concatenate([off, extra]))
    ranges = [None] * world
    dist.all_gather_object(ranges, (row0, row1), group=group)
    starts = np.array([r[0] for r in ranges] + [n_global])
    assert all(ranges[r][1] == starts[r + 1] for r in range(world)), "row blocks must tile the rows in rank order"
    owner = np.searchsorted(starts, ghosts, side="right") - 1
    want = [ghosts[owner == p] for p in range(world)]  # global rows I need from rank p
    asked = [None] * world
    dist.all_gather_object(asked, want, group=group)  # asked[q][p] = rows rank q needs from rank p
    mycol = np.asarray(colors_owned, np.int64)
    assert len(mycol) == nloc and (nloc == 0 or (mycol.min() >= 0 and mycol.max() < ncolors))
    others = [np.asarray(asked[q][rank], np.int64) for q in range(world) if q != rank]
    needed = np.unique(np.concatenate(others)) if others else np.zeros(0, np.int64)  # my rows that another rank reads
    send_lists = [needed[mycol[needed - row0] == c] for c in range(ncolors)]  # sorted global rows, by colour
    all_lists = [None] * world
    dist.all_gather_object(all_lists, send_lists, group=group)
    counts = np.array([[len(all_lists[r][c]) for r in range(world)] for c in range(ncolors)], np.int64).reshape(ncolors, world)
    offs = np.concatenate([np.zeros((ncolors, 1), np.int64), np.cumsum(counts, axis=1)[:, :-1]], axis=1)
    send_ptr = np.concatenate([[0], np.cumsum([len(l) for l in send_lists])]).astype(np.int64)
    send_rows = (np.concatenate(send_lists) - row0 if send_ptr[-1] else np.zeros(0)).astype(np.int32)
    rsrc, rrow, rptr = [], [], [0]
    for c in range(ncolors):
        for p in range(world):
            if p == rank or not len(want[p]):
                continue
            lst = np.asarray(all_lists[p][c], np.int64)
            if not len(lst):
                continue
            k = np.searchsorted(lst, want[p])
            hit = (k < len(lst)) & (lst[np.minimum(k, len(lst) - 1)] == want[p])  # my ghosts owned by p that have colour c
            rsrc.append(offs[c, p] + k[hit])
            rrow.append(nloc + np.searchsorted(ghosts, want[p][hit]))
        rptr.append(sum(len(a) for a in rsrc))
    plan = dict(send_ptr=send_ptr, send_rows=send_rows, counts=np.ascontiguousarray(counts),
                recv_ptr=np.asarray(rptr, np.int64),
                recv_src=(np.concatenate(rsrc) if rsrc else np.zeros(0)).astype(np.int32),
                recv_rows=(np.concatenate(rrow) if rrow else np.zeros(0)).astype(np.int32))
    assert plan["recv_ptr"][-1] == len(ghosts), "every ghost row is owned by exactly one rank and has exactly one colour"
    return ghosts, plan


def rowblock_hierarchy(operators, interpolations, colorings, rank: int, world: int, starts=None, replicate_below: int = 50000, group=None):
    """This rank's share of a hierarchy distributed by row blocks -- host arrays only, no device, no C call; collective
    (the ghost plans of rowblock_plan).  operators / interpolations as for MGMC.from_hierarchy (level 0 = coarsest),
    colorings[l] = (colours of ALL rows of level l, number of colours) for the levels that will be row blocks (None below).
    Returns dict(fold, starts, n, levels): levels[l] for l >= fold holds the local operator (rp, ci, v: owned rows in
    global order, entries in global CSR order, then one identity row per ghost), colors (owned rows), ncolors, plan,
    ghosts (sorted global rows), nowned, row0, P (owned rows of P_l, columns in the local numbering of level l-1 -- global
    when that level is replicated) and R (the rows of P_l^T this rank owns on level l-1, columns in the local numbering of
    level l, entries by ascending global fine row).  Levels below `fold` are replicated (levels[l] = None)."""
    import numpy as np
    import scipy.sparse as sp

    L = len(operators)
    assert L >= 2 and len(interpolations) == L
    A = [sp.csr_matrix((np.asarray(v, np.float64), np.asarray(ci, np.int64), np.asarray(rp, np.int64)), shape=(len(rp) - 1, len(rp) - 1)) for rp, ci, v in operators]
    P = [None] + [sp.csr_matrix((np.asarray(interpolations[l][2], np.float64), np.asarray(interpolations[l][1], np.int64), np.asarray(interpolations[l][0], np.int64)), shape=(A[l].shape[0], A[l - 1].shape[0])) for l in range(1, L)]
    n = [a.shape[0] for a in A]
    if starts is None:
        starts = [np.linspace(0, n[l], world + 1).astype(np.int64) for l in range(L)]
    starts = [np.asarray(s_, np.int64) for s_ in starts]
    r0 = [int(s_[rank]) for s_ in starts]
    r1 = [int(s_[rank + 1]) for s_ in starts]
    R = [None] + [P[l].T.tocsr() for l in range(1, L)]  # rows of P^T, entries by ascending fine row
    for m in R[1:]:
        m.sort_indices()
    fold = 1  # lowest row-block level; the levels below are replicated
    while fold < L - 1 and n[fold] <= replicate_below:
        fold += 1
    ghosts, nloc = [np.zeros(0, np.int64)] * L, [r1[l] - r0[l] for l in range(L)]
    local_of = [None] * L  # global row -> local row of this rank (owned, then ghosts), -1 elsewhere; replicated levels: identity
    for l in range(fold):
        local_of[l] = np.arange(n[l], dtype=np.int64)
    levels = [None] * L
    for l in range(fold, L):
        col, ncol = colorings[l]
        col = np.asarray(col)
        mine = A[l][r0[l]:r1[l]]
        extra = [R[l][r0[l - 1]:r1[l - 1]].indices.astype(np.int64)]  # fine rows my restriction rows read (level l-1 replicated: the block I restrict into)
        if l + 1 < L:
            extra.append(P[l + 1][r0[l + 1]:r1[l + 1]].indices.astype(np.int64))  # rows of this level the finer level's interpolation reads
        ghosts[l], plan = rowblock_plan(mine.indices, r0[l], r1[l], n[l], col[r0[l]:r1[l]], ncol, np.concatenate(extra), rank, world, group)
        ng = len(ghosts[l])
        lo = np.full(n[l], -1, np.int64)
        lo[r0[l]:r1[l]] = np.arange(nloc[l])
        lo[ghosts[l]] = nloc[l] + np.arange(ng)
        local_of[l] = lo
        # local operator: my rows (entries in the order of the global CSR row) + one identity row per ghost
        rp = np.concatenate([mine.indptr, mine.indptr[-1] + 1 + np.arange(ng)]).astype(np.int32)
        ci = np.concatenate([lo[mine.indices], nloc[l] + np.arange(ng)]).astype(np.int32)
        assert ci.min(initial=0) >= 0
        v = np.concatenate([mine.data, np.ones(ng)])
        levels[l] = dict(rp=rp, ci=ci, v=v, colors=np.ascontiguousarray(col[r0[l]:r1[l]], np.int32), ncolors=int(ncol), plan=plan, ghosts=ghosts[l], nowned=nloc[l], row0=r0[l])
    for l in range(fold, L):
        pm = P[l][r0[l]:r1[l]]  # my rows of P_l (entries in the caller's order), columns -> local numbering of level l-1
        prp, pci, pv = pm.indptr.astype(np.int32), local_of[l - 1][pm.indices].astype(np.int32), np.ascontiguousarray(pm.data)
        rm = R[l][r0[l - 1]:r1[l - 1]]  # the rows of P_l^T I own on level l-1, columns -> local numbering of level l
        rrp, rci, rv = rm.indptr.astype(np.int32), local_of[l][rm.indices].astype(np.int32), np.ascontiguousarray(rm.data)
        assert pci.min(initial=0) >= 0 and rci.min(initial=0) >= 0, "a transfer reads a row that is neither owned nor a ghost"
        levels[l]["P"] = (prp, pci, pv)
        levels[l]["R"] = (rrp, rci, rv)
        levels[l]["ncoarse_local"] = n[l - 1] if l == fold else nloc[l - 1] + len(ghosts[l - 1])
    return dict(fold=fold, starts=starts, n=n, levels=levels)


class DistAIJMGMC:
    """MGMC sampler on a caller-supplied hierarchy of assembled MATAIJ matrices distributed by ROW BLOCKS, one rank per
    device: the reference's PCGAMGMC on a MATMPIAIJ (src/pc_gamgmc.c:157-223; level sampler MCSORApply_MPIAIJ,
    src/mc_sor.c:298-381).  `operators[l]` = global CSR triple of level l (0 = coarsest), `interpolations[l]` (l >= 1) =
    global CSR triple of the prolongation from level l-1 to level l -- what MGMC.from_hierarchy takes, available on every
    rank; `starts[l]` = the world+1 row-block boundaries of level l (default: equal blocks).  Every rank keeps its rows of
    the levels with more than `replicate_below` rows (plus one ghost row per row of another rank that its operator,
    restriction or the finer level's interpolation reads) and the WHOLE of the smaller levels -- at least the coarsest,
    which is factored redundantly: there every rank runs the single-device kernels on identical data, which costs less
    than a ghost update per colour for a few thousand rows.  The whole sample loop -- colour
    sweeps, ghost updates, residuals, transfers, coarse solve -- runs in C (pmg_mgmc.c) on the stream; this class only
    slices the matrices and builds the ghost plans (host set-up, collective).  Bit-identical to MGMC.from_hierarchy."""

    def __init__(self, operators, interpolations, rank: int, world: int, group=None, transport=None, starts=None, replicate_below: int = 50000, coloring: int = 0):
        import ctypes as C
        import os

        import numpy as np
        import scipy.sparse as sp
        import torch
        import torch.distributed as dist

        from .capi import check, lib
        from .wrappers import MCSOR

        self.rank, self.world, self.group = rank, world, group
        L = len(operators)
        assert L >= 2 and len(interpolations) == L
        n = [len(op[0]) - 1 for op in operators]
        # --- transport: the generic all-gather of pmg_dist.c ("ipc" peer stores or RCCL), agreed between the ranks
        on_gpu = dist.get_backend(group) == "nccl"
        want_tr = transport or os.environ.get("PMG_DIST_TRANSPORT")
        self._drv, self.transport = _agree_on_driver([want_tr] if want_tr else ["ipc", "rccl"], lambda cand: IpcSlabDriver(None, rank, world, group=group) if cand == "ipc" else RcclSlabDriver(None, rank, world, group=group), rank, world, group, "row-block hierarchy")
        if self._drv is None:
            raise RuntimeError("DistAIJMGMC needs the 'ipc' or 'rccl' transport of the HIP library")
        # --- the global colouring of the row-block levels: the library's first-fit rule, as MGMC.from_hierarchy applies it
        fold = 1
        while fold < L - 1 and n[fold] <= replicate_below:
            fold += 1
        colorings = [None] * L
        for l in range(fold, L):
            rp, ci, v = operators[l]
            mc = MCSOR(np.ascontiguousarray(rp, np.int32), np.ascontiguousarray(ci, np.int32), np.ascontiguousarray(v, np.float64), coloring).setup()  # capi.COLORING_GREEDY (0) or COLORING_ITERATED: what MGMC.from_hierarchy + set_coloring(rule) sweeps with on one device
            colorings[l] = (mc.get_coloring(), mc.get_num_colors())
            mc.destroy()
        H = rowblock_hierarchy(operators, interpolations, colorings, rank, world, starts=starts, replicate_below=replicate_below, group=group)
        assert H["fold"] == fold
        self.fold, self.starts = fold, H["starts"]
        # --- hand everything to the C hierarchy
        self._keep = [H]
        self._h = C.c_void_p()
        check(lib.pmg_mgmc_create_hierarchy(L, C.byref(self._h)))
        check(lib.pmg_mgmc_set_coloring(self._h, int(coloring)))  # the replicated AIJ levels below the fold
        a0 = sp.csr_matrix((np.asarray(operators[0][2], np.float64), np.asarray(operators[0][1], np.int64), np.asarray(operators[0][0], np.int64)), shape=(n[0], n[0]))
        a0.sort_indices()
        rp0, ci0, v0 = a0.indptr.astype(np.int32), a0.indices.astype(np.int32), np.ascontiguousarray(a0.data)
        self._keep.append((rp0, ci0, v0))
        check(lib.pmg_mgmc_set_level_operator(self._h, 0, n[0], rp0.ctypes.data, ci0.ctypes.data, v0.ctypes.data))
        cs = np.ascontiguousarray(self.starts[fold - 1], np.int64)  # who restricts which rows of the highest replicated level
        self._keep.append(cs)
        check(lib.pmg_mgmc_set_rowblock_transport(self._h, self._drv._h, cs.ctypes.data))
        for l in range(1, fold):  # replicated level: the whole operator and interpolation, as MGMC.from_hierarchy takes them
            rp, ci, v = (np.ascontiguousarray(a_, t_) for a_, t_ in zip(operators[l], (np.int32, np.int32, np.float64)))
            prp, pci, pv = (np.ascontiguousarray(a_, t_) for a_, t_ in zip(interpolations[l], (np.int32, np.int32, np.float64)))
            self._keep += [rp, ci, v, prp, pci, pv]
            check(lib.pmg_mgmc_set_level_operator(self._h, l, n[l], rp.ctypes.data, ci.ctypes.data, v.ctypes.data))
            check(lib.pmg_mgmc_set_level_interpolation(self._h, l, n[l], n[l - 1], prp.ctypes.data, pci.ctypes.data, pv.ctypes.data))
        for l in range(fold, L):
            Lv = H["levels"][l]
            plan, nl = Lv["plan"], Lv["nowned"] + len(Lv["ghosts"])
            check(lib.pmg_mgmc_set_level_operator(self._h, l, nl, Lv["rp"].ctypes.data, Lv["ci"].ctypes.data, Lv["v"].ctypes.data))
            check(lib.pmg_mgmc_set_level_rowblock(self._h, l, Lv["row0"], Lv["nowned"], Lv["ncolors"], Lv["colors"].ctypes.data, plan["send_ptr"].ctypes.data, plan["send_rows"].ctypes.data, plan["counts"].ctypes.data,
                                                  plan["recv_ptr"].ctypes.data, plan["recv_src"].ctypes.data, plan["recv_rows"].ctypes.data))
            (prp, pci, pv), (rrp, rci, rv) = Lv["P"], Lv["R"]
            check(lib.pmg_mgmc_set_level_interpolation(self._h, l, Lv["nowned"], Lv["ncoarse_local"], prp.ctypes.data, pci.ctypes.data, pv.ctypes.data))
            check(lib.pmg_mgmc_set_level_restriction(self._h, l, len(rrp) - 1, nl, rrp.ctypes.data, rci.ctypes.data, rv.ctypes.data))
        top = H["levels"][L - 1]
        self.n_owned, self.n_local = top["nowned"], top["nowned"] + len(top["ghosts"])
        self.row_range = (top["row0"], top["row0"] + top["nowned"])
        self.levels = L

    def set_smoother(self, scaled: bool, omega: float = 1.0, sweep_type: int = SOR_FORWARD_SWEEP, its: int = 1):
        from .capi import check, lib

        check(lib.pmg_mgmc_set_smoother(self._h, int(scaled), omega, sweep_type, its))

    def set_correction_form(self, literal: bool):
        from .capi import check, lib

        check(lib.pmg_mgmc_set_correction_form(self._h, int(literal)))

    def algorithmic_bytes(self):
        """(this rank's algorithmic bytes of one sample, per level): pmg_mgmc_get_algorithmic_bytes"""
        import ctypes as C

        import numpy as np

        from .capi import check, lib

        tot, per = C.c_double(), np.zeros(self.levels)
        check(lib.pmg_mgmc_get_algorithmic_bytes(self._h, C.byref(tot), per.ctypes.data))
        return tot.value, per

    def set_lowrank(self, B_owned, S):
        """MATLRC fine operator A + B diag(S) B^T, propagated to every level with the hierarchy's own restriction
        (reference src/pc_gamgmc.c:157-196): B_owned = this rank's rows of B (n_owned x k).  Before setup()."""
        import numpy as np

        from .capi import check, lib

        B = np.asarray(B_owned, np.float64)
        S = np.ascontiguousarray(S, np.float64)
        assert B.shape == (self.n_owned, len(S))
        Bl = np.zeros((self.n_local, len(S)), order="F")
        Bl[: self.n_owned] = B
        check(lib.pmg_mgmc_set_lowrank(self._h, len(S), Bl.ctypes.data, S.ctypes.data))

    def setup(self):
        from .capi import check, lib

        err = None
        try:
            check(lib.pmg_mgmc_setup(self._h))
        except Exception as e:  # noqa: BLE001
            err = e
        _all_ok(err, self.group, "row-block hierarchy set-up")
        return self

    def sample(self, b_owned, y_owned, its: int, seed: int, counter0: int = 0, guesszero: bool = False) -> int:
        """`its` samples; b_owned, y_owned: this rank's rows of the finest level (device tensors, y updated in place)"""
        import ctypes as C

        import torch

        from .capi import check, lib
        from .wrappers import _ptr, _stream

        b = torch.zeros(self.n_local, dtype=torch.float64, device="cuda")
        y = torch.zeros(self.n_local, dtype=torch.float64, device="cuda")
        b[: self.n_owned] = b_owned
        y[: self.n_owned] = y_owned
        out = C.c_uint64()
        check(lib.pmg_mgmc_sample(self._h, _ptr(b), _ptr(y), its, int(guesszero), seed, counter0, C.byref(out), None, None, _stream()))
        y_owned.copy_(y[: self.n_owned])
        return out.value

    def destroy(self):
        import ctypes as C

        from .capi import lib

        if getattr(self, "_h", None) is not None and self._h:
            lib.pmg_mgmc_destroy(C.byref(self._h))
            self._h = None
        if getattr(self, "_drv", None) is not None:
            self._drv.destroy()  # collective: disconnect, barrier, destroy
            self._drv = None


class CRowBlock:
    """The multi-rank samplers on MATMPIAIJ row blocks reached through the C-ABI ALONE (pmg_rowblock.c): transport bootstrap,
    colouring, ghost plans, local operators and the sample loops are C; Python only supplies the byte all-gather callback
    (torch.distributed here, MPI_Allgather in adapter/, pipes in examples/pmg_bench.c) and device tensors.  This is what a
    PETSc caller with an MPI_Comm uses; DistMCSOR / DistAIJMGMC above are the round-2 Python builders the C plans are pinned to.

      CRowBlock.sampler(...)  MCSORCreate + MCSORSetUp + the sample loop on a MATMPIAIJ (reference src/mc_sor.c:298-381,553-605)
      CRowBlock.mgmc(...)     PCGAMGMC on a hierarchy of MATMPIAIJ levels (src/pc_gamgmc.c:157-264)"""

    def __init__(self, rank: int, world: int, group=None, transport: str = "ipc"):
        import ctypes as C

        from . import capi
        from .capi import check, lib

        self.rank, self.world = rank, world
        self.comm, self._keep = capi.torch_host_comm(rank, world, group)
        self._dist = C.c_void_p()
        path = capi.torch_rccl_path()
        check(lib.pmg_dist_create_comm(C.byref(self.comm), transport.encode(), None, path.encode() if path else None, C.byref(self._dist)))
        self.transport = transport
        self._mc, self._dm, self._mg, self._rbh = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.nowned = 0

    @property
    def dist_handle(self):
        return self._dist

    def sampler(self, row_starts, rowptr, colidx_global, vals, omega=1.0, colors=None, ncolors=0):
        """this rank's rows with GLOBAL columns (pmg_rowblock_merge_mpiaij of MatMPIAIJGetSeqAIJ's blocks); colors None: the
        library's first-fit colouring of the global matrix"""
        import ctypes as C

        import numpy as np

        from .capi import check, lib

        rs = np.ascontiguousarray(row_starts, np.int64)
        rp, ci, v = np.ascontiguousarray(rowptr, np.int64), np.ascontiguousarray(colidx_global, np.int64), np.ascontiguousarray(vals, np.float64)
        col = None if colors is None else np.ascontiguousarray(colors, np.int32)
        self.nowned = int(rs[self.rank + 1] - rs[self.rank])
        check(lib.pmg_rowblock_sampler_create(C.byref(self.comm), self._dist, rs.ctypes.data, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, 64, ncolors, col.ctypes.data if col is not None else None, omega, C.byref(self._mc), C.byref(self._dm)))
        return self

    def sample(self, b_owned, y_owned, its: int, seed: int, counter0: int = 0, scaled: bool = True, sweep_type: int = SOR_FORWARD_SWEEP) -> int:
        import ctypes as C

        from .capi import check, lib
        from .wrappers import _ptr, _stream

        out = C.c_uint64()
        if self._mg:  # the hierarchy's vectors have one entry per LOCAL row of the finest level (ghost entries ignored)
            bp, yp = self._pad(b_owned), self._pad(y_owned)  # both stay referenced until the call has been enqueued
            check(lib.pmg_mgmc_sample(self._mg, _ptr(bp), _ptr(yp), its, 0, seed, counter0, C.byref(out), None, None, _stream()))
            y_owned.copy_(yp[: self.nowned])
        else:
            check(lib.pmg_distmcsor_sample(self._dm, self.nowned, _ptr(b_owned), _ptr(y_owned), its, int(scaled), sweep_type, seed, counter0, C.byref(out), _stream()))
        return out.value

    def apply(self, b_owned, y_owned, sweep_type: int = SOR_FORWARD_SWEEP):
        from .capi import check, lib
        from .wrappers import _ptr, _stream

        check(lib.pmg_distmcsor_apply(self._dm, self.nowned, _ptr(b_owned), _ptr(y_owned), sweep_type, _stream()))

    def _pad(self, t):
        import torch

        p = torch.zeros(self.nlocal, dtype=torch.float64, device="cuda")
        p[: self.nowned] = t
        return p

    def mgmc(self, levels, replicate_below: int = 50000, smoother=(True, 1.0, SOR_FORWARD_SWEEP, 1), lowrank=None):
        """levels[l] = dict(n=global rows, row0=, A=(rowptr, colidx_global, vals) of this rank's rows, P=(rowptr,
        colidx_global_coarse, vals) of this rank's rows of P_l (l >= 1)); level 0 = coarsest"""
        import ctypes as C

        import numpy as np

        from .capi import RbhLevelView, check, lib

        L = len(levels)
        check(lib.pmg_rbh_create(C.byref(self.comm), L, replicate_below, C.byref(self._rbh)))
        keep = []
        for l, Lv in enumerate(levels):
            rp, ci, v = (np.ascontiguousarray(a_, t_) for a_, t_ in zip(Lv["A"], (np.int64, np.int64, np.float64)))
            keep += [rp, ci, v]
            check(lib.pmg_rbh_set_level_operator(self._rbh, l, Lv["n"], Lv["row0"], len(rp) - 1, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, 64))
            if l >= 1:
                prp, pci, pv = (np.ascontiguousarray(a_, t_) for a_, t_ in zip(Lv["P"], (np.int64, np.int64, np.float64)))
                keep += [prp, pci, pv]
                check(lib.pmg_rbh_set_level_interpolation(self._rbh, l, len(prp) - 1, prp.ctypes.data, pci.ctypes.data, pv.ctypes.data, 64))
        check(lib.pmg_rbh_build(self._rbh))
        check(lib.pmg_rbh_create_mgmc(self._rbh, self._dist, C.byref(self._mg)))
        check(lib.pmg_mgmc_set_smoother(self._mg, int(smoother[0]), smoother[1], smoother[2], smoother[3]))
        view = RbhLevelView()
        check(lib.pmg_rbh_get_level(self._rbh, L - 1, C.byref(view)))
        self.nowned, self.nlocal = view.nowned, view.nlocal
        if lowrank is not None:
            B, S = lowrank
            Bl = np.zeros((self.nlocal, len(S)), order="F")
            Bl[: self.nowned] = B
            S = np.ascontiguousarray(S, np.float64)
            check(lib.pmg_mgmc_set_lowrank(self._mg, len(S), Bl.ctypes.data, S.ctypes.data))
        err = None
        try:
            check(lib.pmg_mgmc_setup(self._mg))
        except Exception as e:  # noqa: BLE001
            err = e
        _all_ok(err, None, "row-block hierarchy set-up")
        lib.pmg_rbh_destroy(C.byref(self._rbh))  # the arrays were borrowed until set-up
        return self

    def destroy(self):
        """collective"""
        import ctypes as C

        from .capi import lib

        if self._mg:
            lib.pmg_mgmc_destroy(C.byref(self._mg))
        if self._dm:
            lib.pmg_distmcsor_destroy(C.byref(self._dm))
        if self._mc:
            lib.pmg_mcsor_destroy(C.byref(self._mc))
        if self._rbh:
            lib.pmg_rbh_destroy(C.byref(self._rbh))
        if self._dist:
            lib.pmg_dist_destroy_comm(C.byref(self.comm), C.byref(self._dist))
