"""world_size-2 `gloo` test (CPU) of the multi-device sample loop: slab ownership, the per-colour halo exchange
through `SlabHalo` with the offsets the C library reports (`pmg_grid_halo_plane`, no GPU needed for that), and the
colour/draw schedule of `run_samples`.  The local colour sweep is injected from the CPU oracle (test-only), so the
test checks the distributed SCHEDULE: the two-rank chain must equal the one-domain chain bit for bit."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class CvecLayout:
    """numpy restatement of the documented cvec layout for one slab."""

    def __init__(self, nx, ny, nzg, kz0, nz):
        from parmgmc_amd.capi import lib

        self.nx, self.ny, self.nzg, self.kz0, self.nz = nx, ny, nzg, kz0, nz
        h = C.c_void_p()
        assert lib.pmg_grid_create(nx, ny, nzg, kz0, nz, 1.0, C.byref(h)) == 0
        ln = C.c_int64()
        lib.pmg_grid_cvec_len(h, C.byref(ln))
        self.len = ln.value
        self.planes = []
        for c in (0, 1):
            row = []
            for s in (0, 1):
                a, b, n = C.c_int64(), C.c_int64(), C.c_int64()
                assert lib.pmg_grid_halo_plane(h, c, s, C.byref(a), C.byref(b), C.byref(n)) == 0
                row.append((a.value, b.value, n.value))
            self.planes.append(row)
        lib.pmg_grid_destroy(C.byref(h))
        self.sp = self.planes[0][0][2]
        self.sx = self.sp // ny
        self.cs = self.len // 2
        # position of every point of the EXTENDED slab (ghost planes kz0-1 .. kz0+nz) in the cvec
        k, j, i = np.meshgrid(np.arange(-1, nz + 1), np.arange(ny), np.arange(nx), indexing="ij")
        c = (i + j + k + kz0) & 1
        self.pos = (c * self.cs + (k + 1) * self.sp + j * self.sx + (i >> 1)).ravel()

    def ext_from_cvec(self, cv):
        return cv[self.pos]  # natural order over planes kz0-1 .. kz0+nz

    def cvec_from_ext(self, ext):
        cv = np.zeros(self.len)
        cv[self.pos] = ext
        return cv


def _worker(rank, world, port, nx, ny, nz, kappa, omega, sweep_type, its, b, y0, out_queue):
    import torch
    import torch.distributed as dist

    from parmgmc_amd.dist import SlabHalo, run_samples
    from parmgmc_amd.slab import slab_cuts

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cuts = slab_cuts(nz, world)
    lo, hi = cuts[rank], cuts[rank + 1]
    lay = CvecLayout(nx, ny, nz, lo, hi - lo)
    A = O.shifted_laplace(nx, ny, nz, kappa)
    dp, idg, sd = O.diag_pointers(A), O.idiag(A, omega), O.sqrtdiag(A, omega, True)
    colors = O.coloring_redblack(nx, ny, nz)
    plane = nx * ny
    elo, ehi = max(lo - 1, 0), min(hi + 1, nz)  # extended slab clipped to the domain

    def ext_to_glob(ext, glob):
        full = ext.reshape(hi - lo + 2, plane)
        glob[elo * plane:ehi * plane] = full[(elo - (lo - 1)):(ehi - (lo - 1))].ravel()

    def glob_to_ext(glob):
        ext = np.zeros((hi - lo + 2, plane))
        ext[(elo - (lo - 1)):(ehi - (lo - 1))] = glob[elo * plane:ehi * plane].reshape(-1, plane)
        return ext.ravel()

    bglob = b.copy()
    yg0 = np.zeros(A.n)
    yg0[lo * plane:hi * plane] = y0[lo * plane:hi * plane]  # a rank knows ONLY its owned planes
    y = torch.from_numpy(lay.cvec_from_ext(glob_to_ext(yg0)))
    owned = np.arange(lo * plane, hi * plane, dtype=np.int32)
    L = O.lib()

    def sweep_planes(c, k0, nk, _b, yt, ctr):
        glob = np.zeros(A.n)
        ext_to_glob(lay.ext_from_cvec(yt.numpy()), glob)
        w = O.prepare_rhs(O.noise_grid(nx, ny, nz, 77, ctr), sd, bglob)
        sel = owned[(lo + k0) * plane - lo * plane:(lo + k0 + nk) * plane - lo * plane]
        rows = np.ascontiguousarray(sel[colors[sel] == c])
        L.orc_parsor_rows(len(rows), rows, A.rowptr, A.colidx, A.vals, dp, idg, omega, w, glob, None, None, None, None)
        # like the HIP kernel, write ONLY the swept points (an exchange into the ghost planes may be in flight)
        ext_idx = rows - (lo - 1) * plane  # natural index inside the extended slab (planes lo-1 .. hi)
        yt[torch.from_numpy(lay.pos[ext_idx])] = torch.from_numpy(glob[rows])

    halo = SlabHalo(rank, world, lay.planes)
    ctr = run_samples(sweep_planes, halo, hi - lo, None, y, its, sweep_type, 3)
    glob = np.zeros(A.n)
    ext_to_glob(lay.ext_from_cvec(y.numpy()), glob)
    out_queue.put((rank, lo, hi, glob[lo * plane:hi * plane].copy(), ctr))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sweep_type,world", [(O.SOR_FORWARD, 2), (O.SOR_SYMMETRIC, 2), (O.SOR_BACKWARD, 3)])
def test_two_rank_chain_equals_one_domain_chain(sweep_type, world):
    import torch.multiprocessing as mp

    nx, ny, nz, kappa, omega, its = 6, 5, 7, 2.0, 1.2, 2
    rng = np.random.default_rng(9)
    n = nx * ny * nz
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nx, ny, nz, kappa, omega, sweep_type, its, b, y0, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = np.concatenate([x[3] for x in parts])
    A = O.shifted_laplace(nx, ny, nz, kappa)
    want = O.gibbs_samples(A, O.coloring_redblack(nx, ny, nz), b, y0, its, lambda d: O.noise_grid(nx, ny, nz, 77, 3 + d), omega, sweep_type, True)
    assert np.array_equal(got, want)
    assert all(x[4] == 3 + its * (2 if sweep_type == O.SOR_SYMMETRIC else 1) for x in parts)


def test_slab_cuts_follow_petsc_ownership():
    from parmgmc_amd.slab import slab_cuts

    assert slab_cuts(512, 8) == [0, 64, 128, 192, 256, 320, 384, 448, 512]
    assert slab_cuts(10, 4) == [0, 3, 6, 8, 10]  # PETSC_DECIDE: the first n % size ranks get one more
    assert slab_cuts(3, 1) == [0, 3]
