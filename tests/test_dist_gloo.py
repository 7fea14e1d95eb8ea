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


# ------------------------------------------------------------------------------------------------------------
# The distributed V-cycle (pmg_mgmc_create_dmda_slab) restated rank by rank on the CPU: ownership of coarse planes,
# halos per sweep phase, residual halo + restriction, all-gather into the replicated coarse part, prolongation onto
# ghost planes.  Every rank keeps full-size arrays but poisons (NaN) everything that is neither owned nor a current
# ghost plane, so a read outside what the algorithm says it has shows up in the result.
# ------------------------------------------------------------------------------------------------------------
GOLD = 0x9E3779B97F4A7C15
M64 = (1 << 64) - 1


def _lvl_seed(seed, l):
    return (seed + GOLD * (l + 1)) & M64


def _mg_levels(grid, kappa, levels):
    dims = [grid]
    for _ in range(levels - 1):
        dims.append(tuple((d - 1) // 2 + 1 if d > 1 else 1 for d in dims[-1]))
    dims = dims[::-1]
    lv = [None] * levels
    lv[levels - 1] = dict(A=O.shifted_laplace(*grid, kappa).scipy(), P=None, dims=dims[-1])
    for l in range(levels - 1, 0, -1):
        lv[l]["P"] = O.q1_interp(*dims[l - 1])
        lv[l - 1] = dict(A=O.galerkin(lv[l]["A"], lv[l]["P"]), P=None, dims=dims[l - 1])
    return lv


def _slab_vcycle_worker(rank, world, port, grid, kappa, levels, nrep, b, y0, seed, sample, q):
    """one MGMC sample (in-place form: V-cycle started from the guess y) with `nrep` replicated coarse levels"""
    import torch
    import torch.distributed as dist

    from parmgmc_amd.slab import slab_cuts

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lv = _mg_levels(grid, kappa, levels)
    top = levels - 1
    L = O.lib()
    csr = [O.CSR.from_scipy(x["A"]) for x in lv]
    dp = [O.diag_pointers(a) for a in csr]
    idg = [O.idiag(a, 1.0) for a in csr]
    sd = [O.sqrtdiag(a, 1.0, False) for a in csr]
    cols = [O.coloring_parity8(*x["dims"]) for x in lv]
    cols[top] = O.coloring_redblack(*grid)
    plane = [x["dims"][0] * x["dims"][1] for x in lv]
    nzl = [x["dims"][2] for x in lv]
    cuts = [None] * levels
    cuts[top] = slab_cuts(grid[2], world)
    for l in range(top, 0, -1):
        cuts[l - 1] = [(c + 1) // 2 for c in cuts[l]]  # coarse plane K belongs to the owner of fine plane 2K
    dist_lvl = [l >= nrep for l in range(levels)]  # levels 0 .. nrep-1 are replicated
    ctr = {l: 64 * sample for l in range(levels)}

    def owned(l):
        return (cuts[l][rank], cuts[l][rank + 1]) if dist_lvl[l] else (0, nzl[l])

    def rows_of(l, k0, k1):
        return np.arange(k0 * plane[l], k1 * plane[l], dtype=np.int32)

    def poison(l, v):
        """everything but the owned planes and the in-domain ghost planes becomes NaN"""
        lo, hi = owned(l)
        out = np.full_like(v, np.nan)
        a, e = max(lo - 1, 0) * plane[l], min(hi + 1, nzl[l]) * plane[l]
        out[a:e] = v[a:e]
        return out

    def halo(l, v):
        """boundary planes to the neighbours' ghost planes (blocking pairwise exchange, even ranks send first)"""
        if not dist_lvl[l]:
            return
        lo, hi = owned(l)
        p = plane[l]
        for nb, mine, ghost in ((rank - 1, lo, lo - 1), (rank + 1, hi - 1, hi)):
            if nb < 0 or nb >= world:
                continue
            snd = torch.from_numpy(v[mine * p:(mine + 1) * p].copy())
            rcv = torch.zeros(p, dtype=torch.float64)
            if rank % 2 == 0:
                dist.send(snd, nb)
                dist.recv(rcv, nb)
            else:
                dist.recv(rcv, nb)
                dist.send(snd, nb)
            v[ghost * p:(ghost + 1) * p] = rcv.numpy()

    def sweep_rows(l, rows, w, x):
        rows = np.ascontiguousarray(rows, np.int32)
        L.orc_parsor_rows(len(rows), rows, csr[l].rowptr, csr[l].colidx, csr[l].vals, dp[l], idg[l], 1.0, w, x, None, None, None, None)

    def smooth(l, bb, x):
        lo, hi = owned(l)
        mine = rows_of(l, lo, hi)
        xi = O.noise_grid(*grid, _lvl_seed(seed, l), ctr[l]) if l == top else O.noise_rows(csr[l].n, _lvl_seed(seed, l), ctr[l])
        ctr[l] += 1
        w = O.prepare_rhs(xi, sd[l], np.nan_to_num(bb))  # only owned rows of w are used
        if l == top:  # red-black: one halo per colour
            phases = [[0], [1]]
        else:  # parity colours: even planes (0..3), halo, odd planes (4..7), halo
            phases = [[0, 1, 2, 3], [4, 5, 6, 7]]
        for ph in phases:
            for c in ph:
                sweep_rows(l, mine[cols[l][mine] == c], w, x)
            halo(l, x)

    def level_cycle(l, bb, x):
        """x holds the guess (owned + ghost planes current); returns after the post-smoothing"""
        if l == 0:
            xi = O.noise_rows(csr[0].n, _lvl_seed(seed, 0), ctr[0])
            return O.chol_sample(O.potrf_lower(csr[0].dense()), bb, xi)
        lo, hi = owned(l)
        mine = rows_of(l, lo, hi)
        smooth(l, bb, x)
        r = np.full(csr[l].n, np.nan)
        r[mine] = bb[mine] - lv[l]["A"][mine] @ np.nan_to_num(x, nan=1e300)  # structural zeros skip the poison; 1e300 would show
        halo(l, r)
        # restriction into the coarse planes this rank owns (all of them on a replicated coarse level it reaches alone)
        clo, chi = (cuts[l - 1][rank], cuts[l - 1][rank + 1]) if dist_lvl[l] else (0, nzl[l - 1])
        crow = rows_of(l - 1, clo, chi)
        bc = np.full(csr[l - 1].n, np.nan)
        bc[crow] = lv[l]["P"].T.tocsr()[crow] @ np.nan_to_num(r, nan=1e300)
        if dist_lvl[l] and not dist_lvl[l - 1]:  # fold: all-gather the owned planes
            parts = [torch.zeros((cuts[l - 1][q + 1] - cuts[l - 1][q]) * plane[l - 1], dtype=torch.float64) for q in range(world)]
            for src in range(world):
                if src == rank:
                    parts[src] = torch.from_numpy(bc[crow].copy())
                dist.broadcast(parts[src], src)
            bc = np.concatenate([t.numpy() for t in parts])
        xc = poison(l - 1, np.zeros(csr[l - 1].n)) if dist_lvl[l - 1] else np.zeros(csr[l - 1].n)
        xc = level_cycle(l - 1, bc, xc)
        # prolongation onto the owned planes AND the in-domain ghost planes
        ext = rows_of(l, max(lo - 1, 0), min(hi + 1, nzl[l])) if dist_lvl[l] else mine
        x[ext] = x[ext] + lv[l]["P"].tocsr()[ext] @ np.nan_to_num(xc, nan=1e300)
        smooth(l, bb, x)
        return x

    lo, hi = owned(top)
    bt = np.full(csr[top].n, np.nan)
    bt[lo * plane[top]:hi * plane[top]] = b[lo * plane[top]:hi * plane[top]]
    xt = np.full(csr[top].n, np.nan)
    xt[lo * plane[top]:hi * plane[top]] = y0[lo * plane[top]:hi * plane[top]]
    halo(top, xt)  # what pmg_dist_sample_cvec does first
    xt = level_cycle(top, bt, xt)
    q.put((rank, xt[lo * plane[top]:hi * plane[top]].copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("grid,levels,world,nrep", [((5, 5, 9), 3, 2, 1), ((5, 5, 17), 3, 3, 1), ((9, 5, 9), 3, 2, 2)])
def test_slab_vcycle_algorithm_equals_one_domain_vcycle(grid, levels, world, nrep):
    """the distributed V-cycle's data motion on `world` gloo ranks against the oracle's
    one-domain V-cycle with the same noise: distributed coarse level(s) for nrep = 1, replicated from level 1 for 2"""
    import torch.multiprocessing as mp

    kappa, seed, sample = 1.5, 99, 2
    n = int(np.prod(grid))
    rng = np.random.default_rng(4)
    b, y0 = rng.standard_normal(n), rng.standard_normal(n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slab_vcycle_worker, args=(r, world, port, grid, kappa, levels, nrep, b, y0, seed, sample, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = np.concatenate([x[1] for x in parts])
    # one-domain oracle cycle, in place on (b, y0): V-cycle started from the guess
    lv = _mg_levels(grid, kappa, levels)
    top = levels - 1
    csr = [O.CSR.from_scipy(x["A"]) for x in lv]
    cols = [O.coloring_parity8(*x["dims"]) for x in lv]
    cols[top] = O.coloring_redblack(*grid)
    ctr = {l: 64 * sample for l in range(levels)}

    def noise(l):
        c = ctr[l]
        ctr[l] += 1
        return O.noise_grid(*grid, _lvl_seed(seed, l), c) if l == top else O.noise_rows(csr[l].n, _lvl_seed(seed, l), c)

    smooth = lambda l, rhs, x, leg: O.gibbs_samples(csr[l], cols[l], rhs, x, 1, lambda d: noise(l), 1.0, O.SOR_FORWARD, False)
    Lc = O.potrf_lower(csr[0].dense())
    want = O.vcycle(lv, top, b, y0.copy(), smooth, lambda rhs: O.chol_sample(Lc, rhs, noise(0)))
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-13
