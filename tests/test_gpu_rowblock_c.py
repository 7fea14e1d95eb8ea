"""The multi-rank samplers reached through the C-ABI alone (pmg_rowblock.c, pmg_woodbury.c; parmgmc_amd.dist.CRowBlock is
a ctypes shell around them): 2-3 ranks sharing the ONE GPU of the test box, "ipc" transport bootstrapped through the byte
all-gather callback, against the single-device objects.

  pmg_rowblock_sampler_create + pmg_distmcsor_sample / _apply   == MCSOR (MCSORApply_MPIAIJ, reference src/mc_sor.c:298-381), bit for bit
  pmg_rbh_* + pmg_rbh_create_mgmc                                == MGMC.from_hierarchy (src/pc_gamgmc.c:157-264), bit for bit; MATLRC 1e-12
  pmg_woodbury_* with row-distributed B (PCWOODBURY, src/woodbury.c) on row blocks and on z-slabs == the one-device chain, 1e-12
"""
import os
import socket
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
MESH = Path(__file__).resolve().parent / "golden" / "lshape.msh"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def lshape(refine):
    from parmgmc_amd.unstructured import assemble_p1, read_gmsh41_triangles, refine_uniform

    xy, tris = read_gmsh41_triangles(MESH)
    for _ in range(refine):
        xy, tris = refine_uniform(xy, tris)
    return assemble_p1(xy, tris, 1.0)


def lowrank_factors(n):
    rng = np.random.default_rng(11)
    B = np.zeros((n, 3))
    for c, (lo, hi) in enumerate([(0.05, 0.45), (0.3, 0.8), (0.55, 0.98)]):
        rows = np.arange(int(lo * n), int(hi * n), 3)
        B[rows, c] = rng.uniform(0.5, 1.5, len(rows)) / len(rows)
    return B, np.array([30.0, 80.0, 50.0])


def _run(target, world, *args, timeout=240):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q, *args)) for r in range(world)]
    for p in procs:
        p.start()
    parts = sorted((q.get(timeout=timeout) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for x in parts:
        assert not isinstance(x[1], str), x[1]
    return parts


def _init(rank, world, port):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _sampler_worker(rank, world, port, q, omega, sweep_type, its, woodbury):
    import torch
    import torch.distributed as dist

    try:
        _init(rank, world, port)
        from parmgmc_amd.dist import CRowBlock
        from parmgmc_amd.wrappers import WoodburySampler

        A = lshape(1)
        n = A.shape[0]
        rs = np.array([round(n * r / world) for r in range(world + 1)], np.int64)
        mine = A[rs[rank]:rs[rank + 1]].tocsr()
        mine.sort_indices()
        rb = CRowBlock(rank, world).sampler(rs, mine.indptr, mine.indices, mine.data, omega=omega)
        rng = np.random.default_rng(5)
        b_all, y_all = rng.standard_normal(n), rng.standard_normal(n)
        b = torch.as_tensor(b_all[rs[rank]:rs[rank + 1]], device="cuda")
        y = torch.as_tensor(y_all[rs[rank]:rs[rank + 1]], device="cuda")
        if not woodbury:
            ya = y.clone()
            rb.apply(b, ya, sweep_type)  # MCSORApply
            ctr = rb.sample(b, y, its - 1, seed=42, counter0=1, sweep_type=sweep_type)
            ctr = rb.sample(b, y, 1, seed=42, counter0=ctr, sweep_type=sweep_type)
            torch.cuda.synchronize()
            q.put((rank, y.cpu().numpy(), ya.cpu().numpy(), ctr))
        else:
            B, S = lowrank_factors(n)

            def solve(bb, xx):  # 12 deterministic symmetric sweeps from the zero guess
                for _ in range(12):
                    rb.apply(bb, xx, 3)

            wb = WoodburySampler(B[rs[rank]:rs[rank + 1]], S, solve, lambda w, yy, c: rb.sample(w, yy, 1, seed=42, counter0=c, sweep_type=sweep_type), dist_handle=rb.dist_handle)
            ctr = wb.run(b, y, its, seed=42, counter0=1)
            torch.cuda.synchronize()
            q.put((rank, y.cpu().numpy(), wb.correction(), ctr))
            wb.destroy()
        dist.barrier()
        rb.destroy()
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,sweep_type", [(2, 1), (3, 3)], ids=["2ranks", "3ranks_symmetric"])
def test_c_row_block_sampler_is_the_single_device_chain(world, sweep_type):
    import torch

    from parmgmc_amd import MCSOR

    omega, its = 1.15, 3
    parts = _run(_sampler_worker, world, omega, sweep_type, its, False)
    A = lshape(1)
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    one = MCSOR(A.indptr, A.indices, A.data).setup()
    one.set_omega(omega)
    one.set_sweep_type(sweep_type)
    bd = torch.as_tensor(b_all, device="cuda")
    ya = torch.as_tensor(y_all, device="cuda")
    one.apply(bd, ya)
    yd = torch.as_tensor(y_all, device="cuda")
    ctr = one.sample(bd, yd, its, seed=42, counter0=1)
    assert all(x[3] == ctr for x in parts)
    assert np.array_equal(np.concatenate([x[2] for x in parts]), ya.cpu().numpy())  # MCSORApply
    assert np.array_equal(np.concatenate([x[1] for x in parts]), yd.cpu().numpy())  # the chain


@pytest.mark.parametrize("world", [2, 3])
def test_woodbury_on_row_blocks(world):
    """PCWOODBURY (reference src/woodbury.c) with B's rows distributed like A's: G = C (S^-1 + B^T C)^-1 and every B^T y are
    summed over the ranks in rank order"""
    import torch

    from parmgmc_amd import MCSOR
    from parmgmc_amd.wrappers import WoodburySampler

    omega, its = 1.0, 3
    parts = _run(_sampler_worker, world, omega, 1, its, True)
    A = lshape(1)
    n = A.shape[0]
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(n), rng.standard_normal(n)
    one = MCSOR(A.indptr, A.indices, A.data).setup()
    B, S = lowrank_factors(n)

    def solve(bb, xx):
        one.set_sweep_type(3)
        for _ in range(12):
            one.apply(bb, xx)
        one.set_sweep_type(1)

    wb = WoodburySampler(B, S, solve, lambda w, yy, c: one.sample(w, yy, 1, seed=42, counter0=c))
    yd = torch.as_tensor(y_all, device="cuda")
    ctr = wb.run(torch.as_tensor(b_all, device="cuda"), yd, its, seed=42, counter0=1)
    ref, G = yd.cpu().numpy(), wb.correction()
    assert all(x[3] == ctr for x in parts)
    got, Gd = np.concatenate([x[1] for x in parts]), np.concatenate([x[2] for x in parts])
    assert np.abs(Gd - G).max() / np.abs(G).max() < 1e-12
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-12
    plain = torch.as_tensor(y_all, device="cuda")  # the update is not a no-op
    one.sample(torch.as_tensor(b_all, device="cuda"), plain, its, seed=42, counter0=1)
    assert np.abs(plain.cpu().numpy() - ref).max() / np.abs(ref).max() > 1e-6
    # and G is what the reference defines: C = solver(B), G = C (S^-1 + B^T C)^-1
    Cm = np.zeros((n, 3))
    for c in range(3):
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        solve(torch.as_tensor(np.ascontiguousarray(B[:, c]), device="cuda"), x)
        Cm[:, c] = x.cpu().numpy()
    want = Cm @ np.linalg.inv(np.diag(1.0 / S) + B.T @ Cm)
    assert np.abs(G - want).max() / np.abs(want).max() < 1e-12


def _hier_levels(ops, ps, rank, world):
    L = len(ops)
    n = [len(o[0]) - 1 for o in ops]
    out = []
    for l in range(L):
        A = sp.csr_matrix((ops[l][2], ops[l][1], ops[l][0]), shape=(n[l], n[l]))
        r0, r1 = (round(n[l] * r / world) for r in (rank, rank + 1))
        mine = A[r0:r1].tocsr()
        mine.sort_indices()
        d = dict(n=n[l], row0=r0, A=(mine.indptr, mine.indices, mine.data))
        if l >= 1:
            P = sp.csr_matrix((ps[l][2], ps[l][1], ps[l][0]), shape=(n[l], n[l - 1]))[r0:r1].tocsr()
            d["P"] = (P.indptr, P.indices, P.data)
        out.append(d)
    return out, n


def _mgmc_worker(rank, world, port, q, lowrank, its):
    import torch
    import torch.distributed as dist

    try:
        _init(rank, world, port)
        from parmgmc_amd.dist import CRowBlock
        from parmgmc_amd.unstructured import build_hierarchy

        ops, ps = build_hierarchy(lshape(2), coarse_max=60)
        levels, n = _hier_levels(ops, ps, rank, world)
        r0, r1 = levels[-1]["row0"], levels[-1]["row0"] + len(levels[-1]["A"][0]) - 1
        lr = None
        if lowrank:
            B, S = lowrank_factors(n[-1])
            lr = (B[r0:r1], S)
        rb = CRowBlock(rank, world).mgmc(levels, replicate_below=400, smoother=(True, 1.1, 3, 1), lowrank=lr)
        rng = np.random.default_rng(5)
        b_all, y_all = rng.standard_normal(n[-1]), rng.standard_normal(n[-1])
        b = torch.as_tensor(b_all[r0:r1], device="cuda")
        y = torch.as_tensor(y_all[r0:r1], device="cuda")
        ctr = rb.sample(b, y, its - 1, seed=42, counter0=1)
        ctr = rb.sample(b, y, 1, seed=42, counter0=ctr)
        torch.cuda.synchronize()
        q.put((rank, y.cpu().numpy(), ctr))
        dist.barrier()
        rb.destroy()
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,lowrank", [(2, False), (3, False), (2, True)], ids=["2ranks", "3ranks", "2ranks_lowrank"])
def test_c_row_block_hierarchy_is_the_single_device_chain(world, lowrank):
    import torch

    from parmgmc_amd import MGMC
    from parmgmc_amd.unstructured import build_hierarchy

    its = 3
    parts = _run(_mgmc_worker, world, lowrank, its)
    ops, ps = build_hierarchy(lshape(2), coarse_max=60)
    n = len(ops[-1][0]) - 1
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(n), rng.standard_normal(n)
    one = MGMC.from_hierarchy(ops, ps)
    one.set_smoother(True, 1.1, 3, 1)
    if lowrank:
        one.set_lowrank(*lowrank_factors(n))
    one.setup()
    yd = torch.as_tensor(y_all, device="cuda")
    ctr = one.sample(torch.as_tensor(b_all, device="cuda"), yd, its, seed=42, counter0=1)
    assert all(x[2] == ctr for x in parts)
    got, ref = np.concatenate([x[1] for x in parts]), yd.cpu().numpy()
    if lowrank:
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-12
    else:
        assert np.array_equal(got, ref)


def _slab_woodbury_worker(rank, world, port, q, grid, levels, its):
    import ctypes as C

    import torch
    import torch.distributed as dist

    try:
        _init(rank, world, port)
        from parmgmc_amd import make_observation_mats
        from parmgmc_amd.capi import check, lib
        from parmgmc_amd.dist import DistMGMC
        from parmgmc_amd.wrappers import WoodburySampler, _ptr, _stream

        nx, ny, nz = grid
        mg = DistMGMC(nx, ny, nz, 3.0, levels, rank, world, transport="ipc")
        mg.setup()
        k0, k1 = mg.plane_range
        B, S, f = make_observation_mats(nx, ny, nz, np.array([0.3, 0.3, 0.3, 0.7, 0.6, 0.7]), [0.25, 0.3], [1.0, -1.0], 1e-2, kz0=k0, nz_owned=k1 - k0)
        gs, drv = mg.grid_sampler.grid, mg.grid_sampler.rccl

        def solve(bb, xx):  # 10 deterministic symmetric slab sweeps (pmg_dist_apply_cvec) from the zero guess
            bc, xc = gs.to_cvec(bb), gs.to_cvec(xx)
            for _ in range(10):
                check(lib.pmg_dist_apply_cvec(drv._h, _ptr(bc), _ptr(xc), 3, _stream()))
            xx.copy_(gs.from_cvec(xc))

        wb = WoodburySampler(B, S, solve, lambda w, yy, c: mg.sample(w, yy, 1, seed=42, counter0=c), dist_handle=drv._h)
        rng = np.random.default_rng(5)
        b_all, y_all = rng.standard_normal(nx * ny * nz), rng.standard_normal(nx * ny * nz)
        lo, hi = k0 * nx * ny, k1 * nx * ny
        b = torch.as_tensor(b_all[lo:hi], device="cuda")
        y = torch.as_tensor(y_all[lo:hi], device="cuda")
        ctr = wb.run(b, y, its, seed=42, counter0=1)
        torch.cuda.synchronize()
        q.put((rank, y.cpu().numpy(), ctr))
        wb.destroy()
        dist.barrier()
        mg.destroy()
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_woodbury_on_z_slabs(world):
    """BASELINE config 5's other route (obs.c / woodbury.c on a DMDA split in z over 4 GPUs): the MGMC V-cycle of the PRIOR as
    A-sampler, deterministic slab sweeps as solver, B's rows on the slabs"""
    import torch

    from parmgmc_amd import MGMC, GridMCSOR, make_observation_mats
    from parmgmc_amd.wrappers import WoodburySampler

    grid, levels, its = (17, 17, 33), 3, 3
    parts = _run(_slab_woodbury_worker, world, grid, levels, its)
    nx, ny, nz = grid
    mg = MGMC(nx, ny, nz, 3.0, levels).setup()
    g = GridMCSOR(nx, ny, nz, 3.0)
    g.set_sweep_type(3)
    B, S, f = make_observation_mats(nx, ny, nz, np.array([0.3, 0.3, 0.3, 0.7, 0.6, 0.7]), [0.25, 0.3], [1.0, -1.0], 1e-2)

    def solve(bb, xx):
        bc, xc = g.to_cvec(bb), g.to_cvec(xx)
        for _ in range(10):
            g.apply_cvec(bc, xc)
        xx.copy_(g.from_cvec(xc))

    wb = WoodburySampler(B, S, solve, lambda w, yy, c: mg.sample(w, yy, 1, seed=42, counter0=c))
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(nx * ny * nz), rng.standard_normal(nx * ny * nz)
    yd = torch.as_tensor(y_all, device="cuda")
    ctr = wb.run(torch.as_tensor(b_all, device="cuda"), yd, its, seed=42, counter0=1)
    ref = yd.cpu().numpy()
    assert all(x[2] == ctr for x in parts)
    got = np.concatenate([x[1] for x in parts])
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-12
    plain = torch.as_tensor(y_all, device="cuda")
    mg.sample(torch.as_tensor(b_all, device="cuda"), plain, its, seed=42, counter0=1)
    assert np.abs(plain.cpu().numpy() - ref).max() / np.abs(ref).max() > 1e-6
