"""Two ranks sharing the ONE GPU of the test box: the production multi-device driver (`DistGridSampler`:
slab ownership, boundary-first overlap schedule, HIP plane-range kernels, halo exchange through
torch.distributed) against the single-device chain.  RCCL cannot run two ranks on one device, so the exchange
goes either through the `gloo` backend (torch transport, staged through host memory by SlabHalo) or through the
"ipc" transport of pmg_dist.c (hipIpc peer copies + interprocess events -- the SAME code path the multi-GPU run
uses, only the peer happens to be the same device); the kernels and the schedule are the ones `bench.py --gpus N`
runs."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nx, ny, nz, kappa, omega, sweep_type, its, q, transport=None):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parmgmc_amd.dist import DistGridSampler

    smp = DistGridSampler(nx, ny, nz, kappa, rank, world, omega=omega, sweep_type=sweep_type, transport=transport)
    assert smp.transport == (transport or "torch")
    g = smp.grid
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(nx * ny * nz), rng.standard_normal(nx * ny * nz)
    lo, hi = g.kz0 * nx * ny, (g.kz0 + g.nz) * nx * ny
    b = g.to_cvec(torch.as_tensor(b_all[lo:hi], device="cuda"))
    y = g.to_cvec(torch.as_tensor(y_all[lo:hi], device="cuda"))
    ctr = smp.sample_cvec(b, y, its - 1, seed=42, counter0=1)
    ctr = smp.sample_cvec(b, y, 1, seed=42, counter0=ctr)  # a second call continues the chain
    torch.cuda.synchronize()
    q.put((rank, g.from_cvec(y).cpu().numpy(), ctr))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sweep_type,world,transport,nz", [(1, 2, None, 11), (3, 3, None, 11), (1, 2, "ipc", 11), (3, 3, "ipc", 11), (2, 4, "ipc", 11), (3, 5, "ipc", 6)], ids=["fwd-2-torch", "sym-3-torch", "fwd-2-ipc", "sym-3-ipc", "bwd-4-ipc", "sym-5-ipc-one-plane-slabs"])
def test_ranks_on_one_gpu_reproduce_the_single_device_chain(sweep_type, world, transport, nz):
    import torch
    import torch.multiprocessing as mp

    from parmgmc_amd import GridMCSOR

    nx, ny, kappa, omega, its = 40, 18, 1.5, 1.1, 3  # nz = 6 on 5 ranks: slabs of 2, 1, 1, 1, 1 planes (one plane = both faces)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nx, ny, nz, kappa, omega, sweep_type, its, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    parts = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = np.concatenate([x[1] for x in parts])
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(nx * ny * nz), rng.standard_normal(nx * ny * nz)
    one = GridMCSOR(nx, ny, nz, kappa)
    one.set_omega(omega)
    one.set_sweep_type(sweep_type)
    yd = torch.as_tensor(y_all, device="cuda")
    ctr = one.sample(torch.as_tensor(b_all, device="cuda"), yd, its, seed=42, counter0=1)
    assert np.array_equal(got, yd.cpu().numpy())  # bit-identical chain for any number of ranks
    assert all(x[2] == ctr for x in parts)


def test_rccl_driver_loopback_on_one_gpu():
    """The C/RCCL sample loop (pmg_dist.c) on ONE GPU: a single rank made its own z-neighbour for the halo.  This
    exercises dlopen of RCCL, ncclCommInitRank, grouped ncclSend/ncclRecv on the comm stream and the event
    choreography; both slab faces are physical boundaries, so the result must equal the plain sampler bit for bit,
    and after the call each ghost plane must hold the matching boundary plane (the send really happened)."""
    import torch

    from parmgmc_amd import GridMCSOR
    from parmgmc_amd.dist import RcclSlabDriver

    nx, ny, nz, kappa = 36, 10, 7, 2.0
    rng = np.random.default_rng(1)
    b0, y0 = rng.standard_normal(nx * ny * nz), rng.standard_normal(nx * ny * nz)
    g = GridMCSOR(nx, ny, nz, kappa)
    g.set_omega(1.2)
    drv = RcclSlabDriver(g, 0, 1, loopback=True)
    b, y = g.to_cvec(torch.as_tensor(b0, device="cuda")), g.to_cvec(torch.as_tensor(y0, device="cuda"))
    ctr = drv.sample_cvec(b, y, 3, True, 3, 17, 5)  # symmetric sweeps
    torch.cuda.synchronize()
    ref = GridMCSOR(nx, ny, nz, kappa)
    ref.set_omega(1.2)
    ref.set_sweep_type(3)
    yr = ref.to_cvec(torch.as_tensor(y0, device="cuda"))
    assert ref.sample_cvec(b, yr, 3, 17, 5) == ctr == 11
    assert np.array_equal(g.from_cvec(y).cpu().numpy(), ref.from_cvec(yr).cpu().numpy())
    for c in (0, 1):
        for side in (0, 1):
            own, ghost, n = g.halo_plane(c, side)
            assert torch.equal(y[ghost:ghost + n], y[own:own + n]) and float(y[own:own + n].abs().sum()) > 0


def _mg_worker(rank, world, port, grid, kappa, levels, opts, its, q, transport):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for k, v in opts.get("env", {}).items():
        os.environ[k] = v
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parmgmc_amd.dist import DistMGMC

    nx, ny, nz = grid
    mg = DistMGMC(nx, ny, nz, kappa, levels, rank, world, transport=transport)
    mg.set_smoother(opts["scaled"], opts["omega"], opts["sweep"], opts["nu"])
    mg.set_coarse(opts["coarse"], opts["coarse_its"])
    mg.set_correction_form(opts["literal"])
    lo, hi = mg.plane_range[0] * nx * ny, mg.plane_range[1] * nx * ny
    if opts.get("lowrank"):
        B, S = _ball_factors(grid)
        mg.set_lowrank(B[lo:hi], S)  # this rank's rows of the observation vectors
    mg.setup()
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(nx * ny * nz), rng.standard_normal(nx * ny * nz)
    b = torch.as_tensor(b_all[lo:hi], device="cuda")
    y = torch.as_tensor(y_all[lo:hi], device="cuda")
    seen = []
    ctr = mg.sample(b, y, its - 1, seed=42, counter0=1, callback=lambda it, yy: seen.append(yy.cpu().numpy().copy()))
    ctr = mg.sample(b, y, 1, seed=42, counter0=ctr)  # a second call continues the chain
    torch.cuda.synchronize()
    q.put((rank, y.cpu().numpy(), ctr, seen))
    dist.barrier()
    mg.destroy()
    dist.destroy_process_group()


def _ball_factors(grid):
    """three ball-indicator observation vectors (reference src/obs.c:39-50) straddling the slab faces"""
    nx, ny, nz = grid
    X, Y, Z = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny), np.linspace(0, 1, nz), indexing="ij")
    pts = np.stack([X.ravel(order="F"), Y.ravel(order="F"), Z.ravel(order="F")], 1)
    B = np.zeros((nx * ny * nz, 3))
    for c, (ctr, r) in enumerate([((0.3, 0.4, 0.5), 0.22), ((0.7, 0.6, 0.25), 0.2), ((0.5, 0.3, 0.8), 0.18)]):
        inside = ((pts - np.asarray(ctr)) ** 2).sum(1) < r * r
        B[inside, c] = 1.0 / inside.sum()
    return B, np.array([40.0, 90.0, 60.0])


MG_DEFAULT = dict(scaled=False, omega=1.0, sweep=1, nu=1, coarse="cholsampler", coarse_its=1, literal=False, env={})


@pytest.mark.parametrize("grid,levels,world,opts", [
    ((17, 17, 17), 3, 2, {}),                                                             # 17 -> 9 -> 5: everything below the grid level replicated
    ((17, 17, 33), 4, 3, {"env": {"PMG_MG_REPLICATE_BELOW": "400"}}),                      # distributed class-stencil levels, uneven slabs
    ((33, 17, 33), 4, 4, {"env": {"PMG_MG_REPLICATE_BELOW": "2000"}, "scaled": True, "omega": 1.2, "sweep": 3, "nu": 2, "coarse": "gibbs", "coarse_its": 2}),
    ((17, 9, 33), 3, 2, {"env": {"PMG_MG_REPLICATE_BELOW": "100"}, "literal": True, "sweep": 2, "scaled": True}),
    ((9, 9, 33), 4, 5, {"env": {"PMG_MG_REPLICATE_BELOW": "50"}, "scaled": True, "sweep": 3}),  # three distributed levels; on the 9-plane level the ranks own 2,2,2,1,2 planes (5 ranks + this process = the box's limit of 6 GPU processes)
    ((17, 17, 33), 4, 3, {"env": {"PMG_MG_REPLICATE_BELOW": "400"}, "lowrank": True, "scaled": True, "sweep": 3}),  # config 5 across ranks: low-rank update on distributed and replicated levels
    ((17, 17, 17), 3, 2, {"lowrank": True, "literal": True, "coarse": "gibbs", "coarse_its": 2, "scaled": True}),
], ids=["replicated", "slab_levels_3ranks", "symmetric_gibbs_coarse_4ranks", "literal_backward", "one_plane_per_rank_5ranks", "lowrank_3ranks", "lowrank_literal_2ranks"])
def test_distributed_vcycle_reproduces_the_single_device_chain(grid, levels, world, opts):
    """z-slab MGMC (pmg_mgmc_create_dmda_slab) with `world` ranks sharing the one GPU over the ipc transport: sweeps
    with per-phase halos, residual halo + restriction, all-gather into the replicated coarse part, prolongation onto
    ghost planes -- bit-identical to the single-device sampler, sample by sample."""
    import torch
    import torch.multiprocessing as mp

    from parmgmc_amd import MGMC

    o = dict(MG_DEFAULT, **opts)
    kappa, its = 1.5, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mg_worker, args=(r, world, port, grid, kappa, levels, o, its, q, "ipc")) for r in range(world)]
    for p in procs:
        p.start()
    parts = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = int(np.prod(grid))
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(n), rng.standard_normal(n)
    one = MGMC(*grid, kappa, levels)
    one.set_smoother(o["scaled"], o["omega"], o["sweep"], o["nu"])
    one.set_coarse(o["coarse"], o["coarse_its"])
    one.set_correction_form(o["literal"])
    if o.get("lowrank"):
        one.set_lowrank(*_ball_factors(grid))
    one.setup()
    yd = torch.as_tensor(y_all, device="cuda")
    want = []
    ctr = one.sample(torch.as_tensor(b_all, device="cuda"), yd, its, seed=42, counter0=1, callback=lambda it, yy: want.append(yy.cpu().numpy().copy()))
    assert all(x[2] == ctr for x in parts)
    if o.get("lowrank"):  # the k-vectors B^T y are summed per rank, then over the ranks: equal to rounding, not bit for bit
        got, ref = np.concatenate([x[1] for x in parts]), yd.cpu().numpy()
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-12
        return
    for it in range(its - 1):  # every intermediate sample, as the callback saw it
        assert np.array_equal(np.concatenate([x[3][it] for x in parts]), want[it])
    assert np.array_equal(np.concatenate([x[1] for x in parts]), yd.cpu().numpy())


def _csr_lowrank_factors(n):
    """three observation-like vectors with supports that straddle every row-block cut, and their precisions"""
    rng = np.random.default_rng(11)
    B = np.zeros((n, 3))
    for c, (lo, hi) in enumerate([(0.05, 0.45), (0.3, 0.8), (0.55, 0.98)]):
        rows = np.arange(int(lo * n), int(hi * n), 3)
        B[rows, c] = rng.uniform(0.5, 1.5, len(rows)) / len(rows)
    return B, np.array([30.0, 80.0, 50.0])


def _csr_worker(rank, world, port, which, omega, sweep_type, its, q, transport=None, lowrank=False):
    import torch
    import torch.distributed as dist

    import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parmgmc_amd.dist import DistMCSOR

    A, colors = _csr_problem(which)
    n = A.n
    cuts = [round(n * r / world) for r in range(world + 1)]
    r0, r1 = cuts[rank], cuts[rank + 1]
    rp = A.rowptr[r0:r1 + 1] - A.rowptr[r0]
    sl = slice(A.rowptr[r0], A.rowptr[r1])
    smp = DistMCSOR(rp, A.colidx[sl], A.vals[sl], r0, r1, n, colors[r0:r1], int(colors.max()) + 1, rank, world, omega=omega, sweep_type=sweep_type, transport=transport)
    assert smp.transport == (transport or "torch") and (smp._c is not None) == (transport == "ipc")
    if lowrank:
        B, S = _csr_lowrank_factors(n)
        smp.set_lowrank(B[r0:r1], S)
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(n), rng.standard_normal(n)
    b = smp.to_layout(torch.as_tensor(b_all[r0:r1], device="cuda"))
    y = smp.to_layout(torch.as_tensor(y_all[r0:r1], device="cuda"))
    ctr = smp.sample_layout(b, y, its - 1, seed=42, counter0=1)
    ctr = smp.sample_layout(b, y, 1, seed=42, counter0=ctr)
    torch.cuda.synchronize()
    q.put((rank, smp.from_layout(y).cpu().numpy(), ctr))
    dist.barrier()
    dist.destroy_process_group()


def _csr_problem(which):
    import oracle as O

    if which == "lshape":
        from pathlib import Path

        from parmgmc_amd.unstructured import assemble_p1, read_gmsh41_triangles, refine_uniform

        xy, tris = read_gmsh41_triangles(Path(__file__).resolve().parent / "golden" / "lshape.msh")
        xy, tris = refine_uniform(xy, tris)
        A = O.CSR.from_scipy(assemble_p1(xy, tris, 1.0))
        return A, O.coloring_greedy(A)
    A = O.shifted_laplace(9, 7, 6, 2.0)
    return A, O.coloring_redblack(9, 7, 6)


@pytest.mark.parametrize("which,world,sweep_type,transport", [("lshape", 2, 1, None), ("lshape", 3, 3, None), ("grid", 4, 2, None), ("lshape", 2, 1, "ipc"), ("lshape", 3, 3, "ipc"), ("grid", 4, 2, "ipc")], ids=["lshape-2-torch", "lshape-3-sym-torch", "grid-4-bwd-torch", "lshape-2-c-ipc", "lshape-3-sym-c-ipc", "grid-4-bwd-c-ipc"])
def test_row_block_distributed_csr_sampler_reproduces_the_single_device_chain(which, world, sweep_type, transport):
    """MCSORApply_MPIAIJ (reference src/mc_sor.c:298-381) on the device: the matrix split into contiguous row blocks,
    ghost values exchanged before every colour -- by the Python loop of DistMCSOR over torch.distributed (gloo here) or by
    the C driver pmg_distmcsor.c over the "ipc" transport (no Python between colours) --, per-colour sliced-ELL
    sweeps with noise keyed on the global row -- against the single-device multicolour sampler with the same colouring,
    bit for bit (BASELINE config 4's matrix: the P1 operator of the reference's lshape.msh, refined once)."""
    import torch
    import torch.multiprocessing as mp

    from parmgmc_amd import MCSOR

    omega, its = 1.15, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_csr_worker, args=(r, world, port, which, omega, sweep_type, its, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    parts = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    A, colors = _csr_problem(which)
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(A.n), rng.standard_normal(A.n)
    one = MCSOR(A.rowptr, A.colidx, A.vals, user_colors=colors).setup()
    one.set_omega(omega)
    one.set_sweep_type(sweep_type)
    yd = torch.as_tensor(y_all, device="cuda")
    ctr = one.sample(torch.as_tensor(b_all, device="cuda"), yd, its, seed=42, counter0=1)
    assert all(x[2] == ctr for x in parts)
    assert np.array_equal(np.concatenate([x[1] for x in parts]), yd.cpu().numpy())


def _aij_hierarchy(refine, coarse_max):
    from pathlib import Path

    from parmgmc_amd.unstructured import assemble_p1, build_hierarchy, read_gmsh41_triangles, refine_uniform

    xy, tris = read_gmsh41_triangles(Path(__file__).resolve().parent / "golden" / "lshape.msh")
    for _ in range(refine):
        xy, tris = refine_uniform(xy, tris)
    return build_hierarchy(assemble_p1(xy, tris, 1.0), coarse_max=coarse_max)


def _aij_mg_worker(rank, world, port, refine, coarse_max, opts, its, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parmgmc_amd.dist import DistAIJMGMC

    ops, ps = _aij_hierarchy(refine, coarse_max)
    mg = DistAIJMGMC(ops, ps, rank, world, transport="ipc", replicate_below=opts["replicate_below"], coloring=opts.get("coloring", 0))
    mg.set_smoother(opts["scaled"], opts["omega"], opts["sweep"], opts["nu"])
    mg.set_correction_form(opts["literal"])
    n = len(ops[-1][0]) - 1
    r0, r1 = mg.row_range
    if opts.get("lowrank"):
        B, S = _csr_lowrank_factors(n)
        mg.set_lowrank(B[r0:r1], S)
    mg.setup()
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(n), rng.standard_normal(n)
    b = torch.as_tensor(b_all[r0:r1], device="cuda")
    y = torch.as_tensor(y_all[r0:r1], device="cuda")
    ctr = mg.sample(b, y, its - 1, seed=42, counter0=1)
    ctr = mg.sample(b, y, 1, seed=42, counter0=ctr)  # a second call continues the chain
    torch.cuda.synchronize()
    q.put((rank, y.cpu().numpy(), ctr, mg.levels))
    dist.barrier()
    mg.destroy()
    dist.destroy_process_group()


@pytest.mark.parametrize("refine,coarse_max,world,opts", [
    (2, 300, 1, {"scaled": True}),
    (2, 300, 2, {}),
    (2, 300, 3, {"scaled": True, "omega": 1.2, "sweep": 3, "nu": 2}),
    (3, 500, 4, {"scaled": True, "sweep": 2}),
    (2, 300, 2, {"literal": True, "scaled": True}),
    (2, 300, 3, {"lowrank": True, "scaled": True, "sweep": 3}),
    (3, 500, 2, {"lowrank": True, "literal": True, "scaled": True}),
    (3, 300, 3, {"replicate_below": 2000, "scaled": True, "sweep": 3}),                     # 5 levels: 114, 408, 1549 replicated, 6033 and 23809 by row blocks
    (3, 300, 2, {"replicate_below": 500, "lowrank": True, "scaled": True}),                  # the low-rank block is handed to the replicated levels at the fold
    (2, 300, 2, {"replicate_below": 10 ** 9}),                                                # only the finest level by row blocks
    (3, 300, 2, {"replicate_below": 2000, "scaled": True, "coloring": 3}),                   # PMG_COLORING_ITERATED on row-block and replicated levels (round 4)
], ids=["1rank", "2ranks", "3ranks_symmetric_nu2", "4ranks_backward_4levels", "2ranks_literal", "3ranks_lowrank", "2ranks_lowrank_literal", "3ranks_replicated_small_levels", "2ranks_replicated_lowrank", "2ranks_only_finest_distributed", "2ranks_iterated_colouring"])
def test_row_block_distributed_aij_vcycle_reproduces_the_single_device_chain(refine, coarse_max, world, opts):
    """PCGAMGMC on a MATMPIAIJ hierarchy (reference src/pc_gamgmc.c:157-223 over MCSORApply_MPIAIJ, src/mc_sor.c:298-381):
    the aggregation hierarchy of the P1 matrix of the reference's lshape.msh, every level above the coarsest split into
    contiguous row blocks over `world` ranks sharing the one GPU ("ipc" transport) -- per-colour sweeps with ghost updates,
    residual, ghost update of the residual, the owned rows of P^T and P, all-gathered right-hand side of the replicated
    exact coarse sampler, all in the C loop of pmg_mgmc.c -- against MGMC.from_hierarchy on one device, bit for bit."""
    import torch
    import torch.multiprocessing as mp

    from parmgmc_amd import MGMC

    o = dict(scaled=False, omega=1.0, sweep=1, nu=1, literal=False, lowrank=False, replicate_below=0)
    o.update(opts)
    its = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_aij_mg_worker, args=(r, world, port, refine, coarse_max, o, its, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = sorted((q.get(timeout=150) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ops, ps = _aij_hierarchy(refine, coarse_max)
    assert parts[0][3] == len(ops) >= 4
    n = len(ops[-1][0]) - 1
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(n), rng.standard_normal(n)
    one = MGMC.from_hierarchy(ops, ps)
    one.set_coloring(o.get("coloring", 0))
    one.set_smoother(o["scaled"], o["omega"], o["sweep"], o["nu"])
    one.set_correction_form(o["literal"])
    if o.get("lowrank"):
        one.set_lowrank(*_csr_lowrank_factors(n))
    one.setup()
    yd = torch.as_tensor(y_all, device="cuda")
    ctr = one.sample(torch.as_tensor(b_all, device="cuda"), yd, its, seed=42, counter0=1)
    assert all(x[2] == ctr for x in parts)
    got, ref = np.concatenate([x[1] for x in parts]), yd.cpu().numpy()
    assert np.isfinite(got).all()
    if o.get("lowrank"):  # the k-vectors B^T y are summed per rank, then over the ranks: equal to rounding, not bit for bit
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-12
        plain = MGMC.from_hierarchy(ops, ps)  # ... and the update is not a no-op
        plain.set_smoother(o["scaled"], o["omega"], o["sweep"], o["nu"])
        plain.setup()
        yp = torch.as_tensor(y_all, device="cuda")
        plain.sample(torch.as_tensor(b_all, device="cuda"), yp, its, seed=42, counter0=1)
        assert np.abs(yp.cpu().numpy() - ref).max() / np.abs(ref).max() > 1e-6
        return
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("world,sweep_type", [(2, 1), (3, 3), (4, 2)], ids=["2ranks", "3ranks_symmetric", "4ranks_backward"])
def test_row_block_sampler_with_a_low_rank_update(world, sweep_type):
    """MATLRC operator A + B S B^T on a row-block distributed MATAIJ base (MCSORSetUp's LRC branch, reference
    src/mc_sor.c:572-595; per-sweep repair :101-112, noise term src/pc_mcgibbs.c:130-140): the correction is built with
    distributed deterministic sweeps and rank-ordered all-reduces, the k-vectors B^T y are summed per rank and then over
    the ranks -- equal to the single-device sampler to rounding (1e-12), not bit for bit."""
    import torch
    import torch.multiprocessing as mp

    from parmgmc_amd import MCSOR

    omega, its = 1.1, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_csr_worker, args=(r, world, port, "lshape", omega, sweep_type, its, q, "ipc", True)) for r in range(world)]
    for p in procs:
        p.start()
    parts = sorted((q.get(timeout=150) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    A, colors = _csr_problem("lshape")
    rng = np.random.default_rng(5)
    b_all, y_all = rng.standard_normal(A.n), rng.standard_normal(A.n)
    one = MCSOR(A.rowptr, A.colidx, A.vals, user_colors=colors).setup()
    one.set_omega(omega)
    one.set_sweep_type(sweep_type)
    one.set_lowrank(*_csr_lowrank_factors(A.n))
    yd = torch.as_tensor(y_all, device="cuda")
    ctr = one.sample(torch.as_tensor(b_all, device="cuda"), yd, its, seed=42, counter0=1)
    assert all(x[2] == ctr for x in parts)
    got, ref = np.concatenate([x[1] for x in parts]), yd.cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got - ref).max() / np.abs(ref).max() < 1e-12
    plain = MCSOR(A.rowptr, A.colidx, A.vals, user_colors=colors).setup()  # the update is not a no-op
    plain.set_omega(omega)
    plain.set_sweep_type(sweep_type)
    yp = torch.as_tensor(y_all, device="cuda")
    plain.sample(torch.as_tensor(b_all, device="cuda"), yp, its, seed=42, counter0=1)
    assert np.abs(yp.cpu().numpy() - ref).max() / np.abs(ref).max() > 1e-6


def _chain_of_objects_worker(rank, world, port, shapes, q):
    """several sampler objects one after the other in ONE process, as bench.py builds them: every one gets the pooled
    receive block of its predecessor (zeroed) and the peers' kept mappings"""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parmgmc_amd.dist import DistGridSampler

    out = []
    for it, (nx, ny, nz, sweeps) in enumerate(shapes):
        smp = DistGridSampler(nx, ny, nz, 1.5, rank, world, omega=1.0, sweep_type=1, transport="ipc")
        assert smp.transport == "ipc"
        g = smp.grid
        rng = np.random.default_rng(5 + it)
        b_all, y_all = rng.standard_normal(nx * ny * nz), rng.standard_normal(nx * ny * nz)
        lo, hi = g.kz0 * nx * ny, (g.kz0 + g.nz) * nx * ny
        b = g.to_cvec(torch.as_tensor(b_all[lo:hi], device="cuda"))
        y = g.to_cvec(torch.as_tensor(y_all[lo:hi], device="cuda"))
        smp.sample_cvec(b, y, sweeps, seed=42, counter0=1)
        torch.cuda.synchronize()
        out.append(g.from_cvec(y).cpu().numpy())
        smp.destroy()  # collective: disconnect, barrier, back to the pool
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_a_chain_of_ipc_objects_in_one_process_reuses_its_receive_block():
    """The ipc receive block holds sequence numbers that only grow.  A second object that inherits the block of the first
    (pmg_dist.c pools the blocks; before, the allocator could hand the same memory back) must start from zeros: the first
    object here runs 40 sweeps, so its flag words stand at 80 when the second one, with the same block size, begins to wait
    for 1, 2, 3 ...  Three objects, two shapes (two pool entries), 3 ranks: every chain bit-identical to one device."""
    import torch
    import torch.multiprocessing as mp

    from parmgmc_amd import GridMCSOR

    shapes = [(40, 18, 12, 40), (40, 18, 12, 3), (24, 10, 9, 5), (40, 18, 12, 2)]
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chain_of_objects_worker, args=(r, world, port, shapes, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = sorted((q.get(timeout=200) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for it, (nx, ny, nz, sweeps) in enumerate(shapes):
        rng = np.random.default_rng(5 + it)
        b_all, y_all = rng.standard_normal(nx * ny * nz), rng.standard_normal(nx * ny * nz)
        one = GridMCSOR(nx, ny, nz, 1.5)
        yd = torch.as_tensor(y_all, device="cuda")
        one.sample(torch.as_tensor(b_all, device="cuda"), yd, sweeps, seed=42, counter0=1)
        got = np.concatenate([x[1][it] for x in parts])
        assert np.array_equal(got, yd.cpu().numpy()), f"object {it} of the chain"


def _bytes_worker(rank, world, port, grid, levels, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PMG_MG_REPLICATE_BELOW"] = "400"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parmgmc_amd.capi import lib
    from parmgmc_amd.dist import DistMGMC

    nx, ny, nz = grid
    out = {}
    for lowrank in (False, True):
        mg = DistMGMC(nx, ny, nz, 1.5, levels, rank, world, transport="ipc")
        mg.set_smoother(True, 1.0, 1, 1)
        lo, hi = mg.plane_range[0] * nx * ny, mg.plane_range[1] * nx * ny
        if lowrank:  # ONE ball well inside the first slab: the second rank holds none of B's support
            X, Y, Z = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny), np.linspace(0, 1, nz), indexing="ij")
            pts = np.stack([X.ravel(order="F"), Y.ravel(order="F"), Z.ravel(order="F")], 1)
            inside = ((pts - np.asarray((0.5, 0.5, 0.15))) ** 2).sum(1) < 0.1 ** 2
            B = np.zeros((nx * ny * nz, 1))
            B[inside, 0] = 1.0 / inside.sum()
            assert not inside[nx * ny * (nz // 2 - 2):].any()
            mg.set_lowrank(B[lo:hi], np.array([50.0]))
        mg.setup()
        y = torch.zeros(hi - lo, dtype=torch.float64, device="cuda")
        mg.sample(torch.ones(hi - lo, dtype=torch.float64, device="cuda"), y, 2, seed=3, counter0=0)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y).all())
        out[lowrank] = (mg.algorithmic_bytes()[1][-1], lib.pmg_last_error_string().decode())  # the (distributed) finest level
        mg.destroy()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_a_slab_without_observations_is_charged_no_lowrank_bytes():
    """advisor (round 3): pmg_mgmc_get_algorithmic_bytes charged a rank whose slab misses B's support the DENSE form's
    (24 k + 48) N bytes per sweep (and left a stale error message): bench.py sums the ranks' bytes into the roofline of the
    multi-GPU low-rank line.  Two z-slabs, one ball inside the first: the second rank's finest level must count exactly
    the bytes of the plain cycle, the first rank's a little more."""
    import torch.multiprocessing as mp

    grid, levels, world = (17, 17, 33), 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bytes_worker, args=(r, world, port, grid, levels, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    plain0, with0 = parts[0][False][0], parts[0][True][0]
    plain1, with1 = parts[1][False][0], parts[1][True][0]
    assert with1 == plain1, (plain1, with1)          # no support on rank 1: nothing launched, nothing charged
    assert plain0 < with0 < 1.5 * plain0, (plain0, with0)  # rank 0: a few support rows, far from the dense form
    assert "dense" not in parts[1][True][1]          # and the query is not routed through the error path
