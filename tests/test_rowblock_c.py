"""The row-block set-up in C (parmgmc_amd/csrc/pmg_rowblock.c) against the Python builders of round 2
(parmgmc_amd/dist.py: rowblock_plan, rowblock_hierarchy), index for index, under `gloo` with 1, 2 and 3 ranks on the CPU --
no GPU: the plan builders are host code.  The C side sees ONLY what a PETSc caller holds (MatMPIAIJGetSeqAIJ's diagonal and
off-diagonal blocks + garray, MatGetOwnershipRanges, reference src/mc_sor.c:152-214,308) and one byte all-gather callback
(here torch.distributed; MPI_Allgather in adapter/).  Also: the merge of the two blocks, and the distributed first-fit
colouring == the library's colouring of the global matrix (tests/test_gpu_mcsor.py pins that one to the oracle's)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp

import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def problem():
    """P1 matrix of the reference's lshape.msh refined twice + a 4-level aggregation hierarchy (5 985 ... rows)"""
    from pathlib import Path

    from parmgmc_amd.unstructured import assemble_p1, build_hierarchy, read_gmsh41_triangles, refine_uniform

    xy, tris = read_gmsh41_triangles(Path(__file__).resolve().parent / "golden" / "lshape.msh")
    for _ in range(2):
        xy, tris = refine_uniform(xy, tris)
    A = assemble_p1(xy, tris, 1.0)
    return build_hierarchy(A, coarse_max=60)


def split_mpiaij(A, r0, r1, dtype):
    """what MatMPIAIJGetSeqAIJ returns for rows [r0, r1) of the square matrix A: Ad (local columns), Ao (compact columns), garray"""
    mine = A[r0:r1].tocsr()
    mine.sort_indices()
    ad = mine[:, r0:r1].tocsr()
    off = mine.copy().tolil()
    off[:, r0:r1] = 0
    off = off.tocsr()
    off.eliminate_zeros()
    garray = np.unique(off.indices)
    ao = sp.csr_matrix((off.data, np.searchsorted(garray, off.indices), off.indptr), shape=(r1 - r0, len(garray)))
    cast = lambda a: np.ascontiguousarray(a, dtype)
    return (cast(ad.indptr), cast(ad.indices), np.ascontiguousarray(ad.data)), (cast(ao.indptr), cast(ao.indices), np.ascontiguousarray(ao.data)), cast(garray)


def arr(ptr, n, dt):
    return np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[: int(n)].astype(dt).copy() if n else np.zeros(0, dt)


def _worker(rank, world, port, idx_width, q):
    import torch.distributed as dist

    from parmgmc_amd import capi
    from parmgmc_amd.capi import check, lib
    from parmgmc_amd.dist import rowblock_hierarchy

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm, keep = capi.torch_host_comm(rank, world)
        ops, ps = problem()
        L = len(ops)
        n = [len(o[0]) - 1 for o in ops]
        mats = [sp.csr_matrix((o[2], o[1], o[0]), shape=(n[l], n[l])) for l, o in enumerate(ops)]
        Ps = [None] + [sp.csr_matrix((p[2], p[1], p[0]), shape=(n[l], n[l - 1])) for l, p in enumerate(ps) if p is not None]
        starts = [np.linspace(0, n[l], world + 1).astype(np.int64) for l in range(L)]
        it = np.int64 if idx_width == 64 else np.int32
        replicate_below = 400
        # ---- C: every level from its MPIAIJ blocks ----
        h = C.c_void_p()
        check(lib.pmg_rbh_create(C.byref(comm), L, replicate_below, C.byref(h)))
        merged = {}
        for l in range(L):
            r0, r1 = int(starts[l][rank]), int(starts[l][rank + 1])
            (arp, aci, av), (orp, oci, ov), garray = split_mpiaij(mats[l], r0, r1, it)
            nnz = len(av) + len(ov)
            rp, ci, v = np.zeros(r1 - r0 + 1, np.int64), np.zeros(nnz, np.int64), np.zeros(nnz)
            check(lib.pmg_rowblock_merge_mpiaij(r1 - r0, r0, arp.ctypes.data, aci.ctypes.data, av.ctypes.data, orp.ctypes.data, oci.ctypes.data, ov.ctypes.data, garray.ctypes.data, idx_width, 0, rp.ctypes.data, ci.ctypes.data, v.ctypes.data))
            mine = mats[l][r0:r1].tocsr()
            mine.sort_indices()
            assert np.array_equal(rp, mine.indptr) and np.array_equal(ci, mine.indices) and np.array_equal(v, mine.data)  # the sequential row
            if l == L - 1:  # the reference's own visiting order: diagonal block, then off-diagonal block (src/mc_sor.c:331-333)
                rp2, ci2, v2 = np.zeros_like(rp), np.zeros_like(ci), np.zeros_like(v)
                check(lib.pmg_rowblock_merge_mpiaij(r1 - r0, r0, arp.ctypes.data, aci.ctypes.data, av.ctypes.data, orp.ctypes.data, oci.ctypes.data, ov.ctypes.data, garray.ctypes.data, idx_width, 1, rp2.ctypes.data, ci2.ctypes.data, v2.ctypes.data))
                assert np.array_equal(rp2, rp)
                for r in range(0, r1 - r0, 97):
                    seg = ci2[rp2[r]:rp2[r + 1]]
                    own = (seg >= r0) & (seg < r1)
                    k = int(own.sum())
                    assert own[:k].all() and not own[k:].any() and sorted(seg) == list(ci[rp[r]:rp[r + 1]])
            merged[l] = (rp, ci, v)
            check(lib.pmg_rbh_set_level_operator(h, l, n[l], r0, r1 - r0, rp.ctypes.data, ci.ctypes.data, v.ctypes.data, 64))
            if l >= 1:
                pm = Ps[l][r0:r1].tocsr()
                prp, pci, pv = np.ascontiguousarray(pm.indptr, it), np.ascontiguousarray(pm.indices, it), np.ascontiguousarray(pm.data)
                check(lib.pmg_rbh_set_level_interpolation(h, l, r1 - r0, prp.ctypes.data, pci.ctypes.data, pv.ctypes.data, idx_width))
        check(lib.pmg_rbh_build(h))
        nl, fold = C.c_int32(), C.c_int32()
        check(lib.pmg_rbh_get_info(h, C.byref(nl), C.byref(fold)))
        # ---- Python reference: the global colouring by the library's first-fit rule = the oracle's ----
        colorings = [None] * L
        for l in range(fold.value, L):
            col = O.coloring_greedy(O.CSR.from_scipy(mats[l]))
            colorings[l] = (col, int(col.max()) + 1)
        H = rowblock_hierarchy(ops, ps, colorings, rank, world, starts=starts, replicate_below=replicate_below)
        assert (nl.value, fold.value) == (L, H["fold"]) and 1 <= fold.value < L - 1  # at least two row-block levels
        for l in range(L):
            v = capi.RbhLevelView()
            check(lib.pmg_rbh_get_level(h, l, C.byref(v)))
            assert v.n_global == n[l] and np.array_equal(arr(v.starts, world + 1, np.int64), starts[l])
            if l < fold.value:  # replicated: the whole matrices
                assert v.replicated == 1 and v.nlocal == n[l]
                ref = mats[l].copy()
                if l == 0:
                    ref.sort_indices()
                nnz = ref.nnz
                assert np.array_equal(arr(v.rp, n[l] + 1, np.int64), ref.indptr) and np.array_equal(arr(v.ci, nnz, np.int64), ref.indices) and np.array_equal(arr(v.v, nnz, np.float64), ref.data)
                if l >= 1:
                    assert np.array_equal(arr(v.P_rp, n[l] + 1, np.int64), Ps[l].indptr) and np.array_equal(arr(v.P_ci, Ps[l].nnz, np.int64), Ps[l].indices) and np.array_equal(arr(v.P_v, Ps[l].nnz, np.float64), Ps[l].data)
                continue
            R = H["levels"][l]
            plan = R["plan"]
            ncol = R["ncolors"]
            assert v.replicated == 0 and (v.nowned, v.ncolors, v.row0) == (R["nowned"], ncol, R["row0"])
            assert np.array_equal(arr(v.colors, v.nowned, np.int32), R["colors"])  # distributed first-fit == global first-fit
            assert v.nghost == len(R["ghosts"]) and np.array_equal(arr(v.ghosts, v.nghost, np.int64), R["ghosts"])
            assert v.nlocal == R["nowned"] + len(R["ghosts"])
            nnz = len(R["v"])
            assert np.array_equal(arr(v.rp, v.nlocal + 1, np.int32), R["rp"]) and np.array_equal(arr(v.ci, nnz, np.int32), R["ci"]) and np.array_equal(arr(v.v, nnz, np.float64), R["v"])
            assert np.array_equal(arr(v.send_ptr, ncol + 1, np.int64), plan["send_ptr"]) and np.array_equal(arr(v.send_rows, plan["send_ptr"][-1], np.int32), plan["send_rows"])
            assert np.array_equal(arr(v.counts, ncol * world, np.int64).reshape(ncol, world), plan["counts"])
            assert np.array_equal(arr(v.recv_ptr, ncol + 1, np.int64), plan["recv_ptr"]) and np.array_equal(arr(v.recv_src, v.nghost, np.int32), plan["recv_src"]) and np.array_equal(arr(v.recv_rows, v.nghost, np.int32), plan["recv_rows"])
            (prp, pci, pv), (rrp, rci, rv) = R["P"], R["R"]
            assert v.P_nrows == len(prp) - 1 and np.array_equal(arr(v.P_rp, len(prp), np.int32), prp) and np.array_equal(arr(v.P_ci, len(pci), np.int32), pci) and np.array_equal(arr(v.P_v, len(pv), np.float64), pv)
            assert v.R_nrows == len(rrp) - 1 and np.array_equal(arr(v.R_rp, len(rrp), np.int32), rrp) and np.array_equal(arr(v.R_ci, len(rci), np.int32), rci) and np.array_equal(arr(v.R_v, len(rv), np.float64), rv)
            assert v.ncoarse_local == R["ncoarse_local"]
        # ---- the stand-alone plan (no transfer ghosts) and an error that must surface on EVERY rank ----
        top = L - 1
        rp, ci, _ = merged[top]
        cols_ = np.ascontiguousarray(colorings[top][0][starts[top][rank]:starts[top][rank + 1]], np.int32)
        pl = C.c_void_p()
        check(lib.pmg_rowblock_plan_create(C.byref(comm), starts[top].ctypes.data, len(ci), ci.ctypes.data, 0, None, colorings[top][1], cols_.ctypes.data, C.byref(pl)))
        ng = C.c_int32()
        ptrs = [C.c_void_p() for _ in range(7)]
        check(lib.pmg_rowblock_plan_get(pl, C.byref(ng), *[C.byref(p_) for p_ in ptrs]))
        gh = arr(C.cast(ptrs[0], C.POINTER(C.c_int64)), ng.value, np.int64)
        off = ci[(ci < starts[top][rank]) | (ci >= starts[top][rank + 1])]
        assert np.array_equal(gh, np.unique(off))
        lib.pmg_rowblock_plan_destroy(C.byref(pl))
        bad = cols_.copy()
        if rank == world - 1 and world > 1:
            bad[:] = 99  # a colour outside [0, ncolors) on ONE rank
        st = lib.pmg_rowblock_plan_create(C.byref(comm), starts[top].ctypes.data, len(ci), ci.ctypes.data, 0, None, colorings[top][1], bad.ctypes.data, C.byref(pl))
        assert (st != 0) == (world > 1), "a failure on one rank must fail on all"
        lib.pmg_rbh_destroy(C.byref(h))
        q.put((rank, "ok"))
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,idx_width", [(1, 32), (2, 32), (3, 64)])
def test_c_plans_equal_the_python_plans(world, idx_width):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, idx_width, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r, msg in sorted(res):
        assert msg == "ok", f"rank {r}:\n{msg}"


def _clique_worker(rank, world, port, q):
    """rank 0 owns a K4 clique (rows 0..3, colours 0..3) and a fifth row; rank 1 owns ONE row coupled to rows 0, 1, 2 and to
    nothing else: first fit must give it colour 3.  Round 3 sized its mark array by the number of LOCAL rows (1 + 2 slots),
    dropped the neighbour colour 2 and handed out colour 2 -- the colour of its neighbour row 2 (advisor finding)."""
    import torch.distributed as dist

    from parmgmc_amd import capi
    from parmgmc_amd.capi import check, lib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm, keep = capi.torch_host_comm(rank, world)
        n = 6
        D = np.zeros((n, n))
        for i in range(4):
            for j in range(4):
                D[i, j] = -1.0
        D[4, 3] = D[3, 4] = -1.0  # rank 0's fifth row: a neighbour of row 3 only
        for j in (0, 1, 2):
            D[5, j] = D[j, 5] = -1.0  # rank 1's only row
        D[np.arange(n), np.arange(n)] = 10.0
        A = sp.csr_matrix(D)
        A.sort_indices()
        starts = np.array([0, 5, 6], np.int64)
        r0, r1 = int(starts[rank]), int(starts[rank + 1])
        mine = A[r0:r1].tocsr()
        rp, ci, v = np.ascontiguousarray(mine.indptr, np.int64), np.ascontiguousarray(mine.indices, np.int64), np.ascontiguousarray(mine.data)
        cols = np.full(r1 - r0, -7, np.int32)
        nc = C.c_int32()
        check(lib.pmg_rowblock_color_greedy(C.byref(comm), starts.ctypes.data, rp.ctypes.data, ci.ctypes.data, cols.ctypes.data, C.byref(nc)))
        want = O.coloring_greedy(O.CSR.from_scipy(A))
        assert np.array_equal(cols, want[r0:r1]), (rank, cols, want)
        assert nc.value == int(want.max()) + 1 == 4
        # the plan + the validity check behind every constructor: the right colouring passes, a colouring with the round-3
        # fault (row 5 in the colour of its neighbour row 2) is refused ON EVERY RANK, although only rank 1 can see it
        transport = None
        for bad in (False, True):
            c2 = cols.copy()
            if bad and rank == 1:
                c2[0] = 2
            pl = C.c_void_p()
            check(lib.pmg_rowblock_plan_create(C.byref(comm), starts.ctypes.data, len(ci), ci.ctypes.data, 0, None, 4, c2.ctypes.data, C.byref(pl)))
            st = lib.pmg_rowblock_check_coloring(C.byref(comm), pl, rp.ctypes.data, ci.ctypes.data, c2.ctypes.data)
            assert (st != 0) == bad, (rank, bad, st)
            lib.pmg_rowblock_plan_destroy(C.byref(pl))
        q.put((rank, "ok"))
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    dist.barrier()
    dist.destroy_process_group()


def test_tiny_block_beside_a_clique():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_clique_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for r, msg in sorted(res):
        assert msg == "ok", f"rank {r}:\n{msg}"


def _iterated_worker(rank, world, port, q):
    """the distributed second colouring round (pmg_rowblock_color_iterated) == the one-process rule (oracle twin of
    PMG_COLORING_ITERATED) on the P1 matrix of lshape.msh refined twice and on its first Galerkin level, uneven row blocks"""
    import torch.distributed as dist

    from parmgmc_amd import capi
    from parmgmc_amd.capi import check, lib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm, keep = capi.torch_host_comm(rank, world)
        ops, _ = problem()
        fewer = 0
        for op in ops[-2:]:
            n = len(op[0]) - 1
            A = sp.csr_matrix((op[2], op[1], op[0]), shape=(n, n))
            A.sort_indices()
            cuts = [0] + [int(n * f) for f in np.cumsum([0.5, 0.1, 0.25, 0.15][:world - 1])] + [n] if world > 1 else [0, n]
            starts = np.array(cuts, np.int64)
            r0, r1 = int(starts[rank]), int(starts[rank + 1])
            mine = A[r0:r1].tocsr()
            rp, ci = np.ascontiguousarray(mine.indptr, np.int64), np.ascontiguousarray(mine.indices, np.int64)
            csr = O.CSR.from_scipy(A)
            for fn, want in ((lib.pmg_rowblock_color_greedy, O.coloring_greedy(csr)), (lib.pmg_rowblock_color_iterated, O.coloring_iterated(csr))):
                cols = np.full(r1 - r0, -7, np.int32)
                nc = C.c_int32()
                check(fn(C.byref(comm), starts.ctypes.data, rp.ctypes.data, ci.ctypes.data, cols.ctypes.data, C.byref(nc)))
                assert np.array_equal(cols, want[r0:r1]) and nc.value == int(want.max()) + 1, (rank, n, nc.value)
            fewer += int(O.coloring_iterated(csr).max() < O.coloring_greedy(csr).max())
        assert fewer >= 1  # the second round saves a class on at least one of the two matrices
        q.put((rank, "ok"))
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4])
def test_distributed_iterated_colouring_equals_the_one_process_rule(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_iterated_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r, msg in sorted(res):
        assert msg == "ok", f"rank {r}:\n{msg}"
