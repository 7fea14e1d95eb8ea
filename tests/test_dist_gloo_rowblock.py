"""world_size 2 / 3 `gloo` test (CPU) of the ghost-update plan of a row-block distributed level (parmgmc_amd.dist.
rowblock_plan: what MatCreateScatters builds per colour in the reference, src/mc_sor.c:152-214): executed with numpy
exactly as pmg_distmcsor.c executes it -- per colour: gather the send rows, concatenate the ranks' blocks in rank order,
scatter recv_src -> recv_rows -- every ghost row must end up with its owner's value, for the operator's own off-process
columns and for extra ghosts (the rows a restriction or an interpolation reads)."""
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


def _problem():
    A = O.shifted_laplace(7, 6, 5, 1.0)
    col = O.coloring_greedy(A)
    rng = np.random.default_rng(3)
    extra = [np.sort(rng.choice(A.n, 25, replace=False)) for _ in range(4)]  # per rank: rows it reads beyond its operator
    return A, col, extra


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parmgmc_amd.dist import rowblock_plan

    A, col, extra = _problem()
    n = A.n
    cuts = [round(n * r / world) for r in range(world + 1)]
    r0, r1 = cuts[rank], cuts[rank + 1]
    ci = A.colidx[A.rowptr[r0]:A.rowptr[r1]]
    ncol = int(col.max()) + 1
    ghosts, plan = rowblock_plan(ci, r0, r1, n, col[r0:r1], ncol, extra[rank], rank, world)
    nloc = r1 - r0
    truth = 1.5 * np.arange(n) + 0.25
    v = np.concatenate([truth[r0:r1], np.full(len(ghosts), np.nan)])
    for c in range(ncol):  # pmg_distmcsor.c: distmcsor_update
        mine = v[plan["send_rows"][plan["send_ptr"][c]:plan["send_ptr"][c + 1]]]
        assert len(mine) == plan["counts"][c, rank]
        blocks = [None] * world
        dist.all_gather_object(blocks, mine)
        assert [len(b) for b in blocks] == list(plan["counts"][c])
        buf = np.concatenate(blocks) if sum(len(b) for b in blocks) else np.zeros(0)
        sl = slice(plan["recv_ptr"][c], plan["recv_ptr"][c + 1])
        v[plan["recv_rows"][sl]] = buf[plan["recv_src"][sl]]
    want_ghosts = np.unique(np.concatenate([ci[(ci < r0) | (ci >= r1)], extra[rank][(extra[rank] < r0) | (extra[rank] >= r1)]]))
    q.put((rank, bool(np.array_equal(ghosts, want_ghosts)), bool(np.array_equal(v[nloc:], truth[ghosts])), bool((plan["send_rows"] < nloc).all() and (plan["recv_rows"] >= nloc).all())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rowblock_plan_delivers_every_ghost_row(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] and r[3] for r in res), res


# ---- the row-block V-cycle as an algorithm (host logic of parmgmc_amd.dist.rowblock_hierarchy) ------------------------------
# numpy restatement of what pmg_mgmc.c does with the arrays rowblock_hierarchy hands it -- per-colour Gauss-Seidel sweeps with a
# ghost update after every colour, residual on the owned rows, ghost refresh of the residual, the owned rows of P^T, the
# all-gather into the replicated levels, the replicated levels as on one device, exact coarse solve, the owned rows of P --
# run with 1, 2 and 3 gloo ranks: the fine-level result must not depend on the number of ranks (bit for bit), with every
# level by row blocks and with the small levels replicated.

def _hier():
    from pathlib import Path

    from parmgmc_amd.unstructured import assemble_p1, build_hierarchy, read_gmsh41_triangles, refine_uniform

    xy, tris = read_gmsh41_triangles(Path(__file__).resolve().parent / "golden" / "lshape.msh")
    xy, tris = refine_uniform(xy, tris)
    return build_hierarchy(assemble_p1(xy, tris, 1.0), coarse_max=40)


def _gs_color(rp, ci, v, rows, b, x):
    for r in rows:
        s, d = b[r], 0.0
        for k in range(rp[r], rp[r + 1]):
            if ci[k] == r:
                d = v[k]
            else:
                s = s - v[k] * x[ci[k]]
        x[r] = s / d


def _vcycle_worker(rank, world, port, replicate_below, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parmgmc_amd.dist import rowblock_hierarchy

    ops, ps = _hier()
    L = len(ops)
    cols = [None] + [(lambda c: (c, int(c.max()) + 1))(O.coloring_greedy(O.CSR(o[0], o[1], o[2]))) for o in ops[1:]]
    H = rowblock_hierarchy(ops, ps, cols, rank, world, replicate_below=replicate_below)
    fold, st = H["fold"], H["starts"]
    Afull = [sp.csr_matrix((o[2], o[1], o[0]), shape=(len(o[0]) - 1,) * 2) for o in ops]
    Pfull = [None] + [sp.csr_matrix((p[2], p[1], p[0]), shape=(Afull[l].shape[0], Afull[l - 1].shape[0])) for l, p in enumerate(ps) if l >= 1]
    Rfull = [None] + [p.T.tocsr() for p in Pfull[1:]]
    for m in Rfull[1:]:
        m.sort_indices()

    def exchange(l, vec, colours):
        plan = H["levels"][l]["plan"]
        for c in colours:
            mine = vec[plan["send_rows"][plan["send_ptr"][c]:plan["send_ptr"][c + 1]]]
            blocks = [None] * world
            dist.all_gather_object(blocks, mine)
            buf = np.concatenate(blocks) if sum(len(b_) for b_ in blocks) else np.zeros(0)
            sl = slice(plan["recv_ptr"][c], plan["recv_ptr"][c + 1])
            vec[plan["recv_rows"][sl]] = buf[plan["recv_src"][sl]]

    def smooth(l, b, x):
        if l >= fold:
            Lv = H["levels"][l]
            exchange(l, x, range(Lv["ncolors"]))
            for c in range(Lv["ncolors"]):
                _gs_color(Lv["rp"], Lv["ci"], Lv["v"], np.nonzero(Lv["colors"] == c)[0], b, x)
                exchange(l, x, [c])
        else:
            col, nc = cols[l]
            a = Afull[l]
            for c in range(nc):
                _gs_color(a.indptr, a.indices, a.data, np.nonzero(col == c)[0], b, x)

    def rows_apply(rp, ci, v, src, nrows):
        out = np.zeros(nrows)
        for i in range(nrows):
            s = 0.0
            for k in range(rp[i], rp[i + 1]):
                s = s + v[k] * src[ci[k]]
            out[i] = s
        return out

    def cycle(l, b, x):  # b, x: level vectors in the local numbering (replicated levels: all rows)
        if l == 0:
            x[:] = np.linalg.solve(Afull[0].toarray(), b)
            return
        smooth(l, b, x)
        if l >= fold:
            Lv = H["levels"][l]
            no, nl = Lv["nowned"], Lv["nowned"] + len(Lv["ghosts"])
            r = np.zeros(nl)
            r[:no] = b[:no] - rows_apply(Lv["rp"], Lv["ci"], Lv["v"], x, no)
            exchange(l, r, range(Lv["ncolors"]))
            rrp, rci, rv = Lv["R"]
            mine = rows_apply(rrp, rci, rv, r, len(rrp) - 1)
            if l == fold:  # all-gather into the replicated level
                blocks = [None] * world
                dist.all_gather_object(blocks, mine)
                bc = np.concatenate(blocks)
            else:
                Cv = H["levels"][l - 1]
                bc = np.zeros(Cv["nowned"] + len(Cv["ghosts"]))
                bc[:Cv["nowned"]] = mine
            xc = np.zeros(len(bc))
            cycle(l - 1, bc, xc)
            prp, pci, pv = Lv["P"]
            x[:no] = x[:no] + rows_apply(prp, pci, pv, xc, no)
        else:
            a = Afull[l]
            r = b - rows_apply(a.indptr, a.indices, a.data, x, a.shape[0])
            bc = rows_apply(Rfull[l].indptr, Rfull[l].indices, Rfull[l].data, r, Rfull[l].shape[0])
            xc = np.zeros(len(bc))
            cycle(l - 1, bc, xc)
            x[:] = x + rows_apply(Pfull[l].indptr, Pfull[l].indices, Pfull[l].data, xc, a.shape[0])
        smooth(l, b, x)

    top = H["levels"][L - 1]
    n, r0, no = H["n"][L - 1], top["row0"], top["nowned"]
    rng = np.random.default_rng(9)
    b_all = rng.standard_normal(n)
    b = np.zeros(no + len(top["ghosts"]))
    b[:no] = b_all[r0:r0 + no]
    x = np.zeros(len(b))
    cycle(L - 1, b, x)
    cycle(L - 1, b, x)  # a second cycle starts from a non-zero iterate
    q.put((rank, x[:no].copy(), fold))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("replicate_below", [0, 150], ids=["all_levels_by_row_blocks", "small_levels_replicated"])
def test_rowblock_vcycle_does_not_depend_on_the_number_of_ranks(replicate_below):
    import torch.multiprocessing as mp

    results = {}
    for world in (1, 2, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_vcycle_worker, args=(r, world, port, replicate_below, q)) for r in range(world)]
        for p in procs:
            p.start()
        parts = sorted((q.get(timeout=150) for _ in range(world)), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        results[world] = (np.concatenate([p_[1] for p_ in parts]), parts[0][2])
    ops, _ = _hier()
    assert len(ops) >= 4 and results[1][1] == (1 if replicate_below == 0 else 2), ([len(o[0]) - 1 for o in ops], results[1][1])
    assert np.isfinite(results[1][0]).all() and np.abs(results[1][0]).max() > 0
    assert np.array_equal(results[1][0], results[2][0]) and np.array_equal(results[1][0], results[3][0])
    # ... and the cycle is a convergent one: two cycles reduce the residual of A x = b
    A = sp.csr_matrix((ops[-1][2], ops[-1][1], ops[-1][0]), shape=(len(ops[-1][0]) - 1,) * 2)
    b_all = np.random.default_rng(9).standard_normal(A.shape[0])
    assert np.linalg.norm(b_all - A @ results[1][0]) < 0.2 * np.linalg.norm(b_all)
