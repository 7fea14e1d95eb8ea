"""CPU restatement of PCPARSOR's multi-process sweep (reference src/pc_parsor.c) -- TEST INFRASTRUCTURE ONLY.

A literal, sequential emulation of what ``nparts`` MPI ranks do: every rank keeps its own ghost vector ``lvec``
(one slot per off-process column it references), the scatters copy values between the ranks' arrays at the points
of the schedule where the reference starts / ends them, and the MID phase is an event loop with explicit messages.
The device path derives a data-flow graph from the same rules and level-schedules it; this file does NOT, so the
two agree only if that derivation is right.

Parity status: the reference cannot be built here (PETSc), and it ships no fixture for this path, so the
emulation is pinned only through properties the algorithm must have (one rank == lexicographic SOR, the solution of
A x = b is a fixed point, convergence for SPD A) -- "parity unpinned" against the reference binary.  The processor
colouring of the reference is PETSc's randomised JP colouring (src/pc_parsor.c:240-243), unpinned by any reference
test: here a first-fit colouring in rank order, or the caller's.
"""
from __future__ import annotations

import numpy as np

INT, TOP, MID, BOT = 0, 1, 2, 3


def owner_of(row_starts, c):
    """PetscLayoutFindOwner for contiguous row blocks"""
    return int(np.searchsorted(np.asarray(row_starts), c, side="right") - 1)


def color_processors(A, row_starts):
    """first-fit colouring, in rank order, of the graph "rank p references a column owned by rank q"
    (the graph of ColorProcessors, src/pc_parsor.c:187-270; the reference colours it with MATCOLORINGJP)"""
    nparts = len(row_starts) - 1
    adj = [set() for _ in range(nparts)]
    for p in range(nparts):
        for r in range(row_starts[p], row_starts[p + 1]):
            for k in range(A.rowptr[r], A.rowptr[r + 1]):
                q = owner_of(row_starts, A.colidx[k])
                if q != p:
                    adj[p].add(q)
                    adj[q].add(p)
    cols = [-1] * nparts
    for p in range(nparts):
        used = {cols[q] for q in adj[p] if cols[q] >= 0}
        c = 0
        while c in used:
            c += 1
        cols[p] = c
    return np.asarray(cols, np.int32)


class _Rank:
    """what one MPI rank holds after ParallelSORSetUp (src/pc_parsor.c:272-620), restated"""

    def __init__(self, A, row_starts, proccols, p):
        self.p, self.r0, self.r1 = p, int(row_starts[p]), int(row_starts[p + 1])
        n = self.r1 - self.r0
        # MatMPIAIJGetSeqAIJ: diagonal block (local columns) and off-diagonal block (compressed ghost columns, ascending
        # global index = colmap)
        self.ad, self.ao = [[] for _ in range(n)], [[] for _ in range(n)]
        ghosts = sorted({int(A.colidx[k]) for r in range(self.r0, self.r1) for k in range(A.rowptr[r], A.rowptr[r + 1]) if not (self.r0 <= A.colidx[k] < self.r1)})
        self.colmap = ghosts
        slot = {g: s for s, g in enumerate(ghosts)}
        self.diag = np.zeros(n)
        for r in range(self.r0, self.r1):
            ent = sorted((int(A.colidx[k]), float(A.vals[k])) for k in range(A.rowptr[r], A.rowptr[r + 1]))
            for c, v in ent:
                if self.r0 <= c < self.r1:
                    self.ad[r - self.r0].append((c - self.r0, v))
                    if c == r:
                        self.diag[r - self.r0] = v
                else:
                    self.ao[r - self.r0].append((slot[c], v))
        self.ghost_owner = [owner_of(row_starts, g) for g in ghosts]
        # ParallelSORPartitionNodes :311-331
        mycol = proccols[p]
        self.cls = np.zeros(n, np.int32)
        for i in range(n):
            istop = any(proccols[self.ghost_owner[s]] < mycol for s, _ in self.ao[i])
            isbot = any(proccols[self.ghost_owner[s]] > mycol for s, _ in self.ao[i])
            self.cls[i] = INT if not (istop or isbot) else BOT if not istop else TOP if not isbot else MID
        self.top = [i for i in range(n) if self.cls[i] == TOP]
        self.bot = [i for i in range(n) if self.cls[i] == BOT]
        self.mid = [i for i in range(n) if self.cls[i] == MID]
        ints = [i for i in range(n) if self.cls[i] == INT]
        # the INT1 / INT2 split :353-362 -- costs are row lengths of the OFF-DIAGONAL block, which is 0 for every INT
        # row, so the loop either never breaks (all INT rows in INT1) or breaks at once (all in INT2)
        cost = lambda rows: sum(len(self.ao[i]) for i in rows)  # noqa: E731
        tgt = np.float32(0.5) * np.float32(cost(ints) + cost(self.bot) - cost(self.top))
        tgt = int(np.sign(tgt) * np.floor(np.abs(tgt) + np.float32(0.5)))  # roundf: halves away from zero
        split, cur = 0, 0
        while split < len(ints):
            cur += len(self.ao[ints[split]])
            if cur > tgt:
                break
            split += 1
        self.int1, self.int2 = ints[:split], ints[split:]
        self.lvec = np.zeros(len(ghosts))


def parsor_apply(A, row_starts, b, x, omega=1.0, its=1, zero_initial_guess=False, proccols=None):
    """ParallelSORApply (src/pc_parsor.c:703-878) for ``len(row_starts)-1`` ranks owning contiguous row blocks; returns
    the new x (global)."""
    row_starts = [int(v) for v in row_starts]
    nparts = len(row_starts) - 1
    proccols = color_processors(A, row_starts) if proccols is None else np.asarray(proccols, np.int32)
    R = [_Rank(A, row_starts, proccols, p) for p in range(nparts)]
    b = np.asarray(b, np.float64)
    x = np.array(x, np.float64, copy=True)
    xs = [x[r.r0:r.r1] for r in R]  # views: the ranks' local arrays
    bs = [b[r.r0:r.r1] for r in R]

    def is_mid_global(g):
        q = owner_of(row_starts, g)
        return R[q].cls[g - R[q].r0] == MID

    def sweep_rows(r, rows, use_ghosts):
        """SORLocalForwardSweepIS :666-701: entries left of the diagonal, right of it, then the ghost block"""
        xl, bl = xs[r.p], bs[r.p]
        for i in rows:
            s = bl[i]
            for c, v in r.ad[i]:
                if c < i:
                    s = s - v * xl[c]
            for c, v in r.ad[i]:
                if c > i:
                    s = s - v * xl[c]
            if use_ghosts:
                for sl, v in r.ao[i]:
                    s = s - v * r.lvec[sl]
            xl[i] = (1.0 - omega) * xl[i] + s * (omega / r.diag[i])

    def scatter(r, rows, src):
        """VecScatter built from the off-diagonal entries of `rows` (:373-413): ghost slot <- owner's value in `src`"""
        for i in rows:
            for sl, _ in r.ao[i]:
                r.lvec[sl] = src[r.colmap[sl]]

    first = zero_initial_guess
    for _ in range(its):
        for r in R:
            r.lvec[:] = 0.0
        if first:
            x[:] = 0.0
        else:
            snap = x.copy()  # topsct: values at the start of the iteration
            for r in R:
                scatter(r, r.top, snap)
        first = False
        for r in R:
            sweep_rows(r, r.top, True)
        snap = x.copy()  # VecCopy(xx, parsor->xx) on every rank, then botsct: values after everybody's TOP phase
        for r in R:
            scatter(r, r.bot + r.mid, snap)
        for r in R:
            sweep_rows(r, r.int1, False)
        # ---- MID phase :745-865: event loop with messages {global row, value} to lower-coloured ranks
        deps, done, inbox = [], [], [[] for _ in R]
        for r in R:
            row_to_mid = {row: m for m, row in enumerate(r.mid)}
            d = []
            for m, row in enumerate(r.mid):
                cnt = 0
                for sl, _ in r.ao[row]:
                    if proccols[r.ghost_owner[sl]] > proccols[r.p] and is_mid_global(r.colmap[sl]):
                        cnt += 1  # one per stored entry (:451-456)
                for c, _ in r.ad[row]:
                    m2 = row_to_mid.get(c, -1)
                    if m2 >= 0 and m2 != m and r.mid[m2] < row:
                        cnt += 1  # adjacent MID rows of this rank with a smaller index go first (:547-557)
                d.append(cnt)
            deps.append(d)
            done.append([False] * len(r.mid))
        remaining = sum(len(r.mid) for r in R)
        while remaining > 0:
            progress = False
            for r in R:
                row_to_mid = {row: m for m, row in enumerate(r.mid)}
                moved = True
                while moved:
                    moved = False
                    for m, row in enumerate(r.mid):
                        if done[r.p][m] or deps[r.p][m] > 0:
                            continue
                        done[r.p][m] = True
                        remaining -= 1
                        moved = progress = True
                        sweep_rows(r, [row], True)
                        seen = set()
                        for c, _ in r.ad[row]:
                            m2 = row_to_mid.get(c, -1)
                            if m2 >= 0 and m2 != m and r.mid[m2] > row and m2 not in seen:
                                seen.add(m2)
                        # mid_local_deps holds each later neighbour once, but its counter was raised once per stored
                        # entry; a symmetric-pattern row stores the neighbour once, so the two counts agree
                        for m2 in seen:
                            deps[r.p][m2] -= sum(1 for c, _ in r.ad[r.mid[m2]] if c == row)
                        dests = []
                        for sl, _ in r.ao[row]:
                            q = r.ghost_owner[sl]
                            if proccols[q] < proccols[r.p] and is_mid_global(r.colmap[sl]) and q not in dests:
                                dests.append(q)
                        for q in dests:
                            inbox[q].append((r.r0 + row, xs[r.p][row]))
                # deliver what was sent (MPI_Waitany + the receive handler :826-846)
                for q, rq in enumerate(R):
                    if not inbox[q]:
                        continue
                    slot_of = {g: s for s, g in enumerate(rq.colmap)}
                    mid_slots = {sl for row in rq.mid for sl, _ in rq.ao[row] if proccols[rq.ghost_owner[sl]] > proccols[q] and is_mid_global(rq.colmap[sl])}
                    for gid, val in inbox[q]:
                        sl = slot_of[gid]
                        assert sl in mid_slots, "MID update for a column no MID row of the receiver references"
                        rq.lvec[sl] = val
                        for m, row in enumerate(rq.mid):
                            deps[q][m] -= sum(1 for s2, _ in rq.ao[row] if s2 == sl)
                    inbox[q] = []
            assert progress, "MID phase dead-locked"
        for r in R:
            sweep_rows(r, r.int2, False)
            sweep_rows(r, r.bot, True)
    return x


def node_classes(A, row_starts, proccols=None):
    proccols = color_processors(A, row_starts) if proccols is None else np.asarray(proccols, np.int32)
    out = np.zeros(A.n, np.int32)
    for p in range(len(row_starts) - 1):
        r = _Rank(A, row_starts, proccols, p)
        out[r.r0:r.r1] = r.cls
    return out
