"""z-slab domain decomposition of a DMDA grid over several `GridMCSOR` objects.

One slab per device in production (see ``parmgmc_amd.dist``); :class:`SlabSet` keeps all slabs in ONE process
on one device and moves the ghost planes with device-to-device copies.  It exists to check the decomposition
logic -- the per-colour ghost exchange that replaces the reference's per-colour VecScatter
(reference src/mc_sor.c:317-340) -- without needing several GPUs."""
from __future__ import annotations

import numpy as np

from .capi import SOR_BACKWARD_SWEEP, SOR_FORWARD_SWEEP, SOR_SYMMETRIC_SWEEP
from .wrappers import GridMCSOR


def slab_cuts(nz: int, parts: int) -> list[int]:
    """PETSc-style ownership split of nz planes over `parts` ranks: the first nz % parts ranks get one more."""
    base, rem = divmod(nz, parts)
    cuts = [0]
    for r in range(parts):
        cuts.append(cuts[-1] + base + (1 if r < rem else 0))
    return cuts


class SlabSet:
    def __init__(self, nx, ny, nz, kappa, cuts):
        self.nx, self.ny, self.nz, self.cuts = nx, ny, nz, list(cuts)
        self.slabs = [GridMCSOR(nx, ny, nz, kappa, kz0=lo, nz_owned=hi - lo) for lo, hi in zip(cuts[:-1], cuts[1:])]
        self.type = SOR_FORWARD_SWEEP

    def set_omega(self, omega):
        for s in self.slabs:
            s.set_omega(omega)

    def set_sweep_type(self, t):
        self.type = t

    def exchange(self, ys, color):
        """ghost planes of `color`: slab d's high ghost <- slab d+1's low owned plane and vice versa."""
        for d in range(len(self.slabs) - 1):
            lo, hi = self.slabs[d], self.slabs[d + 1]
            own_hi, ghost_hi, n = lo.halo_plane(color, 1)
            own_lo, ghost_lo, _ = hi.halo_plane(color, 0)
            ys[d][ghost_hi:ghost_hi + n].copy_(ys[d + 1][own_lo:own_lo + n])
            ys[d + 1][ghost_lo:ghost_lo + n].copy_(ys[d][own_hi:own_hi + n])

    def one_sweep(self, bs, ys, direction, noisy, scaled, seed, counter):
        order = (0, 1) if direction == SOR_FORWARD_SWEEP else (1, 0)
        for c in order:
            self.exchange(ys, 1 - c)  # colour c reads colour 1-c across the slab faces
            for s, b, y in zip(self.slabs, bs, ys):
                s.sweep_color_cvec(c, b, y, noisy, scaled, seed, counter)

    def sample_natural(self, b, y0, its, seed, counter0=0, scaled=True):
        import torch

        plane = self.nx * self.ny
        bs, ys = [], []
        for s, lo, hi in zip(self.slabs, self.cuts[:-1], self.cuts[1:]):
            bs.append(s.to_cvec(torch.as_tensor(np.ascontiguousarray(b[lo * plane:hi * plane]), device="cuda")))
            ys.append(s.to_cvec(torch.as_tensor(np.ascontiguousarray(y0[lo * plane:hi * plane]), device="cuda")))
        ctr = counter0
        for _ in range(its):
            if self.type == SOR_SYMMETRIC_SWEEP:
                self.one_sweep(bs, ys, SOR_FORWARD_SWEEP, True, scaled, seed, ctr)
                self.one_sweep(bs, ys, SOR_BACKWARD_SWEEP, True, scaled, seed, ctr + 1)
                ctr += 2
            else:
                self.one_sweep(bs, ys, self.type, True, scaled, seed, ctr)
                ctr += 1
        return np.concatenate([s.from_cvec(y).cpu().numpy() for s, y in zip(self.slabs, ys)])
