"""Generates the golden fixtures under tests/golden/ from the CPU oracle (oracle/pmg_oracle.c).

Run from the repo root:  python tests/golden/make_golden.py
The reference ships no numeric golden files and cannot be built here (no PETSc), so
these vectors are outputs of the restatement, pinned by the reference's known-answer tests in
tests/test_oracle_reference_kat.py.  Inputs are seeded numpy draws stored next to the expected outputs, so
the fixtures stay valid if numpy's generator changes.  Nothing here reads /root/reference.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent


def main():
    rng = np.random.default_rng(20240607)
    # 1. operator of src/problems.c on 3x3 and 9x9 (kappa 10 and 1) + 3-D 5x5x5
    ops = {}
    for name, (nx, ny, nz, kappa) in {"lap_3x3_k10": (3, 3, 1, 10.0), "lap_9x9_k10": (9, 9, 1, 10.0), "lap_9x9_k1": (9, 9, 1, 1.0), "lap_5x5x5_k10": (5, 5, 5, 10.0), "lap_6x5x4_k2": (6, 5, 4, 2.0)}.items():
        A = O.shifted_laplace(nx, ny, nz, kappa)
        ops[name + "_rowptr"], ops[name + "_colidx"], ops[name + "_vals"] = A.rowptr, A.colidx, A.vals
        ops[name + "_diagptr"] = O.diag_pointers(A)
        ops[name + "_redblack"] = O.coloring_redblack(nx, ny, nz)
        ops[name + "_greedy"] = O.coloring_greedy(A)
        ops[name + "_lexlevels"] = O.coloring_lexlevels(A)
    A = O.ex6_matrix(10, 1e-4)
    ops["ex6_10_rowptr"], ops["ex6_10_colidx"], ops["ex6_10_vals"] = A.rowptr, A.colidx, A.vals
    np.savez_compressed(OUT / "operators.npz", **ops)

    # 2. deterministic sweeps (MCSORApply) for fixed b, y: fwd / bwd / sym at omega 1 and 1.2, three colourings
    sw = {}
    for name, (nx, ny, nz, kappa) in {"9x9": (9, 9, 1, 10.0), "5x5x5": (5, 5, 5, 10.0), "6x5x4": (6, 5, 4, 2.0)}.items():
        A = O.shifted_laplace(nx, ny, nz, kappa)
        n = A.n
        b, y = rng.standard_normal(n), rng.standard_normal(n)
        sw[f"{name}_b"], sw[f"{name}_y"] = b, y
        for cname, col in {"redblack": O.coloring_redblack(nx, ny, nz), "single": O.coloring_single(n), "greedy": O.coloring_greedy(A)}.items():
            for om in (1.0, 1.2):
                for tname, t in {"fwd": O.SOR_FORWARD, "bwd": O.SOR_BACKWARD, "sym": O.SOR_SYMMETRIC}.items():
                    sw[f"{name}_{cname}_om{om}_{tname}"] = O.mcsor_apply(A, col, b, y, om, t)
    np.savez_compressed(OUT / "sweeps.npz", **sw)

    # 3. noise: Philox known answers (Random123 kat), normal pairs, grid and row streams, Box-Muller pairing
    nz_ = {}
    nz_["philox_ctr"] = np.array([[0, 0, 0, 0], [0xFFFFFFFF] * 4, [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], np.uint32)
    nz_["philox_key"] = np.array([[0, 0], [0xFFFFFFFF] * 2, [0xA4093822, 0x299F31D0]], np.uint32)
    nz_["philox_out"] = np.stack([O.philox4x32_10(c, k) for c, k in zip(nz_["philox_ctr"], nz_["philox_key"])])
    nz_["rows_seed51966_sweep7_n33"] = O.noise_rows(33, 0xCAFE, 7)
    nz_["grid_9x9x1_seed51966_sweep3"] = O.noise_grid(9, 9, 1, 0xCAFE, 3)
    nz_["grid_6x5x4_seed51966_sweep3"] = O.noise_grid(6, 5, 4, 0xCAFE, 3)
    u = rng.random(8)
    nz_["bm_uniforms"] = u
    nz_["bm_n7"] = O.box_muller_vec(7, u)
    np.savez_compressed(OUT / "noise.npz", **nz_)

    # 4. noisy sample chains (PCApplyRichardson_MulticolorGibbs / _SORGibbs) with the library's noise streams
    ch = {}
    for name, (nx, ny, nz, kappa) in {"9x9": (9, 9, 1, 10.0), "6x5x4": (6, 5, 4, 2.0)}.items():
        A = O.shifted_laplace(nx, ny, nz, kappa)
        n = A.n
        b, y0 = rng.standard_normal(n), rng.standard_normal(n)
        ch[f"{name}_b"], ch[f"{name}_y0"] = b, y0
        rb = O.coloring_redblack(nx, ny, nz)
        for om, scaled, tname, t in [(1.0, True, "fwd", O.SOR_FORWARD), (1.3, True, "sym", O.SOR_SYMMETRIC), (1.0, False, "bwd", O.SOR_BACKWARD)]:
            ch[f"{name}_grid_om{om}_{'mc' if scaled else 'sor'}_{tname}"] = O.gibbs_samples(A, rb, b, y0, 3, lambda d: O.noise_grid(nx, ny, nz, 0xCAFE, 5 + d), om, t, scaled)
            ch[f"{name}_rows_om{om}_{'mc' if scaled else 'sor'}_{tname}"] = O.gibbs_samples(A, O.coloring_greedy(A), b, y0, 3, lambda d: O.noise_rows(n, 0xCAFE, 5 + d), om, t, scaled)
    np.savez_compressed(OUT / "chains.npz", **ch)
    print("wrote", [p.name for p in OUT.glob("*.npz")])


if __name__ == "__main__":
    main()
