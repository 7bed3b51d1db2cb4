"""Band structure along a k-path from saved Hamiltonian rows: the non-SOC, spin-unpolarised branch of the reference's post-processing script
`DFT_interfaces/openmx/band_cal.py` (:64-108 input handling, :296-392 the per-crystal loop), without its plotting / cif output.

    bands = band_structure(graphs, H_rows, nao_max=19, ham_type="openmx", k_path=[[0,0,0],[0.5,0,0],...], nk=120)

`H_rows`: what `Model.test()` saves as `prediction_hamiltonian.npy` (hamgnn/models/Model.py:315-340) -- the rows of all crystals in the per-crystal
[on-site; off-site] order -- or None to use the targets `Hon / Hoff` stored in the graphs (the script's `hamiltonian_path: null`).
Per crystal the script builds H(k), S(k) by phase-factor sums over the edges, masks them to the atoms' orbitals and solves the generalised
eigenproblem through the Cholesky factor of S(k); here that is `kspace.band_energies` (the `hg_hk_assemble` kernel + hipSOLVER through
`torch.linalg`), the k-path is `kspace.k_path_points`, energies come out in eV relative to the valence-band maximum exactly as the script prints
them.  `auto_mode` (pymatgen's KPathSeek) is not available: pass the nodes."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import kspace
from .models.hamgnn_output import HamGNNPlusPlusOut

AU2EV = 27.211386245988        # Hartree -> eV (DFT_interfaces/openmx/utils.py: au2ev)


def band_structure(graphs: Sequence, hamiltonian_rows=None, nao_max: int = 19, ham_type: str = "openmx", k_path=None, nk: int = 120,
                   device: str = "cuda") -> List[dict]:
    """one dict per crystal: {"k_vec" [nk, 3] reduced, "k_dist" [nk], "k_node" [nodes], "bands_eV" [nbands, nk] (0 = valence-band maximum),
    "vbm_eV", "band_gap_eV"}"""
    if not isinstance(k_path, (list, tuple)) or len(k_path) < 2:
        raise ValueError("band_structure: pass the k-path nodes in reduced coordinates (auto_mode needs pymatgen's KPathSeek)")
    head = HamGNNPlusPlusOut("1x0e", "1x0e", nao_max=nao_max, ham_type=ham_type, ham_only=True, symmetrize=True, add_H0=False, soc_switch=False,
                             calculate_band_energy=True, num_k=nk, k_path=list(k_path), calculate_sparsity=False)
    head.compile(torch.device(device))                        # only its basis tables are used (orbital ranks, valence electrons)
    rows = None if hamiltonian_rows is None else torch.as_tensor(np.asarray(hamiltonian_rows), dtype=torch.float32)
    out, r0 = [], 0
    for g in graphs:
        N, E = int(g.z.shape[0]), int(g.edge_index.shape[1])
        if rows is None:
            on, off = g.Hon.float(), g.Hoff.float()
        else:
            on, off = rows[r0:r0 + N], rows[r0 + N:r0 + N + E]
            r0 += N + E
        gd = g.to(device)
        lat = gd.cell.detach().cpu().double().numpy().reshape(3, 3)
        k_red, lat_per_inv = kspace.k_path_points(k_path, nk, lat)
        nodes = np.asarray(k_path, dtype=np.float64)
        metric = np.linalg.inv(lat @ lat.T)
        seg = np.sqrt(np.einsum("ni,ij,nj->n", np.diff(nodes, axis=0), metric, np.diff(nodes, axis=0)))
        k_node = np.concatenate([[0.0], np.cumsum(seg)])
        pins = np.rint(k_node / k_node[-1] * (nk - 1))
        pins[0], pins[-1] = 0, nk - 1
        k_dist = np.interp(np.arange(nk), pins, k_node)
        k_cart = torch.from_numpy(k_red @ lat_per_inv).float().reshape(1, nk, 3)
        be = kspace.band_energies(head, on.to(device).contiguous(), off.to(device).contiguous(), gd, k_vecs=k_cart)[0]
        eig = be.double().cpu().numpy() * AU2EV                                      # [bands, nk]
        nel = float(head._num_valence[g.z.cpu()].sum())
        half = math.ceil(nel / 2)
        vbm, cbm = float(eig[half - 1].max()), float(eig[half].min())
        out.append({"k_vec": k_red, "k_dist": k_dist, "k_node": k_node, "bands_eV": eig - vbm, "vbm_eV": vbm, "band_gap_eV": cbm - vbm})
    if rows is not None and r0 != rows.shape[0]:
        raise ValueError(f"band_structure: {rows.shape[0]} Hamiltonian rows for crystals that need {r0}")
    return out
