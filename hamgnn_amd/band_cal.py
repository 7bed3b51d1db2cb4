"""Band structure along a k-path from saved Hamiltonian rows: the three branches of the reference's post-processing script
`DFT_interfaces/openmx/band_cal.py` -- spin-unpolarised (:453-621), spin-orbit (`soc_switch`, :101-283) and collinear spin (`spin_colinear`,
:284-452) -- without its plotting / cif output.

    bands = band_structure(graphs, H_rows, nao_max=19, ham_type="openmx", k_path=[[0,0,0],[0.5,0,0],...], nk=120)

`H_rows`: what `Model.test()` saves as `prediction_hamiltonian.npy` (hamgnn/models/Model.py:315-340) -- the rows of all crystals in the per-crystal
[on-site; off-site] order -- or None to use the targets `Hon / Hoff` stored in the graphs (the script's `hamiltonian_path: null`).
Per crystal the script builds H(k), S(k) by phase-factor sums over the edges, masks them to the atoms' orbitals and solves the generalised
eigenproblem through the Cholesky factor of S(k); here that is `kspace.band_energies` (the `hg_hk_assemble` kernel + hipSOLVER through
`torch.linalg`), the k-path is `kspace.k_path_points`, energies come out in eV relative to the valence-band maximum exactly as the script prints
them.  `auto_mode=True` (script :135-145) takes the nodes of every crystal from pymatgen's KPathSeek (`kspace.auto_k_path`; raises where pymatgen is not installed).

    band_structure(..., soc_switch=True):    H_rows per crystal = [real rows (N + E); imaginary rows (N + E)] of width (2 nao)^2, as the SOC heads
        write them (hamgnn_output.py:3621-3626) or Hon / Hoff / iHon / iHoff of the graphs; four spin blocks of H(k), kron(1_2, S(k)),
        every band singly occupied: the valence-band maximum is band number sum(valence electrons) (script :222-225);
    band_structure(..., spin_colinear=True): rows of width 2 nao^2 = [spin][nao][nao] (script :311-312): the spin-free calculation once per spin
        channel; the result dict then holds "bands_eV" / "vbm_eV" / "band_gap_eV" as lists of two."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import kspace
from .models.hamgnn_output import HamGNNPlusPlusOut

AU2EV = 27.211386245988        # Hartree -> eV (DFT_interfaces/openmx/utils.py: au2ev)


def band_structure(graphs: Sequence, hamiltonian_rows=None, nao_max: int = 19, ham_type: str = "openmx", k_path=None, nk: int = 120,
                   device: str = "cuda", soc_switch: bool = False, spin_colinear: bool = False, auto_mode: bool = False) -> List[dict]:
    """one dict per crystal: {"k_vec" [nk, 3] reduced, "k_dist" [nk], "k_node" [nodes], "bands_eV" [nbands, nk] (0 = valence-band maximum),
    "vbm_eV", "band_gap_eV"} (+ "k_labels" with auto_mode)"""
    if not auto_mode and (not isinstance(k_path, (list, tuple)) or len(k_path) < 2):
        raise ValueError("band_structure: pass the k-path nodes in reduced coordinates, or auto_mode=True (pymatgen's KPathSeek)")
    head = HamGNNPlusPlusOut("1x0e", "1x0e", nao_max=nao_max, ham_type=ham_type, ham_only=True, symmetrize=True, add_H0=False, soc_switch=False,
                             calculate_band_energy=True, num_k=nk, k_path=None if auto_mode else list(k_path), calculate_sparsity=False)
    head.compile(torch.device(device))                        # only its basis tables are used (orbital ranks, valence electrons)
    if soc_switch and spin_colinear:
        raise ValueError("band_structure: soc_switch and spin_colinear exclude each other (the script's if / elif)")
    rows = None if hamiltonian_rows is None else torch.as_tensor(np.asarray(hamiltonian_rows), dtype=torch.float32)
    out, r0 = [], 0
    for g in graphs:
        N, E = int(g.z.shape[0]), int(g.edge_index.shape[1])
        if soc_switch:
            if rows is None:
                on, off, ion, ioff = g.Hon.float(), g.Hoff.float(), g.iHon.float(), g.iHoff.float()
            else:                                              # [real (N + E); imaginary (N + E)] rows of one crystal (script :103-117, :148)
                blk = rows[r0:r0 + 2 * (N + E)]
                on, off, ion, ioff = blk[:N], blk[N:N + E], blk[N + E:2 * N + E], blk[2 * N + E:]
                r0 += 2 * (N + E)
        elif rows is None:
            on, off = g.Hon.float(), g.Hoff.float()
        else:
            on, off = rows[r0:r0 + N], rows[r0 + N:r0 + N + E]
            r0 += N + E
        gd = g.to(device)
        lat = gd.cell.detach().cpu().double().numpy().reshape(3, 3)
        labels = None
        if auto_mode:                                          # script :135-145 (and :322-332, :491-501): the crystal's own high-symmetry path
            labels, k_path = kspace.auto_k_path(lat, g.pos.detach().cpu().double().numpy(), g.z.detach().cpu().numpy())
        k_red, lat_per_inv = kspace.k_path_points(k_path, nk, lat)
        nodes = np.asarray(k_path, dtype=np.float64)
        metric = np.linalg.inv(lat @ lat.T)
        seg = np.sqrt(np.einsum("ni,ij,nj->n", np.diff(nodes, axis=0), metric, np.diff(nodes, axis=0)))
        k_node = np.concatenate([[0.0], np.cumsum(seg)])
        pins = np.rint(k_node / k_node[-1] * (nk - 1))
        pins[0], pins[-1] = 0, nk - 1
        k_dist = np.interp(np.arange(nk), pins, k_node)
        k_cart = torch.from_numpy(k_red @ lat_per_inv).float().reshape(1, nk, 3)
        nel = float(head._num_valence[g.z.cpu()].sum())
        base = {"k_vec": k_red, "k_dist": k_dist, "k_node": k_node}
        if labels is not None:
            base["k_labels"] = labels
        dv = lambda t: t.to(device).contiguous()
        if soc_switch:
            be = kspace.band_energies_soc(head, dv(on), dv(ion), dv(off), dv(ioff), gd, k_vecs=k_cart)[0]
            eig = be.double().cpu().numpy() * AU2EV                                  # [2 bands, nk]
            occ = int(round(nel))                                                    # spinor bands hold one electron each (script :222-224)
            vbm, cbm = float(eig[occ - 1].max()), float(eig[occ].min())
            out.append(dict(base, bands_eV=eig - vbm, vbm_eV=vbm, band_gap_eV=cbm - vbm))
        elif spin_colinear:
            res = {"bands_eV": [], "vbm_eV": [], "band_gap_eV": []}
            half = math.ceil(nel / 2)
            for ispin in range(2):                                                   # rows [., spin, nao, nao] (script :311-312, :346-392)
                sel = lambda t: t.reshape(-1, 2, nao_max * nao_max)[:, ispin].contiguous()
                be = kspace.band_energies(head, dv(sel(on)), dv(sel(off)), gd, k_vecs=k_cart)[0]
                eig = be.double().cpu().numpy() * AU2EV
                vbm, cbm = float(eig[half - 1].max()), float(eig[half].min())
                res["bands_eV"].append(eig - vbm)
                res["vbm_eV"].append(vbm)
                res["band_gap_eV"].append(cbm - vbm)
            out.append(dict(base, **res))
        else:
            be = kspace.band_energies(head, dv(on), dv(off), gd, k_vecs=k_cart)[0]
            eig = be.double().cpu().numpy() * AU2EV                                  # [bands, nk]
            half = math.ceil(nel / 2)
            vbm, cbm = float(eig[half - 1].max()), float(eig[half].min())
            out.append(dict(base, bands_eV=eig - vbm, vbm_eV=vbm, band_gap_eV=cbm - vbm))
    if rows is not None and r0 != rows.shape[0]:
        raise ValueError(f"band_structure: {rows.shape[0]} Hamiltonian rows for crystals that need {r0}")
    return out
