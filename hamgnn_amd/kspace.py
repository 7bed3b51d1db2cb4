"""The k-space step after the read-out: `HamGNNPlusPlusOut(calculate_band_energy=True)` of the reference
(hamgnn/models/hamgnn_output.py:1675-1996 calculate_band_energies; k-point generation :3802-3854; hamgnn/physics/kpoints.py:26-165).

Per crystal: k-points (random, or a path through given nodes in reduced coordinates) -> H(k), S(k) in the compact orbital basis by the
HIP kernel `hg_hk_assemble` (phase-factor sums over the edges of every atom pair, fixed order) -> generalized eigenproblem
H(k) psi = E S(k) psi through the Cholesky factor of S(k) exactly as the reference does it, on hipSOLVER via `torch.linalg`
(cholesky / inv / eigh on complex64: library calls, not kernels of this repository) -> band energies, wavefunctions, band gap,
optional band window.  The spin-free branch with the reference overlaps `Son / Soff` (also what the reference runs with ham_only=False: its
overlap-network variant :1368-1673 factorises the REFERENCE overlap too and is only reached with export_reciprocal_values) and the spinor
branch `band_energies_soc` (:1998-2286)."""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch

from . import ops
from .topo import gget


# ------------------------------------------------------------------------------------------------ k-points (host geometry)
def k_path_points(kpts, nk: int, lat: np.ndarray):
    """The k-path of the reference (kpoints_generator(dim_k=3, lat).k_path(kpts, nk), hamgnn/physics/kpoints.py:26-165) as one piecewise
    linear interpolation: the path is a polyline through the reduced-coordinate nodes `kpts`, parametrised by its arc length in the
    reciprocal metric of `lat`; every node is pinned to the sample index nearest to its share of the total length, and between two pinned
    indices the samples are equally spaced.  That is `np.interp` of the three coordinates over the sample index with the pinned indices as
    abscissae.  Returns (k_vec [nk, 3] reduced, inv(lat).T)."""
    nodes = np.asarray(kpts, dtype=np.float64)
    if nodes.ndim != 2 or nodes.shape[1] != 3:
        raise ValueError("k_path: the nodes must be an [n, 3] list of reduced coordinates")
    if nk < nodes.shape[0]:
        raise ValueError("k_path: fewer sample points than nodes")
    cellm = np.asarray(lat, dtype=np.float64)
    metric = np.linalg.inv(cellm @ cellm.T)                                    # |dk|^2 = dk^T (A A^T)^-1 dk for reduced dk
    steps = np.diff(nodes, axis=0)
    arc = np.concatenate([[0.0], np.cumsum(np.sqrt(np.einsum("ni,ij,nj->n", steps, metric, steps)))])
    pins = np.rint(arc / arc[-1] * (nk - 1)).astype(np.int64)                  # sample index of every node (ends: 0 and nk - 1 exactly)
    pins[0], pins[-1] = 0, nk - 1
    if np.any(np.diff(pins) <= 0):
        raise ValueError("k_path: two nodes fall onto the same sample point (coincident nodes or too few points)")
    idx = np.arange(nk, dtype=np.float64)
    k_vec = np.stack([np.interp(idx, pins.astype(np.float64), nodes[:, d]) for d in range(3)], axis=1)
    return k_vec, np.linalg.inv(cellm).T


AU2ANG = 0.5291772083       # hamgnn/utils/constants.py:2


def auto_k_path_nodes(lat_bohr: np.ndarray, pos_bohr: np.ndarray, z) -> list:
    return auto_k_path(lat_bohr, pos_bohr, z)[1]


def auto_k_path(lat_bohr: np.ndarray, pos_bohr: np.ndarray, z):
    """-> (labels, nodes).  k_path='auto' of the reference (hamgnn_output.py:3812-3833): the high-symmetry path of the crystal from pymatgen's KPathSeek -- a Structure in
    Angstrom with the species' symbols, the labels of all path segments in a row with consecutive repeats dropped, their reduced coordinates as nodes.
    pymatgen is a third-party dependency of the reference (imported at :19-21) that this image does not have: without it the call raises."""
    try:
        from pymatgen.core.periodic_table import Element
        from pymatgen.core.structure import Structure
        from pymatgen.symmetry.kpath import KPathSeek
    except ImportError as e:
        raise NotImplementedError("k_path='auto' needs pymatgen (Structure / Element / KPathSeek), as in the reference (hamgnn_output.py:19-21, 3812-3837); "
                                  "give a list of reduced k-points or None (random) instead") from e
    structure = Structure(lattice=np.asarray(lat_bohr) * AU2ANG, species=[Element.from_Z(int(k)).symbol for k in z], coords=np.asarray(pos_bohr) * AU2ANG,
                          coords_are_cartesian=True)
    seek = KPathSeek(structure=structure)
    labels = [lab for group in seek.kpath["path"] for lab in group]
    unique = [labels[0]]
    for x in labels[1:]:
        if x != unique[-1]:
            unique.append(x)
    return unique, [seek.kpath["kpoints"][k] for k in unique]


def make_k_vectors(k_path, num_k: int, cell: torch.Tensor, rng=np.random, data=None) -> torch.Tensor:
    """data.k_vecs of the reference (:3802-3854): [n_crystals, num_k, 3] Cartesian-reciprocal k-vectors (no 2 pi: it sits in the phase).
    k_path: list of nodes in reduced coordinates; None -> uniformly random reduced k in [-1, 1)^3 (numpy's global RNG, as the reference); 'auto' -> the
    crystal's high-symmetry path (auto_k_path_nodes; needs `data` for positions / species and pymatgen; as in the reference a path that cannot be sampled
    falls back to random points)."""
    out = []
    cells = cell.detach().cpu().double().numpy().reshape(-1, 3, 3)
    auto = isinstance(k_path, str) and k_path.lower() == "auto"
    if auto:
        if data is None:
            raise ValueError("k_path='auto' needs the graph (positions and species of every crystal)")
        counts = gget(data, "node_counts")
        counts = [int(c) for c in counts.tolist()] if counts is not None else [int(data.z.shape[0])]
        pos_all, z_all = data.pos.detach().cpu().double().numpy(), data.z.detach().cpu().numpy()
        starts = np.concatenate([[0], np.cumsum(counts)])
    for ci, lat in enumerate(cells):
        if isinstance(k_path, (list, tuple)):
            k_vec, lat_per_inv = k_path_points(k_path, num_k, lat)
        elif auto:
            nodes = auto_k_path_nodes(lat, pos_all[starts[ci]:starts[ci + 1]], z_all[starts[ci]:starts[ci + 1]])
            try:
                k_vec, lat_per_inv = k_path_points(nodes, num_k, lat)
            except Exception:                                  # noqa: BLE001  (the reference's bare except, :3709-3713 / :3838-3841: random k-points)
                lat_per_inv = np.linalg.inv(lat).T
                k_vec = 2.0 * rng.rand(num_k, 3) - 1.0
        elif k_path is None:
            lat_per_inv = np.linalg.inv(lat).T
            k_vec = 2.0 * rng.rand(num_k, 3) - 1.0
        else:
            raise NotImplementedError(f"k_path={k_path!r}: a list of reduced k-points, 'auto' or None (random)")
        out.append(torch.from_numpy(k_vec.dot(lat_per_inv[np.newaxis, :, :]).reshape(-1, 3)).float())
    return torch.stack(out, 0)


# ------------------------------------------------------------------------------------------------ per-crystal index plumbing
def _crystal_slices(data):
    node_counts = gget(data, "node_counts")
    src = data.edge_index[0]
    if node_counts is None:
        return [(0, int(data.z.shape[0]), 0, int(src.shape[0]))]
    ncs = [int(v) for v in node_counts.tolist()]
    batch = gget(data, "batch")
    ecs = torch.bincount(batch[src], minlength=len(ncs)).tolist() if batch is not None else [int(src.shape[0])]
    out, n0, e0 = [], 0, 0
    for n, e in zip(ncs, ecs):
        out.append((n0, n, e0, int(e)))
        n0, e0 = n0 + n, e0 + int(e)
    return out


def assemble_k(on, off, data, k_vecs_c, n0, n, e0, e, orank_all, nao):
    """[nk, M, M] complex64 of one crystal (rows n0:n0+n, edges e0:e0+e) from planar [.., nao^2] blocks"""
    dev = on.device
    src = (data.edge_index[0][e0:e0 + e] - n0).contiguous()
    dst = (data.edge_index[1][e0:e0 + e] - n0).contiguous()
    key = src * n + dst
    order = torch.sort(key, stable=True).indices
    uniq, counts = torch.unique_consecutive(key[order], return_counts=True)
    ptr = torch.zeros(uniq.numel() + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(counts, 0)
    pij = torch.stack([uniq // n, uniq % n], 1).contiguous()
    orank = orank_all[n0:n0 + n].contiguous()
    norb = (orank >= 0).sum(1)
    ooff = (torch.cumsum(norb, 0) - norb).to(torch.int32).contiguous()
    M = int(norb.sum())
    return ops.hk_assemble(on[n0:n0 + n].contiguous(), off[e0:e0 + e].contiguous(), data.nbr_shift[e0:e0 + e].contiguous().float(),
                           k_vecs_c.contiguous().float(), ptr, order.contiguous(), pij, n, nao, orank, ooff, M), M


def _compact_index(orank_all, n0, n):
    """[n, nao] compact orbital index of (atom, orbital) inside one crystal, or -1"""
    orank = orank_all[n0:n0 + n]
    norb = (orank >= 0).sum(1)
    ooff = torch.cumsum(norb, 0) - norb
    return torch.where(orank >= 0, ooff[:, None] + orank, torch.full_like(orank, -1)).long()


def assemble_k_adjoint(G, data, k_vecs_c, n0, n, e0, e, orank_all, nao):
    """Adjoint of assemble_k (hg_hk_assemble) for one crystal: G [nk, M, M] complex = the gradient of a real loss with respect to H(k) in
    torch's convention (d/dRe + i d/dIm) -> (g_on [n, nao^2], g_off [e, nao^2]) real.  With H(k)[(i a), (j b)] = on_i[a, b] delta_ij +
    sum_{e: i -> j} exp(2 pi i k . shift_e) off_e[a, b]:  g_off_e[a, b] = sum_k Re(conj(phase_k(e)) G_k[(i a), (j b)]).  Gathers and
    one complex multiply-reduce per chunk of k (torch tensor ops; device-agnostic, checked on CPU against autograd)."""
    comp = _compact_index(orank_all, n0, n)                    # [n, nao]
    src = (data.edge_index[0][e0:e0 + e] - n0).long()
    dst = (data.edge_index[1][e0:e0 + e] - n0).long()
    nk = G.shape[0]
    rdt = G.real.dtype
    ok_on = ((comp[:, :, None] >= 0) & (comp[:, None, :] >= 0)).to(rdt)
    Gon = G[:, comp.clamp(min=0)[:, :, None], comp.clamp(min=0)[:, None, :]]                # [nk, n, nao, nao]
    g_on = (Gon.real.sum(0) * ok_on).reshape(n, nao * nao)
    R, Cc = comp[src], comp[dst]
    ok = ((R[:, :, None] >= 0) & (Cc[:, None, :] >= 0)).to(rdt)
    shift = data.nbr_shift[e0:e0 + e].to(torch.float64)
    ph = 2.0 * math.pi * (k_vecs_c.to(torch.float64)[:, None, :] * shift[None, :, :]).sum(-1)      # [nk, e] in double (see the kernel)
    cph = torch.complex(torch.cos(ph), -torch.sin(ph)).to(G.dtype)                               # conj(phase)
    g_off = torch.zeros(e, nao, nao, device=G.device, dtype=rdt)
    for k0 in range(0, nk, 8):
        Ge = G[k0:k0 + 8][:, R.clamp(min=0)[:, :, None], Cc.clamp(min=0)[:, None, :]]          # [<=8, e, nao, nao]
        g_off += (cph[k0:k0 + 8][:, :, None, None] * Ge).real.sum(0)
    return g_on, (g_off * ok).reshape(e, nao * nao)


def _eig_chain(head, Hk, Sk, val_c, z_c):
    """generalized eigenproblem through the Cholesky factor of S(k), as the reference (:1911-1928) -> (evals, evecs, Ht, gap)"""
    L = torch.linalg.cholesky(Sk)
    Linv = torch.linalg.inv(L)
    LHinv = torch.linalg.inv(L.conj().transpose(-1, -2))
    Ht = torch.bmm(torch.bmm(Linv, Hk), LHinv)
    evals, evecs = torch.linalg.eigh(Ht)
    evecs = torch.einsum("ijk,ika->iaj", LHinv, evecs)
    half = math.ceil(float(val_c.sum()) / 2)
    gap = (evals[:, half].min() - evals[:, half - 1].max()).reshape(1)
    bnc = head.band_num_control
    if bnc is not None:
        if isinstance(bnc, dict):
            nb = int(sum(int(bnc.get(int(zz), bnc.get(str(int(zz)), 0))) for zz in z_c.tolist()))
            evals, evecs = evals[:, :nb], evecs[:, :nb, :]
        else:
            win = max(1, int(bnc * half)) if isinstance(bnc, float) else min(int(bnc), half)
            evals, evecs = evals[:, half - win:half + win], evecs[:, half - win:half + win, :]
    return evals, evecs, Ht, gap


def band_energies(head, onsite_hamiltonian, offsite_hamiltonian, data, k_vecs: Optional[torch.Tensor] = None):
    """calculate_band_energies(onsite, offsite, data) of the reference (:1675-1996, export_reciprocal_values=False): returns
    (band_energy [sum_c bands_c, num_k], wavefunction (flattened), band_gap [n_crystals], H_sym (flattened))."""
    nao = head.nao_max
    dev = onsite_hamiltonian.device
    k_vecs = gget(data, "k_vecs") if k_vecs is None else k_vecs
    if k_vecs is None:
        raise ValueError("band_energies: no k-vectors (data.k_vecs)")
    k_vecs = k_vecs.to(dev)
    z = data.z
    orank_all = head._orank.to(dev)[z]                         # [N, nao] rank of an orbital in its element's valid set or -1
    val = head._num_valence.to(dev)[z].to(torch.float64)
    energies, waves, gaps, hsyms = [], [], [], []
    Son, Soff = data.Son.contiguous().float(), data.Soff.contiguous().float()
    for c, (n0, n, e0, e) in enumerate(_crystal_slices(data)):
        Hk, M = assemble_k(onsite_hamiltonian, offsite_hamiltonian, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
        Sk, _ = assemble_k(Son, Soff, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
        evals, evecs, Ht, gap = _eig_chain(head, Hk, Sk, val[n0:n0 + n], z[n0:n0 + n])
        gaps.append(gap)
        energies.append(evals.transpose(-1, -2))
        waves.append(evecs.reshape(-1))
        hsyms.append(Ht.reshape(-1))
    return torch.cat(energies, 0), torch.cat(waves, 0), torch.cat(gaps, 0), torch.cat(hsyms, 0)


def band_energies_export(head, onsite_hamiltonian, offsite_hamiltonian, data, overlap=None, k_vecs: Optional[torch.Tensor] = None):
    """The `export_reciprocal_values=True` form of the k-space step: calculate_band_energies(..., True) (hamgnn_output.py:1675-1996; `overlap`
    None) and calculate_band_energies_with_overlap(..., True) (:1368-1673; `overlap` = the PREDICTED (onsite, offsite) overlap rows of the
    overlap networks).  Returns (band_energy [sum_c bands_c, num_k], wavefunction [C, num_k, bands, M] normalised to <psi|S(k)|psi> = 1,
    HK [C, num_k, M, M], SK [C, num_k, M, M] (the reference overlap, or the predicted one), dSK [C, num_k, M, M, 3] from data.dSon / dSoff,
    band_gap [C]).  As in the reference the eigenproblem is always solved with the REFERENCE overlap (:1603), and the per-crystal results are
    stacked, i.e. every crystal of the batch must have the same number of orbitals."""
    nao = head.nao_max
    dev = onsite_hamiltonian.device
    k_vecs = gget(data, "k_vecs") if k_vecs is None else k_vecs
    if k_vecs is None:
        raise ValueError("band_energies_export: no k-vectors (data.k_vecs)")
    k_vecs = k_vecs.to(dev)
    z = data.z
    orank_all = head._orank.to(dev)[z]
    val = head._num_valence.to(dev)[z].to(torch.float64)
    Son, Soff = data.Son.contiguous().float(), data.Soff.contiguous().float()
    dSon, dSoff = data.dSon.float().reshape(Son.shape[0], nao * nao, 3), data.dSoff.float().reshape(Soff.shape[0], nao * nao, 3)
    energies, waves, gaps, HKs, SKs, dSKs = [], [], [], [], [], []
    for c, (n0, n, e0, e) in enumerate(_crystal_slices(data)):
        Hk, M = assemble_k(onsite_hamiltonian, offsite_hamiltonian, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
        Sk, _ = assemble_k(Son, Soff, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
        Sp = Sk if overlap is None else assemble_k(overlap[0].contiguous().float(), overlap[1].contiguous().float(), data, k_vecs[c], n0, n, e0, e, orank_all, nao)[0]
        dSk = torch.stack([assemble_k(dSon[..., d].contiguous(), dSoff[..., d].contiguous(), data, k_vecs[c], n0, n, e0, e, orank_all, nao)[0] for d in range(3)], -1)
        evals, evecs, _, gap = _eig_chain(head, Hk, Sk, val[n0:n0 + n], z[n0:n0 + n])
        norm = torch.einsum("nai,nij,naj->na", evecs.conj(), Sk, evecs).real
        evecs = evecs * (1.0 / torch.sqrt(norm)).unsqueeze(-1)
        energies.append(evals.transpose(-1, -2))
        waves.append(evecs)
        gaps.append(gap)
        HKs.append(Hk)
        SKs.append(Sp)
        dSKs.append(dSk)
    if len({tuple(w.shape) for w in waves}) != 1:
        raise ValueError("export_reciprocal_values stacks the crystals' H(k) / S(k) / wavefunctions (hamgnn_output.py:1984-1990): every crystal of the "
                         "batch must have the same number of orbitals and bands")
    return torch.cat(energies, 0), torch.stack(waves, 0), torch.stack(HKs, 0), torch.stack(SKs, 0), torch.stack(dSKs, 0), torch.cat(gaps, 0)


def band_energies_soc(head, real_onsite, imag_onsite, real_offsite, imag_offsite, data, k_vecs: Optional[torch.Tensor] = None):
    """calculate_band_energies_with_spin_orbit_coupling of the reference (hamgnn_output.py:1998-2286): spinor Hamiltonian rows
    [., (2 nao)^2] (real and imaginary part, on-site and off-site) -> (band_energy [sum_c bands_c, num_k], wavefunction (flattened)).
    Per crystal the four spin blocks (uu, ud, du, dd) of H(k) are phase-factor sums like the spin-free case -- each block = assembly of its
    real part + i x assembly of its imaginary part (the assembly is linear), eight launches of hg_hk_assemble -- stacked to [2 M, 2 M];
    S(k) is the spin-free overlap on both spin diagonals (kron(1_2, S(k)), :2165-2167); generalised eigenproblem through the Cholesky
    factor of S(k) (:2236-2252, hipSOLVER through torch.linalg); band window :2254-2263 (dict: leading bands; int: +- that many bands
    around the number of valence electrons)."""
    nao = head.nao_max
    dev = real_onsite.device
    k_vecs = (gget(data, "k_vecs") if k_vecs is None else k_vecs)
    if k_vecs is None:
        raise ValueError("band_energies_soc: no k-vectors (data.k_vecs)")
    k_vecs = k_vecs.to(dev)
    z = data.z
    orank_all = head._orank.to(dev)[z]
    val = head._num_valence.to(dev)[z].to(torch.float64)
    Son, Soff = data.Son.contiguous().float(), data.Soff.contiguous().float()
    energies, waves = [], []
    for c, (n0, n, e0, e) in enumerate(_crystal_slices(data)):
        Sk, M = assemble_k(Son, Soff, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
        Hk = _soc_hk(real_onsite, imag_onsite, real_offsite, imag_offsite, data, k_vecs[c], n0, n, e0, e, orank_all, nao)   # [nk, 2 M, 2 M]
        Ssoc = torch.zeros_like(Hk)
        Ssoc[:, :M, :M] = Sk
        Ssoc[:, M:, M:] = Sk
        L = torch.linalg.cholesky(Ssoc)
        Linv = torch.linalg.inv(L)
        LHinv = torch.linalg.inv(L.conj().transpose(-1, -2))
        evals, evecs = torch.linalg.eigh(torch.bmm(torch.bmm(Linv, Hk), LHinv))
        evecs = torch.bmm(LHinv, evecs)
        bnc = head.band_num_control
        if bnc is not None:
            if isinstance(bnc, dict):
                nb = int(sum(int(bnc.get(int(zz), bnc.get(str(int(zz)), 0))) for zz in z[n0:n0 + n].tolist()))
                evals, evecs = evals[:, :nb], evecs[:, :nb, :]
            else:
                nval = int(val[n0:n0 + n].sum())
                evals, evecs = evals[:, nval - int(bnc):nval + int(bnc)], evecs[:, nval - int(bnc):nval + int(bnc), :]
        energies.append(evals.transpose(-1, -2))
        waves.append(evecs.reshape(-1))
    return torch.cat(energies, 0), torch.cat(waves, 0)


def band_energy_backward(head, onsite_hamiltonian, offsite_hamiltonian, data, cotangent, k_vecs: Optional[torch.Tensor] = None):
    """gradient of sum(band_energy * cotangent) with respect to the real-space blocks (the band-energy loss of the reference's second
    training stage, Model.py:150-196 with prediction: band_energy): H(k) from the assembly kernel, the Cholesky / eigh chain
    differentiated by torch.autograd (library solvers, as in the forward), then the assembly's adjoint.  Returns (g_on, g_off)."""
    nao = head.nao_max
    dev = onsite_hamiltonian.device
    k_vecs = (gget(data, "k_vecs") if k_vecs is None else k_vecs).to(dev)
    z = data.z
    orank_all = head._orank.to(dev)[z]
    val = head._num_valence.to(dev)[z].to(torch.float64)
    Son, Soff = data.Son.contiguous().float(), data.Soff.contiguous().float()
    g_on = torch.zeros_like(onsite_hamiltonian)
    g_off = torch.zeros_like(offsite_hamiltonian)
    row = 0
    for c, (n0, n, e0, e) in enumerate(_crystal_slices(data)):
        Hk, M = assemble_k(onsite_hamiltonian, offsite_hamiltonian, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
        Sk, _ = assemble_k(Son, Soff, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
        with torch.enable_grad():
            Hk = Hk.detach().requires_grad_()
            evals = _eig_chain(head, Hk, Sk, val[n0:n0 + n], z[n0:n0 + n])[0].transpose(-1, -2)     # [bands, nk]
            nb = evals.shape[0]
            (G,) = torch.autograd.grad((evals * cotangent[row:row + nb].to(evals.dtype)).sum(), Hk)
        row += nb
        a, b = assemble_k_adjoint(G, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
        g_on[n0:n0 + n] = a
        g_off[e0:e0 + e] = b
    return g_on, g_off


def _soc_hk(real_onsite, imag_onsite, real_offsite, imag_offsite, data, k_vecs_c, n0, n, e0, e, orank_all, nao):
    """spinor H(k) [nk, 2 M, 2 M] of one crystal: four spin blocks, each the assembly of its real part + i x the assembly of its imaginary part"""
    blk = lambda t, a, b: t.reshape(-1, 2, nao, 2, nao)[:, a, :, b, :].reshape(-1, nao * nao).contiguous().float()
    rows = []
    for a in (0, 1):
        cols = []
        for b in (0, 1):
            Hr, _ = assemble_k(blk(real_onsite, a, b), blk(real_offsite, a, b), data, k_vecs_c, n0, n, e0, e, orank_all, nao)
            Hi, _ = assemble_k(blk(imag_onsite, a, b), blk(imag_offsite, a, b), data, k_vecs_c, n0, n, e0, e, orank_all, nao)
            cols.append(Hr + 1j * Hi)
        rows.append(torch.cat(cols, -1))
    return torch.cat(rows, -2)


def band_energy_backward_soc(head, real_onsite, imag_onsite, real_offsite, imag_offsite, data, cotangent, k_vecs: Optional[torch.Tensor] = None):
    """gradient of sum(band_energy * cotangent) of band_energies_soc with respect to the four spinor row sets (a band-energy loss on a
    spin-orbit head; Model.py:150-196 with prediction: band_energy, bands from hamgnn_output.py:1998-2286).  As band_energy_backward: the
    Cholesky / eigh chain on the stacked [2 M, 2 M] H(k) is differentiated by torch.autograd (library solvers), then every spin block of
    the gradient G goes through the assembly's adjoint -- G_ab for the real rows, -i G_ab for the imaginary rows (H_ab = A(real) + i A(imag)
    with the real-linear assembly A).  Returns (g_real_on, g_imag_on, g_real_off, g_imag_off) in the rows' [., 2, nao, 2, nao] layout."""
    nao = head.nao_max
    dev = real_onsite.device
    k_vecs = (gget(data, "k_vecs") if k_vecs is None else k_vecs).to(dev)
    z = data.z
    orank_all = head._orank.to(dev)[z]
    val = head._num_valence.to(dev)[z].to(torch.float64)
    Son, Soff = data.Son.contiguous().float(), data.Soff.contiguous().float()
    g = [torch.zeros(t.shape[0], 2, nao, 2, nao, device=dev, dtype=torch.float32) for t in (real_onsite, imag_onsite, real_offsite, imag_offsite)]
    row = 0
    for c, (n0, n, e0, e) in enumerate(_crystal_slices(data)):
        Sk, M = assemble_k(Son, Soff, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
        Hk = _soc_hk(real_onsite, imag_onsite, real_offsite, imag_offsite, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
        Ssoc = torch.zeros_like(Hk)
        Ssoc[:, :M, :M] = Sk
        Ssoc[:, M:, M:] = Sk
        with torch.enable_grad():
            Hk = Hk.detach().requires_grad_()
            L = torch.linalg.cholesky(Ssoc)
            Linv = torch.linalg.inv(L)
            LHinv = torch.linalg.inv(L.conj().transpose(-1, -2))
            evals = torch.linalg.eigvalsh(torch.bmm(torch.bmm(Linv, Hk), LHinv))
            bnc = head.band_num_control
            if bnc is not None:
                if isinstance(bnc, dict):
                    nb = int(sum(int(bnc.get(int(zz), bnc.get(str(int(zz)), 0))) for zz in z[n0:n0 + n].tolist()))
                    evals = evals[:, :nb]
                else:
                    nval = int(val[n0:n0 + n].sum())
                    evals = evals[:, nval - int(bnc):nval + int(bnc)]
            evals = evals.transpose(-1, -2)                    # [bands, nk]
            nb = evals.shape[0]
            (G,) = torch.autograd.grad((evals * cotangent[row:row + nb].to(evals.dtype)).sum(), Hk)
        row += nb
        for a in (0, 1):
            for b in (0, 1):
                Gab = G[:, a * M:(a + 1) * M, b * M:(b + 1) * M]
                r_on, r_off = assemble_k_adjoint(Gab, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
                i_on, i_off = assemble_k_adjoint(-1j * Gab, data, k_vecs[c], n0, n, e0, e, orank_all, nao)
                g[0][n0:n0 + n, a, :, b, :] = r_on.reshape(n, nao, nao)
                g[1][n0:n0 + n, a, :, b, :] = i_on.reshape(n, nao, nao)
                g[2][e0:e0 + e, a, :, b, :] = r_off.reshape(e, nao, nao)
                g[3][e0:e0 + e, a, :, b, :] = i_off.reshape(e, nao, nao)
    return tuple(t.reshape(t.shape[0], -1) for t in g)
