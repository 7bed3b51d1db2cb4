"""Synthetic periodic crystals in the reference's graph format: positions in Bohr, every periodic pair with
|r_ij| < r_i + r_j (OpenMX PAO cut-off radii, data table from hamgnn/models/base_model.py:25-61), directed edges emitted
centre-major (edge_index[0] = centre j, edge_index[1] = neighbour i, sorted by centre like read_openmx.c:846-893 does),
integer cell_shift, Cartesian nbr_shift = cell_shift @ cell, graph-local inv_edge_idx."""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial import cKDTree

from ..basis import atomic_radii
from .graph import Graph

SYMBOLS = ("X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo Tc "
           "Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi").split()


def build_graph(pos: np.ndarray, cell: np.ndarray, z: np.ndarray, radius_type="openmx", radius_scale=1.0, pbc=(True, True, True)) -> Graph:
    radii_tab = atomic_radii(radius_type)
    rad = np.array([radii_tab[SYMBOLS[int(Z)]] for Z in z], dtype=np.float64) * radius_scale
    rmax = 2 * rad.max()
    # number of images needed along each lattice vector
    vol = abs(np.linalg.det(cell))
    heights = vol / np.array([np.linalg.norm(np.cross(cell[1], cell[2])), np.linalg.norm(np.cross(cell[2], cell[0])),
                              np.linalg.norm(np.cross(cell[0], cell[1]))])
    nimg = [int(np.ceil(rmax / h)) if p else 0 for h, p in zip(heights, pbc)]
    shifts = np.array([(a, b, c) for a in range(-nimg[0], nimg[0] + 1) for b in range(-nimg[1], nimg[1] + 1)
                       for c in range(-nimg[2], nimg[2] + 1)], dtype=np.int64)
    n = len(z)
    img_pos = (pos[None, :, :] + (shifts @ cell)[:, None, :]).reshape(-1, 3)
    tree = cKDTree(img_pos)
    src, dst, sh = [], [], []
    for j in range(n):
        cand = np.asarray(tree.query_ball_point(pos[j], rad[j] + rad.max() + 1e-9), dtype=np.int64)
        if cand.size == 0:
            continue
        ci, cs = cand % n, cand // n
        d = np.linalg.norm(img_pos[cand] - pos[j], axis=1)
        keep = (d < rad[j] + rad[ci]) & ~((ci == j) & (np.abs(shifts[cs]).sum(1) == 0))
        ci, cs = ci[keep], cs[keep]
        order = np.lexsort((shifts[cs][:, 2], shifts[cs][:, 1], shifts[cs][:, 0], ci))
        src.append(np.full(order.size, j, dtype=np.int64))
        dst.append(ci[order])
        sh.append(shifts[cs[order]])
    src, dst, sh = np.concatenate(src), np.concatenate(dst), np.concatenate(sh)
    # inverse edge: (i -> j, -shift)
    off = 2 * max(nimg) + 1 if max(nimg) > 0 else 1
    R = max(nimg)
    key = ((src * n + dst) * off + (sh[:, 0] + R)) * off * off + (sh[:, 1] + R) * off + (sh[:, 2] + R)
    ikey = ((dst * n + src) * off + (-sh[:, 0] + R)) * off * off + (-sh[:, 1] + R) * off + (-sh[:, 2] + R)
    order = np.argsort(key)
    pos_in_sorted = np.searchsorted(key[order], ikey)
    inv = order[pos_in_sorted]
    assert np.array_equal(key[inv], ikey), "edge list is not symmetric"
    g = Graph(z=torch.from_numpy(np.asarray(z, dtype=np.int64)), pos=torch.from_numpy(pos.astype(np.float32)),
              cell=torch.from_numpy(cell.astype(np.float32))[None], edge_index=torch.from_numpy(np.stack([src, dst])),
              cell_shift=torch.from_numpy(sh), nbr_shift=torch.from_numpy((sh @ cell).astype(np.float32)),
              inv_edge_idx=torch.from_numpy(inv), batch=torch.zeros(n, dtype=torch.long), node_counts=torch.tensor([n]))
    return g


def add_random_targets(g: Graph, nao, seed=0, soc=False, basis_def=None):
    """H0 ~ N(0, 0.1^2), Hermitian-consistent through inv_edge_idx and masked by basis_def; L ~ N(0,1) antisymmetrised."""
    rng = np.random.default_rng(seed)
    N, E = g.num_nodes, g.num_edges
    inv = g.inv_edge_idx.numpy()
    dim = 2 * nao if soc else nao

    def herm(rows, inv_):
        A = rng.normal(0, 0.1, size=(rows, dim, dim)).astype(np.float32)
        B = A if inv_ is None else A[inv_]
        return (0.5 * (A + B.transpose(0, 2, 1))).reshape(rows, dim * dim)          # (explicit width: rows may be 0)
    g["Hon0"], g["Hoff0"] = torch.from_numpy(herm(N, None)), torch.from_numpy(herm(E, inv))
    g["Hon"], g["Hoff"] = g["Hon0"].clone(), g["Hoff0"].clone()
    g["Son"], g["Soff"] = torch.zeros(N, nao * nao), torch.zeros(E, nao * nao)
    if soc:
        g["iHon0"], g["iHoff0"] = torch.from_numpy(herm(N, None)), torch.from_numpy(herm(E, inv))
        g["iHon"], g["iHoff"] = g["iHon0"].clone(), g["iHoff0"].clone()
        g["Lon"] = torch.from_numpy(rng.normal(size=(N, nao * nao, 3)).astype(np.float32))
        g["Loff"] = torch.from_numpy(rng.normal(size=(E, nao * nao, 3)).astype(np.float32))
    return g


def si_diamond(nx=1, ny=1, nz=1, jitter=0.0, seed=0, primitive=False) -> Graph:
    a = 10.263  # Bohr
    if primitive:
        cell = 0.5 * a * np.array([[0, 1, 1], [1, 0, 1], [1, 1, 0]], dtype=np.float64)
        frac = np.array([[0, 0, 0], [0.25, 0.25, 0.25]])
        pos = frac @ cell
        cell = cell * np.array([nx, ny, nz])[:, None]
        if (nx, ny, nz) != (1, 1, 1):
            raise ValueError("primitive cell generator is 1x1x1 only")
    else:
        base = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0], [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])
        reps = np.array([(i, j, k) for i in range(nx) for j in range(ny) for k in range(nz)], dtype=np.float64)
        pos = ((base[None] + reps[:, None]) * a).reshape(-1, 3)
        cell = np.diag([nx * a, ny * a, nz * a]).astype(np.float64)
    rng = np.random.default_rng(seed)
    if jitter > 0:
        pos = pos + rng.normal(0, jitter, size=pos.shape)
    return build_graph(pos, cell, np.full(len(pos), 14))


def mos2_monolayer(nx=20, ny=20, seed=0, jitter=0.02) -> Graph:
    a, h, vac = 5.97, 5.9, 40.0
    a1, a2 = np.array([a, 0, 0]), np.array([-0.5 * a, np.sqrt(3) / 2 * a, 0])
    basis = [(42, np.array([0, 0, 0.0])), (16, (a1 + 2 * a2) / 3 + np.array([0, 0, h / 2])), (16, (a1 + 2 * a2) / 3 - np.array([0, 0, h / 2]))]
    pos, z = [], []
    for i in range(nx):
        for j in range(ny):
            for Z, b in basis:
                pos.append(i * a1 + j * a2 + b + np.array([0, 0, vac / 2]))
                z.append(Z)
    pos = np.array(pos) + np.random.default_rng(seed).normal(0, jitter, size=(len(z), 3))
    cell = np.array([nx * a1, ny * a2, [0, 0, vac]])
    return build_graph(pos, cell, np.array(z), pbc=(True, True, False))


def amorphous_sio2(n_atoms=10002, seed=1, density=0.00978, min_dist=2.8) -> Graph:
    """random-packed Si:O = 1:2 at `density` atoms/Bohr^3 (grid-jitter placement keeps a minimum distance cheaply)."""
    rng = np.random.default_rng(seed)
    L = (n_atoms / density) ** (1 / 3)
    m = int(np.ceil(n_atoms ** (1 / 3)))
    grid = np.array([(i, j, k) for i in range(m) for j in range(m) for k in range(m)], dtype=np.float64)
    rng.shuffle(grid)
    spacing = L / m
    amp = max(0.0, 0.5 * (spacing - min_dist))
    pos = (grid[:n_atoms] + 0.5) * spacing + rng.uniform(-amp, amp, size=(n_atoms, 3))
    z = np.where(np.arange(n_atoms) % 3 == 0, 14, 8)
    return build_graph(pos, np.diag([L, L, L]), z)


def random_cell(n_atoms, zs, seed=0, density=0.012) -> Graph:
    rng = np.random.default_rng(seed)
    L = (n_atoms / density) ** (1 / 3)
    m = int(np.ceil(n_atoms ** (1 / 3)))
    grid = np.array([(i, j, k) for i in range(m) for j in range(m) for k in range(m)], dtype=np.float64)
    rng.shuffle(grid)
    pos = (grid[:n_atoms] + 0.5) * (L / m) + rng.uniform(-0.6, 0.6, size=(n_atoms, 3))
    z = rng.choice(np.asarray(zs), size=n_atoms)
    return build_graph(pos, np.diag([L, L, L]) + rng.normal(0, 0.2, size=(3, 3)), z)
