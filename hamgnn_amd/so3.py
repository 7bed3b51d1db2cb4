"""Host-side SO(3) constants for the MI355X hot path (numpy, float64; runs once at plan creation).

This is PRODUCT code (it never imports oracle/).  It provides
  * irreps parsing/sorting with e3nn 0.5.0 semantics (needed to reproduce the reference's instruction tables and
    flat weight layouts: /root/reference/hamgnn/nn/message_passing.py:136-171),
  * real Wigner-3j tensors in e3nn's basis/phase convention (Racah formula + real<->complex change of basis),
  * the *edge-aligned frame* tables the fused kernel is built on:
        - ``aligned_path(l_in, l_sh, l_out)``: with the edge direction rotated onto the pole, Y^l(pole) = sqrt(2l+1) e_0 and
          the CG contraction x (x) Y collapses to   t[m] = coef[m] * x[src(m)]   (|m| <= min(l_in, l_out)),
        - ``wigner_tables(lmax)``: constant matrices J^l and sign patterns so that the device can build
          D^l(R_e) = J Z(beta) J^T Z(alpha)  from (cos, sin) of two angles with Chebyshev recurrences only.
"""
from __future__ import annotations

import math
from fractions import Fraction
from functools import lru_cache
from typing import List, Sequence, Tuple

import numpy as np

# ------------------------------------------------------------------------------------------------ irreps


class Irreps:
    """Minimal ordered list of (mul, l, p); string grammar as e3nn ("64x0e+32x1o", "0e + 1o")."""

    def __init__(self, spec=None):
        items: List[Tuple[int, int, int]] = []
        if spec is None:
            pass
        elif isinstance(spec, Irreps):
            items = list(spec.items)
        elif isinstance(spec, str):
            for tok in spec.split("+"):
                tok = tok.strip()
                if not tok:
                    continue
                mul, ir = tok.split("x") if "x" in tok else ("1", tok)
                ir = ir.strip()
                items.append((int(mul), int(ir[:-1]), {"e": 1, "o": -1}[ir[-1]]))
        else:
            for it in spec:
                if len(it) == 3:
                    items.append((int(it[0]), int(it[1]), int(it[2])))
                else:
                    mul, ir = it
                    if isinstance(ir, str):
                        items.append((int(mul), int(ir[:-1]), {"e": 1, "o": -1}[ir[-1]]))
                    else:
                        items.append((int(mul), int(ir[0]), int(ir[1])))
        self.items = items

    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def __eq__(self, other):
        return isinstance(other, Irreps) and self.items == other.items

    def __add__(self, other):
        return Irreps(self.items + Irreps(other).items)

    @property
    def dim(self):
        return sum(m * (2 * l + 1) for m, l, _ in self.items)

    @property
    def num_irreps(self):
        return sum(m for m, _, _ in self.items)

    @property
    def lmax(self):
        return max(l for _, l, _ in self.items)

    def offsets(self):
        o, out = 0, []
        for m, l, _ in self.items:
            out.append(o)
            o += m * (2 * l + 1)
        return out

    def simplify(self):
        out = []
        for m, l, p in self.items:
            if m == 0:
                continue
            if out and out[-1][1:] == (l, p):
                out[-1] = (out[-1][0] + m, l, p)
            else:
                out.append((m, l, p))
        return Irreps(out)

    def sort(self):
        """Stable sort by (l, p) with p=-1 before p=+1 (plain tuple order).  Returns (irreps, perm) with perm[old]=new."""
        order = sorted(range(len(self.items)), key=lambda i: ((self.items[i][1], self.items[i][2]), i))
        perm = [0] * len(order)
        for new, old in enumerate(order):
            perm[old] = new
        return Irreps([self.items[i] for i in order]), perm

    def __repr__(self):
        return "+".join(f"{m}x{l}{'e' if p == 1 else 'o'}" for m, l, p in self.items)


# ------------------------------------------------------------------------------------------------ wigner 3j


def _fact(n):
    return math.factorial(n)


def _cg(j1, m1, j2, m2, j3, m3):
    if m3 != m1 + m2:
        return 0.0
    vmin = max(-j1 + j2 + m3, -j1 + m1, 0)
    vmax = min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3)
    pref = Fraction((2 * j3 + 1) * _fact(j3 + j1 - j2) * _fact(j3 - j1 + j2) * _fact(j1 + j2 - j3) * _fact(j3 + m3) * _fact(j3 - m3),
                    _fact(j1 + j2 + j3 + 1) * _fact(j1 - m1) * _fact(j1 + m1) * _fact(j2 - m2) * _fact(j2 + m2))
    tot = Fraction(0)
    for v in range(vmin, vmax + 1):
        tot += Fraction((-1) ** (v + j2 + m2) * _fact(j2 + j3 + m1 - v) * _fact(j1 - m1 + v),
                        _fact(v) * _fact(j3 - j1 + j2 - v) * _fact(j3 + m3 - v) * _fact(v + j1 - j2 - m3))
    return math.sqrt(pref) * float(tot)


def _real2complex(l):
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    r = 1 / math.sqrt(2)
    for m in range(-l, 0):
        q[l + m, l - m] = r
        q[l + m, l + m] = -1j * r
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + m] = (-1) ** m * r
        q[l + m, l - m] = 1j * (-1) ** m * r
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def wigner_3j(l1, l2, l3) -> np.ndarray:
    """Real [2l1+1,2l2+1,2l3+1] tensor, Frobenius norm 1, e3nn basis (m=-l..l, y polar axis) and phase convention."""
    assert abs(l1 - l2) <= l3 <= l1 + l2
    C = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            if abs(m1 + m2) <= l3:
                C[l1 + m1, l2 + m2, l3 + m1 + m2] = _cg(l1, m1, l2, m2, l3, m1 + m2)
    Q1, Q2, Q3 = _real2complex(l1), _real2complex(l2), _real2complex(l3)
    R = np.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, np.conj(Q3.T), C.astype(np.complex128))
    assert np.abs(R.imag).max() < 1e-9
    R = R.real
    R = R / np.linalg.norm(R)
    R.setflags(write=False)
    return R


# ------------------------------------------------------------------------------------------------ spherical harmonics


@lru_cache(maxsize=None)
def _sh_k(l):
    Yl = sph_harm(l, np.array([[0.0, 1.0, 0.0]]))[0]
    T = np.einsum("ijk,j,k->i", wigner_3j(l + 1, 1, l), np.array([0.0, math.sqrt(3.0), 0.0]), Yl)
    return math.sqrt(2 * l + 3) / T[l + 1]


def sph_harm(l, u: np.ndarray) -> np.ndarray:
    """Y^l of unit vectors u[..., 3] in e3nn axis order, 'component' normalisation (|Y^l|^2 = 2l+1)."""
    if l == 0:
        return np.ones(u.shape[:-1] + (1,))
    if l == 1:
        return math.sqrt(3.0) * u
    return _sh_k(l - 1) * np.einsum("ijk,...j,...k->...i", wigner_3j(l, 1, l - 1), math.sqrt(3.0) * u, sph_harm(l - 1, u))


def wigner_D(l, R: np.ndarray) -> np.ndarray:
    """D^l(R) with Y^l(R u) = D^l(R) Y^l(u)  (least-squares fit on sample directions; exact to ~1e-14)."""
    rng = np.random.default_rng(777 + l)
    u = rng.normal(size=(6 * (2 * l + 1) + 8, 3))
    u /= np.linalg.norm(u, axis=-1, keepdims=True)
    A, B = sph_harm(l, u), sph_harm(l, u @ R.T)
    return np.linalg.lstsq(A, B, rcond=None)[0].T


# ------------------------------------------------------------------------------------------------ aligned frame


@lru_cache(maxsize=None)
def aligned_path(l_in, l_sh, l_out):
    """In the frame where the edge direction is the pole:  t[m_out] = sum_a K[a, m_out] x[a],
    K = sqrt(2 l_sh+1) * w3j(l_in, l_sh, l_out)[:, l_sh(center), :].
    Returns (src, coef): for every output component index c (0..2 l_out) the single input component index
    src[c] (or -1) and the coefficient coef[c].  Asserts the at-most-one-nonzero-per-column structure."""
    K = math.sqrt(2 * l_sh + 1) * wigner_3j(l_in, l_sh, l_out)[:, l_sh, :]
    src = np.full(2 * l_out + 1, -1, dtype=np.int64)
    coef = np.zeros(2 * l_out + 1)
    for c in range(2 * l_out + 1):
        nz = np.nonzero(np.abs(K[:, c]) > 1e-12)[0]
        assert len(nz) <= 1, (l_in, l_sh, l_out, c, nz)
        if len(nz):
            a = int(nz[0])
            m_out, m_in = c - l_out, a - l_in
            assert abs(m_in) == abs(m_out)
            src[c], coef[c] = a, K[a, c]
    return src, coef


def _rot_y(t):  # rotation about the e3nn polar axis (y)
    c, s = math.cos(t), math.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rot_x(t):
    c, s = math.cos(t), math.sin(t)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


@lru_cache(maxsize=None)
def wigner_tables(l):
    """Constants for building D^l of  R(alpha,beta) = Rx(beta) Ry(alpha)  on the device:
         D^l(Ry(t))[l+m, l+m] = D[l-m,l-m] = cos(m t),  D[l+m, l-m] = sgn * sin(m t),  D[l-m, l+m] = -sgn * sin(m t)
         D^l(Rx(t)) = J D^l(Ry(t)) J^T           with J = D^l(Rz(-90deg)) (maps the y axis onto the x axis)
    Returns (J [2l+1,2l+1], sgn [l+1])."""
    t = 0.37
    D = wigner_D(l, _rot_y(t))
    sgn = np.ones(l + 1)
    for m in range(1, l + 1):
        assert abs(D[l + m, l + m] - math.cos(m * t)) < 1e-10 and abs(D[l - m, l - m] - math.cos(m * t)) < 1e-10
        s = D[l + m, l - m] / math.sin(m * t)
        assert abs(abs(s) - 1) < 1e-10 and abs(D[l - m, l + m] + s * math.sin(m * t)) < 1e-10
        sgn[m] = round(s)
    Z = np.zeros_like(D)
    for m in range(0, l + 1):
        Z[l + m, l + m] = Z[l - m, l - m] = math.cos(m * t)
        if m:
            Z[l + m, l - m], Z[l - m, l + m] = sgn[m] * math.sin(m * t), -sgn[m] * math.sin(m * t)
    assert np.abs(Z - D).max() < 1e-10
    Rz = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])   # Rz(-90): y -> x
    J = wigner_D(l, Rz)
    assert np.abs(J @ Z @ J.T - wigner_D(l, _rot_x(t))).max() < 1e-9
    return J, sgn


def z_matrix(l, c1, s1, sgn):
    """D^l(Ry(t)) from cos t, sin t via the Chebyshev recurrence (mirrors the device code)."""
    Z = np.zeros((2 * l + 1, 2 * l + 1))
    Z[l, l] = 1
    cm, sm = 1.0, 0.0
    for m in range(1, l + 1):
        cm, sm = cm * c1 - sm * s1, sm * c1 + cm * s1
        Z[l + m, l + m] = Z[l - m, l - m] = cm
        Z[l + m, l - m], Z[l - m, l + m] = sgn[m] * sm, -sgn[m] * sm
    return Z


def edge_frame_angles(n: np.ndarray):
    """n = unit edge direction in e3nn axis order (= physical (y, z, x)).  R = Rx(beta) Ry(alpha) maps n onto the pole:
    Ry(alpha) zeroes the x component, Rx(beta) then tilts onto +y.  Returns (cos a, sin a, cos b, sin b)."""
    x, y, z = n[..., 0], n[..., 1], n[..., 2]
    rho = np.sqrt(x * x + z * z)
    safe = rho > 1e-12
    ca = np.where(safe, z / np.where(safe, rho, 1), 1.0)
    sa = np.where(safe, -x / np.where(safe, rho, 1), 0.0)
    # after Ry(alpha): (0, y, rho);  Rx(beta): y' = c y - s z = 1, z' = s y + c z = 0  -> c = y, s = -rho
    return ca, sa, y, -rho


def edge_wigner(l, n: np.ndarray) -> np.ndarray:
    """D^l(R_e) for one unit vector (reference implementation of the device routine)."""
    J, sgn = wigner_tables(l)
    ca, sa, cb, sb = (float(v) for v in edge_frame_angles(n))
    return J @ z_matrix(l, cb, sb, sgn) @ J.T @ z_matrix(l, ca, sa, sgn)
