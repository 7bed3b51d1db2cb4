"""oracle/e3.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).  Never imported by the product path.

An e3nn-free restatement, in plain torch-CPU, of the subset of ``e3nn==0.5.0`` that the HamGNN hot path
calls (pin: /root/reference/HamGNN.yaml:40; e3nn itself is NOT vendored in /root/reference and is not
installed in this image, so this file restates its *published algorithms*):

  o3.Irrep / o3.Irreps (parse, sort, simplify, slices)      call sites: hamgnn/nn/message_passing.py:53-57,158-171
  o3.wigner_3j                                                call site : hamgnn/physics/Clebsch_Gordan_coefficients.py:27
  o3.SphericalHarmonics (normalize=True, 'component')         call site : hamgnn/toolbox/nequip/nn/embedding/_edge.py:55-57
  o3.TensorProduct (uvw / uvu / uuu, component+element norm)  call sites: hamgnn/nn/message_passing.py:81-96, tensor_products.py:31-38
  o3.Linear                                                   call sites: hamgnn/nn/convolution.py:112-114, interaction_blocks.py:306-309
  nn.Gate / nn.Activation / normalize2mom                     call site : hamgnn/nn/interaction_blocks.py:317-323
  nn.FullyConnectedNet                                        call site : hamgnn/nn/message_passing.py:186-189

PARITY STATUS: "parity unpinned" w.r.t. e3nn itself -- the reference ships no tests / golden vectors and e3nn
cannot be run here.  This restatement is pinned instead by (a) known-answer / invariant tests in
tests/test_oracle_e3.py (w3j orthogonality + selection rules + known values, SH norms / explicit polynomials /
scipy cross-check, equivariance under random rotations) and (b) wiring fixtures produced by running the
reference's *own* hamgnn/nn modules on top of this file (oracle/gen_golden.py -> tests/golden/).

The module layout mimics e3nn's (``e3.o3.X`` / ``e3.nn.X`` / ``e3.util.jit.compile_mode``) so that the
reference's hamgnn/nn files can be imported against it by oracle/gen_golden.py (this container only).
"""
from __future__ import annotations

import collections
import math
import types
from fractions import Fraction
from functools import lru_cache
from typing import List, Optional, Tuple

import torch

# --------------------------------------------------------------------------------------------------------------
# Irrep / Irreps
# --------------------------------------------------------------------------------------------------------------


class Irrep(tuple):
    """(l, p) with p in {+1 (e), -1 (o)}.  Plain tuple ordering => 0o < 0e < 1o < 1e < ... [e3nn-recall]."""

    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                s = l.strip()
                ll = int(s[:-1])
                pp = {"e": 1, "o": -1, "y": (-1) ** ll}[s[-1]]
                return super().__new__(cls, (ll, pp))
            if isinstance(l, tuple):
                l, p = l
        assert isinstance(l, int) and l >= 0 and p in (-1, 1), (l, p)
        return super().__new__(cls, (l, p))

    @property
    def l(self):
        return self[0]

    @property
    def p(self):
        return self[1]

    @property
    def dim(self):
        return 2 * self[0] + 1

    def __repr__(self):
        return f"{self.l}{'e' if self.p == 1 else 'o'}"

    def __mul__(self, other):
        other = Irrep(other)
        p = self.p * other.p
        return [Irrep(l, p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]

    def __rmul__(self, mul):
        assert isinstance(mul, int)
        return Irreps([(mul, self)])

    def __add__(self, other):
        return Irreps(self) + Irreps(other)

    def is_scalar(self):
        return self.l == 0 and self.p == 1


class _MulIr(tuple):
    def __new__(cls, mul, ir=None):
        if ir is None:
            mul, ir = mul
        return super().__new__(cls, (int(mul), Irrep(ir)))

    @property
    def mul(self):
        return self[0]

    @property
    def ir(self):
        return self[1]

    @property
    def dim(self):
        return self[0] * self[1].dim

    def __repr__(self):
        return f"{self.mul}x{self.ir}"


class Irreps(tuple):
    def __new__(cls, irreps=None):
        if isinstance(irreps, Irreps):
            return super().__new__(cls, irreps)
        out = []
        if irreps is None:
            pass
        elif isinstance(irreps, Irrep):
            out.append(_MulIr(1, irreps))
        elif isinstance(irreps, str):
            if irreps.strip() != "":
                for tok in irreps.split("+"):
                    tok = tok.strip()
                    if "x" in tok:
                        mul, ir = tok.split("x")
                        out.append(_MulIr(int(mul), Irrep(ir.strip())))
                    else:
                        out.append(_MulIr(1, Irrep(tok)))
        else:
            for item in irreps:
                if isinstance(item, _MulIr):
                    out.append(item)
                elif isinstance(item, Irrep):
                    out.append(_MulIr(1, item))
                elif isinstance(item, str):
                    out.append(_MulIr(1, Irrep(item)))
                else:
                    mul, ir = item
                    out.append(_MulIr(mul, ir))
        return super().__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, (l, p ** l)) for l in range(lmax + 1)])

    @property
    def dim(self):
        return sum(mi.dim for mi in self)

    @property
    def num_irreps(self):
        return sum(mi.mul for mi in self)

    @property
    def ls(self):
        return [mi.ir.l for mi in self for _ in range(mi.mul)]

    @property
    def lmax(self):
        if len(self) == 0:
            raise ValueError("empty irreps")
        return max(mi.ir.l for mi in self)

    def slices(self):
        s, i = [], 0
        for mi in self:
            s.append(slice(i, i + mi.dim))
            i += mi.dim
        return s

    def __getitem__(self, i):
        x = super().__getitem__(i)
        if isinstance(i, slice):
            return Irreps(x)
        return x

    def __contains__(self, ir):
        ir = Irrep(ir)
        return ir in (mi.ir for mi in self)

    def count(self, ir):
        ir = Irrep(ir)
        return sum(mi.mul for mi in self if mi.ir == ir)

    def __add__(self, other):
        return Irreps(tuple(self) + tuple(Irreps(other)))

    def __radd__(self, other):
        return Irreps(other) + self

    def __mul__(self, n):
        assert isinstance(n, int)
        return Irreps(tuple(self) * n)

    def __rmul__(self, n):
        assert isinstance(n, int)
        return Irreps(tuple(self) * n)

    def simplify(self):
        """Merge ADJACENT equal irreps only (e3nn semantics); drop mul==0."""
        out = []
        for mi in self:
            if mi.mul == 0:
                continue
            if out and out[-1][1] == mi.ir:
                out[-1] = (out[-1][0] + mi.mul, mi.ir)
            else:
                out.append((mi.mul, mi.ir))
        return Irreps(out)

    def remove_zero_multiplicities(self):
        return Irreps([(m, ir) for m, ir in self if m > 0])

    def sort(self):
        """Stable sort by (ir, original index).  Returns namedtuple(irreps, p, inv) with p[i_old] = i_new."""
        Ret = collections.namedtuple("sort", ["irreps", "p", "inv"])
        out = sorted([(mi.ir, i, mi.mul) for i, mi in enumerate(self)])
        inv = tuple(i for _, i, _ in out)
        p = [0] * len(inv)
        for new, old in enumerate(inv):
            p[old] = new
        return Ret(Irreps([(mul, ir) for ir, _, mul in out]), tuple(p), inv)

    def __repr__(self):
        return "+".join(repr(mi) for mi in self)

    def randn(self, *size, dtype=None, generator=None):
        size = [s if s != -1 else self.dim for s in size]
        return torch.randn(*size, dtype=dtype, generator=generator)


# --------------------------------------------------------------------------------------------------------------
# wigner_3j  
# --------------------------------------------------------------------------------------------------------------


def _f(n):
    return math.factorial(n)


def _su2_cg_coeff(j1, m1, j2, m2, j3, m3) -> float:
    """<j1 m1 j2 m2 | j3 m3>, standard Racah sum with exact rationals under the square root."""
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max(-j1 + j2 + m3, -j1 + m1, 0))
    vmax = int(min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3))
    C = Fraction((2 * j3 + 1) * _f(j3 + j1 - j2) * _f(j3 - j1 + j2) * _f(j1 + j2 - j3) * _f(j3 + m3) * _f(j3 - m3),
                 _f(j1 + j2 + j3 + 1) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2))
    S = Fraction(0)
    for v in range(vmin, vmax + 1):
        S += Fraction((-1) ** (v + j2 + m2) * _f(j2 + j3 + m1 - v) * _f(j1 - m1 + v),
                      _f(v) * _f(j3 - j1 + j2 - v) * _f(j3 + m3 - v) * _f(v + j1 - j2 - m3))
    return math.sqrt(float(C)) * float(S)


def _su2_cg(j1, j2, j3) -> torch.Tensor:
    mat = torch.zeros(2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1, dtype=torch.float64)
    if abs(j1 - j2) <= j3 <= j1 + j2:
        for m1 in range(-j1, j1 + 1):
            for m2 in range(-j2, j2 + 1):
                if abs(m1 + m2) <= j3:
                    mat[j1 + m1, j2 + m2, j3 + m1 + m2] = _su2_cg_coeff(j1, m1, j2, m2, j3, m1 + m2)
    return mat


def _real_to_complex(l) -> torch.Tensor:
    """Change of basis real -> complex spherical harmonics, e3nn phase convention ((-i)^l prefactor)."""
    q = torch.zeros(2 * l + 1, 2 * l + 1, dtype=torch.complex128)
    s2 = 1 / math.sqrt(2)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = s2
        q[l + m, l - abs(m)] = -1j * s2
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m * s2
        q[l + m, l - abs(m)] = 1j * (-1) ** m * s2
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def _wigner_3j_f64(l1, l2, l3) -> torch.Tensor:
    assert abs(l1 - l2) <= l3 <= l1 + l2
    Q1, Q2, Q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    C = _su2_cg(l1, l2, l3).to(torch.complex128)
    C = torch.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, torch.conj(Q3.T), C)
    assert C.imag.abs().max() < 1e-9
    C = C.real
    return C / C.norm()


def wigner_3j(l1, l2, l3, dtype=None, device=None) -> torch.Tensor:
    """Real 3j tensor [2l1+1, 2l2+1, 2l3+1], Frobenius norm 1 (e3nn.o3.wigner_3j)."""
    return _wigner_3j_f64(l1, l2, l3).to(dtype=dtype or torch.get_default_dtype(), device=device).clone()


# --------------------------------------------------------------------------------------------------------------
# Spherical harmonics: CG recursion, 'component' normalisation => ||Y^l||^2 = 2l+1
# --------------------------------------------------------------------------------------------------------------


@lru_cache(maxsize=None)
def _sh_recursion_consts(l):
    """constant k_l s.t.  Y^{l+1} = k_l * einsum(w3j(l+1,1,l), Y^1, Y^l)   (component normalisation, pole m=0 > 0)."""
    # evaluate at the pole (0,1,0) in e3nn coordinates in float64
    Y1 = torch.tensor([0.0, math.sqrt(3.0), 0.0], dtype=torch.float64)
    Yl = _sh_f64(l, torch.tensor([[0.0, 1.0, 0.0]], dtype=torch.float64))[0]
    T = torch.einsum("ijk,j,k->i", _wigner_3j_f64(l + 1, 1, l), Y1, Yl)
    assert T[l + 1].abs() > 1e-12
    k = math.sqrt(2 * (l + 1) + 1) / T[l + 1].item()  # sign fixes m=0 component positive at the pole
    return k


def _sh_f64(l, v):
    """Y^l of UNIT vectors v[...,3] (e3nn axis order), float64, component normalisation."""
    if l == 0:
        return torch.ones(v.shape[:-1] + (1,), dtype=v.dtype)
    Y1 = math.sqrt(3.0) * v
    if l == 1:
        return Y1
    Yp = _sh_f64(l - 1, v)
    k = _sh_recursion_consts(l - 1)
    return k * torch.einsum("ijk,...j,...k->...i", _wigner_3j_f64(l, 1, l - 1).to(v.dtype), Y1, Yp)


def spherical_harmonics(ls, x, normalize=True, normalization="component"):
    ls = [ls] if isinstance(ls, int) else list(ls)
    dt = x.dtype
    v = x.to(torch.float64) if dt != torch.float64 else x
    if normalize:
        v = torch.nn.functional.normalize(v, dim=-1)
        outs = [_sh_f64(l, v) for l in ls]
    else:
        r = v.norm(dim=-1, keepdim=True)
        u = v / r.clamp_min(1e-300)
        outs = [_sh_f64(l, u) * r ** l for l in ls]
    if normalization == "integral":
        outs = [o / math.sqrt(4 * math.pi) for o in outs]
    elif normalization == "norm":
        outs = [o / math.sqrt(2 * l + 1) for o, l in zip(outs, ls)]
    return torch.cat(outs, dim=-1).to(dt)


class SphericalHarmonics(torch.nn.Module):
    """NOTE: evaluated in float64 internally and rounded to the input dtype (the oracle is the accuracy anchor)."""

    def __init__(self, irreps_out, normalize, normalization="integral", irreps_in=None):
        super().__init__()
        self.irreps_out = Irreps(irreps_out) if not isinstance(irreps_out, int) else Irreps.spherical_harmonics(irreps_out)
        for mul, ir in self.irreps_out:
            assert ir.p == (-1) ** ir.l, "SH irreps must have parity (-1)^l"
        self.ls = self.irreps_out.ls
        self.normalize = normalize
        self.normalization = normalization

    def forward(self, x):
        return spherical_harmonics(self.ls, x, self.normalize, self.normalization)


# --------------------------------------------------------------------------------------------------------------
# TensorProduct
# --------------------------------------------------------------------------------------------------------------

Instruction = collections.namedtuple("Instruction", "i_in1 i_in2 i_out connection_mode has_weight path_weight path_shape")


CONTRACTION = "outer"      # "optimized": the contraction order opt_einsum_fx would pick for the uvw paths (TensorProduct.forward); timing only, see there


class TensorProduct(torch.nn.Module):
    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, in1_var=None, in2_var=None, out_var=None,
                 irrep_normalization="component", path_normalization="element", internal_weights=None,
                 shared_weights=None):
        super().__init__()
        self.irreps_in1 = Irreps(irreps_in1)
        self.irreps_in2 = Irreps(irreps_in2)
        self.irreps_out = Irreps(irreps_out)
        assert irrep_normalization == "component" and path_normalization == "element"
        ins = []
        for t in instructions:
            t = tuple(t)
            if len(t) == 5:
                t = t + (1.0,)
            i1, i2, io, mode, hw, pw = t
            m1, m2, mo = self.irreps_in1[i1].mul, self.irreps_in2[i2].mul, self.irreps_out[io].mul
            shape = {"uvw": (m1, m2, mo), "uvu": (m1, m2), "uvv": (m1, m2), "uuw": (m1, mo), "uuu": (m1,), "uvuv": (m1, m2)}[mode]
            ins.append(Instruction(i1, i2, io, mode, hw, pw, shape))

        def num_elements(i):
            m1, m2 = self.irreps_in1[i.i_in1].mul, self.irreps_in2[i.i_in2].mul
            return {"uvw": m1 * m2, "uvu": m2, "uvv": m1, "uuw": m1, "uuu": 1, "uvuv": 1}[i.connection_mode]

        normed = []
        for i in ins:
            alpha = self.irreps_out[i.i_out].ir.dim
            x = sum(num_elements(j) for j in ins if j.i_out == i.i_out)
            if x > 0:
                alpha /= x
            alpha *= i.path_weight
            normed.append(i._replace(path_weight=math.sqrt(alpha)))
        self.instructions = normed

        if shared_weights is False and internal_weights is None:
            internal_weights = False
        if shared_weights is None:
            shared_weights = True
        if internal_weights is None:
            internal_weights = shared_weights and any(i.has_weight for i in self.instructions)
        assert shared_weights or not internal_weights
        self.internal_weights = internal_weights
        self.shared_weights = shared_weights
        self.weight_numel = sum(math.prod(i.path_shape) for i in self.instructions if i.has_weight)
        if internal_weights and self.weight_numel > 0:
            self.weight = torch.nn.Parameter(torch.randn(self.weight_numel))
        else:
            self.register_buffer("weight", torch.Tensor())
        self._s1, self._s2, self._so = self.irreps_in1.slices(), self.irreps_in2.slices(), self.irreps_out.slices()

    def forward(self, x, y, weight=None):
        if weight is None:
            assert self.internal_weights or self.weight_numel == 0
            weight = self.weight
        batch_w = weight.dim() > 1 and not self.shared_weights
        Z = x.shape[0]
        outs = [None] * len(self.irreps_out)
        off = 0
        for ins in self.instructions:
            m1, ir1 = self.irreps_in1[ins.i_in1]
            m2, ir2 = self.irreps_in2[ins.i_in2]
            mo, iro = self.irreps_out[ins.i_out]
            x1 = x[:, self._s1[ins.i_in1]].reshape(Z, m1, ir1.dim)
            x2 = y[:, self._s2[ins.i_in2]].reshape(Z, m2, ir2.dim)
            C = wigner_3j(ir1.l, ir2.l, iro.l, dtype=x.dtype)
            w = None
            if ins.has_weight:
                n = math.prod(ins.path_shape)
                w = weight[..., off:off + n].reshape(((Z,) if batch_w else ()) + ins.path_shape)
                off += n
            zw = "z" if batch_w else ""
            mode = ins.connection_mode
            if CONTRACTION == "optimized" and mode == "uvw" and m2 == 1 and w is not None and not batch_w:
                # TIMING form (bench.py's cpu_baseline leg only; the parity oracle keeps the naive order below): e3nn 0.5.0 hands its generated einsums to
                # opt_einsum_fx, which reorders `uvw,ijk,zuvij->zwk` for the example shapes -- with one copy of the second operand (the spherical harmonics: v = 1)
                # and shared weights the cheap order is weights first, 3j tensor with the harmonics second, and no [Z, u, v, i, j] outer product is ever formed
                t = torch.einsum("zui,uw->zwi", x1, w[:, 0, :])
                cy = torch.einsum("zj,ijk->zik", x2[:, 0, :], C)
                r = torch.bmm(t, cy)
                r = ins.path_weight * r.flatten(1)
                outs[ins.i_out] = r if outs[ins.i_out] is None else outs[ins.i_out] + r
                continue
            # contraction order of e3nn's generated code: outer product of the inputs, then the 3j tensor, then the weights
            if mode == "uvw":
                xx = torch.einsum("zui,zvj->zuvij", x1, x2)
                r = torch.einsum("zuvij,ijk->zuvk", xx, C)
                r = torch.einsum(f"zuvk,{zw}uvw->zwk", r, w)
            elif mode == "uvu":
                xx = torch.einsum("zui,zvj->zuvij", x1, x2)
                r = torch.einsum("zuvij,ijk->zuvk", xx, C)
                r = torch.einsum(f"zuvk,{zw}uv->zuk", r, w) if w is not None else r.sum(2)
            elif mode == "uuu":
                xx = torch.einsum("zui,zuj->zuij", x1, x2)
                r = torch.einsum("zuij,ijk->zuk", xx, C)
                if w is not None:
                    r = r * (w[:, :, None] if batch_w else w[None, :, None])
            else:
                raise NotImplementedError(mode)
            r = ins.path_weight * r.flatten(1)                  # (flatten, not reshape(Z, -1): Z may be 0)
            outs[ins.i_out] = r if outs[ins.i_out] is None else outs[ins.i_out] + r
        res = []
        for i, (mul, ir) in enumerate(self.irreps_out):
            res.append(outs[i] if outs[i] is not None else x.new_zeros(Z, mul * ir.dim))
        return torch.cat(res, dim=-1) if res else x.new_zeros(Z, 0)


class ElementwiseTensorProduct(TensorProduct):
    def __init__(self, irreps_in1, irreps_in2):
        irreps_in1, irreps_in2 = Irreps(irreps_in1).simplify(), Irreps(irreps_in2).simplify()
        assert irreps_in1.num_irreps == irreps_in2.num_irreps
        a, b = list(irreps_in1), list(irreps_in2)
        i = 0
        while i < len(a):  # align multiplicities
            m1, ir1 = a[i]
            m2, ir2 = b[i]
            if m1 < m2:
                b[i] = _MulIr(m1, ir2)
                b.insert(i + 1, _MulIr(m2 - m1, ir2))
            if m2 < m1:
                a[i] = _MulIr(m2, ir1)
                a.insert(i + 1, _MulIr(m1 - m2, ir1))
            i += 1
        out, instr = [], []
        for i, ((mul, ir1), (mul2, ir2)) in enumerate(zip(a, b)):
            assert mul == mul2
            for ir in ir1 * ir2:
                instr.append((i, i, len(out), "uuu", False))
                out.append((mul, ir))
        super().__init__(Irreps(a), Irreps(b), Irreps(out), instr)


# --------------------------------------------------------------------------------------------------------------
# Linear
# --------------------------------------------------------------------------------------------------------------


class Linear(torch.nn.Module):
    def __init__(self, irreps_in, irreps_out, internal_weights=None, shared_weights=None, **_):
        super().__init__()
        self.irreps_in = Irreps(irreps_in)
        self.irreps_out = Irreps(irreps_out)
        self.paths = [(i, o) for i, (_, iri) in enumerate(self.irreps_in) for o, (_, iro) in enumerate(self.irreps_out) if iri == iro]
        self.fan_in = {o: sum(self.irreps_in[i].mul for i, oo in self.paths if oo == o) for _, o in self.paths}
        self.weight_numel = sum(self.irreps_in[i].mul * self.irreps_out[o].mul for i, o in self.paths)
        self.weight = torch.nn.Parameter(torch.randn(self.weight_numel))
        self._si, self._so = self.irreps_in.slices(), self.irreps_out.slices()

    def forward(self, x, weight=None):
        w = self.weight if weight is None else weight
        Z = x.shape[0]
        outs = [None] * len(self.irreps_out)
        off = 0
        for i, o in self.paths:
            mi, ir = self.irreps_in[i]
            mo, _ = self.irreps_out[o]
            W = w[off:off + mi * mo].reshape(mi, mo)
            off += mi * mo
            xi = x[:, self._si[i]].reshape(Z, mi, ir.dim)
            r = torch.einsum("uw,zui->zwi", W, xi).flatten(1) / math.sqrt(self.fan_in[o])
            outs[o] = r if outs[o] is None else outs[o] + r
        res = [outs[o] if outs[o] is not None else x.new_zeros(Z, mul * ir.dim) for o, (mul, ir) in enumerate(self.irreps_out)]
        return torch.cat(res, dim=-1)


# --------------------------------------------------------------------------------------------------------------
# normalize2mom / Activation / Gate / FullyConnectedNet (SURVEY 8c-C,F,G)
# --------------------------------------------------------------------------------------------------------------

_N2M_CACHE = {}


def normalize2mom_const(f) -> float:
    """E_{z~N(0,1)}[f(z)^2]^{-1/2} by e3nn's Monte-Carlo recipe (1e6 samples, CPU generator seed 0, float64)."""
    key = getattr(f, "__name__", repr(f))
    if key not in _N2M_CACHE:
        gen = torch.Generator(device="cpu").manual_seed(0)
        z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
        c = f(z).pow(2).mean().pow(-0.5).item()
        if abs(c - 1) < 1e-4:
            c = 1.0
        _N2M_CACHE[key] = c
    return _N2M_CACHE[key]


class normalize2mom(torch.nn.Module):
    def __init__(self, f):
        super().__init__()
        self.f = f
        self.cst = normalize2mom_const(f)

    def forward(self, x):
        return self.f(x) * self.cst if self.cst != 1.0 else self.f(x)


def _act_parity(act):
    x = torch.linspace(0, 10, 256)
    a1, a2 = act(x), act(-x)
    if (a1 - a2).abs().max() < 1e-5:
        return 1
    if (a1 + a2).abs().max() < 1e-5:
        return -1
    return 0


class Activation(torch.nn.Module):
    def __init__(self, irreps_in, acts):
        super().__init__()
        irreps_in = Irreps(irreps_in)
        assert len(irreps_in) == len(acts), (irreps_in, acts)
        acts = [normalize2mom(a) if a is not None else None for a in acts]
        out = []
        for (mul, (l, p_in)), act in zip(irreps_in, acts):
            if act is not None:
                assert l == 0
                p_act = _act_parity(act)
                p_out = p_act if p_in == -1 else p_in
                assert p_out != 0
                out.append((mul, (0, p_out)))
            else:
                out.append((mul, (l, p_in)))
        self.irreps_in, self.irreps_out = irreps_in, Irreps(out)
        self.acts = torch.nn.ModuleList([a if a is not None else torch.nn.Identity() for a in acts])
        self._has = [a is not None for a in acts]

    def forward(self, x):
        outs, i = [], 0
        for (mul, ir), act, has in zip(self.irreps_in, self.acts, self._has):
            blk = x[:, i:i + mul * ir.dim]
            outs.append(act(blk) if has else blk)
            i += mul * ir.dim
        return torch.cat(outs, dim=-1) if outs else x


class Extract(torch.nn.Module):
    def __init__(self, irreps_in, irreps_outs, instructions):
        super().__init__()
        self.irreps_in = Irreps(irreps_in)
        self.irreps_outs = tuple(Irreps(i) for i in irreps_outs)
        self.instructions = instructions
        self._s = self.irreps_in.slices()

    def forward(self, x):
        outs = []
        for ins in self.instructions:
            parts = [x[:, self._s[i]] for i in ins]
            outs.append(torch.cat(parts, dim=-1) if parts else x[:, :0])
        return tuple(outs)


class _Sortcut(torch.nn.Module):
    """e3nn.nn._gate._Sortcut: concatenation of the three groups, SORTED (stable) then simplified [e3nn-recall]."""

    def __init__(self, *irreps_outs):
        super().__init__()
        self.irreps_outs = tuple(Irreps(i).simplify() for i in irreps_outs)
        irreps_in = sum(self.irreps_outs, Irreps([]))
        i, instructions = 0, []
        for io in self.irreps_outs:
            instructions.append(tuple(range(i, i + len(io))))
            i += len(io)
        irreps_in, p, _ = irreps_in.sort()
        instructions = [tuple(p[i] for i in x) for x in instructions]
        self.cut = Extract(irreps_in, self.irreps_outs, instructions)
        self.irreps_in = irreps_in.simplify()

    def forward(self, x):
        return self.cut(x)


class Gate(torch.nn.Module):
    def __init__(self, irreps_scalars, act_scalars, irreps_gates, act_gates, irreps_gated):
        super().__init__()
        irreps_scalars, irreps_gates, irreps_gated = Irreps(irreps_scalars), Irreps(irreps_gates), Irreps(irreps_gated)
        assert irreps_gates.num_irreps == irreps_gated.num_irreps
        self.sc = _Sortcut(irreps_scalars, irreps_gates, irreps_gated)
        self.irreps_scalars, self.irreps_gates, self.irreps_gated = self.sc.irreps_outs
        self._irreps_in = self.sc.irreps_in
        self.act_scalars = Activation(self.irreps_scalars, act_scalars)
        self.act_gates = Activation(self.irreps_gates, act_gates)
        self.mul = ElementwiseTensorProduct(self.irreps_gated, self.act_gates.irreps_out)
        self._irreps_out = self.act_scalars.irreps_out + self.mul.irreps_out

    @property
    def irreps_in(self):
        return self._irreps_in

    @property
    def irreps_out(self):
        return self._irreps_out

    def forward(self, features):
        scalars, gates, gated = self.sc(features)
        scalars = self.act_scalars(scalars)
        if gates.shape[-1]:
            gates = self.act_gates(gates)
            gated = self.mul(gated, gates)
            return torch.cat([scalars, gated], dim=-1)
        return scalars


class _FCLayer(torch.nn.Module):
    def __init__(self, h_in, h_out, act):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(h_in, h_out))
        self.act = act
        self.h_in = h_in

    def forward(self, x):
        x = x @ (self.weight / math.sqrt(self.h_in))
        return self.act(x) if self.act is not None else x


class FullyConnectedNet(torch.nn.Sequential):
    def __init__(self, hs, act=None, **_):
        super().__init__()
        self.hs = list(hs)
        act = normalize2mom(act) if act is not None else None
        n = len(hs) - 1
        for i, (h1, h2) in enumerate(zip(hs, hs[1:])):
            setattr(self, f"layer{i}", _FCLayer(h1, h2, act if i < n - 1 else None))


class NormActivation(torch.nn.Module):
    """e3nn.nn.NormActivation (e3nn 0.5.0, nn/_normact.py) [e3nn-recall]: every irrep COPY (channel) is scaled by f(|x|) / |x| (normalize=True), the
    norm clamped from below at epsilon: norms = o3.Norm(irreps, squared=True)(x); norms[norms < eps^2] = eps^2; norms = sqrt(norms);
    scalings = f(norms [+ bias]) / norms; out = ElementwiseTensorProduct(norms' irreps (0e per channel), irreps)(scalings, x) -- whose 0e x l -> l
    "uuu" path has coefficient sqrt(2 l + 1) * w3j(0, l, l) = 1 under e3nn's default component normalisation, i.e. a plain product.  The scalar
    nonlinearity is applied AS GIVEN (no normalize2mom wrapper, unlike Gate / Activation).  irreps_out = irreps_in."""

    def __init__(self, irreps_in, scalar_nonlinearity, normalize=True, epsilon=None, bias=False):
        super().__init__()
        self.irreps_in = Irreps(irreps_in)
        self.irreps_out = Irreps(irreps_in)
        if epsilon is None and normalize:
            epsilon = 1e-8
        elif epsilon is not None and not normalize:
            raise ValueError("epsilon and normalize = False don't make sense together")
        self._eps_squared = epsilon * epsilon if epsilon is not None else 0.0
        self.scalar_nonlinearity = scalar_nonlinearity
        self.normalize = normalize
        self.bias = bias
        if bias:
            self.biases = torch.nn.Parameter(torch.zeros(self.irreps_in.num_irreps))

    def forward(self, features):
        out = []
        ch = 0
        for (mul, ir), sl in zip(self.irreps_in, self.irreps_in.slices()):
            x = features[..., sl].reshape(*features.shape[:-1], mul, ir.dim)
            n2 = (x * x).sum(-1)
            if self._eps_squared > 0:
                n2 = torch.where(n2 < self._eps_squared, torch.full_like(n2, self._eps_squared), n2)
            n = n2.sqrt()
            arg = n + self.biases[ch:ch + mul] if self.bias else n
            ch += mul
            sc = self.scalar_nonlinearity(arg)
            if self.normalize:
                sc = sc / n
            out.append((x * sc[..., None]).reshape(*features.shape[:-1], mul * ir.dim))
        return torch.cat(out, dim=-1)


def compile_mode(mode):
    def deco(cls):
        return cls
    return deco


# --------------------------------------------------------------------------------------------------------------
# Wigner-D by fitting SH (used by the tests to check equivariance; independent of the product's own routine)
# --------------------------------------------------------------------------------------------------------------


def wigner_D_from_matrix(l, R, n=None) -> torch.Tensor:
    """D^l(R) with Y^l(R v) = D^l(R) Y^l(v); R is a 3x3 matrix in e3nn axis order.  Least squares in float64."""
    g = torch.Generator().manual_seed(1234 + l)
    v = torch.nn.functional.normalize(torch.randn(4 * (2 * l + 1) + 8, 3, generator=g, dtype=torch.float64), dim=-1)
    A = _sh_f64(l, v)                     # [n, 2l+1]
    B = _sh_f64(l, v @ R.to(torch.float64).T)
    # B = A @ D^T
    Dt = torch.linalg.lstsq(A, B).solution
    return Dt.T


def rand_rotation(generator=None) -> torch.Tensor:
    q = torch.randn(4, generator=generator, dtype=torch.float64)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    return torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=torch.float64)


def block_D(irreps, R, parity_flip=False) -> torch.Tensor:
    irreps = Irreps(irreps)
    blocks = []
    for mul, ir in irreps:
        D = wigner_D_from_matrix(ir.l, R)
        if parity_flip and ir.p == -1:
            D = -D
        blocks.append(torch.kron(torch.eye(mul, dtype=torch.float64), D))
    return torch.block_diag(*blocks)


# e3nn-style namespaces so that `from e3nn import o3`, `from e3nn.nn import Gate` ... can be aliased to this file.
o3 = types.SimpleNamespace(Irrep=Irrep, Irreps=Irreps, wigner_3j=wigner_3j, SphericalHarmonics=SphericalHarmonics,
                           spherical_harmonics=spherical_harmonics, TensorProduct=TensorProduct, Linear=Linear,
                           ElementwiseTensorProduct=ElementwiseTensorProduct)
nn = types.SimpleNamespace(Gate=Gate, Activation=Activation, FullyConnectedNet=FullyConnectedNet,
                           NormActivation=NormActivation, Extract=Extract)
