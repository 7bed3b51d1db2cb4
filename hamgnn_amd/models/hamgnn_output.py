"""HamGNNPlusPlusOut -- MI355X drop-in for the reference pair read-out head (hamgnn/models/hamgnn_output.py:96-123 ctor,
:2916-4021 forward).  Same constructor keywords, parameter names ({onsite,offsite}_hamiltonian_network.{residual_block,
linear_transform}, ..._ksi_network, ..._overlap_network) and result dict.  In scope this round: the non-SOC branch
(:3772-3799) incl. overlap networks, SOC/so3 (:3026-3144), SOC/su2 (:3146-3178; E3TensorDecomposition.get_H,
hamgnn/nn/tensor_decomposition.py:553-603), masks, symmetrisation, H0, per-crystal concatenation, sparsity ratio.
The k-space step `calculate_band_energy` is built for the spin-free and the spin-orbit branches (hamgnn_amd/kspace.py).
Out of scope (raise NotImplementedError): spin-constrained / collinear branches; return_forces is carried, as in the reference it computes nothing."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .. import basis as B
from .. import nn as hnn
from .. import ops
from .. import plan as P
from ..so3 import Irreps
from ..topo import get_topology, gget, ghas, gset


class HamGNNPlusPlusOut(nn.Module):
    def __init__(self, irreps_in_node=None, irreps_in_edge=None, nao_max=14, return_forces=False, create_graph=False,
                 ham_type="openmx", ham_only=False, symmetrize=True, include_triplet=False, calculate_band_energy=False,
                 num_k=8, k_path=None, band_num_control=None, soc_switch=True, nonlinearity_type="gate",
                 export_reciprocal_values=False, add_H0=False, soc_basis="so3", spin_constrained=False, use_learned_weight=True,
                 minMagneticMoment=0.5, collinear_spin=False, zero_point_shift=False, add_H_nonsoc=False,
                 get_nonzero_mask_tensor=False, calculate_sparsity=True):
        super().__init__()
        self.derivative, self.create_graph = return_forces, create_graph
        self.nao_max, self.ham_type, self.ham_only = nao_max, ham_type.lower(), ham_only
        self.symmetrize, self.soc_switch, self.add_H0 = symmetrize, soc_switch, add_H0
        self.soc_basis = soc_basis.lower()
        if soc_switch and self.ham_type != "openmx":
            self.soc_basis = "su2"                                           # hamgnn_output.py:151-153
        self.zero_point_shift, self.add_H_nonsoc = zero_point_shift, add_H_nonsoc
        self.calculate_sparsity = calculate_sparsity
        self.get_nonzero_mask_tensor = get_nonzero_mask_tensor
        self.calculate_band_energy, self.num_k, self.k_path, self.band_num_control = calculate_band_energy, num_k, k_path, band_num_control
        # return_forces / create_graph: stored as `derivative` / `create_graph` and read by nothing in the reference's head (hamgnn_output.py:127-128);
        # its Model only switches autograd on for `pos` (Model.py:103, 227, 285, 459-460) -- no force is computed anywhere: accepted, no effect
        assert nonlinearity_type in ("gate", "norm"), "Invalid nonlinearity_type. Choose either 'gate' or 'norm'."      # interaction_blocks.py:289-290
        self.nonlinearity_type = nt = nonlinearity_type                     # of every HamLayer's ResidualBlock (hamgnn_output.py:38-58, 847)
        for flag, name in ((spin_constrained, "spin_constrained"), (collinear_spin, "collinear_spin")):
            if flag:
                raise NotImplementedError(f"HamGNNPlusPlusOut({name}) is outside the MI355X hot-path scope of this round")
        self.export_reciprocal_values = export_reciprocal_values
        if export_reciprocal_values and soc_switch:
            raise NotImplementedError("HamGNNPlusPlusOut(export_reciprocal_values) with soc_switch: the reference exports H(k) / S(k) / dS(k) on its non-SOC branch only")
        if soc_switch and self.soc_basis not in ("so3", "su2"):
            raise NotImplementedError("Unsupported SOC basis")                  # hamgnn_output.py:3180-3181
        t = B.basis_table(self.ham_type, nao_max)
        self.num_valence = t["num_valence"]
        self.row = self.col = Irreps(t["row"])
        self.index_change, self.minus_index, self.basis_def = t["index_change"], t["minus_index"], t["basis_def"]
        self.hamiltonian_irreps = P.ham_irreps(self.row)
        self.node_layout = P.PlanarLayout(irreps_in_node)
        self.edge_layout = P.PlanarLayout(irreps_in_edge)
        self._compiled_for = None
        if soc_switch and self.soc_basis == "su2":
            # hamgnn_output.py:281-293 + :189-198: irreps_out = 2 * (required + required); get_H reads copies 0 (re) and 2 (im)
            half = P.su2_irreps(self.row)
            if half.lmax > 7 or Irreps(irreps_in_node).lmax > 6 or Irreps(irreps_in_edge).lmax > 6:
                raise NotImplementedError("su2 SOC head: features beyond l = 6 / couplings beyond l = 7 are not instantiated")
            self.hamiltonian_irreps_su2 = Irreps(list(half) * 2)
            keep = [c in (0, 2) for c in range(4) for _ in range(len(half))]
            self.onsite_hamiltonian_network = hnn.HamLayer(irreps_in_node, Irreps(list(half) * 4), keep, nonlinearity_type=nt)
            self.offsite_hamiltonian_network = hnn.HamLayer(irreps_in_edge, Irreps(list(half) * 4), keep, nonlinearity_type=nt)
            if not ham_only:
                self.onsite_overlap_network = hnn.HamLayer(irreps_in_node, self.hamiltonian_irreps, nonlinearity_type=nt)
                self.offsite_overlap_network = hnn.HamLayer(irreps_in_edge, self.hamiltonian_irreps, nonlinearity_type=nt)
            return
        self.onsite_hamiltonian_network = hnn.HamLayer(irreps_in_node, self.hamiltonian_irreps, nonlinearity_type=nt)
        self.offsite_hamiltonian_network = hnn.HamLayer(irreps_in_edge, self.hamiltonian_irreps, nonlinearity_type=nt)
        if soc_switch:
            ksi = Irreps([(nao_max ** 2, 0, 1)])
            self.onsite_ksi_network = hnn.HamLayer(irreps_in_node, ksi, nonlinearity_type=nt)
            self.offsite_ksi_network = hnn.HamLayer(irreps_in_edge, ksi, nonlinearity_type=nt)
        if not ham_only:
            self.onsite_overlap_network = hnn.HamLayer(irreps_in_node, self.hamiltonian_irreps, nonlinearity_type=nt)
            self.offsite_overlap_network = hnn.HamLayer(irreps_in_edge, self.hamiltonian_irreps, nonlinearity_type=nt)

    # ------------------------------------------------------------------------------------------------------------
    def compile(self, device):
        dev = torch.device(device)
        self._adj_tabs = None
        for m in self.children():
            if isinstance(m, hnn.HamLayer):
                m.compile(dev)
        su2 = self.soc_switch and self.soc_basis == "su2"
        if getattr(self, "_struct_for", None) == dev:          # everything below is STRUCTURAL (basis tables, CG maps, layouts): built once per device;
            self._compiled_for = dev                           # a recompile after an optimiser step only repacks the HamLayers' weights above
            return self
        net = self.onsite_overlap_network if (su2 and not self.ham_only) else self.onsite_hamiltonian_network
        if not su2 or not self.ham_only:
            st, ptr, idx, val = P.ham_merge_tables(self.row, self.nao_max, self.index_change, self.minus_index, net.girr, net.slot_pos)
            self._slot = torch.from_numpy(st).to(dev)
            self._cg = tuple(torch.from_numpy(a).to(dev) for a in (ptr, idx, val))
        if su2:
            net = self.onsite_hamiltonian_network
            st, ptr, idx, val = P.su2_merge_tables(self.row, self.nao_max, self.index_change, self.minus_index, net.girr, net.slot_pos)
            self._slot_su2 = torch.from_numpy(st).to(dev)
            self._cg_su2 = tuple(torch.from_numpy(a).to(dev) for a in (ptr, idx, val))
        mask = np.zeros((119, self.nao_max), dtype=np.float32)
        for Z, orb in self.basis_def.items():
            mask[Z, orb] = 1.0
        self._mask = torch.from_numpy(mask).to(dev)
        orank = np.full((119, self.nao_max), -1, dtype=np.int32)           # k-space step: rank of an orbital inside its element's valid set
        for Z, orb in self.basis_def.items():
            orank[Z, sorted(orb)] = np.arange(len(orb), dtype=np.int32)
        nval = np.zeros(119, dtype=np.float32)
        for Z, cnt in self.num_valence.items():
            nval[int(Z)] = cnt
        self._orank, self._num_valence = torch.from_numpy(orank).to(dev), torch.from_numpy(nval).to(dev)
        norb = np.full(256, self.nao_max, dtype=np.int64)
        defined = np.zeros(256, dtype=bool)
        for Z, orb in self.basis_def.items():
            norb[Z], defined[Z] = len(orb), True
        self._norb, self._defined = torch.from_numpy(norb).to(dev), torch.from_numpy(defined).to(dev)
        self._basis_sig = hash((self.nao_max, norb.tobytes(), defined.tobytes()))      # (key of per-graph memos that depend on the basis tables only)
        self._n_imap = torch.from_numpy(self.node_layout.index_map().astype(np.int32)).to(dev)
        self._e_imap = torch.from_numpy(self.edge_layout.index_map().astype(np.int32)).to(dev)
        self._rot_tab = torch.from_numpy(P.rotate_table(self.edge_layout)).to(dev)
        self._lmax = max(self.edge_layout.irreps.lmax, self.hamiltonian_irreps.lmax, P.su2_irreps(self.row).lmax if su2 else 0)
        self._jtab = torch.from_numpy(P.wigner_jtab(self._lmax)).to(dev)
        self._blk = torch.from_numpy(P.shell_block_table(self.row, self.nao_max)).to(dev)
        self._compiled_for = self._struct_for = dev
        return self

    # -- index preparation (integer plumbing; hamgnn_output.py:2874-2914, 2985-2990, 1187-1229, 2784-2872)
    def _validate(self, data):
        get_topology(data).check_basis(self._defined, self.basis_def)

    @staticmethod
    def _global_inverse(data):
        return get_topology(data).global_inverse(data)

    @staticmethod
    def _cat_by_crystal(data, on, off, edge_counts):
        if edge_counts is None or edge_counts.numel() <= 1:
            return torch.cat([on, off], 0)
        nn_, ne = get_topology(data).crystal_sizes(data)
        out = []
        for a, b in zip(torch.split(on, nn_), torch.split(off, ne)):
            out += [a, b]
        return torch.cat(out, 0)

    @staticmethod
    def _split_by_crystal(data, H, edge_counts):
        """inverse of _cat_by_crystal: rows in the result's per-crystal [on-site; off-site] order -> (all on-site rows, all off-site rows)"""
        if edge_counts is None or edge_counts.numel() <= 1:
            N = data.z.shape[0]
            return H[:N], H[N:]
        sizes = [v for pair in zip(*get_topology(data).crystal_sizes(data)) for v in pair]
        parts = torch.split(H, sizes)
        return torch.cat(parts[0::2], 0), torch.cat(parts[1::2], 0)

    def _soc_bands(self, data, on_r, on_i, off_r, off_i, dev):
        """calculate_band_energy of the spin-orbit branches (hamgnn_output.py:3629-3662): k-vectors (a path when k_path is given, else
        random), bands of the prediction and -- attached to the batch -- of the target blocks; with zero_point_shift the predicted bands
        are aligned by their mean (:3920-3922)"""
        if not self.calculate_band_energy:
            return None, None
        from .. import kspace
        f32c = lambda t: t.contiguous().float()
        data["k_vecs"] = kspace.make_k_vectors(self.k_path if self.k_path is not None else None, self.num_k, data.cell, data=data).to(dev)
        be, wf = kspace.band_energies_soc(self, on_r, on_i, off_r, off_i, data)
        with torch.no_grad():
            tb, tw = kspace.band_energies_soc(self, f32c(data.Hon), f32c(data.iHon), f32c(data.Hoff), f32c(data.iHoff), data)
        data["band_energy"], data["wavefunction"] = tb, tw
        if self.zero_point_shift:
            be = be - torch.mean(be - tb)
        return be, wf

    def _apply_zero_point_shift(self, data, H, edge_counts, soc):
        """hamgnn_output.py:3971-3981 / :3892-3913; targets as the reference prepares them (:2975-2978, :3617-3618)."""
        f32c = lambda t: t.contiguous().float()
        S = gget(data, "overlap") if ghas(data, "overlap") else self._cat_by_crystal(data, data.Son, data.Soff, edge_counts)
        if not soc and ghas(data, "hamiltonian"):
            Href = gget(data, "hamiltonian")
        else:
            Href = self._cat_by_crystal(data, data.Hon, data.Hoff, edge_counts)
        from .. import parallel
        if parallel.is_sharded(data):
            # edge-sharded crystal: ONE dE for the whole crystal -- the replicated on-site rows counted once, the ranks' off-site sums added
            # (hamgnn_output.py:3971-3981; SOC :3892-3913 shifts the two spin-diagonal real blocks).  Reductions + one axpy: tensor algebra.
            import torch.distributed as dist
            n, N = self.nao_max, int(data.z.shape[0])
            Sf, Hf = f32c(S), f32c(Href)
            sel = Sf > 1e-6
            if soc:
                D = (H - Hf).reshape(-1, 2, n, 2, n)
                diff = 0.5 * (D[:, 0, :, 0, :] + D[:, 1, :, 1, :]).reshape(-1, n * n)
            else:
                diff = H - Hf
            num = torch.where(sel, diff, torch.zeros_like(diff)).double().sum(1)
            den = torch.where(sel, Sf, torch.zeros_like(Sf)).double().sum(1)
            part = torch.stack([num[N:].sum(), den[N:].sum()])
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
            dE = ((part[0] + num[:N].sum()) / (part[1] + den[:N].sum())).to(H.dtype)
            if soc:
                Hv = H.view(-1, 2, n, 2, n)
                S3 = Sf.reshape(-1, n, n)
                Hv[:, 0, :, 0, :] -= dE * S3
                Hv[:, 1, :, 1, :] -= dE * S3
            else:
                H -= dE * Sf
            return H
        ops.zero_point_shift(H, f32c(Href), f32c(S), self.nao_max, soc)
        return H

    def _zero_point_shift_adjoint(self, data, gH, edge_counts, threshold: float = 1e-6):
        """adjoint of hg_zero_point_shift (hamgnn_output.py:3971-3981, SOC :3892-3913): H' = H - dE S with ONE dE per batch,
        dE = sum_{S > thr} (H - Href) / sum_{S > thr} S  (SOC: the two spin-diagonal real blocks, denominator doubled)  =>
        g_H = g - [S > thr] (sum g S) / den.  Element-wise tensor algebra + two global sums on the gradient rows."""
        n = self.nao_max
        S = (gget(data, "overlap") if ghas(data, "overlap") else self._cat_by_crystal(data, data.Son, data.Soff, edge_counts)).float()
        from .. import parallel
        sel = (S > threshold).to(gH.dtype)
        sharded = parallel.is_sharded(data)
        N = int(data.z.shape[0])

        def total(rows):                                       # sum over the crystal's rows: replicated on-site rows once + all ranks' off-site rows
            if not sharded:
                return rows.double().sum()
            import torch.distributed as dist
            off = rows[N:].double().sum()
            dist.all_reduce(off, op=dist.ReduceOp.SUM)
            return off + rows[:N].double().sum()

        den = total((S * sel).sum(1))
        if not self.soc_switch:
            c = (total((gH * S).sum(1)) / den).to(gH.dtype)
            return gH - sel * c
        half = gH.shape[0] // 2                                # [real rows; imaginary rows]: only the real spin-diagonal blocks are shifted
        R = gH[:half].reshape(-1, 2, n, 2, n).clone()
        S3, sel3 = S.reshape(-1, n, n), sel.reshape(-1, n, n)
        c = (total(((R[:, 0, :, 0, :] + R[:, 1, :, 1, :]) * S3).sum((1, 2))) / (2.0 * den)).to(gH.dtype)
        R[:, 0, :, 0, :] -= sel3 * c
        R[:, 1, :, 1, :] -= sel3 * c
        return torch.cat([R.reshape(half, -1), gH[half:]], 0)

    def build_interaction_masks(self, data, edge_counts=None, soc=False):
        """bool masks of the matrix elements that exist for the atoms' basis sets (index plumbing, no arithmetic):
        non-SOC build_interaction_masks (hamgnn_output.py:2616-2665: [on-site rows; off-site rows], NOT per crystal),
        SOC build_spin_orbit_interaction_masks (:2716-2783: every spin block carries the orbital mask; per-crystal order)."""
        m = self._mask[data.z] > 0                                           # [N, nao]
        src, dst = data.edge_index
        on = m[:, :, None] & m[:, None, :]
        off = m[src][:, :, None] & m[dst][:, None, :]
        if not soc:
            return torch.cat([on.reshape(on.shape[0], -1), off.reshape(off.shape[0], -1)], 0)
        n2 = (2 * self.nao_max) ** 2
        return self._cat_by_crystal(data, on.repeat(1, 2, 2).reshape(-1, n2), off.repeat(1, 2, 2).reshape(-1, n2), edge_counts)

    def calculate_sparsity_ratio(self, data):
        """hamgnn_output.py:2894-2930: a function of the graph's z / edge_index and this head's basis tables only -- kept on the graph's Topology
        (re-evaluated when z or edge_index are replaced or mutated: topo.get_topology), 19 element-wise launches per forward otherwise"""
        topo = get_topology(data)
        memo = topo.__dict__.setdefault("_sparsity", {})
        key = (self._basis_sig, str(data.z.device))
        if key not in memo:
            memo[key] = self._sparsity_ratio(data)
        return memo[key]

    def _sparsity_ratio(self, data):
        z = data.z
        n2 = self.nao_max ** 2
        src, dst = data.edge_index
        ni = self._norb[z]
        both = self._defined[z[src]] & self._defined[z[dst]]
        eff = (ni * ni).sum() + torch.where(both, ni[src] * ni[dst], torch.full_like(src, n2)).sum()
        total = float((z.numel() + src.numel()) * n2)
        return (total / eff.to(torch.float64)).to(torch.float32)

    # ------------------------------------------------------------------------------------------------------------
    def _blocks(self, net_on, net_off, node_pl, edge_rot, geo, data, inv, H0_on, H0_off, into=None):
        """into: optional [N+E, nao^2] buffer (single crystal: the result rows [on-site; off-site] are written in place)"""
        n, n2 = self.nao_max, self.nao_max ** 2
        src, dst = geo.src, geo.dst
        z = data.z.contiguous()
        N = z.shape[0]
        E = src.shape[0]
        on = into[:N] if into is not None else torch.empty(N, n2, device=z.device, dtype=torch.float32)
        off = into[N:] if into is not None else torch.empty(E, n2, device=z.device, dtype=torch.float32)
        # one pass per row set: CG merge + reorder + symmetrise (against the inverse edge, same block) + H0 + mask, written in place
        ops.ham_readout(net_on(node_pl), None, self._slot, *self._cg, n, None, H0_on, self._mask, z, None, None, on, self.hamiltonian_irreps.lmax, 1.0, self.symmetrize)
        pairs = get_topology(data).inverse_pairs(data)
        ops.ham_readout(net_off(edge_rot), geo, self._slot, *self._cg, n, pairs, H0_off, self._mask, z, src, dst, off, self.hamiltonian_irreps.lmax, 1.0, self.symmetrize)
        return on, off

    # ---- backward of the read-out (K6 data gradient + the head's weight gradients): non-SOC and SOC / so3
    def backward(self, data, graph_representation, grad_hamiltonian, grad_unshifted=None):
        """grad_hamiltonian: gradient with respect to result["hamiltonian"] in the forward's row order -- non-SOC: [N + E, nao^2];
        SOC / so3: [2 (N + E), (2 nao)^2] = [real rows; imaginary rows].  Returns (g_node_planar [N, Dp], g_edge_planar_rot [E, Dp] --
        gradients of the representation's planar node rows and edge-frame edge rows --, {parameter name: gradient}).
        Chain of the spin-free blocks: mask and symmetrisation are their own adjoint (hg_ham_finish on the gradient), the CG merge +
        reorder is a CSR map applied transposed (hg_ham_merge with plan.ham_merge_adjoint_tables), the un-rotation's adjoint is the
        rotation (hg_rotate_gather), then HamLayer.backward (Linear / Gate adjoints, weight gradients as GEMMs).
        SOC / so3 (hamgnn_output.py:3026-3144; csrc/head.hip soc_assemble_kernel): the (2 nao)^2 gradient rows are folded back onto the
        spin-free block (uu + dd of the real part) and onto ksi (the three L components x the antihermitised blocks), the shell-block
        mean is its own adjoint (hg_block_mean), then the ksi networks' HamLayer.backward.  With add_H_nonsoc the spin-free block is an
        input (the non-SOC model's prediction) and only the ksi path carries gradients -- the Uni-HamGNN SOC training mode."""
        if not self.ham_only:
            raise NotImplementedError("head backward: ham_only=True (reference overlaps)")
        rep = graph_representation
        ops.require_fp32(self, data)                           # `precision: 64` raises instead of returning fp32-accurate rows
        dev = data.z.device
        if self._compiled_for != dev:
            self.compile(dev)
        geo = rep["_geometry"]
        node_pl, edge_rot = rep["_node_planar"], self._edge_rows_of(rep)
        inv, edge_counts = self._global_inverse(data)
        n = self.nao_max
        gH = grad_hamiltonian.float()
        if self.zero_point_shift:                              # the shift is the last step of the forward: its adjoint comes first
            gH = self._zero_point_shift_adjoint(data, gH, edge_counts)
        if grad_unshifted is not None:                         # gradient with respect to the blocks BEFORE the shift (the band energies read those)
            gH = gH + grad_unshifted.float()
        g_node = g_edge = None
        grads = {}
        if self.soc_switch and self.soc_basis == "su2":
            # SOC / su2 (hamgnn_output.py:3146-3178): the two (2 nao)^2 planes are finished separately (sign +1 real / -1 imaginary; mask and
            # (anti)symmetrisation are again their own adjoint), side by side they are the gradient of the su2 CG merge's [real | imag] rows
            half = gH.shape[0] // 2
            z = data.z.contiguous()
            gr_on, gr_off = self._split_by_crystal(data, gH[:half], edge_counts)
            gi_on, gi_off = self._split_by_crystal(data, gH[half:], edge_counts)
            fin = lambda g_, inv_, ia, ib, sign: ops.ham_finish(g_.contiguous(), inv_, None, self._mask, z, ia, ib, 2 * n, sign, self.symmetrize, True)
            g_on = torch.cat([fin(gr_on, None, None, None, 1.0), fin(gi_on, None, None, None, -1.0)], 1)
            g_off = torch.cat([fin(gr_off, inv, geo.src, geo.dst, 1.0), fin(gi_off, inv, geo.src, geo.dst, -1.0)], 1)
            return self._backward_merge("su2", (self._slot_su2,) + self._cg_su2, geo, node_pl, edge_rot, g_on, g_off)
        if self.soc_switch:
            half = gH.shape[0] // 2
            gr_on, gr_off = self._split_by_crystal(data, gH[:half], edge_counts)
            gi_on, gi_off = self._split_by_crystal(data, gH[half:], edge_counts)
            f32c = lambda t: t.contiguous().float()
            gk_on, gh_on = self._soc_fold(gr_on, gi_on, f32c(data.Lon), None)
            gk_off, gh_off = self._soc_fold(gr_off, gi_off, f32c(data.Loff), inv)
            ksi_lay = P.PlanarLayout(self.onsite_ksi_network.ham_irreps)
            pad = lambda t: torch.nn.functional.pad(t, (0, ksi_lay.dim - t.shape[1])).contiguous()
            g_node, gw = self.onsite_ksi_network.backward(node_pl, pad(ops.block_mean(gk_on.contiguous(), self._blk, n)))
            grads.update({"onsite_ksi_network." + k: v for k, v in gw.items()})
            g_edge, gw = self.offsite_ksi_network.backward(edge_rot, pad(ops.block_mean(gk_off.contiguous(), self._blk, n)))
            grads.update({"offsite_ksi_network." + k: v for k, v in gw.items()})
            if self.add_H_nonsoc:                              # the spin-free networks are not evaluated (their parameters get zeros)
                for name in ("onsite_hamiltonian_network", "offsite_hamiltonian_network"):
                    grads.update({f"{name}.{k}": torch.zeros_like(p).reshape(-1) for k, p in getattr(self, name).named_parameters()})
                return g_node, g_edge, grads
            gH_on, gH_off = gh_on.contiguous(), gh_off.contiguous()
        else:
            gH_on, gH_off = (t.contiguous() for t in self._split_by_crystal(data, gH, edge_counts))
        gn, ge, gw = self._backward_spin_free(data, geo, node_pl, edge_rot, inv, gH_on, gH_off)
        grads.update(gw)
        return (gn if g_node is None else g_node + gn), (ge if g_edge is None else g_edge + ge), grads

    def _soc_fold(self, gr, gi, L, inv):
        """adjoint of hg_soc_assemble for one row set: gradient rows of the real / imaginary (2 nao)^2 matrices -> (g_ksi [rows, nao^2],
        g_H [rows, nao^2]).  real = [[H, A_y], [A_y, H]], imag = [[A_z, A_x], [-A_x, -A_z]],  A_k = antiherm(ksi L_k) (element-wise
        torch ops on [rows, nao, nao] views: HBM-bound bookkeeping, a few MB per thousand rows)."""
        n = self.nao_max
        R = gr.reshape(-1, 2, n, 2, n)
        I = gi.reshape(-1, 2, n, 2, n)
        g_h = (R[:, 0, :, 0, :] + R[:, 1, :, 1, :]).reshape(-1, n * n)
        g_a = torch.stack([I[:, 0, :, 1, :] - I[:, 1, :, 0, :],          # k = 0 (x): imaginary off-diagonal blocks
                           R[:, 0, :, 1, :] + R[:, 1, :, 0, :],          # k = 1 (y): real off-diagonal blocks
                           I[:, 0, :, 0, :] - I[:, 1, :, 1, :]], -1)     # k = 2 (z): imaginary diagonal blocks      -> [rows, n, n, 3]
        if self.symmetrize:                                    # A_k[e] = (ksi L_k)[e] / 2 - (ksi L_k)[inv e]^T / 2
            other = g_a if inv is None else g_a[inv]
            g_a = 0.5 * (g_a - other.transpose(1, 2))
        return (g_a * L.reshape(-1, n, n, 3)).sum(-1).reshape(-1, n * n), g_h

    def _backward_spin_free(self, data, geo, node_pl, edge_rot, inv, gH_on, gH_off):
        z = data.z.contiguous()
        n = self.nao_max
        # mask . symmetrise is self-adjoint (the orbital mask of an edge equals the transposed mask of its inverse edge)
        g_on = ops.ham_finish(gH_on, None, None, self._mask, z, None, None, n, 1.0, self.symmetrize)
        g_off = ops.ham_finish(gH_off, inv, None, self._mask, z, geo.src, geo.dst, n, 1.0, self.symmetrize)
        return self._backward_merge("so3", (self._slot,) + self._cg, geo, node_pl, edge_rot, g_on, g_off)

    def _backward_merge(self, key, tables, geo, node_pl, edge_rot, g_on, g_off):
        """adjoint of the CG merge + reorder (a CSR map applied transposed: hg_ham_merge with plan.ham_merge_adjoint_tables), of the
        off-site un-rotation (hg_rotate_gather) and of the two Hamiltonian HamLayers"""
        dev = g_on.device
        # structural tables (CG merge maps: functions of the basis, not of the weights): built once per (key, table object) -- rebuilding them
        # after every optimiser step put a device -> host copy in the middle of the step
        cache = self.__dict__.setdefault("_adj_tabs_by", {})
        key = (key, str(tables[0].device))                     # (per device: an id()-keyed entry outlives its tables and could be recycled)
        if key not in cache:
            glay = P.PlanarLayout(self.onsite_hamiltonian_network.girr)
            st, ptr_, idx_, val_ = (t.cpu().numpy() for t in tables)
            sid, pT, iT, vT, scat = P.ham_merge_adjoint_tables(st, ptr_, idx_, val_, glay.dim)
            cache[key] = tuple(torch.from_numpy(a).to(dev) for a in (sid, pT, iT, vT, scat)) + (
                torch.from_numpy(P.rotate_table(glay)).to(dev), int(st.shape[0]))
        sid, pT, iT, vT, scat, rot_g, ncoef = cache[key]
        gc_on = ops.from_planar(ops.ham_merge(g_on, None, sid, pT, iT, vT, ncoef), scat)
        gc_off = ops.rotate_gather(ops.from_planar(ops.ham_merge(g_off, None, sid, pT, iT, vT, ncoef), scat), None, geo, rot_g)
        g_node, gw_on = self.onsite_hamiltonian_network.backward(node_pl, gc_on)
        g_edge, gw_off = self.offsite_hamiltonian_network.backward(edge_rot, gc_off)
        grads = {"onsite_hamiltonian_network." + k: v for k, v in gw_on.items()}
        grads.update({"offsite_hamiltonian_network." + k: v for k, v in gw_off.items()})
        return g_node, g_edge, grads

    def edge_irreps_read(self):
        """the (l, parity) classes of the representation's EDGE rows this head's result depends on: the union over its off-site networks
        (HamLayer.input_irreps_read).  A backbone that knows its only consumer (HamGNNConvE3.declare_consumer, called by Model) need not compute the others
        in its last PairInteractionBlock; e.g. openmx nao_max 19 (s3 p2 d2): l <= 4 with parity (-1)^l ... never 0o, 4o, 5o, 5e, 6e of the shipped features."""
        nets = [n_ for n_ in (getattr(self, k, None) for k in ("offsite_hamiltonian_network", "offsite_overlap_network", "offsite_ksi_network")) if n_ is not None]
        return frozenset().union(*(n_.input_irreps_read() for n_ in nets))

    def _edge_rows_of(self, rep):
        """the planar edge-frame rows of the representation, complete in every irrep this head reads.  A backbone that skipped unread irreps says which
        ones it did compute (`_edge_alive`); if that does not cover this head (another head than the declared consumer), the complete rows are asked for
        (`_edge_planar_rot_full`: evaluated on first access)."""
        if not hasattr(rep, "get"):
            return None
        alive = rep.get("_edge_alive")
        if alive is not None and not (self.edge_irreps_read() & frozenset((int(l), int(p)) for _, l, p in self.edge_layout.irreps)) <= alive:
            return rep["_edge_planar_rot_full"]
        return rep.get("_edge_planar_rot")

    def forward(self, data, graph_representation=None):
        rep = graph_representation
        ops.require_fp32(self, data)                           # `precision: 64` raises instead of returning fp32-accurate rows
        dev = data.z.device
        if self._compiled_for != dev:
            self.compile(dev)
        self._validate(data)
        geo = rep.get("_geometry") if hasattr(rep, "get") else None
        if geo is None or geo.lmax < self._lmax:
            c = 1.0  # cutoff / radial basis are irrelevant for the frames
            geo = ops.Geometry(data.pos, data.edge_index, data.nbr_shift, c, 1, self._lmax, self._jtab)
        node_pl = rep.get("_node_planar") if hasattr(rep, "get") else None
        if node_pl is None:
            node_pl = ops.to_planar(rep["node_attr"], self._n_imap, self.node_layout.dim)
        edge_rot = self._edge_rows_of(rep)
        if edge_rot is None:
            edge_rot = ops.rotate_gather(ops.to_planar(rep["edge_attr"], self._e_imap, self.edge_layout.dim), None, geo, self._rot_tab)
        inv, edge_counts = self._global_inverse(data)
        f32c = lambda t: t.contiguous().float()
        # the reference attaches the combined TARGETS to the batch on the first forward (hamgnn_output.py:2975-2978): losses, the
        # zero-point shift and the test stage's target_hamiltonian.npy read data.hamiltonian / data.overlap (non-SOC layout)
        if not self.soc_switch:
            if not ghas(data, "hamiltonian") and ghas(data, "Hon") and ghas(data, "Hoff"):
                gset(data, "hamiltonian", self._cat_by_crystal(data, gget(data, "Hon"), gget(data, "Hoff"), edge_counts))
            if not ghas(data, "overlap") and ghas(data, "Son") and ghas(data, "Soff"):
                gset(data, "overlap", self._cat_by_crystal(data, gget(data, "Son"), gget(data, "Soff"), edge_counts))
        result = {}
        if not self.ham_only:
            s_on, s_off = self._blocks(self.onsite_overlap_network, self.offsite_overlap_network, node_pl, edge_rot, geo, data, inv, None, None)
            result["overlap"] = self._cat_by_crystal(data, s_on, s_off, edge_counts)
        if self.soc_switch and not ghas(data, "hamiltonian") and all(ghas(data, k) for k in ("Hon", "Hoff", "iHon", "iHoff")):
            # SOC targets as the reference attaches them (hamgnn_output.py:3621-3626): [real rows; imaginary rows], per crystal each
            tr = self._cat_by_crystal(data, gget(data, "Hon"), gget(data, "Hoff"), edge_counts)
            ti = self._cat_by_crystal(data, gget(data, "iHon"), gget(data, "iHoff"), edge_counts)
            gset(data, "hamiltonian_real", tr)
            gset(data, "hamiltonian_imag", ti)
            gset(data, "hamiltonian", torch.cat([tr, ti], 0))
        if self.soc_switch and self.soc_basis == "su2":                      # ---- SOC / su2 (hamgnn_output.py:3146-3178)
            n, big = self.nao_max, 4 * self.nao_max ** 2
            z = data.z.contiguous()
            H0 = (f32c(data.Hon0), f32c(data.Hoff0), f32c(data.iHon0), f32c(data.iHoff0)) if self.add_H0 else (None,) * 4
            raw_on = ops.ham_merge(self.onsite_hamiltonian_network(node_pl), None, self._slot_su2, *self._cg_su2, 2 * big)
            raw_off = ops.ham_merge(self.offsite_hamiltonian_network(edge_rot), geo, self._slot_su2, *self._cg_su2, 2 * big)
            fin = lambda raw, plane, inv_, h0, ia, ib: ops.ham_finish(raw[:, plane * big:(plane + 1) * big], inv_, h0, self._mask, z, ia, ib,
                                                                       2 * n, 1.0 if plane == 0 else -1.0, self.symmetrize, True)
            on_r, on_i = fin(raw_on, 0, None, H0[0], None, None), fin(raw_on, 1, None, H0[2], None, None)
            off_r, off_i = fin(raw_off, 0, inv, H0[1], geo.src, geo.dst), fin(raw_off, 1, inv, H0[3], geo.src, geo.dst)
            Hr = self._cat_by_crystal(data, on_r, off_r, edge_counts)
            Hi = self._cat_by_crystal(data, on_i, off_i, edge_counts)
            be, wf = self._soc_bands(data, on_r, on_i, off_r, off_i, dev)    # from the blocks BEFORE the shift (hamgnn_output.py:3642-3662)
            if self.zero_point_shift:
                # a training step with a band-energy loss re-evaluates the bands from the unshifted rows (training.training_step)
                self._unshifted = torch.cat([Hr, Hi], 0) if (self.calculate_band_energy and "_tape" in rep) else None
                Hr = self._apply_zero_point_shift(data, Hr, edge_counts, True)
            result.update({"hamiltonian": torch.cat([Hr, Hi], 0), "hamiltonian_real": Hr, "hamiltonian_imag": Hi, "band_energy": be,
                           "wavefunction": wf})
            if self.get_nonzero_mask_tensor:
                result["mask_real_imag"] = self.build_interaction_masks(data, edge_counts, soc=True)
            if self.calculate_sparsity:
                result["sparsity_ratio"] = self.calculate_sparsity_ratio(data)
            return result
        if self.soc_switch:                                                  # ---- SOC / so3 (hamgnn_output.py:3026-3144)
            n = self.nao_max
            if self.add_H_nonsoc:
                on, off = f32c(data.Hon_nonsoc), f32c(data.Hoff_nonsoc)
            else:
                on, off = self._blocks(self.onsite_hamiltonian_network, self.offsite_hamiltonian_network, node_pl, edge_rot, geo, data, inv, None, None)
            ksi_on = ops.block_mean(self.onsite_ksi_network(node_pl), self._blk, n)
            ksi_off = ops.block_mean(self.offsite_ksi_network(edge_rot), self._blk, n)
            H0 = (f32c(data.Hon0), f32c(data.Hoff0), f32c(data.iHon0), f32c(data.iHoff0)) if self.add_H0 else (None,) * 4
            on_r, on_i = ops.soc_assemble(on, ksi_on, f32c(data.Lon), None, H0[0], H0[2], n, self.symmetrize, self.add_H_nonsoc)
            off_r, off_i = ops.soc_assemble(off, ksi_off, f32c(data.Loff), inv, H0[1], H0[3], n, self.symmetrize, self.add_H_nonsoc)
            Hr = self._cat_by_crystal(data, on_r, off_r, edge_counts)
            Hi = self._cat_by_crystal(data, on_i, off_i, edge_counts)
            be, wf = self._soc_bands(data, on_r, on_i, off_r, off_i, dev)    # from the blocks BEFORE the shift (hamgnn_output.py:3642-3662)
            if self.zero_point_shift:
                # a training step with a band-energy loss re-evaluates the bands from the unshifted rows (training.training_step)
                self._unshifted = torch.cat([Hr, Hi], 0) if (self.calculate_band_energy and "_tape" in rep) else None
                Hr = self._apply_zero_point_shift(data, Hr, edge_counts, True)
            result.update({"hamiltonian": torch.cat([Hr, Hi], 0), "hamiltonian_real": Hr, "hamiltonian_imag": Hi, "band_energy": be,
                           "wavefunction": wf})
            if self.get_nonzero_mask_tensor:
                result["mask_real_imag"] = self.build_interaction_masks(data, edge_counts, soc=True)
            if self.calculate_sparsity:
                result["sparsity_ratio"] = self.calculate_sparsity_ratio(data)
            return result
        H0_on = f32c(data.Hon0) if self.add_H0 else None
        H0_off = f32c(data.Hoff0) if self.add_H0 else None
        single = edge_counts is None or edge_counts.numel() <= 1
        H = torch.empty(node_pl.shape[0] + edge_rot.shape[0], self.nao_max ** 2, device=dev, dtype=torch.float32) if single else None
        on, off = self._blocks(self.onsite_hamiltonian_network, self.offsite_hamiltonian_network, node_pl, edge_rot, geo, data, inv, H0_on, H0_off, into=H)
        if not single:
            H = self._cat_by_crystal(data, on, off, edge_counts)
        result.update({"band_energy": None, "wavefunction": None, "band_gap": None, "H_sym": None})
        if self.calculate_band_energy:                                        # hamgnn_output.py:3800-3880 (non-SOC, reference overlaps)
            # BEFORE the zero-point shift, as the reference (:3802-3880 precede :3971-3981): the bands come from the unshifted blocks
            # (`on` / `off` are views of H for a single crystal, and the shift below works in place)
            from .. import kspace
            data["k_vecs"] = kspace.make_k_vectors(self.k_path, self.num_k, data.cell, data=data).to(dev)
            if self.export_reciprocal_values:                                    # :3856-3869: H(k), S(k), dS(k) next to the bands, H_sym = None;
                be, wf, HK, SK, dSK, gap = kspace.band_energies_export(           # with overlap networks S(k) is the PREDICTED overlap
                    self, on, off, data, overlap=None if self.ham_only else (s_on, s_off))
                hs = None
                result.update({"HK": HK, "SK": SK, "dSK": dSK})
            else:
                be, wf, gap, hs = kspace.band_energies(self, on, off, data)
            with torch.no_grad():                                             # reference bands from the target blocks (:3876-3879)
                tb, tw, tg, th = kspace.band_energies(self, f32c(data.Hon), f32c(data.Hoff), data)
            data["band_energy"], data["wavefunction"], data["band_gap"], data["H_sym"] = tb, tw, tg, th
            if self.zero_point_shift:                                         # :3983-3985: the bands are aligned by their mean instead
                be = be - torch.mean(be - tb)
                # what the bands were computed from: a backward re-evaluates them (the shift below is in place).  Kept on the head -- not in the
                # public result dict -- and only when a backward will follow (the representation carries the backbone's tape)
                self._unshifted = H.clone() if "_tape" in rep else None
            result.update({"band_energy": be, "wavefunction": wf, "band_gap": gap, "H_sym": hs})
        if self.zero_point_shift:
            H = self._apply_zero_point_shift(data, H, edge_counts, False)
        result["hamiltonian"] = H
        if self.get_nonzero_mask_tensor:
            result["mask"] = self.build_interaction_masks(data)
        if self.calculate_sparsity:
            result["sparsity_ratio"] = self.calculate_sparsity_ratio(data)
        return result
