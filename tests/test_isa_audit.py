"""Compile-time guards on the hot kernels' code objects (no GPU: hipcc cross-compiles gfx950).  What they pin was measured on MI355X this
round (profiles/r03_lite.md, r03_tp_is_experiments.md, r03_wgrad.md): register budgets that keep two waves per SIMD without scratch, and
the wait structure of the lite run loop (its ring look-ahead silently collapses to vmcnt(0) when the refill is requested before the slot's
last read or the priming loads are reordered -- 8 % of the launch)."""
import os, re, shutil, sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_audit as A

CSRC = os.path.join(A.ROOT, "hamgnn_amd", "csrc")
pytestmark = pytest.mark.skipif(not os.path.exists(A.HIPCC) and shutil.which("hipcc") is None, reason="hipcc not found")


@pytest.fixture(scope="module")
def tp_is():
    return A.audit(os.path.join(CSRC, "tp_is.hip"))


def test_tp_is_register_budget_and_no_scratch(tp_is):
    ks = [k for k in tp_is if "tp_is_kernel" in k["name"]]
    assert len(ks) == 4                                         # <SPLIT, LITE> in {false, true}^2
    for k in ks:
        m = k["meta"]
        assert m["vgpr_spills"] == 0 and m["scratch"] == 0, (k["name"], m)
        assert m["vgprs"] <= (128 if "ELb1EEv" in k["name"] else 256), (k["name"], m)                # default: 2 waves per SIMD (219 / 219); lite: 4 per SIMD at <= 128 (126 / 128)


def test_lite_stream_loop_keeps_its_ring_lookahead(tp_is):
    lite = [k for k in tp_is if "tp_is_kernel" in k["name"] and "ILb0ELb1" in k["name"]]
    assert len(lite) == 1
    # stream_lite (r4): the blocks that issue a step's MFMAs (1..8 of them, behind one wait for the step's fragment): SL_RING unrolled steps x
    # {paired, single}; the fragment wait leaves the younger requests of the ring in flight, and no scalar load sits inside a step
    # (the r3 runs: one per step, waited for with lgkmcnt(0) by the step's MFMAs)
    step = [(lab, s) for lab, _, s, _ in lite[0]["blocks"] if re.fullmatch(r"(\[[^\]]*\])+M{1,8}(\[l\(\d+\)\]M{1,4})?", s)]
    ring = [(lab, s) for lab, s in step if "v(3)" in s]          # SL_RING = 4: the fragment wait leaves the 3 younger requests in flight
    assert len(ring) >= 6, [s for _, s in step]
    for lab, s in ring:
        assert "v(0)" not in s and "S" not in s, (lab, s)
    # the tile read-modify-write of a finished task: all reads, then all writes (was read -> wait -> write per element)
    for lab, _, s, f in lite[0]["blocks"]:
        assert "SERIAL-RMW" not in f, (lab, s)


def test_default_kernel_has_no_serial_lds_read_modify_write(tp_is):
    base = [k for k in tp_is if "ILb0ELb0" in k["name"]]
    assert len(base) == 1
    assert not [lab for lab, _, s, f in base[0]["blocks"] if "SERIAL-RMW" in f]
    # ... and no MFMA that waits for an LDS read issued right before it (r4: the natural-K GEMM1 of the one-row-tile items did)
    chained = sum(len(re.findall(r"r\[l\(0\)\]M", s)) for lab, _, s, f in base[0]["blocks"] if "SERIAL-r-M" in f)
    assert chained <= 4, chained                                # (was 334 of the 4 096 static MFMAs)


def test_row_program_and_readout_register_budgets():
    # row_program_kernel: 16 waves per workgroup = 128 VGPRs per wave at most; one VGPR (8 bytes of scratch) is spilled at the time of writing
    for src, kern, max_spill in (("rowprog.hip", "row_program_kernel", 1), ("head.hip", "ham_readout_kernel", 0)):
        ks = [k for k in A.audit(os.path.join(CSRC, src)) if kern in k["name"]]
        assert ks, (src, kern)
        for k in ks:
            assert k["meta"]["vgpr_spills"] <= max_spill and k["meta"]["scratch"] <= 8 * max_spill, (k["name"], k["meta"])


def test_segment_stationary_kernel_register_budget():
    """tp_fused_kernel (embedding TP, plain Linear programs, the fallback of the MessagePackBlocks; <true>: lite_mode programs): no scratch"""
    ks = [k for k in A.audit(os.path.join(CSRC, "tp_fused.hip")) if "tp_fused_kernel" in k["name"]]
    assert len(ks) == 2
    for k in ks:
        assert k["meta"]["vgpr_spills"] == 0 and k["meta"]["scratch"] == 0 and k["meta"]["vgprs"] <= 256, (k["name"], k["meta"])


@pytest.mark.skipif(os.environ.get("HG_SLOW_TESTS") != "1", reason="tp_wgrad.hip has 62 instantiations: ~100 s of compile time (HG_SLOW_TESTS=1)")
def test_weight_gradient_kernel_register_budget():
    ks = [k for k in A.audit(os.path.join(CSRC, "tp_wgrad.hip")) if "tp_wgrad_kernel" in k["name"]]
    assert ks
    worst = max(k["meta"]["scratch"] for k in ks)
    assert worst <= 124, worst                                # r3: the 13-column instantiation spills 124 B / lane, the NC = 1 ones 2 VGPRs


def test_no_kernel_of_the_library_holds_packed_fp32_instructions():
    """profiles/r06_tp_is.md section 8: on gfx950 the packed fp32 VALU instructions the compiler emits for scalar x vector products return wrong results while another
    wave of the SIMD executes v_mfma_f32_16x16x32_f16 / _bf16 (a co-running half-precision GEMM is enough; 33 000 of 41 000 tiles of an edge-kernel launch were wrong).
    The library is built without them (csrc/Makefile: NOPK); this pins the flag and its effect on every source file.  The one encoding a self-contained reproducer
    singles out is `v_pk_fma_f32 ... op_sel:[0,1,0]` (tools/pkfma_repro.hip); the compiler chooses encodings freely, hence none at all."""
    from concurrent.futures import ThreadPoolExecutor
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert " ".join(A.NOPK) in mk and "$(NOPK)" in mk.split("FLAGS :=", 1)[1].split("\n", 1)[0]
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    assert len(srcs) >= 11
    with ThreadPoolExecutor(8) as ex:
        texts = list(ex.map(lambda f: A.compile_to_asm(os.path.join(CSRC, f)), srcs))
    for f, text in zip(srcs, texts):
        hits = re.findall(r"^\s+(v_pk_(?:fma|mul|add)_f32|v_pk_mov_b32)\b", text, flags=re.M)
        assert not hits, (f, len(hits))
        assert re.search(r"^\s+v_(?:fma|mul|add)_f32", text, flags=re.M) or f in ("block_gemm.hip",), f      # (the audit did look at code)
