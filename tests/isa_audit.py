"""ISA audit of a HIP source for gfx950 (the tool behind profiles/r03_*: "ISA audit" sections).

    python tests/isa_audit.py hamgnn_amd/csrc/tp_is.hip [-D FLAG ...] [--kernel SUBSTR] [--all]

Compiles the file with `hipcc --cuda-device-only -S`, splits every kernel into basic blocks and prints, per block, the sequence of its
memory / MFMA instructions and waits:  M = v_mfma, G = global load, D = LDS-DMA / buffer load, W = global store / atomic, r / w = LDS
read / write, S = scalar load, | = s_barrier, [v(n)] / [l(n)] = s_waitcnt vmcnt / lgkmcnt.  Patterns it flags (each one cost measurable time
somewhere this round):
  SERIAL-r-M    ds_read -> lgkmcnt(0) -> v_mfma repeated: an LDS round trip per MFMA
  SERIAL-RMW    ds_read -> lgkmcnt(0) -> ds_write repeated: "+=" on LDS rows the compiler cannot prove distinct
  SERIAL-G      global load -> vmcnt(0) repeated: dependent loads, one memory round trip each
  LOOP-VMCNT0   a block with >= 4 MFMAs and global loads that waits vmcnt(0): a register ring whose look-ahead collapses (refill requested
                before the slot's last read -> v_mov rotation at the back-edge; priming loads issued out of order -> the loop head's one
                static wait is sized for the entry)
Also printed per kernel: VGPRs, spilled VGPRs / SGPRs, scratch bytes (from the code object metadata)."""
import argparse, os, re, subprocess, sys, tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOPK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]      # as hamgnn_amd/csrc/Makefile (tests/test_isa_audit.py checks that the two agree)


import functools


@functools.lru_cache(maxsize=None)      # (one compile per (file, defines) and test session: tp_is.hip alone takes 20 s)
def compile_to_asm(src, defines=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", *NOPK, "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out, src]
    cmd += ["-D" + d for d in defines]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text):
    """[(name, [(block label, [instruction, ...]), ...])] and {name: metadata dict}"""
    out, cur, blk = [], None, None
    for l in text.split("\n"):
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", l)
        if m and not l.startswith(".L") and not m.group(1).startswith("__hip_cuid"):
            cur = (m.group(1), [])
            out.append(cur)
            blk = ("entry", [])
            cur[1].append(blk)
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m and cur is not None:
            blk = (m.group(1), [])
            cur[1].append(blk)
            continue
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith("\t.section"):
            cur = None if l.startswith("\t.section") else cur
        if cur is not None and blk is not None and l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
            blk[1].append(l.strip())
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n\s+- \.|\n\.\.\.|\Z)", text, re.S):
        pass
    for sec in re.split(r"\n  - \.agpr_count", text)[1:]:
        name = re.search(r"\.name:\s+(\S+)", sec)
        if not name:
            continue
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", sec).group(1)) if re.search(r"\." + k + r":\s+(\d+)", sec) else -1
        meta[name.group(1)] = dict(vgprs=g("vgpr_count"), vgpr_spills=g("vgpr_spill_count"), sgpr_spills=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"))
    return [k for k in out if any(ins for _, ins in k[1])], meta


def seq_of(ins):
    seq = []
    for i in ins:
        if i.startswith("v_mfma"): seq.append("M")
        elif i.startswith("global_load_lds") or i.startswith("buffer_load"): seq.append("D")
        elif i.startswith("global_load"): seq.append("G")
        elif i.startswith("global_store") or i.startswith("global_atomic"): seq.append("W")
        elif i.startswith("ds_read"): seq.append("r")
        elif i.startswith("ds_write") or i.startswith("ds_add"): seq.append("w")
        elif i.startswith("s_load"): seq.append("S")
        elif i.startswith("s_barrier"): seq.append("|")
        elif i.startswith("s_waitcnt"):
            a = i.split(None, 1)[1]
            seq.append("[" + a.replace("vmcnt", "v").replace("lgkmcnt", "l").replace(" ", "") + "]")
    return "".join(seq)


def flags_of(s):
    f = []
    if re.search(r"(r\[l\(0\)\]M){2,}", s): f.append("SERIAL-r-M")
    if re.search(r"(r\[l\(0\)\]w){2,}", s): f.append("SERIAL-RMW")
    if re.search(r"(G\[v\(0\)\]){2,}", s): f.append("SERIAL-G")
    if s.count("M") >= 4 and "G" in s and re.search(r"\[v\(0\)", s): f.append("LOOP-VMCNT0")
    return f


def audit(src, defines=(), kernel=None):
    ks, meta = kernels(compile_to_asm(src, defines))
    rep = []
    for name, blocks in ks:
        if kernel and kernel not in name:
            continue
        rows = [(lab, len(ins), seq_of(ins)) for lab, ins in blocks]
        rep.append(dict(name=name, meta=meta.get(name, {}), blocks=[(lab, n, s, flags_of(s)) for lab, n, s in rows if s]))
    return rep


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("src"); ap.add_argument("-D", action="append", default=[]); ap.add_argument("--kernel"); ap.add_argument("--all", action="store_true")
    a = ap.parse_args()
    for k in audit(a.src, a.D, a.kernel):
        nm = sum(s.count("M") for _, _, s, _ in k["blocks"])
        ser = sum(len(m.group(0)) // 8 for _, _, s, _ in k["blocks"] for m in re.finditer(r"(r\[l\(0\)\]M){2,}", s))
        print(f"== {k['name']}  {k['meta']}  static MFMAs {nm}, in serial read-wait-MFMA chains {ser}")
        for lab, n, s, f in k["blocks"]:
            if a.all or f:
                print(f"  {lab:12s} {n:4d} {' '.join(f):24s} {s[:200]}")
