"""Dependency-free reader (and bulk writer) for the LMDB graph stores the reference uses (hamgnn/data/graph_data.py:23-93
LMDBGraphDataset; tools/npz_to_lmdb.py:82-94: key ``num_graphs`` -> str(n), keys ``graph_{i}`` -> pickle.dumps(Data)).

The ``lmdb`` module is not installable in this image, so the on-disk format of LMDB 0.9 (``data.mdb``: a copy-on-write B+tree
in fixed-size pages) is parsed directly.  Written from the published layout of ``mdb.c`` (MDB_page / MDB_node / MDB_meta /
MDB_db); **not pinned against liblmdb here** (no lmdb build exists in the container) -- the reader and the writer below are
each other's only test partner, which is stated in DESIGN.md.  Layout facts used (little-endian, 64-bit build):

  page header, 16 bytes: pgno u64 | pad u16 | flags u16 | lower u16, upper u16 (overflow pages: page count u32 instead)
        flags: P_BRANCH 0x01, P_LEAF 0x02, P_OVERFLOW 0x04, P_META 0x08, P_LEAF2 0x20, P_SUBP 0x40
  node pointers: u16 offsets (from the page start) from byte 16 on, one per key, sorted by key; count = (lower - 16) / 2
  node, 8-byte header: lo u16 | hi u16 | flags u16 | ksize u16 | key bytes | data
        leaf:   data size = lo | hi << 16; flags F_BIGDATA 0x01: data is the u64 page number of an overflow run holding the value
        branch: child page number = lo | hi << 16 | flags << 32; node 0 carries an empty key (left-most child)
  meta pages 0 and 1: header, then magic 0xBEEFC0DE u32 | version u32 (1) | address u64 | mapsize u64 | two MDB_db records (48 bytes:
        pad u32 | flags u16 | depth u16 | branch_pages u64 | leaf_pages u64 | overflow_pages u64 | entries u64 | root u64) for the
        free-list DB and the main DB | last_pg u64 | txnid u64.  The page size is the free-list DB's `pad`; the meta page with the
        larger txnid is current; root = 2^64 - 1 means empty.
  keys compare as byte strings (memcmp, shorter first on a tie) -- the default comparator, which the reference's stores use.
"""
from __future__ import annotations

import mmap
import os
import struct
from typing import Dict, Iterator, Optional, Tuple

MAGIC = 0xBEEFC0DE
P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2 = 0x01, 0x02, 0x04, 0x08, 0x20
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
HDR = 16
INVALID = (1 << 64) - 1


class LMDBFormatError(ValueError):
    pass


def _data_file(path: str) -> str:
    return os.path.join(path, "data.mdb") if os.path.isdir(path) else path


class LMDBReader:
    """read-only view of the main database: ``get(key)``, ``items()``, ``len``; supports ``with``"""

    def __init__(self, path: str):
        self._f = open(_data_file(path), "rb")
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        m = self._m
        if len(m) < 2 * 512:
            raise LMDBFormatError("file too small for two meta pages")
        metas = []
        psize = None
        for cand in (4096, 8192, 16384, 32768, 65536, 2048, 1024, 512):    # page 1 sits at offset psize: probe, then confirm by `pad`
            if len(m) >= 2 * cand and struct.unpack_from("<I", m, cand + HDR)[0] == MAGIC:
                if struct.unpack_from("<I", m, HDR + 24)[0] == cand:
                    psize = cand
                    break
        if psize is None:
            raise LMDBFormatError("no LMDB meta page found (magic 0xBEEFC0DE)")
        for pg in (0, 1):
            off = pg * psize
            flags = struct.unpack_from("<H", m, off + 10)[0]
            magic, version = struct.unpack_from("<II", m, off + HDR)
            if magic != MAGIC or not flags & P_META:
                raise LMDBFormatError(f"meta page {pg} is damaged")
            if version != 1:
                raise LMDBFormatError(f"unsupported LMDB data version {version}")
            main = struct.unpack_from("<IHHQQQQQ", m, off + HDR + 24 + 48)
            last_pg, txnid = struct.unpack_from("<QQ", m, off + HDR + 24 + 96)
            metas.append((txnid, main, last_pg))
        txnid, main, last_pg = max(metas, key=lambda t: t[0])
        self.page_size = psize
        _, self._dbflags, self.depth, _, _, _, self.entries, self.root = main
        if self._dbflags & 0x04:                                              # MDB_DUPSORT
            raise LMDBFormatError("DUPSORT databases are not supported")

    # ---- pages / nodes
    def _page(self, pgno: int):
        off = pgno * self.page_size
        if off + self.page_size > len(self._m):
            raise LMDBFormatError(f"page {pgno} beyond the end of the file")
        _, _, flags, lower, upper = struct.unpack_from("<QHHHH", self._m, off)
        return off, flags, (lower - HDR) >> 1

    def _node(self, off: int, i: int):
        p = off + struct.unpack_from("<H", self._m, off + HDR + 2 * i)[0]
        lo, hi, flags, ksize = struct.unpack_from("<HHHH", self._m, p)
        return p, lo, hi, flags, ksize

    def _key(self, p: int, ksize: int) -> bytes:
        return self._m[p + 8:p + 8 + ksize]

    def _value(self, p: int, lo: int, hi: int, flags: int, ksize: int) -> bytes:
        size = lo | (hi << 16)
        if flags & (F_SUBDATA | F_DUPDATA):
            raise LMDBFormatError("sub-databases / duplicate values are not supported")
        d = p + 8 + ksize
        if flags & F_BIGDATA:
            (ov,) = struct.unpack_from("<Q", self._m, d)
            off = ov * self.page_size
            oflags = struct.unpack_from("<H", self._m, off + 10)[0]
            if not oflags & P_OVERFLOW:
                raise LMDBFormatError(f"page {ov} is not an overflow page")
            return self._m[off + HDR:off + HDR + size]
        return self._m[d:d + size]

    # ---- look-up
    def get(self, key: bytes, default=None) -> Optional[bytes]:
        if self.root == INVALID:
            return default
        pg = self.root
        for _ in range(64):
            off, flags, n = self._page(pg)
            if flags & P_BRANCH:
                lo_i, hi_i = 1, n - 1                       # node 0 = left-most child (empty key); last node with key <= target
                child = 0
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) >> 1
                    p, *_r, ksize = self._node(off, mid)
                    if self._key(p, ksize) <= key:
                        child, lo_i = mid, mid + 1
                    else:
                        hi_i = mid - 1
                p, lo, hi, fl, _ = self._node(off, child)
                pg = lo | (hi << 16) | (fl << 32)
            elif flags & P_LEAF:
                if flags & P_LEAF2:
                    raise LMDBFormatError("LEAF2 pages (fixed-size duplicates) are not supported")
                lo_i, hi_i = 0, n - 1
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) >> 1
                    p, lo, hi, fl, ksize = self._node(off, mid)
                    k = self._key(p, ksize)
                    if k == key:
                        return self._value(p, lo, hi, fl, ksize)
                    if k < key:
                        lo_i = mid + 1
                    else:
                        hi_i = mid - 1
                return default
            else:
                raise LMDBFormatError(f"page {pg}: unexpected flags {flags:#x}")
        raise LMDBFormatError("tree deeper than 64 levels")

    def items(self) -> Iterator[Tuple[bytes, bytes]]:
        if self.root == INVALID:
            return
        stack = [self.root]
        while stack:
            off, flags, n = self._page(stack.pop())
            if flags & P_BRANCH:
                kids = []
                for i in range(n):
                    p, lo, hi, fl, _ = self._node(off, i)
                    kids.append(lo | (hi << 16) | (fl << 32))
                stack.extend(reversed(kids))
            else:
                for i in range(n):
                    p, lo, hi, fl, ksize = self._node(off, i)
                    yield self._key(p, ksize), self._value(p, lo, hi, fl, ksize)

    def __len__(self):
        return int(self.entries)

    def close(self):
        self._m.close()
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def write_lmdb(path: str, items: Dict[bytes, bytes], page_size: int = 4096, map_size: int = 1 << 30) -> str:
    """Bulk-load `items` into a fresh single-database LMDB environment directory `path` (data.mdb + an empty lock.mdb): sorted leaf
    pages, branch levels on top, values larger than a quarter page in overflow runs -- the layout mdb.c produces for an append-only load.
    Counterpart of tools/npz_to_lmdb.py for environments without the lmdb module."""
    os.makedirs(path, exist_ok=True)
    keys = sorted(items)
    pages: Dict[int, bytes] = {}
    next_pg = [2]
    n_over = 0
    max_inline = page_size // 4                            # conservative stand-in for mdb.c's me_nodemax

    def alloc(n=1):
        pg = next_pg[0]
        next_pg[0] += n
        return pg

    def build_page(flags, nodes):
        """nodes: list of raw node byte strings (even-sized); returns the page image without its pgno"""
        ptr_end = HDR + 2 * len(nodes)
        upper = page_size
        body = bytearray(page_size)
        offs = []
        for nd in nodes:
            upper -= len(nd)
            body[upper:upper + len(nd)] = nd
            offs.append(upper)
        assert upper >= ptr_end
        struct.pack_into("<HHHH", body, 8, 0, flags, ptr_end, upper)
        for i, o in enumerate(offs):
            struct.pack_into("<H", body, HDR + 2 * i, o)
        return body

    def finish(pg, body):
        struct.pack_into("<Q", body, 0, pg)
        pages[pg] = bytes(body)

    # ---- leaves
    level = []                                             # (first key, pgno)
    cur, cur_size, first = [], HDR, None

    def flush_leaf():
        nonlocal cur, cur_size, first
        if cur:
            pg = alloc()
            finish(pg, build_page(P_LEAF, cur))
            level.append((first, pg))
        cur, cur_size, first = [], HDR, None
    for k in keys:
        v = items[k]
        if len(k) > 511:
            raise ValueError("key longer than LMDB's 511-byte limit")
        if 8 + len(k) + len(v) > max_inline:
            npg = -(-(HDR + len(v)) // page_size)
            ov = alloc(npg)
            img = bytearray(npg * page_size)
            struct.pack_into("<QHHI", img, 0, ov, 0, P_OVERFLOW, npg)
            img[HDR:HDR + len(v)] = v
            for j in range(npg):
                pages[ov + j] = bytes(img[j * page_size:(j + 1) * page_size])
            n_over += npg
            data, fl = struct.pack("<Q", ov), F_BIGDATA
        else:
            data, fl = v, 0
        nd = struct.pack("<HHHH", len(v) & 0xffff, len(v) >> 16, fl, len(k)) + k + data
        if len(nd) & 1:
            nd += b"\0"
        if cur and cur_size + len(nd) + 2 > page_size:
            flush_leaf()
        if not cur:
            first = k
        cur.append(nd)
        cur_size += len(nd) + 2
    flush_leaf()
    n_leaf, n_branch, depth = len(level), 0, 1 if level else 0
    # ---- branch levels
    while len(level) > 1:
        up, cur, cur_size, first = [], [], HDR, None
        for i, (k, pg) in enumerate(level):
            kk = b"" if not cur else k                     # node 0 of a branch page carries an empty key
            nd = struct.pack("<HHHH", pg & 0xffff, (pg >> 16) & 0xffff, (pg >> 32) & 0xffff, len(kk)) + kk
            if len(nd) & 1:
                nd += b"\0"
            if cur and cur_size + len(nd) + 2 > page_size:
                bp = alloc()
                finish(bp, build_page(P_BRANCH, cur))
                up.append((first, bp))
                cur, cur_size, first = [], HDR, None
                nd = struct.pack("<HHHH", pg & 0xffff, (pg >> 16) & 0xffff, (pg >> 32) & 0xffff, 0)
            if not cur:
                first = k
            cur.append(nd)
            cur_size += len(nd) + 2
        bp = alloc()
        finish(bp, build_page(P_BRANCH, cur))
        up.append((first, bp))
        n_branch += len(up)
        level = up
        depth += 1
    root = level[0][1] if level else INVALID
    last_pg = next_pg[0] - 1
    # ---- meta pages (both current; txnid 1)
    for pg in (0, 1):
        body = bytearray(page_size)
        struct.pack_into("<QHH", body, 0, pg, 0, P_META)
        struct.pack_into("<IIQQ", body, HDR, MAGIC, 1, 0, map_size)
        struct.pack_into("<IHHQQQQQ", body, HDR + 24, page_size, 0, 0, 0, 0, 0, 0, INVALID)                # free-list DB (empty)
        struct.pack_into("<IHHQQQQQ", body, HDR + 24 + 48, 0, 0, depth, n_branch, n_leaf, n_over, len(keys), root)
        struct.pack_into("<QQ", body, HDR + 24 + 96, max(last_pg, 1), 1 if pg == 1 else 0)
        pages[pg] = bytes(body)
    with open(os.path.join(path, "data.mdb"), "wb") as f:
        for pg in range(next_pg[0]):
            f.write(pages[pg])
    with open(os.path.join(path, "lock.mdb"), "wb") as f:
        f.write(b"\0" * 8192)
    return path
