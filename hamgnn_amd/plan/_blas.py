"""Host-side cost of a (re)pack.  The builders run on every weight repack (each optimiser step of hamgnn_amd.training).  Their dense algebra is hundreds of TINY
matrix products (L @ Lo per output irrep, fragment packing); a multi-threaded BLAS spends ~30 ms of thread hand-off on each of them (measured: 13 products of
[832, 64] x [64, 64]: 424 ms on 8 OpenBLAS threads, 1.8 ms on one).  `@single_thread_blas` runs a builder single-threaded."""
import functools

try:
    from threadpoolctl import ThreadpoolController as _TPC
except ImportError:                                            # no threadpoolctl: correct, only slower
    _TPC = None
_tpc = None


def single_thread_blas(fn):
    @functools.wraps(fn)
    def wrapped(*a, **k):
        global _tpc
        if _TPC is None:
            return fn(*a, **k)
        if _tpc is None:
            _tpc = _TPC()
        with _tpc.limit(limits=1, user_api="blas"):
            return fn(*a, **k)
    return wrapped
