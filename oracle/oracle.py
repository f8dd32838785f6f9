"""ctypes front-end of the CPU oracle (oracle/ii2_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (inverted_index_2_amd) never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libii2_oracle.so")

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ii2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libii2_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.orc_sort_u32.argtypes = [u32p, C.c_size_t]
        L.orc_compact_u32.argtypes = [u32p, C.c_size_t]
        L.orc_compact_u32.restype = C.c_size_t
        L.orc_merge_term_values.argtypes = [u32p, C.c_size_t, u32p, C.c_size_t, u32p]
        L.orc_merge_term_values.restype = C.c_size_t
        L.orc_compare_terms.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_compare_terms.restype = C.c_int
        L.orc_removed_values.argtypes = [u32p, u64p, C.c_size_t, u32p]
        L.orc_removed_values.restype = C.c_size_t
        L.orc_removed_sync.argtypes = [C.POINTER(C.c_int64), C.c_size_t, C.POINTER(C.c_int64), C.c_size_t, u8p]
        L.orc_filter_removed.argtypes = [u32p, C.c_size_t, u32p, C.c_size_t]
        L.orc_filter_removed.restype = C.c_size_t
        pp64 = C.POINTER(u64p)
        pp32 = C.POINTER(u32p)
        pp8 = C.POINTER(u8p)
        L.orc_merge_segments.argtypes = [C.c_uint32, C.c_uint64, pp64, pp32, pp8, u32p, C.c_size_t, u64p, u32p]
        L.orc_merge_segments.restype = C.c_uint64
        L.orc_merge_segments_mt.argtypes = [C.c_uint32, C.c_uint64, pp64, pp32, u32p, C.c_size_t, u64p, u32p, C.c_int]
        L.orc_merge_segments_mt.restype = C.c_uint64
        L.orc_union.argtypes = [C.c_uint32, pp32, C.POINTER(C.c_size_t), u32p]
        L.orc_union.restype = C.c_size_t
        L.orc_intersect.argtypes = [C.c_uint32, pp32, C.POINTER(C.c_size_t), u32p, C.c_size_t, u32p]
        L.orc_intersect.restype = C.c_size_t
        L.orc_shard_key.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_shard_key.restype = C.c_uint32
        L.orc_dv1_encode.argtypes = [C.c_uint64, u64p, u32p, u32p, C.c_void_p, u8p, u64p]
        L.orc_dv1_encode.restype = C.c_uint32
        L.orc_dv1_decode.argtypes = [C.c_uint64, u32p, C.c_void_p, u8p, u64p, u32p]
        L.orc_dv1_decode.restype = C.c_uint64
    return _lib


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p32(a: np.ndarray):
    return a.ctypes.data_as(u32p)


def _p64(a: np.ndarray):
    return a.ctypes.data_as(u64p)


def sort_u32(v) -> np.ndarray:
    a = _u32(v).copy()
    lib().orc_sort_u32(_p32(a), a.size)
    return a


def merge_term_values(a, b) -> np.ndarray:
    a, b = _u32(a), _u32(b)
    out = np.empty(a.size + b.size, np.uint32)
    n = lib().orc_merge_term_values(_p32(a), a.size, _p32(b), b.size, _p32(out))
    return out[:n].copy()


def compare_terms(a: bytes, b: bytes) -> int:
    return lib().orc_compare_terms(a, len(a), b, len(b))


def removed_values(batches) -> np.ndarray:
    """batches: iterable of u32 sequences (the live RemovedLists batches)."""
    arrs = [_u32(b) for b in batches]
    off = np.zeros(len(arrs) + 1, np.uint64)
    for i, a in enumerate(arrs):
        off[i + 1] = off[i] + a.size
    flat = np.concatenate(arrs) if arrs else np.empty(0, np.uint32)
    flat = _u32(flat)
    out = np.empty(int(off[-1]), np.uint32)
    n = lib().orc_removed_values(_p32(flat), _p64(off), len(arrs), _p32(out))
    return out[:n]


def removed_sync(batch_ts, timestamps) -> np.ndarray:
    bt = np.ascontiguousarray(batch_ts, dtype=np.int64)
    ts = np.ascontiguousarray(timestamps, dtype=np.int64)
    keep = np.empty(bt.size, np.uint8)
    i64p = C.POINTER(C.c_int64)
    lib().orc_removed_sync(bt.ctypes.data_as(i64p), bt.size, ts.ctypes.data_as(i64p), ts.size,
                           keep.ctypes.data_as(u8p))
    return keep.astype(bool)


def filter_removed(values, removed_sorted) -> np.ndarray:
    v = _u32(values).copy()
    r = _u32(removed_sorted)
    n = lib().orc_filter_removed(_p32(v), v.size, _p32(r), r.size)
    return v[:n].copy()


def merge_segments(seg_offs, seg_vals, removed_sorted=(), present=None, threads: int = 0):
    """seg_offs[s]: u64[T+1], seg_vals[s]: u32[...].  Returns (out_off u64[T+1], out_values, n_surviving_terms)."""
    k = len(seg_offs)
    offs = [np.ascontiguousarray(o, dtype=np.uint64) for o in seg_offs]
    vals = [_u32(v) for v in seg_vals]
    T = offs[0].size - 1 if k else 0
    total = int(sum(int(o[-1]) for o in offs))
    r = _u32(removed_sorted)
    out_off = np.zeros(T + 1, np.uint64)
    out_vals = np.empty(max(total, 1), np.uint32)
    OffArr = u64p * max(k, 1)
    ValArr = u32p * max(k, 1)
    po = OffArr(*[_p64(o) for o in offs])
    pv = ValArr(*[_p32(v) for v in vals])
    if threads and present is None:
        n = lib().orc_merge_segments_mt(k, T, po, pv, _p32(r), r.size, _p64(out_off), _p32(out_vals), threads)
    else:
        pres = None
        if present is not None:
            parr = [np.ascontiguousarray(p, dtype=np.uint8) for p in present]
            PArr = u8p * max(k, 1)
            pres = PArr(*[p.ctypes.data_as(u8p) for p in parr])
        n = lib().orc_merge_segments(k, T, po, pv, pres, _p32(r), r.size, _p64(out_off), _p32(out_vals))
    return out_off, out_vals[: int(out_off[-1])].copy(), int(n)


def union(lists) -> np.ndarray:
    arrs = [_u32(a) for a in lists]
    n = len(arrs)
    total = sum(a.size for a in arrs)
    out = np.empty(max(total, 1), np.uint32)
    Ptrs = u32p * max(n, 1)
    Lens = C.c_size_t * max(n, 1)
    m = lib().orc_union(n, Ptrs(*[_p32(a) for a in arrs]), Lens(*[a.size for a in arrs]), _p32(out))
    return out[:m].copy()


def intersect(lists, removed_sorted=()) -> np.ndarray:
    arrs = [_u32(a) for a in lists]
    n = len(arrs)
    r = _u32(removed_sorted)
    cap = arrs[0].size if arrs else 0      # the fold starts from lists[0]
    out = np.empty(max(cap, 1), np.uint32)
    Ptrs = u32p * max(n, 1)
    Lens = C.c_size_t * max(n, 1)
    m = lib().orc_intersect(n, Ptrs(*[_p32(a) for a in arrs]), Lens(*[a.size for a in arrs]),
                            _p32(r), r.size, _p32(out))
    return out[:m].copy()


def shard_key(term: bytes) -> int:
    return lib().orc_shard_key(term, len(term))


SKIP_DTYPE = np.dtype([("first_doc", "<u4"), ("byte_off", "<u4")])


def dv1_encode(post_off, values):
    """Returns (blk_off u32[L+1], skip SKIP_DTYPE[NB+1], payload u8[n_bytes])."""
    po = np.ascontiguousarray(post_off, dtype=np.uint64)
    v = _u32(values)
    L = po.size - 1
    nbytes = C.c_uint64(0)
    nb = lib().orc_dv1_encode(L, _p64(po), _p32(v), None, None, None, C.byref(nbytes))
    blk_off = np.zeros(L + 1, np.uint32)
    skip = np.zeros(nb + 1, SKIP_DTYPE)
    payload = np.zeros(nbytes.value + 16, np.uint8)
    lib().orc_dv1_encode(L, _p64(po), _p32(v), _p32(blk_off), skip.ctypes.data_as(C.c_void_p),
                         payload.ctypes.data_as(u8p), C.byref(nbytes))
    return blk_off, skip, payload[: nbytes.value]


def dv1_decode(blk_off, skip, payload, n_postings_bound: int):
    bo = _u32(blk_off)
    sk = np.ascontiguousarray(skip, dtype=SKIP_DTYPE)
    pl = np.ascontiguousarray(payload, dtype=np.uint8)
    L = bo.size - 1
    po = np.zeros(L + 1, np.uint64)
    out = np.empty(max(n_postings_bound, 1), np.uint32)
    n = lib().orc_dv1_decode(L, _p32(bo), sk.ctypes.data_as(C.c_void_p), pl.ctypes.data_as(u8p), _p64(po), _p32(out))
    return po, out[:n].copy()
