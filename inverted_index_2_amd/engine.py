"""Thin Python host layer over the C ABI (include/ii2.h): contexts, device segments,
tombstones and the three hot-path operators.  numpy arrays are host buffers; DeviceArray
(or any object with .data_ptr(), e.g. a torch CUDA tensor) is a device buffer.

Nothing here computes postings on the CPU — every operator is a call into libii2_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import II2_DEVICE, II2_HOST, MergeStats, SegInfo

SKIP_DTYPE = np.dtype([("first_doc", "<u4"), ("byte_off", "<u4")])


class II2Error(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ii2 error {code} ({_lib.ERRORS.get(code, '?')}): {msg}")
        self.code = code


def _np(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a) -> C.c_void_p:
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    if isinstance(a, int):
        return C.c_void_p(a)
    raise TypeError(type(a))


class DeviceArray:
    """A raw HBM buffer owned by a Context."""

    def __init__(self, ctx: "Context", count: int, dtype=np.uint32):
        self.ctx, self.count, self.dtype = ctx, int(count), np.dtype(dtype)
        p = C.c_void_p()
        ctx._ck(ctx.lib.ii2_dev_alloc(ctx.h, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    @property
    def nbytes(self) -> int:
        return self.count * self.dtype.itemsize

    def data_ptr(self) -> int:
        return self.ptr

    def upload(self, host) -> "DeviceArray":
        a = _np(host, self.dtype)
        assert a.size <= self.count
        if a.size:
            self.ctx._ck(self.ctx.lib.ii2_copy_h2d(self.ctx.h, self.ptr, _ptr(a), a.nbytes))
        return self

    def download(self, count: Optional[int] = None) -> np.ndarray:
        n = self.count if count is None else int(count)
        out = np.empty(n, self.dtype)
        if n:
            self.ctx._ck(self.ctx.lib.ii2_copy_d2h(self.ctx.h, _ptr(out), self.ptr, out.nbytes))
        return out

    def free(self) -> None:
        if self.ptr:
            self.ctx.lib.ii2_dev_free(self.ctx.h, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            if not sys.is_finalizing():      # at interpreter exit the HIP runtime may already be gone
                self.free()
        except Exception:
            pass


class Context:
    """One GPU + one HIP stream (ii2_ctx)."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.ii2_ctx_create(device, 0, C.byref(h))
        if rc:
            raise II2Error(rc, (self.lib.ii2_last_error(None) or b"").decode())
        self.h = h
        self.device = device
        # experiments: II2_OPTIONS="name=value,name=value" sets options on every context of the process (scripts, bench runs)
        for kv in filter(None, os.environ.get("II2_OPTIONS", "").split(",")):
            name, _, value = kv.partition("=")
            self.set_option(name.strip(), int(value))

    def _ck(self, rc: int) -> None:
        if rc:
            raise II2Error(rc, (self.lib.ii2_last_error(self.h) or b"").decode())

    def close(self) -> None:
        if self.h:
            self.lib.ii2_ctx_destroy(self.h)
            self.h = None

    def sync(self) -> None:
        self._ck(self.lib.ii2_ctx_sync(self.h))

    @property
    def stream(self) -> int:
        return self.lib.ii2_ctx_stream(self.h) or 0

    def set_option(self, name: str, value: int) -> None:
        self._ck(self.lib.ii2_set_option(self.h, name.encode(), int(value)))

    def counters(self):
        """(merges repeated on the packing path, look-back launches repeated on their second path, host waits inside the exchange
        entry points) - see ii2_ctx_counters."""
        out = (C.c_uint64 * 3)()
        self._ck(self.lib.ii2_ctx_counters(self.h, out, 3))
        return int(out[0]), int(out[1]), int(out[2])

    def profile_read(self):
        """(total device ms, launches) of the dominant kernel since the last read (option profile.events)."""
        ms, n = C.c_double(), C.c_uint64()
        self._ck(self.lib.ii2_profile_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_region(self, begin: bool) -> None:
        """Record the start (begin=True) / end event of a region on the ctx stream."""
        self._ck(self.lib.ii2_profile_region(self.h, 1 if begin else 0))

    def profile_region_ms(self) -> float:
        ms = C.c_double()
        self._ck(self.lib.ii2_profile_region_ms(self.h, C.byref(ms)))
        return ms.value

    def selftest(self) -> None:
        self._ck(self.lib.ii2_selftest(self.h))

    def empty(self, count: int, dtype=np.uint32) -> DeviceArray:
        return DeviceArray(self, count, dtype)

    # ---- segments -------------------------------------------------------------------------
    def encode(self, post_off, values, where: int = II2_HOST) -> "Segment":
        """Encode step (Writer.Append -> intcomp.CompressUint32, file/writer.go:32-59)."""
        if where == II2_HOST:
            post_off = _np(post_off, np.uint64)
            values = _np(values, np.uint32)
            n_lists = post_off.size - 1
        else:
            n_lists = post_off.count - 1
        s = C.c_void_p()
        self._ck(self.lib.ii2_seg_encode(self.h, n_lists, _ptr(post_off), _ptr(values), where, C.byref(s)))
        return Segment(self, s)

    def encode_lists(self, lists: Sequence) -> "Segment":
        arrs = [_np(l, np.uint32) for l in lists]
        po = np.zeros(len(arrs) + 1, np.uint64)
        if arrs:
            po[1:] = np.cumsum([a.size for a in arrs])
        flat = np.concatenate(arrs) if arrs else np.empty(0, np.uint32)
        return self.encode(po, flat)

    def import_dv1(self, n_postings: int, blk_off, skip, payload) -> "Segment":
        blk_off = _np(blk_off, np.uint32)
        skip = _np(skip, SKIP_DTYPE)
        payload = _np(payload, np.uint8)
        s = C.c_void_p()
        if blk_off.size < 1 or skip.size < 1:
            raise ValueError("import_dv1: blk_off and skip hold at least their closing entry")
        self._ck(self.lib.ii2_seg_import(self.h, blk_off.size - 1, n_postings, skip.size - 1, payload.size, _ptr(blk_off), _ptr(skip),
                                         _ptr(payload) if payload.size else None, II2_HOST, C.byref(s)))
        return Segment(self, s)

    def tombstones(self, removed, where: int = II2_HOST) -> "Tombstones":
        """RemovedLists.Values() as a device bitmap (removed_list.go:44-54, shard.go:183)."""
        if where == II2_HOST:
            removed = _np(removed, np.uint32)
            n = removed.size
        else:
            n = removed.count
        t = C.c_void_p()
        self._ck(self.lib.ii2_tomb_create(self.h, _ptr(removed), n, where, C.byref(t)))
        return Tombstones(self, t)

    # ---- term alignment on the device -------------------------------------------------------
    def align_terms(self, dictionaries) -> "Alignment":
        """k sorted, duplicate-free term dictionaries (lists of bytes) -> their union, on the device (the k-way term walk of
        makeIterator, shard.go:253-278, in bytes.Compare order)."""
        flat = [t for d in dictionaries for t in d]
        off = np.zeros(len(flat) + 1, np.uint64)
        if flat:
            off[1:] = np.cumsum([len(t) for t in flat])
        blob = np.frombuffer(b"".join(flat) + b"\0", dtype=np.uint8).copy()
        first = np.zeros(len(dictionaries) + 1, np.uint64)
        first[1:] = np.cumsum([len(d) for d in dictionaries])
        h = C.c_void_p()
        self._ck(self.lib.ii2_align_terms(self.h, len(dictionaries), _ptr(blob), _ptr(off), _ptr(first), C.byref(h)))
        return Alignment(self, h, flat)

    def align_terms_flat(self, term_bytes: np.ndarray, term_off: np.ndarray, seg_first: np.ndarray) -> "Alignment":
        """Same, from flat arrays (u8 bytes, u64 offsets [n_all + 1], u64 first term of each dictionary [k + 1])."""
        blob, off, first = _np(term_bytes, np.uint8), _np(term_off, np.uint64), _np(seg_first, np.uint64)
        h = C.c_void_p()
        self._ck(self.lib.ii2_align_terms(self.h, first.size - 1, _ptr(blob), _ptr(off), _ptr(first), C.byref(h)))
        return Alignment(self, h, None)

    def dictionary(self, terms) -> "Dictionary":
        """A segment's sorted, duplicate-free term dictionary (list of bytes), made resident in HBM (ii2_dict_create)."""
        terms = list(terms)
        off = np.zeros(len(terms) + 1, np.uint64)
        if terms:
            off[1:] = np.cumsum([len(t) for t in terms])
        blob = np.frombuffer(b"".join(terms) + b"\0", dtype=np.uint8).copy()
        return self.dictionary_flat(blob, off, terms)

    def dictionary_flat(self, term_bytes, term_off, terms=None) -> "Dictionary":
        """Same, from flat arrays: u8 bytes and u64 offsets [n + 1] starting at 0."""
        blob, off = _np(term_bytes, np.uint8), _np(term_off, np.uint64)
        h = C.c_void_p()
        self._ck(self.lib.ii2_dict_create(self.h, _ptr(blob), _ptr(off), off.size - 1, II2_HOST, C.byref(h)))
        return Dictionary(self, h, terms)

    def align_dicts(self, dicts) -> "Alignment":
        """The union of k resident dictionaries and every dictionary's place in it (ii2_align_dicts): no upload, no sort."""
        arr = (C.c_void_p * len(dicts))(*[d.h for d in dicts])
        h = C.c_void_p()
        self._ck(self.lib.ii2_align_dicts(self.h, len(dicts), arr, C.byref(h)))
        flat = None
        if all(d.terms is not None for d in dicts):
            flat = [t for d in dicts for t in d.terms]
        return Alignment(self, h, flat)

    def select_aligned(self, seg: "Segment", alignment: "Alignment", s: int, first_list: int = 0) -> "Segment":
        out = C.c_void_p()
        self._ck(self.lib.ii2_seg_select_aligned(self.h, seg.h, alignment.h, s, first_list, C.byref(out)))
        return Segment(self, out)

    def select_aligned_all(self, segs: Sequence["Segment"], alignment: "Alignment", first_list=None) -> List["Segment"]:
        """The aligned views of all the alignment's dictionaries in one call (ii2_seg_select_aligned_all)."""
        arr = (C.c_void_p * len(segs))(*[s.h for s in segs])
        outs = (C.c_void_p * len(segs))()
        fl = _np(first_list, np.uint64) if first_list is not None else None
        self._ck(self.lib.ii2_seg_select_aligned_all(self.h, arr, alignment.h, _ptr(fl) if fl is not None else None, outs))
        return [Segment(self, C.c_void_p(h)) for h in outs]

    def select(self, seg: "Segment", src_list) -> "Segment":
        src = _np(src_list, np.int64)
        out = C.c_void_p()
        self._ck(self.lib.ii2_seg_select(self.h, seg.h, src.size, _ptr(src), C.byref(out)))
        return Segment(self, out)

    # ---- operators ------------------------------------------------------------------------
    def _listargs(self, lists):
        segs = (C.c_void_p * len(lists))(*[s.h for s, _ in lists])
        idx = (C.c_uint64 * len(lists))(*[int(i) for _, i in lists])
        return segs, idx

    def intersect(self, lists, tomb: Optional["Tombstones"] = None, out: Optional[DeviceArray] = None,
                  cap: Optional[int] = None):
        """lists: [(Segment, list_index), ...].  Returns (DeviceArray ids, count)."""
        segs, idx = self._listargs(lists)
        if cap is None:
            cap = min(s.list_blocks(i, self) for s, i in lists) * 256 if out is None else out.count
        if out is None:
            out = self.empty(max(cap, 1))
        cnt = C.c_uint64()
        self._ck(self.lib.ii2_intersect(self.h, len(lists), segs, idx, tomb.h if tomb else None, _ptr(out), cap, C.byref(cnt)))
        return out, cnt.value

    def intersect_async(self, lists, tomb, out: DeviceArray, d_count: DeviceArray) -> None:
        segs, idx = self._listargs(lists)
        self._ck(self.lib.ii2_intersect_async(self.h, len(lists), segs, idx, tomb.h if tomb else None, _ptr(out), out.count,
                                              _ptr(d_count)))

    def union(self, lists, tomb: Optional["Tombstones"] = None, out: Optional[DeviceArray] = None):
        """PrefixSearch's append + sort + compact (inverted_index.go:274-292)."""
        segs, idx = self._listargs(lists)
        if out is None:
            out = self.empty(max(sum(s.list_blocks(i, self) for s, i in lists) * 256, 1))
        cnt = C.c_uint64()
        self._ck(self.lib.ii2_union(self.h, len(lists), segs, idx, tomb.h if tomb else None, _ptr(out), out.count, C.byref(cnt)))
        return out, cnt.value

    def merge(self, segs: Sequence["Segment"], tomb: Optional["Tombstones"] = None,
              out_off: Optional[DeviceArray] = None, out_values: Optional[DeviceArray] = None):
        """Shard.Merge's loop body (shard.go:163-212).  Returns (out_off u64[T+1], out_values, MergeStats)."""
        T = segs[0].info.n_lists
        if out_off is None:
            out_off = self.empty(T + 1, np.uint64)
        if out_values is None:
            out_values = self.empty(max(sum(s.info.n_postings for s in segs), 1))
        arr = (C.c_void_p * len(segs))(*[s.h for s in segs])
        st = MergeStats()
        self._ck(self.lib.ii2_merge_segments(self.h, len(segs), arr, tomb.h if tomb else None, _ptr(out_off), _ptr(out_values),
                                             out_values.count, C.byref(st)))
        return out_off, out_values, st

    def merge_to_segment(self, segs: Sequence["Segment"], tomb: Optional["Tombstones"] = None):
        arr = (C.c_void_p * len(segs))(*[s.h for s in segs])
        st = MergeStats()
        out = C.c_void_p()
        self._ck(self.lib.ii2_merge_segments_to_seg(self.h, len(segs), arr, tomb.h if tomb else None, C.byref(out), C.byref(st)))
        return (Segment(self, out) if out.value else None), st

    # ---- host-buffer calls (what the cgo binding uses) ---------------------------------------
    def merge_small(self, segs: Sequence["Segment"], dictionaries, removed=()):
        """The common Shard.Merge in one launch (ii2_merge_small): k small segments with their term dictionaries (lists of
        bytes, segs[s] holding one list per term of dictionaries[s]) and RemovedLists.Values().  Returns (merged Segment or
        None, its terms, MergeStats)."""
        flat = [t for d in dictionaries for t in d]
        off = np.zeros(len(flat) + 1, np.uint64)
        if flat:
            off[1:] = np.cumsum([len(t) for t in flat])
        blob = np.frombuffer(b"".join(flat) + b"\0", dtype=np.uint8).copy()
        first = np.zeros(len(dictionaries) + 1, np.uint64)
        first[1:] = np.cumsum([len(d) for d in dictionaries])
        rem = _np(removed, np.uint32)
        arr = (C.c_void_p * len(segs))(*[s.h for s in segs])
        out = C.c_void_p()
        kept = np.zeros(max(len(flat), 1), np.uint64)
        n_kept = C.c_uint64()
        st = MergeStats()
        self._ck(self.lib.ii2_merge_small(self.h, len(segs), arr, _ptr(blob), _ptr(off), _ptr(first), _ptr(rem) if rem.size else None, rem.size,
                                          C.byref(out), kept.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(n_kept), C.byref(st)))
        terms = [flat[int(g)] for g in kept[: n_kept.value]]
        return (Segment(self, out) if out.value else None), terms, st

    def read_small(self, segs: Sequence["Segment"], dictionaries, list_first=None):
        """The small Shard.Read in one launch (ii2_read_small): the merged lists of k small segments (dictionaries[s] names the
        lists list_first[s] .. of segs[s]) on the host.  Returns (terms, post_off, values)."""
        flat = [t for d in dictionaries for t in d]
        off = np.zeros(len(flat) + 1, np.uint64)
        if flat:
            off[1:] = np.cumsum([len(t) for t in flat])
        blob = np.frombuffer(b"".join(flat) + b"\0", dtype=np.uint8).copy()
        first = np.zeros(len(dictionaries) + 1, np.uint64)
        first[1:] = np.cumsum([len(d) for d in dictionaries])
        lf = _np(list_first, np.uint64) if list_first is not None else None
        arr = (C.c_void_p * len(segs))(*[s.h for s in segs])
        cap = int(sum(s.info.n_postings for s in segs))
        rep = np.zeros(max(len(flat), 1), np.uint64)
        post_off = np.zeros(len(flat) + 1, np.uint64)
        values = np.zeros(max(cap, 1), np.uint32)
        n_union = C.c_uint64()
        self._ck(self.lib.ii2_read_small(self.h, len(segs), arr, _ptr(blob), _ptr(off), _ptr(first), _ptr(lf) if lf is not None else None,
                                         rep.ctypes.data_as(C.POINTER(C.c_uint64)), post_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                         _ptr(values), cap, C.byref(n_union)))
        nu = n_union.value
        return [flat[int(g)] for g in rep[:nu]], post_off[: nu + 1].copy(), values[: int(post_off[nu])].copy()

    def merge_host(self, seg_offs, seg_vals, removed=()):
        k = len(seg_offs)
        offs = [_np(o, np.uint64) for o in seg_offs]
        vals = [_np(v, np.uint32) for v in seg_vals]
        T = offs[0].size - 1
        flat_off = np.concatenate(offs) if k else np.zeros(0, np.uint64)
        base = np.zeros(k + 1, np.uint64)
        base[1:] = np.cumsum([v.size for v in vals])
        flat = np.concatenate(vals) if k else np.empty(0, np.uint32)
        rem = _np(removed, np.uint32)
        out_off = np.zeros(T + 1, np.uint64)
        out_vals = np.empty(max(flat.size, 1), np.uint32)
        st = MergeStats()
        self._ck(self.lib.ii2_merge_host(self.h, k, T, _ptr(flat_off), _ptr(base), _ptr(flat), _ptr(rem) if rem.size else None,
                                         rem.size, _ptr(out_off), _ptr(out_vals), out_vals.size, C.byref(st)))
        return out_off, out_vals[: int(out_off[-1])].copy(), st

    def _flat_lists(self, lists):
        arrs = [_np(l, np.uint32) for l in lists]
        off = np.zeros(len(arrs) + 1, np.uint64)
        if arrs:
            off[1:] = np.cumsum([a.size for a in arrs])
        flat = np.concatenate(arrs) if arrs else np.empty(0, np.uint32)
        return off, _np(flat, np.uint32)

    def intersect_host(self, lists, removed=()) -> np.ndarray:
        off, flat = self._flat_lists(lists)
        rem = _np(removed, np.uint32)
        cap = max(min((int(off[i + 1] - off[i]) for i in range(len(lists))), default=0), 1)
        out = np.empty(cap, np.uint32)
        cnt = C.c_uint64()
        self._ck(self.lib.ii2_intersect_host(self.h, len(lists), _ptr(off), _ptr(flat), _ptr(rem) if rem.size else None, rem.size,
                                             _ptr(out), cap, C.byref(cnt)))
        return out[: cnt.value].copy()

    def union_host(self, lists, removed=()) -> np.ndarray:
        off, flat = self._flat_lists(lists)
        rem = _np(removed, np.uint32)
        out = np.empty(max(flat.size, 1), np.uint32)
        cnt = C.c_uint64()
        self._ck(self.lib.ii2_union_host(self.h, len(lists), _ptr(off), _ptr(flat), _ptr(rem) if rem.size else None, rem.size,
                                         _ptr(out), out.size, C.byref(cnt)))
        return out[: cnt.value].copy()

    # ---- multi-GPU ---------------------------------------------------------------------------
    def comm_init(self, world: int, rank: int, unique_id: bytes) -> None:
        buf = C.create_string_buffer(unique_id, _lib.II2_UNIQUE_ID_BYTES)
        self._ck(self.lib.ii2_comm_init(self.h, world, rank, buf))

    def allgatherv(self, local: DeviceArray, n_local: int, out: DeviceArray, world: int):
        counts = (C.c_uint64 * max(world, 1))()
        self._ck(self.lib.ii2_allgatherv(self.h, _ptr(local), n_local, _ptr(out), out.count, counts))
        return [int(c) for c in counts]

    def allgatherv_bytes(self, local, n_bytes: int, out: DeviceArray, world: int):
        """Rank-order concatenation of byte arrays (any device buffers); returns the ranks' byte counts."""
        counts = (C.c_uint64 * max(world, 1))()
        self._ck(self.lib.ii2_allgatherv_bytes(self.h, _ptr(local), n_bytes, _ptr(out), out.nbytes, counts))
        return [int(c) for c in counts]

    def seg_concat(self, segs: Sequence["Segment"]) -> "Segment":
        """The lists of segs[0], then segs[1], ... as one segment on this device (ii2_seg_concat)."""
        arr = (C.c_void_p * len(segs))(*[s.h for s in segs])
        h = C.c_void_p()
        self._ck(self.lib.ii2_seg_concat(self.h, len(segs), arr, C.byref(h)))
        return Segment(self, h)

    def seg_allgather(self, local: "Segment") -> "Segment":
        """Every rank's merged segment, concatenated in rank order into one segment (the postings travel DV1-encoded)."""
        h = C.c_void_p()
        self._ck(self.lib.ii2_seg_allgather(self.h, local.h, C.byref(h)))
        return Segment(self, h)


def seg_gather_plan(shapes):
    """Host arithmetic of seg_allgather: shapes = [(n_lists, n_blocks, n_bytes)] per rank -> (rc, list_off, block_off, byte_off)."""
    world = len(shapes)
    flat = (C.c_uint64 * (3 * max(world, 1)))(*[int(x) for sh in shapes for x in sh])
    lo, bo, qo = ((C.c_uint64 * (world + 1))() for _ in range(3))
    rc = _lib.load().ii2_seg_gather_plan(flat, world, lo, bo, qo)
    return rc, list(lo), list(bo), list(qo)


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(_lib.II2_UNIQUE_ID_BYTES)
    rc = _lib.load().ii2_comm_unique_id(buf)
    if rc:
        raise II2Error(rc, "ncclGetUniqueId failed")
    return buf.raw


class Segment:
    """Device-resident DV1 segment (ii2_seg)."""

    def __init__(self, ctx: Context, h: C.c_void_p):
        self.ctx, self.h = ctx, h
        info = SegInfo()
        ctx.lib.ii2_seg_get_info(h, C.byref(info))
        self.info = info
        self._blk_off = None

    def list_blocks(self, i: int, ctx: Optional["Context"] = None) -> int:
        """Blocks of list i.  A segment belongs to its device, not to the context that made it: any live context works."""
        if self._blk_off is None:
            c = ctx if ctx is not None and ctx.h else self.ctx
            blk = np.zeros(self.info.n_lists + 1, np.uint32)
            c._ck(c.lib.ii2_seg_export(c.h, self.h, _ptr(blk), None, None))
            self._blk_off = blk
        return int(self._blk_off[i + 1] - self._blk_off[i])

    def decode(self):
        """Decode step (Reader.Next -> intcomp.UncompressUint32, file/reader.go:79-100)."""
        po = np.zeros(self.info.n_lists + 1, np.uint64)
        vals = np.empty(max(self.info.n_postings, 1), np.uint32)
        self.ctx._ck(self.ctx.lib.ii2_seg_decode(self.ctx.h, self.h, _ptr(po), _ptr(vals), II2_HOST))
        return po, vals[: self.info.n_postings]

    def export(self):
        blk = np.zeros(self.info.n_lists + 1, np.uint32)
        skip = np.zeros(self.info.n_blocks + 1, SKIP_DTYPE)
        payload = np.zeros(max(self.info.n_bytes, 1), np.uint8)
        self.ctx._ck(self.ctx.lib.ii2_seg_export(self.ctx.h, self.h, _ptr(blk), _ptr(skip), _ptr(payload)))
        return blk, skip, payload[: self.info.n_bytes]

    def free(self) -> None:
        if self.h:
            self.ctx.lib.ii2_seg_free(self.h)      # needs no context: the segment knows its device
            self.h = None

    def __del__(self):
        try:
            if not sys.is_finalizing():      # at interpreter exit the HIP runtime may already be gone
                self.free()
        except Exception:
            pass


class Dictionary:
    """A term dictionary resident in HBM (ii2_dict)."""

    def __init__(self, ctx: Context, h: C.c_void_p, terms=None):
        self.ctx, self.h, self.terms = ctx, h, terms

    def free(self) -> None:
        if self.h:
            self.ctx.lib.ii2_dict_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if not sys.is_finalizing():
                self.free()
        except Exception:
            pass


class Alignment:
    """Device-resident result of Context.align_terms / align_dicts (ii2_align)."""

    def __init__(self, ctx: Context, h: C.c_void_p, flat_terms):
        self.ctx, self.h, self._flat = ctx, h, flat_terms
        n, k = C.c_uint64(), C.c_uint32()
        ctx.lib.ii2_align_info(h, C.byref(n), C.byref(k))
        self.n_union, self.k = n.value, k.value

    def export(self):
        """(union terms as bytes, src_list int64 [k][n_union])."""
        rep = np.zeros(max(self.n_union, 1), np.uint64)
        src = np.zeros(max(self.k * self.n_union, 1), np.int64)
        self.ctx._ck(self.ctx.lib.ii2_align_export(self.ctx.h, self.h, _ptr(rep), _ptr(src)))
        terms = [self._flat[int(g)] for g in rep[: self.n_union]] if self._flat is not None else rep[: self.n_union].copy()
        return terms, src[: self.k * self.n_union].reshape(self.k, self.n_union)

    def free(self) -> None:
        if self.h:
            self.ctx.lib.ii2_align_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if not sys.is_finalizing():
                self.free()
        except Exception:
            pass


class Tombstones:
    def __init__(self, ctx: Context, h: C.c_void_p):
        self.ctx, self.h = ctx, h

    def free(self) -> None:
        if self.h:
            self.ctx.lib.ii2_tomb_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h and not sys.is_finalizing():
                self.free()
        except Exception:
            pass
