"""Python face of the C++ host mirror (host/host_index.cpp, libii2_host.so): the reference's Shard and
InvertedIndex operations — Put / Read / Merge / Remove (PutRemoved) / PrefixSearch, plus the
additive Intersect — with every posting operation executed on the GPU through the C ABI.
Segments stay in HBM; terms are byte strings.  Errors surface as HostError with the
reference-style "s: merge: …" prefix."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib
from .engine import Context

vp = C.c_void_p
u64p = C.POINTER(C.c_uint64)


class HostError(RuntimeError):
    pass


_typed = False


_host = None
HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libii2_host.so")
# (sanitizer runs of the CPU-only file-layer tests load an instrumented build instead: make -C csrc host_asan / host_tsan)
HOST_LIB_PATH = os.environ.get("II2_HOST_LIB", HOST_LIB_PATH)


def _lib_typed():
    global _typed, _host
    if _host is None:
        _lib.load()                      # libii2_hip.so first: the host library links against it
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: build it with __graft_entry__.build()")
        _host = C.CDLL(HOST_LIB_PATH)
    lib = _host
    if not _typed:
        lib.ii2h_create.restype = vp
        lib.ii2h_create.argtypes = [vp, C.c_int]
        lib.ii2h_attach.restype = vp
        lib.ii2h_attach.argtypes = [vp]
        lib.ii2h_open.restype = vp
        lib.ii2h_open.argtypes = [vp, C.c_int, C.c_char_p, C.c_char_p, C.c_uint64]
        lib.ii2h_file_write.argtypes = [vp, vp, C.c_char_p, C.c_int, vp, vp, C.c_uint64, vp, vp, C.c_char_p]
        lib.ii2h_file_read.argtypes = [vp, vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_int, C.c_char_p, C.c_uint64, C.c_int, u64p]
        lib.ii2h_remove_segment.argtypes = [vp, C.c_char_p, C.c_char_p]
        lib.ii2h_terms_read.argtypes = [vp, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), u64p]
        lib.ii2h_removed_write.argtypes = [vp, C.c_char_p, C.c_uint64, vp, vp, vp]
        lib.ii2h_removed_read.argtypes = [vp, C.c_char_p, u64p, u64p]
        lib.ii2h_destroy.argtypes = [vp]
        lib.ii2h_last_error.restype = C.c_char_p
        lib.ii2h_last_error.argtypes = [vp]
        lib.ii2h_put.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32]
        lib.ii2h_remove.argtypes = [vp, vp, C.c_uint64]
        lib.ii2h_merge.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        lib.ii2h_read.argtypes = [vp, C.c_char_p, C.c_uint64, C.c_int, C.c_char_p, C.c_uint64, C.c_int, u64p]
        lib.ii2h_prefix_search.argtypes = [vp, vp, vp, C.c_uint64, u64p]
        lib.ii2h_intersect.argtypes = [vp, vp, vp, C.c_uint64, u64p]
        lib.ii2h_removed_values.argtypes = [vp, u64p]
        for f in ("ii2h_result_term_len", "ii2h_result_values_len"):
            getattr(lib, f).restype = C.c_uint64
            getattr(lib, f).argtypes = [vp, C.c_uint64]
        lib.ii2h_result_copy.argtypes = [vp, C.c_uint64, vp, vp]
        lib.ii2h_ids_copy.argtypes = [vp, vp]
        for f in ("ii2h_segment_count", "ii2h_shard_count"):
            getattr(lib, f).restype = C.c_uint64
            getattr(lib, f).argtypes = [vp]
        _typed = True
    return lib


def _pack(terms: List[bytes]):
    off = np.zeros(len(terms) + 1, np.uint64)
    if terms:
        off[1:] = np.cumsum([len(t) for t in terms])
    blob = np.frombuffer(b"".join(terms) + b"\0", dtype=np.uint8).copy()
    return blob, off


class _Target:
    def __init__(self, ctx: Context, is_index: bool, basedir: Optional[str] = None):
        self.lib = _lib_typed()
        self.ctx = ctx
        self.basedir = basedir
        if basedir is None:
            self.h = self.lib.ii2h_create(ctx.h if ctx is not None else None, 1 if is_index else 0)
        else:               # NewShard(basedir) / NewInvertedIndex(basedir): what the directory holds is loaded
            err = C.create_string_buffer(512)
            self.h = self.lib.ii2h_open(ctx.h, 1 if is_index else 0, os.fsencode(basedir), err, len(err))
            if not self.h:
                raise HostError(err.value.decode())

    def _ck(self, rc: int) -> None:
        if rc:
            raise HostError((self.lib.ii2h_last_error(self.h) or b"").decode())

    def session(self):
        """Another handle on the same shard / index for use from another thread: the operations are thread-safe (like the
        reference's goroutine-safe methods), a handle's result buffers are per handle."""
        other = object.__new__(type(self))
        other.lib, other.ctx, other.basedir = self.lib, self.ctx, self.basedir
        other.h = self.lib.ii2h_attach(self.h)
        return other

    def close(self) -> None:
        if self.h:
            self.lib.ii2h_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx is None or self.ctx.h:
                self.close()
        except Exception:
            pass

    def put(self, terms: List[bytes], val: int) -> None:
        blob, off = _pack(list(terms))
        self._ck(self.lib.ii2h_put(self.h, blob.ctypes.data, off.ctypes.data, len(terms), val))

    def _results(self, n: int) -> List[Tuple[bytes, List[int]]]:
        out = []
        for i in range(n):
            tl = self.lib.ii2h_result_term_len(self.h, i)
            vl = self.lib.ii2h_result_values_len(self.h, i)
            tb = np.zeros(max(tl, 1), np.uint8)
            vb = np.zeros(max(vl, 1), np.uint32)
            self.lib.ii2h_result_copy(self.h, i, tb.ctypes.data, vb.ctypes.data)
            out.append((tb[:tl].tobytes(), vb[:vl].tolist()))
        return out

    def read(self, lo: Optional[bytes] = None, hi: Optional[bytes] = None):
        n = C.c_uint64()
        self._ck(self.lib.ii2h_read(self.h, lo or b"", len(lo or b""), lo is not None, hi or b"", len(hi or b""), hi is not None,
                                    C.byref(n)))
        return self._results(n.value)

    def _merge(self, req: int, m: int, conc: int) -> int:
        out = C.c_int64()
        self._ck(self.lib.ii2h_merge(self.h, req, m, conc, C.byref(out)))
        return out.value

    def _remove(self, values) -> None:
        v = np.ascontiguousarray(values, dtype=np.uint32)
        self._ck(self.lib.ii2h_remove(self.h, v.ctypes.data, v.size))

    def removed_values(self) -> List[int]:
        n = C.c_uint64()
        self._ck(self.lib.ii2h_removed_values(self.h, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint32)
        self.lib.ii2h_ids_copy(self.h, out.ctypes.data)
        return out[: n.value].tolist()


class Shard(_Target):
    """shard.go: Put / Read / Remove / Merge.  With `basedir` the shard lives in that directory (segment files and
    removed.list, host/segment_file.h) and a new Shard on the same directory picks the state up (shard.go:300-358)."""

    def __init__(self, ctx: Context, basedir: Optional[str] = None):
        super().__init__(ctx, False, basedir)

    def merge(self, req_count: int, m_count: int) -> int:
        return self._merge(req_count, m_count, 1)

    def remove(self, values) -> None:
        self._remove(values)

    @property
    def n_segments(self) -> int:
        return self.lib.ii2h_segment_count(self.h)


class InvertedIndex(_Target):
    """inverted_index.go: Put / Read / Merge / PutRemoved / PrefixSearch (+ Intersect)."""

    def __init__(self, ctx: Context, basedir: Optional[str] = None):
        super().__init__(ctx, True, basedir)

    def merge(self, req_count: int, m_count: int, concurrency: int = 1) -> int:
        return self._merge(req_count, m_count, concurrency)

    def put_removed(self, values) -> None:
        self._remove(values)

    def prefix_search(self, prefixes: List[bytes]) -> Dict[bytes, List[int]]:
        blob, off = _pack(list(prefixes))
        n = C.c_uint64()
        self._ck(self.lib.ii2h_prefix_search(self.h, blob.ctypes.data, off.ctypes.data, len(prefixes), C.byref(n)))
        return dict(self._results(n.value))

    def intersect(self, terms: List[bytes]) -> List[int]:
        blob, off = _pack(list(terms))
        n = C.c_uint64()
        self._ck(self.lib.ii2h_intersect(self.h, blob.ctypes.data, off.ctypes.data, len(terms), C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint32)
        self.lib.ii2h_ids_copy(self.h, out.ctypes.data)
        return out[: n.value].tolist()

    @property
    def n_shards(self) -> int:
        return self.lib.ii2h_shard_count(self.h)


class SegmentFiles(_Target):
    """file.Writer / file.Reader / file.RemoveSegment (file/writer.go, file/reader.go) over host/segment_file.h: the
    encode and decode steps run on the device."""

    def __init__(self, ctx: Context):
        super().__init__(ctx, False)

    def write(self, directory: str, term_values: List[Tuple[bytes, List[int]]], direct: bool = False) -> str:
        """NewWriter / NewDirectWriter + Append for every (term, values) + Close; returns GetKey()."""
        blob, off = _pack([t for t, _ in term_values])
        po = np.zeros(len(term_values) + 1, np.uint64)
        if term_values:
            po[1:] = np.cumsum([len(v) for _, v in term_values])
        flat = np.ascontiguousarray([x for _, v in term_values for x in v] + [0], dtype=np.uint32)
        key = C.create_string_buffer(32)
        if self.ctx is None and not direct:
            raise HostError("writer: the encode step needs a device context")
        self._ck(self.lib.ii2h_file_write(self.h, self.ctx.h if self.ctx is not None else None, os.fsencode(directory), 1 if direct else 0,
                                          blob.ctypes.data, off.ctypes.data, len(term_values), po.ctypes.data, flat.ctypes.data, key))
        return key.value.decode()

    def write_arrays(self, directory: str, term_blob: np.ndarray, term_off: np.ndarray, post_off: np.ndarray, values: np.ndarray) -> str:
        """write() for callers that already hold flat arrays (terms: bytes + u64 offsets; lists: u64 offsets + u32 ids)."""
        key = C.create_string_buffer(32)
        term_blob = np.ascontiguousarray(term_blob, np.uint8)
        term_off = np.ascontiguousarray(term_off, np.uint64)
        post_off = np.ascontiguousarray(post_off, np.uint64)
        values = np.ascontiguousarray(values, np.uint32)
        self._ck(self.lib.ii2h_file_write(self.h, self.ctx.h, os.fsencode(directory), 0, term_blob.ctypes.data, term_off.ctypes.data,
                                          term_off.size - 1, post_off.ctypes.data, values.ctypes.data, key))
        return key.value.decode()

    def read_count(self, directory: str, key: str) -> Tuple[int, int]:
        """Reads and decodes a whole segment; returns (terms, postings) without copying the lists into Python."""
        n = C.c_uint64()
        self._ck(self.lib.ii2h_file_read(self.h, self.ctx.h, os.fsencode(directory), key.encode(), b"", 0, 0, b"", 0, 0, C.byref(n)))
        return n.value, sum(self.lib.ii2h_result_values_len(self.h, i) for i in range(0, n.value, max(n.value // 1000, 1)))

    def read(self, directory: str, key: str, lo: Optional[bytes] = None, hi: Optional[bytes] = None):
        """NewReader(dir, key, min, max) drained with Next(); None when the segment has nothing in range."""
        n = C.c_uint64()
        rc = self.lib.ii2h_file_read(self.h, self.ctx.h, os.fsencode(directory), key.encode(), lo or b"", len(lo or b""), lo is not None,
                                     hi or b"", len(hi or b""), hi is not None, C.byref(n))
        if rc == 1:
            return None
        self._ck(rc)
        return self._results(n.value)

    def remove(self, directory: str, key: str) -> None:
        self._ck(self.lib.ii2h_remove_segment(self.h, os.fsencode(directory), key.encode()))

    # ---- the parts of the file layer that involve no device (usable with ctx = None) ----
    def read_terms(self, directory: str, key: str):
        """The term dictionary file alone: (direct, [(term, [value] if direct else [])])."""
        n, direct = C.c_uint64(), C.c_int()
        self._ck(self.lib.ii2h_terms_read(self.h, os.fsencode(directory), key.encode(), C.byref(direct), C.byref(n)))
        return bool(direct.value), self._results(n.value)

    def write_removed(self, directory: str, batches: Dict[int, List[int]]) -> None:
        ts = np.ascontiguousarray(sorted(batches), dtype=np.int64)
        off = np.zeros(ts.size + 1, np.uint64)
        off[1:] = np.cumsum([len(batches[int(t)]) for t in ts])
        vals = np.ascontiguousarray([v for t in ts for v in batches[int(t)]] + [0], dtype=np.uint32)
        self._ck(self.lib.ii2h_removed_write(self.h, os.fsencode(directory), ts.size, ts.ctypes.data, off.ctypes.data, vals.ctypes.data))

    def read_removed(self, directory: str) -> Tuple[int, List[int]]:
        """(batches, RemovedLists.Values()) of the directory's removed.list; (0, []) when there is none."""
        nb, n = C.c_uint64(), C.c_uint64()
        self._ck(self.lib.ii2h_removed_read(self.h, os.fsencode(directory), C.byref(nb), C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint32)
        self.lib.ii2h_ids_copy(self.h, out.ctypes.data)
        return nb.value, out[: n.value].tolist()
