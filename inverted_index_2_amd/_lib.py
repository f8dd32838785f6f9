"""ctypes binding of libii2_hip.so — exactly the entry points include/ii2.h declares.

The library is the product; this module only loads it.  It fails loudly when the shared
object is missing (run `python -c "import __graft_entry__ as g; g.build()"`); there is no
Python / CPU fallback for any operation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libii2_hip.so")

II2_HOST, II2_DEVICE = 0, 1
II2_UNIQUE_ID_BYTES = 128
ERRORS = {0: "OK", -1: "EINVAL", -2: "ENOMEM", -3: "EHIP", -4: "ECAPACITY", -5: "ERANGE", -6: "ECOMM", -7: "ENODEVICE"}


class SegInfo(C.Structure):
    _fields_ = [("n_lists", C.c_uint64), ("n_postings", C.c_uint64), ("n_blocks", C.c_uint64), ("n_bytes", C.c_uint64)]


class MergeStats(C.Structure):
    _fields_ = [("n_in", C.c_uint64), ("n_out", C.c_uint64), ("n_terms_out", C.c_uint64), ("n_tiles", C.c_uint64)]


vp = C.c_void_p
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol ii2.h declares
PROTOTYPES = {
    "ii2_abi_version": (C.c_int, []),
    "ii2_ctx_create": (C.c_int, [C.c_int, C.c_uint32, vpp]),
    "ii2_ctx_destroy": (None, [vp]),
    "ii2_last_error": (C.c_char_p, [vp]),
    "ii2_ctx_sync": (C.c_int, [vp]),
    "ii2_ctx_stream": (vp, [vp]),
    "ii2_ctx_device": (C.c_int, [vp]),
    "ii2_dev_alloc": (C.c_int, [vp, C.c_size_t, vpp]),
    "ii2_dev_free": (C.c_int, [vp, vp]),
    "ii2_copy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ii2_copy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ii2_seg_encode": (C.c_int, [vp, C.c_uint64, vp, vp, C.c_int, vpp]),
    "ii2_seg_import": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_int, vpp]),
    "ii2_seg_decode": (C.c_int, [vp, vp, vp, vp, C.c_int]),
    "ii2_seg_export": (C.c_int, [vp, vp, vp, vp, vp]),
    "ii2_seg_select": (C.c_int, [vp, vp, C.c_uint64, vp, vpp]),
    "ii2_dict_create": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, vpp]),
    "ii2_dict_free": (None, [vp]),
    "ii2_align_dicts": (C.c_int, [vp, C.c_uint32, vpp, vpp]),
    "ii2_align_terms": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vpp]),
    "ii2_align_info": (C.c_int, [vp, u64p, C.POINTER(C.c_uint32)]),
    "ii2_align_export": (C.c_int, [vp, vp, vp, vp]),
    "ii2_seg_select_aligned": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint64, vpp]),
    "ii2_seg_select_aligned_all": (C.c_int, [vp, vpp, vp, vp, vpp]),
    "ii2_align_free": (None, [vp]),
    "ii2_seg_get_info": (C.c_int, [vp, C.POINTER(SegInfo)]),
    "ii2_seg_free": (None, [vp]),
    "ii2_tomb_create": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vpp]),
    "ii2_tomb_free": (None, [vp]),
    "ii2_merge_segments": (C.c_int, [vp, C.c_uint32, vpp, vp, vp, vp, C.c_uint64, C.POINTER(MergeStats)]),
    "ii2_merge_segments_to_seg": (C.c_int, [vp, C.c_uint32, vpp, vp, vpp, C.POINTER(MergeStats)]),
    "ii2_merge_small": (C.c_int, [vp, C.c_uint32, vpp, vp, vp, vp, vp, C.c_uint64, vpp, u64p, u64p, C.POINTER(MergeStats)]),
    "ii2_devmem_stats": (None, [u64p, u64p]),
    "ii2_read_small": (C.c_int, [vp, C.c_uint32, vpp, vp, vp, vp, vp, u64p, u64p, vp, C.c_uint64, u64p]),
    "ii2_intersect": (C.c_int, [vp, C.c_uint32, vpp, u64p, vp, vp, C.c_uint64, u64p]),
    "ii2_intersect_async": (C.c_int, [vp, C.c_uint32, vpp, u64p, vp, vp, C.c_uint64, vp]),
    "ii2_union": (C.c_int, [vp, C.c_uint32, vpp, u64p, vp, vp, C.c_uint64, u64p]),
    "ii2_merge_host": (C.c_int, [vp, C.c_uint32, C.c_uint64, vp, vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.POINTER(MergeStats)]),
    "ii2_intersect_host": (C.c_int, [vp, C.c_uint32, vp, vp, vp, C.c_uint64, vp, C.c_uint64, u64p]),
    "ii2_union_host": (C.c_int, [vp, C.c_uint32, vp, vp, vp, C.c_uint64, vp, C.c_uint64, u64p]),
    "ii2_comm_unique_id": (C.c_int, [vp]),
    "ii2_comm_init": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "ii2_allgatherv": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64, u64p]),
    "ii2_allgatherv_bytes": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64, u64p]),
    "ii2_seg_allgather": (C.c_int, [vp, vp, vpp]),
    "ii2_seg_concat": (C.c_int, [vp, C.c_uint32, vpp, vpp]),
    "ii2_seg_gather_plan": (C.c_int, [u64p, C.c_int, u64p, u64p, u64p]),
    "ii2_gatherv_offsets": (C.c_int, [u64p, C.c_int, C.c_uint64, u64p]),
    "ii2_selftest": (C.c_int, [vp]),
    "ii2_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "ii2_ctx_counters": (C.c_int, [vp, u64p, C.c_uint32]),
    "ii2_debug_read": (C.c_int, [vp, u64p, C.c_uint64]),
    "ii2_profile_read": (C.c_int, [vp, C.POINTER(C.c_double), u64p]),
    "ii2_profile_region": (C.c_int, [vp, C.c_int]),
    "ii2_profile_region_ms": (C.c_int, [vp, C.POINTER(C.c_double)]),
}

_lib = None


def load() -> C.CDLL:
    """Loads libii2_hip.so and types every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                "(make -C inverted_index_2_amd/csrc). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
