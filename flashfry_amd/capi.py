"""ctypes binding of the C ABI in include/flashfry_hip.h (flashfry_amd/lib/libflashfry_hip.so).

This is plumbing only: every call goes straight into the HIP library.  There is NO CPU fallback -- loading fails
loudly if the library has not been built, and `Context()` fails loudly when no GPU is usable."""
import ctypes as C
import os

import numpy as np

from . import _build

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
i64p = C.POINTER(C.c_int64)

FINALIZE_SUMMARIES_ONLY = 1
FINALIZE_JOST = 2
FINALIZE_PRIOR_ON_DEVICE = 4
FINALIZE_NO_POSITIONS = 8
FINALIZE_NO_HIT_SCORES = 16

# every symbol include/flashfry_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "ffh_version", "ffh_device_count", "ffh_debug_pool_errors", "ffh_create", "ffh_destroy", "ffh_last_error", "ffh_db_load_blocks",
    "ffh_db_load_soa", "ffh_db_open", "ffh_db_open_header", "ffh_db_bin_bytes", "ffh_db_info_get", "ffh_db_load_stats", "ffh_host_threads", "ffh_db_write", "ffh_indexer_create", "ffh_indexer_destroy", "ffh_indexer_last_error",
    "ffh_indexer_add_contig", "ffh_indexer_finish", "ffh_db_contig", "ffh_set_plan", "ffh_scan",
    "ffh_scan_bounded", "ffh_set_bounding", "ffh_get_bounding", "ffh_discover_bulge", "ffh_bulge_result_n_guides", "ffh_bulge_result_n_hits", "ffh_bulge_result_guide_offsets",
    "ffh_bulge_result_hit_targets", "ffh_bulge_result_hit_mismatches", "ffh_bulge_result_hit_bulge_type", "ffh_bulge_result_hit_bulge_position",
    "ffh_bulge_result_free", "ffh_exchange_pack", "ffh_exchange_mask", "ffh_exchange_unpack", "ffh_use_stream", "ffh_finalize_shard", "ffh_exchange_prior", "ffh_finalize_shard_fixup", "ffh_shard_totals", "ffh_shard_totals_device", "ffh_summaries_to_device", "ffh_finalize", "ffh_discover", "ffh_score_lists", "ffh_result_n_guides", "ffh_result_n_hits",
    "ffh_result_n_positions", "ffh_result_scores_valid", "ffh_result_summaries", "ffh_result_guide_offsets",
    "ffh_result_hit_targets", "ffh_result_hit_mismatches", "ffh_result_hit_cfd", "ffh_result_pos_offsets",
    "ffh_result_positions", "ffh_result_free", "ffh_get_timings",
    "ffh_comm_unique_id", "ffh_comm_create_rank", "ffh_comm_create_local", "ffh_comm_destroy", "ffh_comm_last_error", "ffh_comm_world",
    "ffh_comm_first_shard", "ffh_comm_local_shards", "ffh_comm_transport", "ffh_discover_sharded", "ffh_comm_exchange", "ffh_comm_shard_lists",
    "ffh_comm_device_summaries", "ffh_comm_timings", "ffh_comm_set_exchange", "ffh_comm_get_exchange", "ffh_ctx_share_db", "ffh_pipe_create", "ffh_pipe_submit", "ffh_pipe_wait",
    "ffh_pipe_last_error", "ffh_pipe_lanes", "ffh_pipe_destroy", "ffh_host_alloc", "ffh_host_free",
]


class GuideSummary(C.Structure):
    _fields_ = [("n_hits", C.c_uint32), ("ot_count", C.c_uint32), ("overflow", C.c_uint32), ("hist", C.c_uint32 * 5),
                ("closest", C.c_uint32), ("closest_count", C.c_uint32), ("in_genome", C.c_uint32), ("n_scored", C.c_uint32),
                ("cfd_max", C.c_double), ("cfd_sum", C.c_double), ("hsu_sum", C.c_double), ("jost_max", C.c_double), ("jost_sum", C.c_double)]


SUMMARY_DTYPE = np.dtype([("n_hits", "<u4"), ("ot_count", "<u4"), ("overflow", "<u4"), ("hist", "<u4", (5,)),
                          ("closest", "<u4"), ("closest_count", "<u4"), ("in_genome", "<u4"), ("n_scored", "<u4"),
                          ("cfd_max", "<f8"), ("cfd_sum", "<f8"), ("hsu_sum", "<f8"), ("jost_max", "<f8"), ("jost_sum", "<f8")])
assert SUMMARY_DTYPE.itemsize == C.sizeof(GuideSummary) == 88


class DbInfo(C.Structure):
    _fields_ = [("n_targets", C.c_uint64), ("n_positions", C.c_uint64), ("n_bins", C.c_uint32), ("bin_begin", C.c_uint32),
                ("bin_end", C.c_uint32), ("enzyme_index", C.c_int), ("prefix_bases", C.c_int), ("suffix_bases", C.c_int),
                ("prepare_ms", C.c_double)]


class LoadStats(C.Structure):
    _fields_ = [("open_ms", C.c_double), ("inflate_ms", C.c_double), ("decode_ms", C.c_double), ("prepare_ms", C.c_double),
                ("compressed_bytes", C.c_uint64), ("raw_bytes", C.c_uint64), ("threads", C.c_uint32), ("reserved", C.c_uint32),
                ("device_inflate_ms", C.c_double), ("alloc_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class IndexStats(C.Structure):
    _fields_ = [("n_bases", C.c_uint64), ("n_sites", C.c_uint64), ("n_targets", C.c_uint64), ("n_positions", C.c_uint64),
                ("n_contigs", C.c_uint32), ("reserved", C.c_uint32), ("scan_ms", C.c_double), ("sort_ms", C.c_double), ("write_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class Timings(C.Structure):
    _fields_ = [("prepare_ms", C.c_double), ("compare_ms", C.c_double),
                ("sort_ms", C.c_double), ("finalize_ms", C.c_double), ("total_scan_ms", C.c_double),
                ("n_raw_hits", C.c_uint64), ("pairs_prefix", C.c_uint64), ("pairs_suffix", C.c_uint64),
                ("items_prefix", C.c_uint64), ("items_suffix", C.c_uint64), ("tiles_prefix", C.c_uint64),
                ("tiles_suffix", C.c_uint64), ("compare_launches", C.c_uint32), ("prefix_bases", C.c_int),
                ("prefix_radius", C.c_int), ("suffix_radius", C.c_int), ("bounded_slabs", C.c_uint32), ("retired_guides", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class FlashFryHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("flashfry_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def library_path():
    return _build.LIB


def load_library(build=True):
    """dlopen the HIP library (building it first if the sources are newer).  Raises if it cannot be had."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels carry their own libamdhip64 / libhsa-runtime64.  Whichever HIP runtime is loaded first serves the
    # whole process (same SONAME); a second HSA runtime cannot open the GPU ("No HIP GPUs are available").  When torch is
    # part of the process (tests, bench.py, torch.distributed) it therefore has to be loaded before this library.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # FFH_LIBRARY names another build of the same library (A/B comparisons of kernel variants on one box)
    path = os.environ.get("FFH_LIBRARY") or (_build.build_hip_library() if build else _build.LIB)
    if not os.path.exists(path):
        raise ImportError("libflashfry_hip.so is missing (%s): build it with `python -m flashfry_amd._build`; "
                          "there is no CPU fallback" % path)
    L = C.CDLL(path)
    L.ffh_version.restype = C.c_int
    L.ffh_device_count.restype = C.c_int
    if hasattr(L, "ffh_debug_pool_errors"):   # (absent from A/B builds of earlier revisions: FFH_LIBRARY)
        L.ffh_debug_pool_errors.restype = C.c_uint64
    L.ffh_create.restype = C.c_void_p
    L.ffh_create.argtypes = [C.c_int, C.c_int]
    L.ffh_destroy.argtypes = [C.c_void_p]
    L.ffh_last_error.restype = C.c_char_p
    L.ffh_last_error.argtypes = [C.c_void_p]
    L.ffh_db_load_blocks.argtypes = [C.c_void_p, i64p, u64p, C.c_uint32]
    L.ffh_db_load_soa.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
    L.ffh_db_open.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32]
    L.ffh_db_open_header.argtypes = [C.c_void_p, C.c_char_p]
    L.ffh_db_bin_bytes.restype = C.c_uint64
    L.ffh_db_bin_bytes.argtypes = [C.c_void_p, C.c_uint32]
    L.ffh_db_info_get.argtypes = [C.c_void_p, C.POINTER(DbInfo)]
    L.ffh_db_load_stats.argtypes = [C.c_void_p, C.POINTER(LoadStats)]
    L.ffh_indexer_create.restype = C.c_void_p
    L.ffh_indexer_create.argtypes = [C.c_int, C.c_int]
    L.ffh_indexer_destroy.argtypes = [C.c_void_p]
    L.ffh_indexer_last_error.restype = C.c_char_p
    L.ffh_indexer_last_error.argtypes = [C.c_void_p]
    L.ffh_indexer_add_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64]
    L.ffh_indexer_finish.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(IndexStats)]
    L.ffh_db_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.ffh_db_contig.restype = C.c_char_p
    L.ffh_db_contig.argtypes = [C.c_void_p, C.c_uint32]
    L.ffh_set_plan.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ffh_scan.argtypes = [C.c_void_p, u64p, C.c_uint32, C.c_int]
    L.ffh_scan_bounded.argtypes = [C.c_void_p, u64p, C.c_uint32, C.c_int, C.c_int]
    L.ffh_set_bounding.argtypes = [C.c_void_p, C.c_int]
    L.ffh_get_bounding.argtypes = [C.c_void_p]
    L.ffh_shard_totals.argtypes = [C.c_void_p, u32p, C.c_uint32]
    L.ffh_discover_bulge.argtypes = [C.c_void_p, u64p, C.c_uint32, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
    L.ffh_bulge_result_n_guides.restype = C.c_uint32
    L.ffh_bulge_result_n_hits.restype = C.c_uint64
    for name, ty in (("guide_offsets", u64p), ("hit_targets", u64p), ("hit_mismatches", C.POINTER(C.c_uint8)), ("hit_bulge_type", C.POINTER(C.c_uint8)),
                     ("hit_bulge_position", C.POINTER(C.c_uint8))):
        getattr(L, "ffh_bulge_result_" + name).restype = ty
    for name in ("n_guides", "n_hits", "guide_offsets", "hit_targets", "hit_mismatches", "hit_bulge_type", "hit_bulge_position", "free"):
        getattr(L, "ffh_bulge_result_" + name).argtypes = [C.c_void_p]
    L.ffh_shard_totals_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.ffh_summaries_to_device.argtypes = [C.c_void_p, C.c_void_p]
    L.ffh_exchange_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ffh_exchange_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.ffh_exchange_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.ffh_use_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ffh_finalize_shard.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
    L.ffh_exchange_prior.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.ffh_finalize_shard_fixup.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ffh_finalize.argtypes = [C.c_void_p, u32p, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
    L.ffh_discover.argtypes = [C.c_void_p, u64p, C.c_uint32, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
    L.ffh_score_lists.argtypes = [C.c_void_p, u64p, C.c_uint32, u64p, u64p, C.POINTER(C.c_void_p)]
    L.ffh_result_n_guides.restype = C.c_uint32
    L.ffh_result_n_guides.argtypes = [C.c_void_p]
    L.ffh_result_n_hits.restype = C.c_uint64
    L.ffh_result_n_hits.argtypes = [C.c_void_p]
    L.ffh_result_n_positions.restype = C.c_uint64
    L.ffh_result_n_positions.argtypes = [C.c_void_p]
    L.ffh_result_scores_valid.argtypes = [C.c_void_p]
    for name, rt in (("ffh_result_summaries", C.c_void_p), ("ffh_result_guide_offsets", u64p), ("ffh_result_hit_targets", u64p),
                     ("ffh_result_hit_mismatches", C.POINTER(C.c_uint8)), ("ffh_result_hit_cfd", C.POINTER(C.c_double)),
                     ("ffh_result_pos_offsets", u64p), ("ffh_result_positions", u64p)):
        getattr(L, name).restype = rt
        getattr(L, name).argtypes = [C.c_void_p]
    L.ffh_result_free.argtypes = [C.c_void_p]
    L.ffh_get_timings.argtypes = [C.c_void_p, C.POINTER(Timings)]
    L.ffh_comm_unique_id.argtypes = [C.c_void_p]
    L.ffh_comm_create_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.ffh_comm_create_local.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
    L.ffh_comm_destroy.argtypes = [C.c_void_p]
    L.ffh_comm_last_error.restype = C.c_char_p
    L.ffh_comm_last_error.argtypes = [C.c_void_p]
    for name in ("ffh_comm_world", "ffh_comm_first_shard", "ffh_comm_local_shards", "ffh_comm_transport"):
        getattr(L, name).argtypes = [C.c_void_p]
    L.ffh_discover_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint, C.c_void_p]
    L.ffh_comm_exchange.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint, C.c_void_p]
    L.ffh_comm_shard_lists.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
    L.ffh_comm_device_summaries.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.ffh_comm_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ffh_comm_set_exchange.argtypes = [C.c_void_p, C.c_int]
    L.ffh_ctx_share_db.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.ffh_pipe_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.ffh_pipe_submit.argtypes = [C.c_void_p, u64p, C.c_uint32, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_uint64)]
    L.ffh_pipe_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.ffh_pipe_last_error.restype = C.c_char_p
    L.ffh_pipe_last_error.argtypes = [C.c_void_p]
    L.ffh_pipe_lanes.argtypes = [C.c_void_p]
    L.ffh_pipe_destroy.argtypes = [C.c_void_p]
    L.ffh_comm_get_exchange.argtypes = [C.c_void_p]
    if hasattr(L, "ffh_host_alloc"):   # (absent from A/B builds of earlier revisions: FFH_LIBRARY)
        L.ffh_host_alloc.restype = C.c_void_p
        L.ffh_host_alloc.argtypes = [C.c_size_t]
        L.ffh_host_free.argtypes = [C.c_void_p]
    _lib = L
    return L


class _HostBlock:
    """a page-locked block from ffh_host_alloc, freed when the last numpy view of it is gone"""

    def __init__(self, L, nbytes):
        self.L, self.p = L, L.ffh_host_alloc(nbytes)
        if not self.p:
            raise MemoryError("ffh_host_alloc(%d)" % nbytes)

    def __del__(self):
        if getattr(self, "p", None):
            self.L.ffh_host_free(self.p)
            self.p = None


def host_summaries(n_guides):
    """n_guides zeroed summaries (SUMMARY_DTYPE) in page-locked memory from ffh_host_alloc: what a caller of Comm.discover /
    ffh_discover_sharded passes as `out` so that the reduced aggregates are copied straight into it (pageable memory is staged by the
    runtime: about twice the time for 100 000 guides).  A library without ffh_host_alloc (an A/B build) gives a plain numpy array."""
    L = load_library()
    n = max(int(n_guides), 1)
    if not hasattr(L, "ffh_host_alloc"):
        return np.zeros(n_guides, dtype=SUMMARY_DTYPE)
    blk = _HostBlock(L, n * SUMMARY_DTYPE.itemsize)
    buf = (C.c_uint8 * (n * SUMMARY_DTYPE.itemsize)).from_address(blk.p)
    arr = np.frombuffer(buf, dtype=SUMMARY_DTYPE, count=n)
    buf._ffh_block = blk   # (the array and every view of it hold `buf`, `buf` holds the block)
    arr[:] = 0
    return arr[:n_guides]


class _ResultOwner:
    """frees the ffh_result when the last numpy view of it is gone"""

    def __init__(self, L, h):
        self.L, self.h = L, h

    def __del__(self):
        try:
            if self.h:
                self.L.ffh_result_free(self.h)
                self.h = None
        except Exception:
            pass


def _view(owner, ptr, n, dtype):
    """zero-copy numpy view of a result array; the view keeps the result alive"""
    dtype = np.dtype(dtype)
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    addr = ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value
    buf = (C.c_char * (n * dtype.itemsize)).from_address(addr)
    buf._owner = owner
    return np.frombuffer(buf, dtype=dtype, count=n)


class Result:
    """An ffh_result: guides in input order, hits in database order, already cut off.  The arrays are read-only views
    of the library's (page-locked) result block, released when the last of them is garbage-collected."""

    def __init__(self, L, h, lists=True, positions=True, hit_scores=True):
        own = _ResultOwner(L, h)
        n = L.ffh_result_n_guides(h)
        self._L, self._own, self.n_guides = L, own, n
        self.scores_valid = bool(L.ffh_result_scores_valid(h))
        self.summaries = _view(own, L.ffh_result_summaries(h), n, SUMMARY_DTYPE)
        if lists:
            H = L.ffh_result_n_hits(h)
            P = L.ffh_result_n_positions(h)
            self.n_hits, self.n_positions = H, P
            self.guide_offsets = _view(own, L.ffh_result_guide_offsets(h), n + 1, np.uint64)
        else:
            self.n_positions = 0  # n_hits / guide_offsets of an aggregates-only result are folded from the summaries on first use
        if lists:
            self.hit_targets = _view(own, L.ffh_result_hit_targets(h), H, np.uint64)
            self.hit_mismatches = _view(own, L.ffh_result_hit_mismatches(h), H, np.uint8)
            self.hit_cfd = _view(own, L.ffh_result_hit_cfd(h), H, np.float64) if hit_scores else None  # FFH_FINALIZE_NO_HIT_SCORES
            if positions:   # pos_offsets: folded by the library on first use (__getattr__), like the offsets of an aggregates-only result
                self.positions = _view(own, L.ffh_result_positions(h), P, np.uint64)
            else:  # FFH_FINALIZE_NO_POSITIONS: the count of a hit is in bits 63:48 of its target long
                self.pos_offsets = self.positions = None

    def __getattr__(self, name):  # only reached for attributes not set yet: the lazy pair of an aggregates-only result
        if name == "n_hits":
            self.n_hits = self._L.ffh_result_n_hits(self._own.h)
            return self.n_hits
        if name == "guide_offsets":
            self.guide_offsets = _view(self._own, self._L.ffh_result_guide_offsets(self._own.h), self.n_guides + 1, np.uint64)
            return self.guide_offsets
        if name == "pos_offsets":
            self.pos_offsets = _view(self._own, self._L.ffh_result_pos_offsets(self._own.h), self.n_hits + 1, np.uint64)
            return self.pos_offsets
        raise AttributeError(name)

    def hits(self, g):
        a, b = int(self.guide_offsets[g]), int(self.guide_offsets[g + 1])
        return self.hit_targets[a:b]

    # the values the reference prints (Doench2016CFDScore.scala:76-87, CrisprMitEduOffTarget.scala:103-105)
    def cfd_specificity(self):
        return 1.0 / (1.0 + self.summaries["cfd_sum"])

    def hsu2013(self):
        return (100.0 / (100.0 + self.summaries["hsu_sum"])) * 100.0


def write_database(path, enzyme_index, targets, positions, contigs, bin_width=7):
    """ffh_db_write: host arrays -> BGZF body + .header in the reference's format (no GPU involved)."""
    L = load_library()
    t = np.ascontiguousarray(targets).view(np.uint64)
    p = np.ascontiguousarray(positions).view(np.uint64)
    names = (C.c_char_p * max(len(contigs), 1))(*[c.encode() for c in contigs])
    rc = L.ffh_db_write(str(path).encode(), enzyme_index, bin_width, names, len(contigs), t.ctypes.data, len(t), p.ctypes.data, len(p))
    if rc:
        raise FlashFryHipError(rc, L.ffh_last_error(None).decode())


def index_contigs(path, enzyme_index, contigs, bin_width=7, device=0):
    """ffh_indexer_*: [(name, sequence bytes/str)] -> database files; returns the IndexStats."""
    L = load_library()
    ix = L.ffh_indexer_create(device, enzyme_index)
    if not ix:
        raise FlashFryHipError(-7, L.ffh_last_error(None).decode())
    try:
        for name, seq in contigs:
            b = seq.encode() if isinstance(seq, str) else bytes(seq)
            rc = L.ffh_indexer_add_contig(ix, name.encode(), b, len(b))
            if rc:
                raise FlashFryHipError(rc, L.ffh_indexer_last_error(ix).decode())
        st = IndexStats()
        rc = L.ffh_indexer_finish(ix, str(path).encode(), bin_width, C.byref(st))
        if rc:
            raise FlashFryHipError(rc, L.ffh_indexer_last_error(ix).decode())
        return st
    finally:
        L.ffh_indexer_destroy(ix)


class BulgeResult:
    """ffh_bulge_result copied into numpy arrays: per guide (CSR) the hit targets in database order with the mismatch count, the
    bulge type (0 none, 1 RNA, 2 DNA) and the bulge position of the best alignment"""

    def __init__(self, L, h):
        n, H = L.ffh_bulge_result_n_guides(h), L.ffh_bulge_result_n_hits(h)
        cp = lambda ptr, cnt, dt: np.ctypeslib.as_array(ptr, shape=(cnt,)).astype(dt, copy=True) if cnt else np.zeros(0, dtype=dt)
        self.n_guides, self.n_hits = n, H
        self.guide_offsets = cp(L.ffh_bulge_result_guide_offsets(h), n + 1, np.uint64)
        self.hit_targets = cp(L.ffh_bulge_result_hit_targets(h), H, np.uint64)
        self.hit_mismatches = cp(L.ffh_bulge_result_hit_mismatches(h), H, np.uint8)
        self.hit_bulge_type = cp(L.ffh_bulge_result_hit_bulge_type(h), H, np.uint8)
        self.hit_bulge_position = cp(L.ffh_bulge_result_hit_bulge_position(h), H, np.uint8)


class Context:
    """One GPU, one HIP stream, one resident database shard."""

    def __init__(self, enzyme_index=3, device=0):
        self.L = load_library()
        self.h = self.L.ffh_create(device, enzyme_index)
        if not self.h:
            raise FlashFryHipError(-7, self.L.ffh_last_error(None).decode())
        self.enzyme_index = enzyme_index

    def close(self):
        if getattr(self, "h", None):
            self.L.ffh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise FlashFryHipError(rc, self.L.ffh_last_error(self.h).decode())

    def share(self):
        """ffh_ctx_share_db: a second context on the same device that scans THIS context's resident database through aliases (nothing is
        copied); the two may be driven from two host threads at once.  Close it before this one."""
        out = C.c_void_p()
        self._check(self.L.ffh_ctx_share_db(self.h, C.byref(out)))
        c = Context.__new__(Context)
        c.L, c.h, c.enzyme_index, c._owner = self.L, out.value, self.enzyme_index, self
        return c

    def pipe(self, lanes=2):
        """ffh_pipe_create: `lanes` discover calls in flight against this context's database (do not use the context directly meanwhile)"""
        return Pipe(self, lanes)

    # ---- database ------------------------------------------------------------------------------------
    def load_blocks(self, longs, bin_offsets):
        longs = np.ascontiguousarray(longs, dtype=np.int64)
        offs = np.ascontiguousarray(bin_offsets, dtype=np.uint64)
        self._check(self.L.ffh_db_load_blocks(self.h, longs.ctypes.data_as(i64p), offs.ctypes.data_as(u64p), len(offs) - 1))

    def load_soa(self, targets, positions):
        t = np.ascontiguousarray(targets).view(np.uint64)
        p = np.ascontiguousarray(positions).view(np.uint64)
        self._check(self.L.ffh_db_load_soa(self.h, t.ctypes.data, len(t), p.ctypes.data, len(p), 0))

    def load_soa_device(self, targets_ptr, n_targets, positions_ptr, n_positions):
        """device pointers (e.g. torch tensors' data_ptr()) on this context's GPU; the data is copied"""
        self._check(self.L.ffh_db_load_soa(self.h, targets_ptr, n_targets, positions_ptr, n_positions, 1))

    def open(self, path, bin_begin=0, bin_end=0):
        self._check(self.L.ffh_db_open(self.h, path.encode(), bin_begin, bin_end))

    def open_header(self, path):
        self._check(self.L.ffh_db_open_header(self.h, path.encode()))

    def bin_bytes(self, n_bins):
        return np.array([self.L.ffh_db_bin_bytes(self.h, b) for b in range(n_bins)], dtype=np.uint64)

    def info(self):
        i = DbInfo()
        self._check(self.L.ffh_db_info_get(self.h, C.byref(i)))
        return i

    def load_stats(self):
        st = LoadStats()
        self._check(self.L.ffh_db_load_stats(self.h, C.byref(st)))
        return st

    def contigs(self):
        out, i = [], 1
        while True:
            c = self.L.ffh_db_contig(self.h, i)
            if c is None:
                return out
            out.append(c.decode())
            i += 1

    def set_plan(self, prefix_bases=-1, prefix_radius=-1):
        self._check(self.L.ffh_set_plan(self.h, prefix_bases, prefix_radius))

    # ---- discover ------------------------------------------------------------------------------------
    def scan(self, guides, max_mismatch=4):
        g = np.ascontiguousarray(guides).view(np.uint64)
        self._n_guides = len(g)
        self._check(self.L.ffh_scan(self.h, g.ctypes.data_as(u64p), len(g), max_mismatch))

    def scan_bounded(self, guides, max_mismatch=4, max_offtargets=2000):
        """ffh_scan_bounded: guides that reach max_offtargets positions are retired from the later slabs of the database"""
        g = np.ascontiguousarray(guides).view(np.uint64)
        self._n_guides = len(g)
        self._check(self.L.ffh_scan_bounded(self.h, g.ctypes.data_as(u64p), len(g), max_mismatch, max_offtargets))

    def set_bounding(self, mode):
        """0 never, 1 always, -1 automatic (the default)"""
        self._check(self.L.ffh_set_bounding(self.h, int(mode)))

    def scan_device(self, guides_ptr, n_guides, max_mismatch=4):
        """ffh_scan with the guides' longs already in device memory (the pointer of a torch tensor, ...)"""
        self._n_guides = int(n_guides)
        self._check(self.L.ffh_scan(self.h, C.cast(C.c_void_p(guides_ptr), u64p), int(n_guides), max_mismatch))

    def discover_device(self, guides_ptr, n_guides, max_mismatch=4, max_offtargets=2000, summaries_only=False, jost=False, positions=True, hit_scores=True):
        """ffh_discover with the guides' longs already in device memory"""
        self._n_guides = int(n_guides)
        out = C.c_void_p()
        self._check(self.L.ffh_discover(self.h, C.cast(C.c_void_p(guides_ptr), u64p), int(n_guides), max_mismatch, max_offtargets,
                                        self._finalize_flags(summaries_only, jost, positions, hit_scores), C.byref(out)))
        return Result(self.L, out.value, lists=not summaries_only, positions=positions, hit_scores=hit_scores)

    def shard_totals(self, clamp):
        t = np.zeros(max(self._n_guides, 1), dtype=np.uint32)
        self._check(self.L.ffh_shard_totals(self.h, t.ctypes.data_as(u32p), clamp))
        return t[:self._n_guides]

    def discover_bulge(self, guides, max_mismatch=3, max_bulge=1, tttv=False, brute_force=False):
        """config C5 (Cas12a): hits with <= max_mismatch mismatches and <= max_bulge one-base bulges; returns a BulgeResult (copies).
        brute_force: every guide against every target (FFH_BULGE_BRUTE_FORCE) instead of the seeded candidate search"""
        g = np.ascontiguousarray(guides).view(np.uint64)
        out = C.c_void_p()
        self._check(self.L.ffh_discover_bulge(self.h, g.ctypes.data_as(u64p), len(g), max_mismatch, max_bulge, (1 if tttv else 0) | (2 if brute_force else 0),
                                              C.byref(out)))
        try:
            return BulgeResult(self.L, out.value)
        finally:
            self.L.ffh_bulge_result_free(out)

    def shard_totals_device(self, device_ptr, clamp):
        """per-guide position totals of this shard (saturated at clamp) written to a device buffer of n_guides uint32"""
        self._check(self.L.ffh_shard_totals_device(self.h, C.c_void_p(device_ptr), clamp))

    def summaries_to_device(self, device_ptr):
        """the summaries of the last finalize copied to a device buffer of n_guides * SUMMARY_DTYPE.itemsize bytes"""
        self._check(self.L.ffh_summaries_to_device(self.h, C.c_void_p(device_ptr)))

    def exchange_pack(self, summ_ptr, n, max_ptr, sum_ptr, fsum_ptr):
        self._check(self.L.ffh_exchange_pack(self.h, summ_ptr, n, max_ptr, sum_ptr, fsum_ptr))

    def exchange_mask(self, summ_ptr, n, max_ptr, sum_ptr):
        self._check(self.L.ffh_exchange_mask(self.h, summ_ptr, n, max_ptr, sum_ptr))

    def exchange_unpack(self, summ_ptr, n, max_ptr, sum_ptr, fsum_all_ptr, world):
        self._check(self.L.ffh_exchange_unpack(self.h, summ_ptr, n, max_ptr, sum_ptr, fsum_all_ptr, world))

    def use_stream(self, hip_stream, on=True):
        """issue this context's work on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream; 0 = the default stream)"""
        self._check(self.L.ffh_use_stream(self.h, C.c_void_p(hip_stream), 1 if on else 0))

    def finalize_shard(self, max_offtargets, summ_ptr, totals_ptr, jost=False):
        """this shard's aggregates as if it were the first shard -> device summaries, and its saturated totals -> device totals"""
        self._check(self.L.ffh_finalize_shard(self.h, max_offtargets, FINALIZE_JOST if jost else 0, C.c_void_p(summ_ptr), C.c_void_p(totals_ptr)))

    def exchange_prior(self, all_totals_ptr, n, rank, clamp, prior_ptr):
        self._check(self.L.ffh_exchange_prior(self.h, C.c_void_p(all_totals_ptr), n, rank, clamp, C.c_void_p(prior_ptr)))

    def finalize_shard_fixup(self, max_offtargets, prior_ptr, totals_ptr, summ_ptr, jost=False):
        """re-aggregate, with the prior, the guides whose cut-off the earlier shards change; their device summaries are overwritten"""
        self._check(self.L.ffh_finalize_shard_fixup(self.h, max_offtargets, FINALIZE_JOST if jost else 0, C.c_void_p(prior_ptr), C.c_void_p(totals_ptr),
                                                    C.c_void_p(summ_ptr)))

    def finalize_device_prior(self, max_offtargets, prior_device_ptr, summaries_only=True, jost=False):
        """finalize with the prior totals taken from device memory (n_guides uint32)"""
        out = C.c_void_p()
        flags = (FINALIZE_SUMMARIES_ONLY if summaries_only else 0) | (FINALIZE_JOST if jost else 0) | FINALIZE_PRIOR_ON_DEVICE
        self._check(self.L.ffh_finalize(self.h, C.cast(C.c_void_p(prior_device_ptr), u32p), max_offtargets, flags, C.byref(out)))
        return Result(self.L, out.value, lists=not summaries_only)

    @staticmethod
    def _finalize_flags(summaries_only, jost, positions, hit_scores):
        return ((FINALIZE_SUMMARIES_ONLY if summaries_only else 0) | (FINALIZE_JOST if jost else 0) | (0 if positions else FINALIZE_NO_POSITIONS) |
                (0 if hit_scores else FINALIZE_NO_HIT_SCORES))

    def finalize(self, max_offtargets=2000, prior_totals=None, summaries_only=False, jost=False, positions=True, hit_scores=True):
        out = C.c_void_p()
        pt = None
        if prior_totals is not None:
            pt = np.ascontiguousarray(prior_totals, dtype=np.uint32)
            assert len(pt) == self._n_guides
        self._check(self.L.ffh_finalize(self.h, pt.ctypes.data_as(u32p) if pt is not None else None, max_offtargets,
                                        (FINALIZE_SUMMARIES_ONLY if summaries_only else 0) | (FINALIZE_JOST if jost else 0) |
                                        (0 if positions else FINALIZE_NO_POSITIONS) | (0 if hit_scores else FINALIZE_NO_HIT_SCORES), C.byref(out)))
        return Result(self.L, out.value, lists=not summaries_only, positions=positions, hit_scores=hit_scores)

    def discover(self, guides, max_mismatch=4, max_offtargets=2000, summaries_only=False, jost=False, positions=True, hit_scores=True):
        """ffh_discover: scan (bounded by max_offtargets when bounding is on for the context) + finalize"""
        g = np.ascontiguousarray(guides).view(np.uint64)
        self._n_guides = len(g)
        out = C.c_void_p()
        self._check(self.L.ffh_discover(self.h, g.ctypes.data_as(u64p), len(g), max_mismatch, max_offtargets, self._finalize_flags(summaries_only, jost, positions, hit_scores),
                                        C.byref(out)))
        return Result(self.L, out.value, lists=not summaries_only, positions=positions, hit_scores=hit_scores)

    def score_lists(self, guides, guide_offsets, hit_targets):
        """the `score` path: score caller-supplied hit lists (CSR) on the device"""
        g = np.ascontiguousarray(guides).view(np.uint64)
        o = np.ascontiguousarray(guide_offsets, dtype=np.uint64)
        t = np.ascontiguousarray(hit_targets).view(np.uint64)
        assert len(o) == len(g) + 1
        out = C.c_void_p()
        self._check(self.L.ffh_score_lists(self.h, g.ctypes.data_as(u64p), len(g), o.ctypes.data_as(u64p), t.ctypes.data_as(u64p), C.byref(out)))
        return Result(self.L, out.value)

    def timings(self):
        t = Timings()
        self._check(self.L.ffh_get_timings(self.h, C.byref(t)))
        return t


class Pipe:
    """ffh_pipe_*: guide batches submitted to a FIFO, `lanes` of them in flight at once against one resident database"""

    def __init__(self, ctx, lanes=2):
        self.L, self.ctx = ctx.L, ctx
        out = C.c_void_p()
        ctx._check(self.L.ffh_pipe_create(ctx.h, int(lanes), C.byref(out)))
        self.h = out.value
        self._kind = {}

    @property
    def lanes(self):
        return self.L.ffh_pipe_lanes(self.h)

    def submit(self, guides, max_mismatch=4, max_offtargets=2000, summaries_only=False, jost=False, positions=True, hit_scores=True):
        g = np.ascontiguousarray(guides).view(np.uint64)
        t = C.c_uint64()
        rc = self.L.ffh_pipe_submit(self.h, g.ctypes.data_as(u64p), len(g), max_mismatch, max_offtargets, self.ctx._finalize_flags(summaries_only, jost, positions, hit_scores), C.byref(t))
        if rc:
            raise FlashFryHipError(rc, "ffh_pipe_submit failed")
        self._kind[t.value] = (not summaries_only, positions, hit_scores)
        return t.value

    def wait(self, ticket):
        out = C.c_void_p()
        rc = self.L.ffh_pipe_wait(self.h, ticket, C.byref(out))
        lists, positions, hit_scores = self._kind.pop(ticket, (True, True, True))
        if rc:
            raise FlashFryHipError(rc, self.L.ffh_pipe_last_error(self.h).decode())
        return Result(self.L, out.value, lists=lists, positions=positions, hit_scores=hit_scores)

    def close(self):
        if getattr(self, "h", None):
            self.L.ffh_pipe_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


TRANSPORTS = {0: "copy", 1: "rccl-all", 2: "rccl-rank"}


def comm_unique_id():
    """ffh_comm_unique_id: the 128 bytes one rank makes and every rank of ffh_comm_create_rank needs (ncclGetUniqueId)"""
    L = load_library()
    buf = (C.c_char * 128)()
    rc = L.ffh_comm_unique_id(buf)
    if rc:
        raise FlashFryHipError(rc, L.ffh_comm_last_error(None).decode())
    return bytes(buf)


class Comm:
    """ffh_comm: the bin-sharded discover with the collectives issued inside the library (RCCL, or device copies when the shards
    share a GPU).  Comm.local(ctxs): one process drives all shards; Comm.rank(ctx, rank, world, id): one process per GPU."""

    def __init__(self, L, h, ctxs):
        self.L, self.h, self.ctxs = L, h, list(ctxs)   # (the contexts must outlive the communicator)

    @classmethod
    def local(cls, ctxs):
        L = load_library()
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        out = C.c_void_p()
        rc = L.ffh_comm_create_local(arr, len(ctxs), C.byref(out))
        if rc:
            raise FlashFryHipError(rc, L.ffh_comm_last_error(None).decode())
        return cls(L, out.value, ctxs)

    @classmethod
    def rank(cls, ctx, rank, world, unique_id):
        L = load_library()
        assert len(unique_id) == 128
        out = C.c_void_p()
        rc = L.ffh_comm_create_rank(ctx.h, rank, world, C.c_char_p(unique_id), C.byref(out))
        if rc:
            raise FlashFryHipError(rc, L.ffh_comm_last_error(None).decode())
        return cls(L, out.value, [ctx])

    def close(self):
        if getattr(self, "h", None):
            self.L.ffh_comm_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise FlashFryHipError(rc, self.L.ffh_comm_last_error(self.h).decode())

    @property
    def world(self):
        return self.L.ffh_comm_world(self.h)

    @property
    def first_shard(self):
        return self.L.ffh_comm_first_shard(self.h)

    @property
    def transport(self):
        return TRANSPORTS.get(self.L.ffh_comm_transport(self.h), "?")

    def discover(self, guides, max_mismatch=4, max_offtargets=2000, jost=False, want_summaries=True):
        """ffh_discover_sharded with host guides: returns the reduced summaries of all shards (numpy, SUMMARY_DTYPE) or None"""
        g = np.ascontiguousarray(guides).view(np.uint64)
        return self._discover(g.ctypes.data, len(g), max_mismatch, max_offtargets, jost, want_summaries)

    def discover_device(self, guides_ptr, n_guides, max_mismatch=4, max_offtargets=2000, jost=False, want_summaries=True, out=None):
        return self._discover(guides_ptr, int(n_guides), max_mismatch, max_offtargets, jost, want_summaries, out)

    def _discover(self, ptr, n, max_mismatch, max_offtargets, jost, want_summaries, out=None):
        for c in self.ctxs:
            c._n_guides = n
        summ = None
        if want_summaries:
            summ = out if out is not None else host_summaries(n)
        self._check(self.L.ffh_discover_sharded(self.h, C.c_void_p(ptr), n, max_mismatch, max_offtargets, FINALIZE_JOST if jost else 0,
                                                C.c_void_p(summ.ctypes.data) if summ is not None else None))
        return summ

    def exchange(self, n_guides, max_offtargets=2000, jost=False):
        """ffh_comm_exchange: the exchange alone, after the caller scanned every local shard with the same guide set"""
        summ = host_summaries(n_guides)
        self._check(self.L.ffh_comm_exchange(self.h, n_guides, max_offtargets, FINALIZE_JOST if jost else 0, C.c_void_p(summ.ctypes.data)))
        return summ

    def shard_lists(self, local_shard, positions=True, hit_scores=True, jost=False):
        out = C.c_void_p()
        self._check(self.L.ffh_comm_shard_lists(self.h, local_shard, Context._finalize_flags(False, jost, positions, hit_scores), C.byref(out)))
        return Result(self.L, out.value, lists=True, positions=positions, hit_scores=hit_scores)

    def set_exchange(self, mode):
        """0 / "gather": one all-gather of all records; 1 / "slice": all-to-all by guide slices (ffh_comm_set_exchange)"""
        self._check(self.L.ffh_comm_set_exchange(self.h, {"gather": 0, "slice": 1}.get(mode, mode)))

    def timings(self):
        a, b = C.c_double(), C.c_double()
        self._check(self.L.ffh_comm_timings(self.h, C.byref(a), C.byref(b)))
        return {"scan_ms": a.value, "exchange_ms": b.value}
