"""ctypes binding of the CPU oracle (oracle/libff_oracle.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the flashfry_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)


class BinAndMask(C.Structure):
    _fields_ = [("bin_long", C.c_uint64), ("guide_mask", C.c_uint64)]


class GuideScores(C.Structure):
    _fields_ = [("cfd_max", C.c_double), ("cfd_spec", C.c_double), ("cfd_valid", C.c_int),
                ("hsu", C.c_double), ("hsu_valid", C.c_int),
                ("closest", C.c_int), ("closest_count", C.c_int), ("hist", C.c_int * 5), ("in_genome", C.c_int),
                ("jost_valid", C.c_int), ("jost_max", C.c_double), ("jost_spec", C.c_double)]


class Site(C.Structure):
    _fields_ = [("start", C.c_int), ("forward", C.c_int), ("bases", C.c_char * 25), ("context", C.c_char * 64),
                ("has_context", C.c_int)]


def build():
    so = os.path.join(ORACLE_DIR, "libff_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h", ".inc"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Oracle:
    def __init__(self, lib):
        self.lib = L = lib
        L.ffo_pack_by_index.restype = C.c_void_p
        L.ffo_pack_by_index.argtypes = [C.c_int]
        L.ffo_bit_encode.argtypes = [C.c_char_p, C.c_int, C.c_int, u64p]
        L.ffo_bit_decode.argtypes = [C.c_uint64, C.c_int, C.c_char_p]
        L.ffo_get_count.argtypes = [C.c_uint64]
        L.ffo_mismatches.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.ffo_bin_to_long_comparitor.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(BinAndMask)]
        L.ffo_mismatch_bin.argtypes = [C.c_void_p, C.POINTER(BinAndMask), C.c_uint64]
        L.ffo_counter_bit_comparisons.restype = C.c_uint64
        L.ffo_counter_all_comparisons.restype = C.c_uint64
        L.ffo_bin_name.argtypes = [C.c_int, C.c_uint32, C.c_char_p]
        L.ffo_longs_to_bytes.argtypes = [i64p, C.c_size_t, C.POINTER(C.c_uint8)]
        L.ffo_bytes_to_longs.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, i64p]
        L.ffo_pos_encode.restype = C.c_uint64
        L.ffo_pos_encode.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_int]
        L.ffo_pos_decode.argtypes = [C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ffo_create_linear_block.restype = C.c_size_t
        L.ffo_create_linear_block.argtypes = [u64p, u64p, C.c_size_t, i64p]
        L.ffo_create_indexed_block.restype = C.c_size_t
        L.ffo_create_indexed_block.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t, C.c_int, C.c_int, i64p]
        L.ffo_db_new.restype = C.c_void_p
        L.ffo_db_new.argtypes = [C.c_int, C.c_int]
        L.ffo_db_free.argtypes = [C.c_void_p]
        L.ffo_db_set_bin.argtypes = [C.c_void_p, C.c_uint32, i64p, C.c_size_t, C.c_int]
        L.ffo_db_add_contig.argtypes = [C.c_void_p, C.c_char_p]
        L.ffo_db_n_bins.argtypes = [C.c_void_p]
        L.ffo_db_checksum.restype = C.c_uint64
        L.ffo_db_checksum.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.ffo_db_seal.argtypes = [C.c_void_p]
        L.ffo_db_bin_width.argtypes = [C.c_void_p]
        L.ffo_db_enzyme.argtypes = [C.c_void_p]
        L.ffo_db_n_contigs.argtypes = [C.c_void_p]
        L.ffo_db_contig.restype = C.c_char_p
        L.ffo_db_contig.argtypes = [C.c_void_p, C.c_int]
        L.ffo_db_bin_longs.restype = C.c_size_t
        L.ffo_db_bin_longs.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(i64p), C.POINTER(C.c_int)]
        L.ffo_db_build_from_sorted.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t, C.c_int]
        L.ffo_db_write.argtypes = [C.c_void_p, C.c_char_p]
        L.ffo_db_read.restype = C.c_void_p
        L.ffo_db_read.argtypes = [C.c_char_p]
        L.ffo_last_error.restype = C.c_char_p
        L.ffo_discover.restype = C.c_void_p
        L.ffo_discover.argtypes = [C.c_void_p, u64p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffo_discover_bin_range.restype = C.c_void_p
        L.ffo_discover_bin_range.argtypes = [C.c_void_p, u64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffo_result_free.argtypes = [C.c_void_p]
        L.ffo_result_n_guides.argtypes = [C.c_void_p]
        L.ffo_result_saturated.argtypes = [C.c_void_p]
        L.ffo_result_n_hits.argtypes = [C.c_void_p, C.c_int]
        L.ffo_result_current_total.argtypes = [C.c_void_p, C.c_int]
        L.ffo_result_full.argtypes = [C.c_void_p, C.c_int]
        L.ffo_result_export.restype = C.c_size_t
        L.ffo_result_export.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p]
        L.ffo_result_total_positions.restype = C.c_size_t
        L.ffo_result_total_positions.argtypes = [C.c_void_p]
        L.ffo_cfd_score_pair.restype = C.c_double
        L.ffo_cfd_score_pair.argtypes = [C.c_char_p, C.c_char_p]
        L.ffo_cfd_pam.restype = C.c_double
        L.ffo_cfd_pam.argtypes = [C.c_char_p]
        L.ffo_hsu_score_offtarget.restype = C.c_double
        L.ffo_hsu_score_offtarget.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.ffo_score_guide.argtypes = [C.c_void_p, C.c_uint64, u64p, C.c_int, C.POINTER(GuideScores), C.POINTER(C.c_double)]
        L.ffo_java_double_to_string.argtypes = [C.c_double, C.c_char_p]
        L.ffo_find_sites.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(Site), C.c_int]
        L.ffo_index_fasta.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.ffo_discover_fasta.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.ffo_score_file.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.ffo_bulge_align.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ffo_jost_calc_score.restype = C.c_double
        L.ffo_jost_calc_score.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]

    # ---- helpers -------------------------------------------------------------------------------
    def pack(self, idx):
        p = self.lib.ffo_pack_by_index(idx)
        assert p, "unknown enzyme %d" % idx
        return p

    def error(self):
        return self.lib.ffo_last_error().decode()

    def encode(self, s, count=1):
        out = C.c_uint64()
        rc = self.lib.ffo_bit_encode(s.encode(), len(s), count, C.byref(out))
        if rc:
            raise ValueError(self.error())
        return out.value

    def decode(self, enc, size):
        buf = C.create_string_buffer(32)
        cnt = self.lib.ffo_bit_decode(enc, size, buf)
        return buf.value.decode(), cnt

    def mismatches(self, enzyme, a, b, mask=0xFFFFFFFFFFFF):
        return self.lib.ffo_mismatches(self.pack(enzyme), a, b, mask)

    def mismatch_bin(self, enzyme, bin_str, guide, rshift=0):
        bm = BinAndMask()
        self.lib.ffo_bin_to_long_comparitor(self.pack(enzyme), bin_str.encode(), len(bin_str), rshift, C.byref(bm))
        return self.lib.ffo_mismatch_bin(self.pack(enzyme), C.byref(bm), guide), bm

    def bin_name(self, width, idx):
        buf = C.create_string_buffer(32)
        self.lib.ffo_bin_name(width, idx, buf)
        return buf.value.decode()

    def linear_block(self, targets, positions):
        t = np.ascontiguousarray(targets, dtype=np.uint64)
        p = np.ascontiguousarray(positions, dtype=np.uint64)
        n = self.lib.ffo_create_linear_block(_ptr(t, u64p), _ptr(p, u64p), len(t), None)
        out = np.zeros(n, dtype=np.int64)
        self.lib.ffo_create_linear_block(_ptr(t, u64p), _ptr(p, u64p), len(t), _ptr(out, i64p))
        return out

    def indexed_block(self, enzyme, targets, positions, prefix_len=7, lookup=4):
        t = np.ascontiguousarray(targets, dtype=np.uint64)
        p = np.ascontiguousarray(positions, dtype=np.uint64)
        n = self.lib.ffo_create_indexed_block(self.pack(enzyme), _ptr(t, u64p), _ptr(p, u64p), len(t), prefix_len, lookup, None)
        out = np.zeros(n, dtype=np.int64)
        self.lib.ffo_create_indexed_block(self.pack(enzyme), _ptr(t, u64p), _ptr(p, u64p), len(t), prefix_len, lookup, _ptr(out, i64p))
        return out

    def db_new(self, enzyme, bin_width=7):
        db = self.lib.ffo_db_new(enzyme, bin_width)
        if not db:
            raise ValueError(self.error())
        return OracleDB(self, db)

    def db_read(self, path):
        db = self.lib.ffo_db_read(path.encode())
        if not db:
            raise IOError(self.error())
        return OracleDB(self, db)

    def db_from_sorted(self, enzyme, targets, positions, bin_width=7, max_linear=500, contigs=()):
        db = self.db_new(enzyme, bin_width)
        for c in contigs:
            db.add_contig(c)
        t = np.ascontiguousarray(targets, dtype=np.uint64)
        p = np.ascontiguousarray(positions, dtype=np.uint64)
        rc = self.lib.ffo_db_build_from_sorted(db.h, _ptr(t, u64p), _ptr(p, u64p), len(t), max_linear)
        if rc:
            raise ValueError(self.error())
        return db

    def score_guide(self, enzyme, guide, hit_targets):
        t = np.ascontiguousarray(hit_targets, dtype=np.uint64)
        s = GuideScores()
        per = np.zeros(max(len(t), 1), dtype=np.float64)
        self.lib.ffo_score_guide(self.pack(enzyme), guide, _ptr(t, u64p), len(t), C.byref(s), per.ctypes.data_as(C.POINTER(C.c_double)))
        return s, per[:len(t)]

    def java_double(self, d):
        buf = C.create_string_buffer(64)
        self.lib.ffo_java_double_to_string(d, buf)
        return buf.value.decode()

    def find_sites(self, enzyme, seq, flank):
        n = self.lib.ffo_find_sites(self.pack(enzyme), seq.encode(), len(seq), flank, None, 0)
        arr = (Site * max(n, 1))()
        self.lib.ffo_find_sites(self.pack(enzyme), seq.encode(), len(seq), flank, arr, n)
        return [(arr[i].bases.decode(), arr[i].start, bool(arr[i].forward), bool(arr[i].has_context), arr[i].context.decode())
                for i in range(n)]


class OracleResult:
    """Flattened copy of an ffo_result: CSR over guides (input order)."""

    def __init__(self, o, h):
        L = o.lib
        n = L.ffo_result_n_guides(h)
        self.n_guides = n
        self.saturated = bool(L.ffo_result_saturated(h))
        H = L.ffo_result_export(h, None, None, None, None)
        P = L.ffo_result_total_positions(h)
        self.guide_offsets = np.zeros(n + 1, dtype=np.uint64)
        self.hit_targets = np.zeros(H, dtype=np.uint64)
        self.pos_offsets = np.zeros(H + 1, dtype=np.uint64)
        self.positions = np.zeros(P, dtype=np.uint64)
        L.ffo_result_export(h, _ptr(self.guide_offsets, u64p), _ptr(self.hit_targets, u64p),
                            _ptr(self.pos_offsets, u64p), _ptr(self.positions, u64p))
        self.current_total = np.array([L.ffo_result_current_total(h, g) for g in range(n)], dtype=np.int64)
        self.full = np.array([L.ffo_result_full(h, g) for g in range(n)], dtype=bool)

    def hits(self, g):
        a, b = int(self.guide_offsets[g]), int(self.guide_offsets[g + 1])
        return self.hit_targets[a:b]


class OracleDB:
    def __init__(self, o, h):
        self.o, self.h = o, h

    def __del__(self):
        try:
            self.o.lib.ffo_db_free(self.h)
        except Exception:
            pass

    @property
    def n_bins(self):
        return self.o.lib.ffo_db_n_bins(self.h)

    def checksums(self):
        """(checksum of the whole database, per-bin checksums): tools/stress_parity.py watches the checker's memory with it"""
        per = np.zeros(self.n_bins, dtype=np.uint64)
        return int(self.o.lib.ffo_db_checksum(self.h, per.ctypes.data_as(C.POINTER(C.c_uint64)), None, None)), per

    def seal(self):
        """all bins into one read-only mapping: a stray CPU store into the checker's database faults (tools/stress_parity.py --seal)"""
        if self.o.lib.ffo_db_seal(self.h):
            raise RuntimeError(self.o.error())

    def first_changed_bin(self, per):
        """index of the first bin whose checksum is not per[bin] any more, or -1"""
        ch = C.c_int(-1)
        self.o.lib.ffo_db_checksum(self.h, None, per.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(ch))
        return ch.value

    @property
    def bin_width(self):
        return self.o.lib.ffo_db_bin_width(self.h)

    @property
    def enzyme(self):
        return self.o.lib.ffo_db_enzyme(self.h)

    def contigs(self):
        return [self.o.lib.ffo_db_contig(self.h, i + 1).decode() for i in range(self.o.lib.ffo_db_n_contigs(self.h))]

    def add_contig(self, name):
        return self.o.lib.ffo_db_add_contig(self.h, name.encode())

    def set_bin(self, idx, longs, n_targets):
        a = np.ascontiguousarray(longs, dtype=np.int64)
        self.o.lib.ffo_db_set_bin(self.h, idx, _ptr(a, i64p), len(a), n_targets)

    def bin(self, idx):
        p = i64p()
        nt = C.c_int()
        n = self.o.lib.ffo_db_bin_longs(self.h, idx, C.byref(p), C.byref(nt))
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.int64), nt.value

    def all_blocks(self):
        """(concatenated longs, offsets[n_bins+1]) -- the payloads LinearTraverser would hand to compareBlock."""
        blocks = [self.bin(b)[0] for b in range(self.n_bins)]
        offs = np.zeros(self.n_bins + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(b) for b in blocks])
        return (np.concatenate(blocks) if blocks else np.zeros(0, np.int64)), offs

    def write(self, path):
        rc = self.o.lib.ffo_db_write(self.h, path.encode())
        if rc:
            raise IOError(self.o.error())

    def discover(self, guides, max_mm=4, max_ot=2000, force_linear=False):
        g = np.ascontiguousarray(guides, dtype=np.uint64)
        self.o.lib.ffo_counters_reset()
        r = self.o.lib.ffo_discover(self.h, _ptr(g, u64p), len(g), max_mm, max_ot, int(force_linear))
        if not r:
            raise RuntimeError(self.o.error())
        try:
            res = OracleResult(self.o, r)
            res.all_comparisons = self.o.lib.ffo_counter_all_comparisons()
            res.bit_comparisons = self.o.lib.ffo_counter_bit_comparisons()
            return res
        finally:
            self.o.lib.ffo_result_free(r)


def load():
    global _LIB
    if _LIB is None:
        # FFO_LIBRARY: another build of the checker (tools/r06_stress_sanitized.sh: the oracle under AddressSanitizer next to the library's host side)
        _LIB = Oracle(C.CDLL(os.environ.get("FFO_LIBRARY") or build()))
    return _LIB
