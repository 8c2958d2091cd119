"""The CPU oracle in a process of its own.  TEST INFRASTRUCTURE ONLY (like oracle_lib.py, which it wraps).

Why: the in-process checker shares its address space with a library that DMA-writes into page-locked host pools and polls host
words.  A checker that lives in another process cannot be reached by a stray device write, a stale page-locked block or a host
thread of the library: a disagreement between the two is then the library's or the oracle's logic, not the neighbourhood's.

    ro = RemoteOracle()                       # starts `python oracle_proc.py --serve`; that process never loads HIP
    odb = ro.db_from_sorted(3, targets, positions, contigs=[...])
    ora = odb.discover(guides, 4, 2000)       # same attributes as oracle_lib.OracleResult
    s, per = ro.score_guide(3, guide, ora.hits(0))

Arrays travel as pickled numpy arrays over the child's stdin / stdout (length-prefixed frames); the child holds the databases."""
import os
import pickle
import struct
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _send(f, obj):
    b = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack("<Q", len(b)))
    f.write(b)
    f.flush()


def _recv(f):
    h = f.read(8)
    if len(h) < 8:
        raise EOFError("the oracle process went away")
    (n,) = struct.unpack("<Q", h)
    b = f.read(n)
    if len(b) < n:
        raise EOFError("the oracle process went away mid-frame")
    return pickle.loads(b)


class RemoteResult:
    """what oracle_lib.OracleResult offers, rebuilt from the child's arrays"""

    def __init__(self, d):
        self.__dict__.update(d)

    def hits(self, g):
        a, b = int(self.guide_offsets[g]), int(self.guide_offsets[g + 1])
        return self.hit_targets[a:b]


class RemoteDB:
    def __init__(self, ro, handle):
        self.ro, self.handle = ro, handle

    def discover(self, guides, max_mm=4, max_ot=2000, force_linear=False):
        g = np.ascontiguousarray(guides, dtype=np.uint64)
        return RemoteResult(self.ro._call("discover", self.handle, g, int(max_mm), int(max_ot), bool(force_linear)))

    def __del__(self):
        try:
            self.ro._call("db_free", self.handle)
        except Exception:
            pass


class RemoteOracle:
    def __init__(self):
        env = dict(os.environ)
        env.pop("LD_PRELOAD", None)
        self._cache = {}
        self.p = subprocess.Popen([sys.executable, os.path.join(HERE, "oracle_proc.py"), "--serve"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
        assert self._call("ping") == "pong"

    def _call(self, *msg):
        _send(self.p.stdin, msg)
        ok, val = _recv(self.p.stdout)
        if not ok:
            raise RuntimeError("oracle process: " + val)
        return val

    def db_from_sorted(self, enzyme, targets, positions, bin_width=7, max_linear=500, contigs=()):
        t = np.ascontiguousarray(targets, dtype=np.uint64)
        p = np.ascontiguousarray(positions, dtype=np.uint64)
        return RemoteDB(self, self._call("db_from_sorted", int(enzyme), t, p, int(bin_width), int(max_linear), list(contigs)))

    def score_guide(self, enzyme, guide, hit_targets):
        ht = np.ascontiguousarray(hit_targets, dtype=np.uint64)
        hit = self._cache.get((int(enzyme), int(guide), ht.tobytes()))
        if hit is not None:
            return hit
        d, per = self._call("score_guide", int(enzyme), int(guide), ht)
        return types.SimpleNamespace(**d), per

    def prefetch(self, enzyme, guides, result):
        """scores of every guide of a discover result in one round trip; score_guide then answers from them"""
        self._cache = {}
        for k, sp in enumerate(self.score_guides(enzyme, guides, result)):
            self._cache[(int(enzyme), int(guides[k]), result.hits(k).tobytes())] = sp

    def score_guides(self, enzyme, guides, result):
        """every guide's (scores, per-hit CFD) of a discover result in ONE round trip -> list"""
        out = self._call("score_guides", int(enzyme), np.ascontiguousarray(guides, dtype=np.uint64), result.guide_offsets, result.hit_targets)
        return [(types.SimpleNamespace(**d), per) for d, per in out]

    def close(self):
        try:
            _send(self.p.stdin, ("quit",))
            self.p.wait(timeout=10)
        except Exception:
            self.p.kill()

    def __del__(self):
        self.close()


def _scores_dict(s):
    return dict(cfd_max=s.cfd_max, cfd_spec=s.cfd_spec, cfd_valid=s.cfd_valid, hsu=s.hsu, hsu_valid=s.hsu_valid, closest=s.closest,
                closest_count=s.closest_count, hist=list(s.hist), in_genome=s.in_genome, jost_valid=s.jost_valid, jost_max=s.jost_max, jost_spec=s.jost_spec)


def _serve():
    sys.path.insert(0, HERE)
    import oracle_lib
    fin, fout = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr   # (nothing but frames on the pipe)
    oracle = oracle_lib.load()
    assert "libflashfry_hip" not in open("/proc/self/maps").read() and "libamdhip64" not in open("/proc/self/maps").read()
    dbs, next_h = {}, 1
    while True:
        try:
            msg = _recv(fin)
        except EOFError:
            return
        try:
            op = msg[0]
            if op == "ping":
                val = "pong"
            elif op == "quit":
                _send(fout, (True, None))
                return
            elif op == "db_from_sorted":
                _, enz, t, p, bw, ml, contigs = msg
                dbs[next_h] = oracle.db_from_sorted(enz, t, p, bin_width=bw, max_linear=ml, contigs=contigs)
                val = next_h
                next_h += 1
            elif op == "db_free":
                dbs.pop(msg[1], None)
                val = None
            elif op == "discover":
                _, h, g, mm, ot, fl = msg
                r = dbs[h].discover(g, mm, ot, force_linear=fl)
                val = dict(n_guides=r.n_guides, saturated=r.saturated, guide_offsets=r.guide_offsets, hit_targets=r.hit_targets, pos_offsets=r.pos_offsets,
                           positions=r.positions, current_total=r.current_total, full=r.full, all_comparisons=r.all_comparisons, bit_comparisons=r.bit_comparisons)
            elif op == "score_guide":
                s, per = oracle.score_guide(msg[1], msg[2], msg[3])
                val = (_scores_dict(s), per.copy())
            elif op == "score_guides":
                _, enz, g, goff, ht = msg
                val = []
                for k in range(len(g)):
                    s, per = oracle.score_guide(enz, int(g[k]), ht[int(goff[k]):int(goff[k + 1])])
                    val.append((_scores_dict(s), per.copy()))
            else:
                raise ValueError("unknown request %r" % (op,))
            _send(fout, (True, val))
        except Exception as e:   # the parent sees the message, the server goes on
            _send(fout, (False, "%s: %s" % (type(e).__name__, e)))


if __name__ == "__main__" and "--serve" in sys.argv:
    _serve()
