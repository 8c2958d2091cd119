"""Seeded synthetic off-target databases and guide sets.

No genome is available on either box, so every test and benchmark runs on data made here.  The generator is
counter-based splitmix64 written in int64 arithmetic that behaves identically in numpy (CPU, tests) and torch
(on the device, bench.py), so a given (seed, size) names the same database everywhere.

Layout of what is produced (bitcoding/BitEncoding.scala:46-67, bitcoding/BitPosition.scala:51-63 of the reference):
  targets[i]   u64: 2 bits/base, first base most significant, 23-mer = 20 guide bases + N + GG; count in bits 63:48
  positions[]  u64: strand[63:60] size[59:52] contig[51:32] pos[31:0]; count(target i) consecutive entries
Targets are distinct, ascending (= database order for 3'-PAM enzymes).
"""
import numpy as np

GOLDEN = -7046029254386353131  # 0x9E3779B97F4A7C15 as int64
M1 = -4658895280553007687      # 0xBF58476D1CE4E5B9
M2 = -7723592293110705685      # 0x94D049BB133111EB
DB_SEED = 0xF1A5F4B1
GUIDE_SEED = 0x6D1DE5
MASK40 = (1 << 40) - 1


class _NP:
    name = "numpy"

    @staticmethod
    def arange(n, device=None):
        return np.arange(n, dtype=np.int64)

    @staticmethod
    def unique(x):
        return np.unique(x)

    @staticmethod
    def unique_counts(x):
        return np.unique(x, return_counts=True)

    @staticmethod
    def searchsorted(a, v):
        return np.searchsorted(a, v, side="right")

    @staticmethod
    def argsort(x):
        return np.argsort(x, kind="stable")

    @staticmethod
    def minimum(a, b):
        return np.minimum(a, b)

    @staticmethod
    def cat(xs):
        return np.concatenate(xs)

    @staticmethod
    def cumsum(x):
        return np.cumsum(x)

    @staticmethod
    def where(c, a, b):
        return np.where(c, a, b)

    @staticmethod
    def zeros(n, device=None):
        return np.zeros(n, dtype=np.int64)

    @staticmethod
    def repeat(x, counts):
        return np.repeat(x, counts)

    @staticmethod
    def scalar(v, like):
        return np.int64(v)

    @staticmethod
    def asint(x):
        return x.astype(np.int64)


def _torch_ns():
    import torch

    class _T:
        name = "torch"

        @staticmethod
        def arange(n, device=None):
            return torch.arange(n, dtype=torch.int64, device=device)

        @staticmethod
        def unique(x):
            return torch.unique(x, sorted=True)

        @staticmethod
        def unique_counts(x):
            return torch.unique(x, sorted=True, return_counts=True)

        @staticmethod
        def searchsorted(a, v):
            return torch.searchsorted(a, v, right=True)

        @staticmethod
        def argsort(x):
            return torch.argsort(x, stable=True)

        @staticmethod
        def minimum(a, b):
            return torch.minimum(a, b)

        @staticmethod
        def cat(xs):
            return torch.cat(xs)

        @staticmethod
        def cumsum(x):
            return torch.cumsum(x, 0)

        @staticmethod
        def where(c, a, b):
            return torch.where(c, a, b)

        @staticmethod
        def zeros(n, device=None):
            return torch.zeros(n, dtype=torch.int64, device=device)

        @staticmethod
        def repeat(x, counts):
            return torch.repeat_interleave(x, counts)

        @staticmethod
        def scalar(v, like):
            return torch.tensor(v, dtype=torch.int64, device=like.device)

        @staticmethod
        def asint(x):
            return x.to(torch.int64)

    return _T


def _lsr(x, s):
    """logical shift right of an int64 array"""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _s64(v):
    """python int -> the same 64 bits as a signed value"""
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def splitmix64(seed, idx):
    """output number idx+1 of splitmix64 seeded with `seed`, vectorised over the int64 array `idx` (wraps mod 2^64)."""
    z = (idx + 1) * GOLDEN + _s64(seed)
    z = (z ^ _lsr(z, 30)) * M1
    z = (z ^ _lsr(z, 27)) * M2
    z = z ^ _lsr(z, 31)
    return z


def _ns(device):
    return _NP if device is None else _torch_ns()


def make_guides(n_guides, seed=GUIDE_SEED, device=None):
    """n uniform random 20-mers + NGG, none starting with CC (GenerateRandomFasta --onlyUnidirectional:
    modules/GenerateRandomFasta.scala:100-101 rejects sequences that also match the reverse regex C C N21).
    Returns int64 array of guide longs with count 1 (bitEncodeString(StringCount(bases, 1)))."""
    xp = _ns(device)
    m = int(n_guides * 1.10) + 64
    i = xp.arange(m, device)
    mer = splitmix64(seed, i) & MASK40
    keep = _lsr(mer, 36) != 0b0101  # first two bases C,C
    mer = mer[keep][:n_guides]
    assert mer.shape[0] == n_guides
    n = splitmix64(seed + 1, xp.arange(n_guides, device)) & 3
    return (mer << 6) | (n << 4) | 0b1010 | (1 << 48)


def _mutate(mers, level, salt, xp, device):
    """copies of the 20-mers with exactly `level` substituted bases (positions and substitutions from a hash)."""
    out = mers
    h = splitmix64(salt + level, xp.arange(mers.shape[0], device))
    # choose `level` distinct positions: start + k*stride (mod 20) with stride in {1,3,7,9} coprime to 20
    start = h % 20
    start = xp.where(start < 0, start + 20, start)
    strides = [1, 3, 7, 9]
    stride_sel = _lsr(h, 8) & 3
    stride = xp.where(stride_sel == 0, xp.scalar(strides[0], h),
                      xp.where(stride_sel == 1, xp.scalar(strides[1], h),
                               xp.where(stride_sel == 2, xp.scalar(strides[2], h), xp.scalar(strides[3], h))))
    for k in range(level):
        pos = (start + k * stride) % 20            # base index 0..19
        delta = (_lsr(h, 16 + 2 * k) % 3) + 1       # 1..3 -> a different base
        sh = 2 * (19 - pos)
        out = out ^ (delta << sh)
    return out


def make_database(n_targets, seed=DB_SEED, plant_guides=None, device=None, with_positions=True):
    """Sorted distinct targets (+counts) and their positions.

    n_targets random 20-mers are drawn (duplicates collapse, so the result is very slightly smaller), every 100th
    guide of `plant_guides` is planted as an exact copy and as copies at 1, 2, 3 and 4 mismatches.
    Returns dict(targets=int64[T], positions=int64[P], pos_offsets=int64[T+1])."""
    xp = _ns(device)
    mer = splitmix64(seed, xp.arange(n_targets, device)) & MASK40
    if plant_guides is not None and plant_guides.shape[0] > 0:
        g = _lsr(plant_guides, 6) & MASK40
        g = g[::100]
        planted = [g] + [_mutate(g, lvl, seed + 17, xp, device) for lvl in (1, 2, 3, 4)]
        mer = xp.cat([mer] + planted)
    mer = xp.unique(mer)
    T = int(mer.shape[0])
    h = splitmix64(seed + 2, mer)
    pam_n = h & 3
    # occurrence count: 1 + geometric tail (p = 0.9), integer thresholds so numpy and torch agree bit-for-bit
    r = _lsr(h, 8) & 0xFFFFFFFF
    count = xp.zeros(T, device) + 1
    thr = 1 << 32
    for _ in range(9):
        thr //= 10
        count = count + xp.asint(r < thr)
    # one target in 4096 is heavy (repeat-like): count 2..1025
    heavy = (_lsr(h, 44) & 0xFFF) == 0
    count = xp.where(heavy, 2 + (_lsr(h, 2) & 0x3FF), count)
    targets = (mer << 6) | (pam_n << 4) | 0b1010 | (count << 48)
    out = {"targets": targets, "T": T}
    if with_positions:
        csum = xp.cumsum(count)
        P = int(csum[-1]) if T else 0
        pos_off = xp.cat([xp.zeros(1, device), csum])
        owner = xp.repeat(xp.arange(T, device), count)
        hp = splitmix64(seed + 3, xp.arange(P, device)) ^ splitmix64(seed + 4, owner)
        contig = (_lsr(hp, 1) % 24) + 1
        pos = _lsr(hp, 8) & ((1 << 27) - 1)
        strand = _lsr(hp, 40) & 1
        positions = (strand << 60) | (23 << 52) | (contig << 32) | pos
        out.update(positions=positions, pos_offsets=pos_off, P=P)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# A repeat-structured genome (the published workload is real hg38, whose bucket occupancy is heavy-tailed: Alu / L1 / satellite
# families, poly-A and other low-complexity tracts).  Three components, all counter-based hashes like the uniform generator:
#   * uniform background: random 20-mers (the unique part of a genome);
#   * repeat families: six levels, level l has 4^l families of cmax / 4^l copies each (at hg38 scale cmax ~ 1e6: one family of a
#     million copies down to a thousand families of a thousand), every family a consensus of REPEAT_WINDOWS 20-mer windows; a copy of a
#     window is the consensus with every base substituted independently at the family's divergence (14 % for the largest, oldest
#     families down to 1 % for the youngest: Alu-like to segmental-duplication-like);
#   * low-complexity tracts: period-1..6 tandem repeats with up to two substitutions (poly-A, (CA)n, (GGAA)n ...): few distinct
#     sequences, very high counts.
# Identical 20-mers collapse into one target whose count is the number of copies (capped at 32767 like BlockReader.scala:147-153).
# Guides are sampled FROM the genome by position (make_guides_from_database), as a tiling library's are, so repeat families get
# their share of guides and those guides collect thousands to hundreds of thousands of raw hits.
# ---------------------------------------------------------------------------------------------------------------------
REPEAT_WINDOWS = 16
REPEAT_LEVELS = 6
REPEAT_DIVERGENCE_64K = [9175, 7209, 5243, 3277, 1966, 655]  # 14, 11, 8, 5, 3, 1 % as thresholds on a 16-bit hash


def _substitute(mers, rate_64k, salt, ids, xp, device):
    """every base of every 20-mer substituted independently with probability rate_64k / 65536 (per-element rates allowed)"""
    out = mers
    for pos in range(20):
        u = splitmix64(salt + pos, ids)
        hit = (u & 0xFFFF) < rate_64k
        delta = (_lsr(u, 16) % 3) + 1
        out = xp.where(hit, out ^ (delta << (2 * (19 - pos))), out)
    return out


def make_repeat_database(n_targets, seed=DB_SEED, device=None, repeat_fraction=0.35, low_fraction=0.005, plant_guides=None):
    """Sorted distinct targets (+counts) and positions of a repeat-structured synthetic genome of ~n_targets drawn 20-mers
    (the number of DISTINCT targets is smaller: copies collapse).  Same return layout as make_database."""
    xp = _ns(device)
    n_rep = int(n_targets * repeat_fraction)
    n_low = int(n_targets * low_fraction)
    n_uni = max(n_targets - n_rep - n_low, 0)
    parts = [splitmix64(seed, xp.arange(n_uni, device)) & MASK40]
    # repeat families
    per_level = max(n_rep // REPEAT_LEVELS, REPEAT_WINDOWS)
    cmax = max(per_level // REPEAT_WINDOWS, 1)                 # copies of the single level-0 family
    n_rep = per_level * REPEAT_LEVELS
    e = xp.arange(n_rep, device)
    level = e // per_level
    r = e - level * per_level
    rate = xp.zeros(n_rep, device)
    fam = xp.zeros(n_rep, device)
    for l in range(REPEAT_LEVELS):
        copies = max(cmax >> (2 * l), 1)                        # cmax / 4^l copies per family, ~4^l families
        fam = xp.where(level == l, r // (copies * REPEAT_WINDOWS), fam)
        rate = xp.where(level == l, xp.scalar(REPEAT_DIVERGENCE_64K[l], e), rate)
    window = r % REPEAT_WINDOWS
    consensus = splitmix64(seed + 10, (level << 40) | (fam << 8) | window) & MASK40
    parts.append(_substitute(consensus, rate, seed + 100, e, xp, device))
    # low-complexity tracts
    i = xp.arange(n_low, device)
    h = splitmix64(seed + 11, i)
    period_sel = h & 7
    unit = _lsr(h, 8)
    tract = xp.zeros(n_low, device)
    for sel, period in enumerate([1, 1, 2, 2, 3, 4, 4, 6]):
        u = unit & ((1 << (2 * period)) - 1)
        rep = xp.zeros(n_low, device)
        for k in range(0, 20, period):
            rep = (rep << (2 * period)) | u
        rep = _lsr(rep, 2 * ((-20) % period)) & MASK40 if 20 % period else rep & MASK40
        tract = xp.where(period_sel == sel, rep, tract)
    parts.append(_substitute(tract, xp.scalar(1966, i), seed + 200, i, xp, device))   # 3 % per base: most copies exact or one off
    if plant_guides is not None and plant_guides.shape[0] > 0:
        g = (_lsr(plant_guides, 6) & MASK40)[::100]
        parts += [g] + [_mutate(g, lvl, seed + 17, xp, device) for lvl in (1, 2, 3, 4)]
    mer, dup = xp.unique_counts(xp.cat(parts))
    T = int(mer.shape[0])
    h = splitmix64(seed + 2, mer)
    pam_n = h & 3
    count = xp.minimum(xp.asint(dup), xp.scalar(32767, mer))
    targets = (mer << 6) | (pam_n << 4) | 0b1010 | (count << 48)
    csum = xp.cumsum(count)
    P = int(csum[-1]) if T else 0
    pos_off = xp.cat([xp.zeros(1, device), csum])
    owner = xp.repeat(xp.arange(T, device), count)
    hp = splitmix64(seed + 3, xp.arange(P, device)) ^ splitmix64(seed + 4, owner)
    contig = (_lsr(hp, 1) % 24) + 1
    pos = _lsr(hp, 8) & ((1 << 27) - 1)
    strand = _lsr(hp, 40) & 1
    positions = (strand << 60) | (23 << 52) | (contig << 32) | pos
    return {"targets": targets, "T": T, "positions": positions, "pos_offsets": pos_off, "P": P}


def make_guides_from_database(db, n_guides, seed=GUIDE_SEED, device=None):
    """n_guides distinct guides drawn from the database BY GENOMIC POSITION (a target with 1000 copies is 1000 times as likely as a
    unique one -- what tiling a genome does), in a hash order that has nothing to do with the database order.  Guide long = the
    target's 23-mer with count 1."""
    xp = _ns(device)
    P, T = db["P"], db["T"]
    m = min(int(n_guides * 1.5) + 64, max(P, 1))
    pick = splitmix64(seed + 7, xp.arange(m, device)) % max(P, 1)
    pick = xp.where(pick < 0, pick + max(P, 1), pick)
    owner = xp.searchsorted(db["pos_offsets"], pick) - 1
    owner = xp.unique(owner)
    order = xp.argsort(splitmix64(seed + 8, owner))
    owner = owner[order][:n_guides]
    t = db["targets"][owner]
    return (t & ((1 << 48) - 1)) | (1 << 48)


def as_u64(x):
    """view an int64 numpy array as uint64 (the C-ABI takes uint64_t*)"""
    return np.ascontiguousarray(x).view(np.uint64)


CONTIGS_24 = ["chr%s" % c for c in list(range(1, 23)) + ["X", "Y"]]
