#!/usr/bin/env python3
"""Write the inputs of cases [first, last] of a tools/stress_parity.py sweep (same seed => same cases) as raw files for
tools/oracle_replay.c -- the checker alone, rebuilt under MemorySanitizer / AddressSanitizer / MALLOC_PERTURB_ (round 6: what made the
in-process checker answer twice differently in sweep B, case 1505 of seed 4101).  CPU only.

  python tools/oracle_case_dump.py <sweep seed> <first case> <last case> <out dir>

File layout (little-endian): u32 enzyme, max_mm, max_ot, kind; u64 n_targets, n_positions, n_guides; then the three u64 arrays."""
import os, sys, struct
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stress_cases
import oracle_lib

seed, first, last, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.makedirs(out, exist_ok=True)
oracle = oracle_lib.load()
rng = np.random.default_rng(seed)
for n in range(1, last + 1):
    c = stress_cases.draw(rng)
    if n < first:
        continue
    odb, t, p, g = stress_cases.build(oracle, c)
    del odb
    path = os.path.join(out, "case_%d_%06d.bin" % (seed, n))
    with open(path, "wb") as f:
        f.write(struct.pack("<4I3Q", c["enz"], c["max_mm"], c["max_ot"], c["kind"], len(t), len(p), len(g)))
        f.write(np.ascontiguousarray(t, dtype=np.uint64).tobytes())
        f.write(np.ascontiguousarray(p, dtype=np.uint64).tobytes())
        f.write(np.ascontiguousarray(g, dtype=np.uint64).tobytes())
    print("case %d kind %d enzyme %d mm %d max_ot %d par %s: T %d P %d G %d -> %s" % (n, c["kind"], c["enz"], c["max_mm"], c["max_ot"], c["par"], len(t), len(p), len(g), path), flush=True)
