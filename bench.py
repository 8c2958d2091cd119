#!/usr/bin/env python3
"""bench.py -- FlashFry `discover` on MI355X: guide x target comparisons per second on the hg38-scale workload.

One "step" = one complete pass of the hot path over the resident database shard with the whole guide batch:
candidate lists -> compare kernel -> hit ordering -> ordered cut-off -> CFD/Hsu2013 scoring -> per-guide aggregates
(device-resident results; only the per-guide summaries travel to the host).  The database and its two bucketed scan
images are already in HBM when the timed region starts (ffh_db_load_soa is outside it; its device time is reported
as db_prepare_ms).  Inputs are synthetic and seeded (flashfry_amd/synth.py) -- there is no genome on the box.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1, --scaling strong (default, BASELINE.json configs[3]): ONE hg38-sized database, its 16384 bins split statically and
contiguously over the ranks (balanced by targets per bin, the synthetic stand-in of BinaryHeader's uncompressedSize); --scaling weak:
every rank owns its own hg38-sized shard (a genome N times larger).  Either way the only exchange is the per-guide totals
all-gather (ordered cut-off across shards) and the per-guide aggregate reduction over RCCL.

After the timed loop the bench verifies its own step (N = 1): the aggregates-only summaries must equal those of one list-delivering
ffh_discover, whose hit lists are checked against an independent brute-force torch scan for sampled guides ("verified": true), and
that complete discover (hit lists + positions on the host) is timed beside the step ("discover_with_lists_ms").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--targets", type=float, default=3.0e8, help="unique targets per GPU shard (hg38 NGG ~ 3e8)")
    ap.add_argument("--guides", type=int, default=100000)
    ap.add_argument("--max-mismatch", type=int, default=4)
    ap.add_argument("--max-offtargets", type=int, default=2000)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--workload", default="hg38-scale")
    ap.add_argument("--plan", default="", help="force the candidate split: PREFIX_BASES,PREFIX_RADIUS (tuning aid; default: the library's cost model)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong", help="N > 1: split one hg38-sized database (strong) or give every rank its own (weak)")
    ap.add_argument("--no-verify", action="store_true", help="skip the self-verification and the list-delivering discover after the timed loop")
    ap.add_argument("--no-skewed", action="store_true", help="skip the second workload (repeat-structured genome, guides sampled from it)")
    ap.add_argument("--no-c2", action="store_true", help="skip the chr22-scale leg (configs[1]: 1000 guides, resident step + CLI wall time)")
    ap.add_argument("--pipelined", action="store_true", help="also time distinct guide batches with two calls in flight (ffh_pipe) against sequential calls")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes that measure the compare kernel's HBM traffic")
    ap.add_argument("--traffic-dir", default=os.path.join(ROOT, "gpurun_out", "traffic"), help="where the PMC passes write their CSVs")
    return ap.parse_args()


def measure_traffic(args):
    """HBM bytes per k_compare launch from the PMC counters, collected as MI355X_MICROARCH.md prescribes: separate
    `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE) with --kernel-trace only, values in KiB, and FETCH_SIZE
    calibrated on a kernel of the same access pattern with a known byte count: ffh::k_image_hist reads exactly 8 B per
    resident target with the same 8-byte-per-lane coalesced loads (gfx950 reports half of the bytes of such streams)."""
    import csv
    import shutil
    import subprocess
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    out = {}
    passes = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"],
              "SQ": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"]}
    for tag, counters in passes.items():
        d = os.path.join(args.traffic_dir, tag)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d, exist_ok=True)
        cmd = [prof, "--kernel-trace", "--pmc"] + counters + ["--kernel-include-regex", "k_compare<|k_image_hist", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-traffic", "--no-verify", "--no-skewed", "--no-c2",
               "--targets", str(args.targets), "--guides", str(args.guides), "--max-mismatch", str(args.max_mismatch), "--max-offtargets", str(args.max_offtargets)]
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, timeout=300)  # a pass takes ~40 s; a hung profiler must not hold the bench
        path = os.path.join(d, "pmc_counter_collection.csv")
        if r.returncode != 0 or not os.path.exists(path):
            if tag == "SQ":
                continue  # the instruction counters are an extra; the traffic figure stands without them
            return None
        for counter in counters:
            vals = {}
            for row in csv.DictReader(open(path)):
                if row["Counter_Name"] == counter:
                    vals.setdefault(row["Kernel_Name"].split("(")[0], []).append(float(row["Counter_Value"]))
            out[counter] = {k: sum(v) / len(v) for k, v in vals.items()}
    try:
        cmp_fetch = [v for k, v in out["FETCH_SIZE"].items() if "k_compare" in k][0]
        cmp_write = [v for k, v in out["WRITE_SIZE"].items() if "k_compare" in k][0]
        cal = [v for k, v in out["FETCH_SIZE"].items() if "k_image_hist" in k][0]
    except (IndexError, KeyError):
        return None
    res = {"fetch_kib": cmp_fetch, "write_kib": cmp_write, "calibration_fetch_kib": cal}
    for counter in passes["SQ"]:
        v = [v for k, v in out.get(counter, {}).items() if "k_compare" in k]
        if v:
            res[counter] = v[0]
    return res


_CPU_MT = None
_CPU_JVM = None
# Issue model of the compare kernel (tools/ubench/op_rate.hip measures the rates): a wave64 VALU instruction
# occupies its SIMD for ~2.8 cycles if it is one of the full-rate integer operations (v_xor, v_or, v_add, shifts, v_bitop3) and for
# ~4.4 otherwise (v_bcnt, v_min*, v_cmp, SDWA / DPP forms, scalar-register operands).  The bit-sliced pair test is made of
# full-rate operations only.
VALU_CYCLES_FULL_RATE = 2.8
VALU_CYCLES_HALF_RATE = 4.4


def epilogue_roofline(raw_hits, ms):
    """the second kernel of the step.  `achieved` / `frac` by SURVEY.md section 8d's figure -- 16 B per hit (the 8-byte record + the 8-byte target
    long it gathers); beside it what the memory system moves for that: one 128-byte line per gathered target long + the 8-byte key
    (`line_*`).  None when the step's epilogue was not timed on its own (the sharded step: it runs inside ffh_discover_sharded)."""
    if not ms or ms <= 0 or not raw_hits:
        return None
    gbps = 16 * raw_hits / (ms * 1e-3) / 1e9
    line = 136 * raw_hits / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "ffh::k_guide_epilogue", "launch_ms": ms, "algorithmic_bytes_per_launch": 16 * raw_hits,
            "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
            "line_bytes_per_launch": 136 * raw_hits, "line_achieved": line, "line_frac": line / 8000.0,
            "note": "frac prices SURVEY 8d's 16 B per hit; line_frac the 128-byte line every randomly gathered target long costs + the 8-byte key (8.5 x as much)"}


def pair_step_valu(rest_bases, far):
    """VALU instructions of one bit-sliced step (64 lanes x 32 targets = 2048 pair tests) with `rest_bases` bases outside the bucket
    id: two per base for the mismatch words, the carry-save adder tree (one v_bitop3 per sum / per carry), four for count <= budget,
    one for the valid word, two more for count > r1 on the suffix image (compile-time form of the per-plan kernel instances)"""
    n, ops = [rest_bases, 0, 0, 0], 0
    for lv in range(4):
        while n[lv] >= 3:
            n[lv] -= 2
            ops += 2 if lv < 3 else 1
            if lv < 3:
                n[lv + 1] += 1
        if n[lv] == 2:
            n[lv] = 1
            ops += 2 if lv < 3 else 1
            if lv < 3:
                n[lv + 1] += 1
    return 2 * rest_bases + ops + 5 + (2 if far else 0)   # (the far condition is two v_bitop3 in the instances compiled for a plan: scan_row<R, FAR>)


def cpu_baseline_all_cores(oracle, run, guides_np, max_mm, max_ot, nb_sample, sample_targets):
    """The same port on all usable host cores as an actual run split over the bins (bins are independent): the
    sample's bins are dealt to one thread per core in contiguous ranges, every thread runs the reference's linear traversal --
    guide x bin prefix filter + block compare -- over ITS bins only (oracle/ff_oracle.c: ffo_discover_bin_range; ctypes releases the
    GIL inside the C call) and the wall time of the slowest thread is what counts.  A full run over the 16384 bins splits the same
    way, so the rate of the sample is the rate of the full run."""
    import ctypes
    import threading
    from flashfry_amd import capi
    threads = max(1, int(capi.load_library().ffh_host_threads()))
    lib = oracle.lib
    g = np.ascontiguousarray(guides_np, dtype=np.uint64)
    gp = g.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    nb = max(threads, nb_sample)
    odb = run.build(nb)
    cuts = [nb * t // threads for t in range(threads + 1)]
    times = [0.0] * threads

    def work(k):
        t0 = time.perf_counter()
        r = lib.ffo_discover_bin_range(odb.h, gp, len(g), max_mm, max_ot, cuts[k], cuts[k + 1])
        times[k] = time.perf_counter() - t0
        if r:
            lib.ffo_result_free(r)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    wall = time.perf_counter() - t0
    rate = len(g) * odb.sample_targets / wall
    return {"value": rate, "unit": "guide*target comparisons/s", "cores": threads, "kind": "port",
            "sample": "oracle/ff_oracle.c, a run split over the bins: %d threads (the usable cores of this box), each running the reference's linear traversal "
                      "(prefix filter + block compare) over its contiguous share of the first %d of 16384 bins (%d targets), all %d guides: %.2f s wall "
                      "(slowest thread %.2f s, fastest %.2f s)" % (threads, nb, odb.sample_targets, len(g), wall, max(times), min(times)),
            "seconds": wall, "projected_full_run_seconds": wall * 16384.0 / nb}


def cpu_baseline(targets_dev, pos_off_dev, positions_dev, guides_np, max_mm, max_ot, budget_s):
    """the oracle (C restatement of the reference algorithm, single thread) timed on a bounded sample: the first
    `nbins` of the 16384 database bins, all guides"""
    import torch
    from tests import oracle_lib
    from flashfry_amd import capi, synth
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_legs
    oracle = oracle_lib.load()
    # the single-thread leg runs pinned to one core, like the reference's own timing harness (`taskset -c 0`,
    # paper/tools/flashfry_off_target.cwl:16-22); its peak RSS is reported as paper/run_timing_collection.py:17-24 does
    core, unpin = bench_legs.pin_to_core()
    sample_arrays = {}

    def build(nbins):
        limit = nbins << 32  # 7-base bin = bits [45:32] of a Cas9 23-mer
        seq = targets_dev & ((1 << 46) - 1)
        idx = int(torch.searchsorted(seq, torch.tensor([limit], device=seq.device, dtype=seq.dtype))[0])
        t = targets_dev[:idx].cpu().numpy().view(np.uint64)
        p = positions_dev[:int(pos_off_dev[idx])].cpu().numpy().view(np.uint64)
        sample_arrays["t"], sample_arrays["p"] = t, p
        odb = oracle.db_from_sorted(3, t, p, contigs=synth.CONTIGS_24)
        odb.sample_targets = idx
        return odb

    def run(nbins):
        odb = build(nbins)
        t0 = time.perf_counter()
        res = odb.discover(guides_np, max_mm, max_ot)
        dt = time.perf_counter() - t0
        return odb.sample_targets, dt, res

    run.build = build

    # The reference's run has a fixed part -- the guide x bin prefix filter over ALL 16384 bins (LinearTraversal.scala:82-97),
    # the same whatever the sample holds -- and a part proportional to the bins scanned.  Timing a few bins and dividing by
    # their targets would charge the whole filter to them (and understate the CPU ~30x), so the two parts are timed apart:
    # F = a run over a database whose bins are all empty, c = (run over the first nb bins - F) / nb; the rate quoted is
    # the one a full run would have, targets_per_bin / (c + F / 16384).
    n_bins = 16384
    _, filt, _ = run(0)
    G = len(guides_np)
    # sample size from a nominal rate (3e8 full comparisons/s; ~0.76 % of the pairs of a bin survive its prefix filters), so the
    # fixed part is paid twice, not three times; a second round only if that guess left most of the budget unused
    est_per_bin = max(G * (targets_dev.numel() / n_bins) * 0.0076 / 3e8, 1e-5)
    nb = int(max(4, min(n_bins // 4, budget_s / est_per_bin)))
    idx, dt, res = run(nb)
    spent = filt + dt
    per_bin = max(dt - filt, 1e-4) / nb
    if (dt - filt) < budget_s / 4 and nb < n_bins // 4:
        nb = int(min(n_bins // 4, budget_s / per_bin))
        idx, dt, res = run(nb)
        spent += dt
        per_bin = max(dt - filt, 1e-4) / nb
    full_run = filt + per_bin * n_bins
    single = G * (idx / nb) * n_bins / full_run
    rss = bench_legs.max_rss_kib()
    unpin()
    global _CPU_MT, _CPU_JVM
    _CPU_MT = None
    _CPU_JVM = None
    try:  # the real reference, when the box has a JVM and the jar; skipped cleanly otherwise
        _CPU_JVM = bench_legs.jvm_leg(capi, sample_arrays["t"], sample_arrays["p"], synth.CONTIGS_24, guides_np, max_mm, max_ot,
                                      "the first %d of %d bins of the synthetic database" % (nb, n_bins))
    except Exception as e:
        _CPU_JVM = {"value": None, "unit": "guide*target comparisons/s", "cores": 1, "kind": "reference", "sample": "failed: %r" % (e,)}
    try:  # SURVEY.md section 8d also asks for the port on all host cores (bins are independent: one thread per bin range)
        _CPU_MT = cpu_baseline_all_cores(oracle, run, guides_np, max_mm, max_ot, nb, idx)
    except Exception as e:
        _CPU_MT = {"value": None, "unit": "guide*target comparisons/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    return {"value": G * (idx / nb) * n_bins / full_run, "unit": "guide*target comparisons/s", "cores": 1, "kind": "port",
            "sample": "oracle/ff_oracle.c (C restatement of the reference loop structure, not the JVM), 1 thread, all %d guides vs the first %d of "
                      "%d bins (%d targets): %.2f s, of which %.2f s is the guide x bin prefix filter over all %d bins (timed alone on an empty "
                      "database); value = the rate of a full run, targets per bin / (scan seconds per bin + filter seconds / %d)"
                      % (G, nb, n_bins, idx, dt, filt, n_bins, n_bins),
            "seconds": spent, "filter_seconds": filt, "scan_seconds_per_bin": per_bin, "projected_full_run_seconds": full_run,
            "sample_only_rate": G * idx / dt, "executed_comparisons": int(res.all_comparisons), "sample_targets": idx,
            "pinned_to_core": core, "max_rss_kib": rss}


def torch_mismatches(torch, guide, targets):
    """BitEncoding.mismatches (bitcoding/BitEncoding.scala:127-132) with torch integer ops over the whole database: an independent
    restatement (no planar keys, no buckets) used to check the step's hit lists"""
    x = (targets ^ guide) & 0x3FFFFFFFFFC0           # comparisonBitEncoding of spCas9-NGG, StandardScanParameters.scala:143
    y = (x | (x << 1)) & 0xAAAAAAAAAAAA              # BitEncoding.scala:205
    y = y - ((y >> 1) & 0x5555555555555555)
    y = (y & 0x3333333333333333) + ((y >> 2) & 0x3333333333333333)
    y = (y + (y >> 4)) & 0x0F0F0F0F0F0F0F0F
    return (y * 0x0101010101010101) >> 56 & 0x7F


def verify_step(torch, ctx, db, guides_np, step_result, args):
    """The timed step only brings the per-guide aggregates to the host.  Here the same call is made once more with the hit lists and
    positions delivered (timed: that is the complete `discover` product), its summaries must be the step's bit for bit, and for a
    sample of guides the delivered hit list must be exactly what a brute-force scan of all targets + the ordered cut-off
    (CRISPRSiteOT.scala:39-46) gives.  Raises on any difference."""
    G = len(guides_np)
    variants = (("no_positions", dict(positions=False, hit_scores=False)),   # what `discover` delivers unless --positionOutput is given
                ("lists", dict(hit_scores=False)),                           # sequences, counts, mismatches AND positions: the reference's product
                ("lists_and_hit_scores", dict()))                            # + the per-hit pam*cfd array (an extra of this library)
    med = {}
    full = None
    for name, kw in variants:
        times = []
        for _ in range(3):
            full = None      # the previous result goes back to the page-locked pool first
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            full = ctx.discover(guides_np, args.max_mismatch, args.max_offtargets, **kw)
            times.append((time.perf_counter() - t0) * 1e3)
        med[name] = float(np.median(times))
        if full.summaries.tobytes() != step_result.summaries.tobytes():
            raise SystemExit("bench verification failed: aggregates-only summaries differ from the list-delivering discover (%s)" % name)
        if name == "no_positions" and full.positions is not None:
            raise SystemExit("bench verification failed: the position-less discover delivered positions")
    sample = sorted(set(list(range(0, G, max(1, G // 12)))[:12] + [1, G // 3 + 1, G - 1]))
    cnt_all = (db["targets"] >> 48) & 0xFFFF
    for g in sample:
        mm = torch_mismatches(torch, int(guides_np[g].astype(np.int64)), db["targets"])
        idx = torch.nonzero(mm <= args.max_mismatch).flatten()
        c = cnt_all[idx]
        before = torch.cumsum(c, 0) - c
        keep = idx[before < args.max_offtargets]                      # kept while the running total before the hit is below the limit
        want = db["targets"][keep].cpu().numpy().view(np.uint64)
        got = full.hits(g)
        if not np.array_equal(got, want):
            raise SystemExit("bench verification failed: hit list of guide %d differs from the brute-force scan" % g)
        a = int(full.guide_offsets[g])
        if len(keep):
            po = full.pos_offsets[a:a + len(keep) + 1]
            k = len(keep) - 1
            src = int(db["pos_offsets"][keep[k]])
            n = int(po[k + 1] - po[k])
            if not np.array_equal(full.positions[int(po[k]):int(po[k + 1])], db["positions"][src:src + n].cpu().numpy().view(np.uint64)):
                raise SystemExit("bench verification failed: positions of guide %d differ" % g)
    note = ("summaries of the timed aggregates-only step == summaries of a list-delivering ffh_discover (bytes); hit lists and positions of %d "
            "sampled guides == brute-force torch scan of all targets + ordered cut-off" % len(sample))
    return True, med, note


def pipelined_leg(torch, capi, synth, ctx, dev, args, n_batches=6, lanes=2):
    """Distinct guide batches against the resident database, one after the other (what GPUTraverser.scala's grouped loop and the reference's
    traverser do) and with `lanes` of them in flight (ffh_pipe_*: sharing contexts on the same database, one host thread each).  Both hand
    the guides over as host buffers (a pipe copies them at submit), both deliver the per-guide aggregates; the results must be the same
    bytes.  Reported BESIDE the synchronous step, never as `value`."""
    G = args.guides
    batches = [synth.make_guides(G, seed=synth.GUIDE_SEED + 1001 + 17 * k, device=dev).cpu().numpy().view(np.uint64) for k in range(n_batches)]
    run = lambda g: ctx.discover(g, args.max_mismatch, args.max_offtargets, summaries_only=True)
    for g in batches[:2]:
        run(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    seq = [run(g) for g in batches]
    t_seq = (time.perf_counter() - t0) * 1e3 / n_batches
    with ctx.pipe(lanes) as pipe:
        for _ in range(2):   # (every lane's buffers, graphs and images warm)
            for t in [pipe.submit(g, args.max_mismatch, args.max_offtargets, summaries_only=True) for g in batches[:2 * lanes]]:
                pipe.wait(t)
        t0 = time.perf_counter()
        got = [pipe.wait(t) for t in [pipe.submit(g, args.max_mismatch, args.max_offtargets, summaries_only=True) for g in batches]]
        t_pipe = (time.perf_counter() - t0) * 1e3 / n_batches
    same = all(a.summaries.tobytes() == b.summaries.tobytes() for a, b in zip(seq, got))
    if not same:
        raise SystemExit("bench: a pipelined batch differs from the sequential call")
    return {"what": "%d distinct batches of %d guides (host buffers), aggregates to the host: sequential ffh_discover calls against %d calls in flight (ffh_pipe)" % (n_batches, G, lanes),
            "lanes": lanes, "batches": n_batches, "sequential_ms_per_batch": t_seq, "pipelined_ms_per_batch": t_pipe, "gain": 1.0 - t_pipe / t_seq, "identical": same}


def skewed_workload(torch, capi, synth, ctx_uniform, dev, local, args):
    """The second workload: the same sizes on a repeat-structured genome (synth.make_repeat_database) with the guides sampled from the
    genome by position, so that repeat families get their share of guides -- the heavy-tailed case real hg38 is, next to the uniform
    one.  Reported beside the headline, never instead of it."""
    T_req, G = int(args.targets), args.guides
    db = synth.make_repeat_database(T_req, seed=synth.DB_SEED + 99, device=dev)
    guides = synth.make_guides_from_database(db, G, device=dev).cpu().numpy().view(np.uint64)
    T, P = db["T"], db["P"]
    cnt = (db["targets"] >> 48) & 0xFFFF
    stats = {"targets": T, "positions": P, "max_count": int(cnt.max()), "targets_with_count_ge_100": int((cnt >= 100).sum())}
    with capi.Context(3, device=local) as ctx:
        torch.cuda.synchronize()
        ctx.load_soa_device(db["targets"].data_ptr(), T, db["positions"].data_ptr(), P)
        del db, cnt
        torch.cuda.empty_cache()
        def run(mode):
            ctx.set_bounding(mode)
            res = ctx.discover(guides, args.max_mismatch, args.max_offtargets, summaries_only=True)   # warm-up (buffers grow, slab images are built here)
            times, tms = [], []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = ctx.discover(guides, args.max_mismatch, args.max_offtargets, summaries_only=True)
                times.append((time.perf_counter() - t0) * 1e3)
                tms.append(ctx.timings().as_dict())
            return res, float(np.median(times)), tms
        # the scan as ffh_discover runs it (bounding switches itself on for a guide set like this one after the first call; forced on
        # here so that the first timed call already is a bounded one), then the same with bounding off
        res, ms, tms = run(1)
        # ... and the complete product of the bounded call: hit lists and positions on the host (what the CLI asks for)
        lists = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            full = ctx.discover(guides, args.max_mismatch, args.max_offtargets, hit_scores=False)
            lists.append((time.perf_counter() - t0) * 1e3)
        if full.summaries.tobytes() != res.summaries.tobytes():
            raise SystemExit("bench: the list-delivering bounded call's aggregates differ from the aggregates-only call's")
        kept_pos = int(full.n_positions)
        del full
        res_u, ms_u, tms_u = run(0)
        if res.summaries.tobytes() != res_u.summaries.tobytes():
            raise SystemExit("bench: the bounded scan's aggregates differ from the unbounded scan's")
        s = res.summaries
        bd = lambda tt: {k: float(np.mean([t[k] for t in tt])) for k in ("prepare_ms", "compare_ms", "sort_ms", "finalize_ms", "total_scan_ms")}
        out = {"workload": "hg38-skewed: %d guides sampled by position from a repeat-structured genome of %d distinct targets (%d positions), <=%d mismatches, "
                           "maximumOffTargets %d" % (len(guides), T, P, args.max_mismatch, args.max_offtargets),
               "ms_per_step": ms, "value": len(guides) * T / (ms * 1e-3), "unit": "comparisons/s", "breakdown_ms": bd(tms),
               "discover_with_lists_ms": float(np.median(lists[1:])), "kept_positions": kept_pos,
               "raw_hits": int(tms[-1]["n_raw_hits"]), "raw_hits_per_guide": tms[-1]["n_raw_hits"] / max(len(guides), 1),
               "bounded_slabs": int(tms[-1]["bounded_slabs"]), "retired_guides": int(tms[-1]["retired_guides"]),
               # ffh_set_bounding(0): every guide meets the whole database, every raw hit is kept and sorted (round 1's only mode)
               "unbounded": {"ms_per_step": ms_u, "breakdown_ms": bd(tms_u), "raw_hits": int(tms_u[-1]["n_raw_hits"])},
               "overflowed_guides": int(s["overflow"].sum()), "kept_hits": int(s["n_hits"].sum()), "genome": stats}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from flashfry_amd import capi, synth
    from flashfry_amd import dist as ffdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("FFH_BENCH_SAME_GPU") == "1":
        local = 0  # test aid: all ranks share GPU 0 (with FFH_BENCH_BACKEND=gloo), to run the multi-rank path on a 1-GPU box
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # FFH_BENCH_FORCE_EXCHANGE=1 runs the sharded step (device-resident exchange over RCCL) with a single rank: a way to
    # exercise and time that code path on a 1-GPU box
    sharded = world > 1 or os.environ.get("FFH_BENCH_FORCE_EXCHANGE") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("FFH_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    T_req, G = int(args.targets), args.guides
    # ---- synthetic inputs, generated on the device (same generator as the tests) ----
    guides_dev = synth.make_guides(G, device=dev)
    strong = world > 1 and args.scaling == "strong"
    db = synth.make_database(T_req, seed=synth.DB_SEED + (0 if strong else rank), plant_guides=guides_dev, device=dev)
    T_db = db["T"]
    if strong:
        # config C4: the SAME database, its 16384 bins (7-base prefix, bits 45:32 of a Cas9 23-mer) split contiguously over the ranks,
        # balanced by the bins' payload (targets + positions longs = BinaryHeader's uncompressedSize without the block header)
        bins = ((db["targets"] >> 32) & 0x3FFF)
        per_bin_t = torch.bincount(bins, minlength=16384)
        first = torch.cumsum(per_bin_t, 0) - per_bin_t
        bin_pos = db["pos_offsets"][torch.clamp(first + per_bin_t, max=T_db)] - db["pos_offsets"][torch.clamp(first, max=T_db)]
        payload = ((per_bin_t + bin_pos) * 8).cpu().numpy()
        b0, b1 = ffdist.shard_bins(payload, world)[rank]
        lo = int(first[b0]) if b0 < 16384 else T_db
        hi = int(first[b1]) if b1 < 16384 else T_db
        plo, phi = int(db["pos_offsets"][lo]), int(db["pos_offsets"][hi])
        db = {"targets": db["targets"][lo:hi].contiguous(), "positions": db["positions"][plo:phi].contiguous(),
              "pos_offsets": (db["pos_offsets"][lo:hi + 1] - plo).contiguous(), "T": hi - lo, "P": phi - plo}
        torch.cuda.empty_cache()
    T, P = db["T"], db["P"]
    guides_np = guides_dev.cpu().numpy().view(np.uint64)
    ctx = capi.Context(3, device=local)
    if args.plan:
        ctx.set_plan(*[int(x) for x in args.plan.split(",")])
    torch.cuda.synchronize()
    ctx.load_soa_device(db["targets"].data_ptr(), T, db["positions"].data_ptr(), P)
    info = ctx.info()

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        try:
            cpu = cpu_baseline(db["targets"], db["pos_offsets"], db["positions"], guides_np, args.max_mismatch, args.max_offtargets, args.cpu_seconds)
        except Exception as e:  # the baseline is a reported extra, never a reason to lose the measurement
            cpu = {"value": None, "unit": "guide*target comparisons/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
    verify_db = None
    if rank == 0 and world == 1 and not args.no_verify:
        verify_db = {"targets": db["targets"], "pos_offsets": db["pos_offsets"], "positions": db["positions"]}
    del db
    torch.cuda.empty_cache()
    pmc = None
    if rank == 0 and world == 1 and not args.no_traffic:
        try:
            pmc = measure_traffic(args)
        except Exception:
            pmc = None

    # device copy rate measured in the same run (SURVEY.md section 8d asks for the roofline fraction against it too)
    stream_gbps = None
    if rank == 0:
        a = torch.empty(1 << 27, dtype=torch.int64, device=dev).random_()
        b = torch.empty_like(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(4):
            e0.record()
            b.copy_(a)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        stream_gbps = 2 * a.numel() * 8 / (best * 1e-3) / 1e9
        del a, b
        torch.cuda.empty_cache()

    # The sharded step: by default ONE library call per step (ffh_discover_sharded: scan + totals all-gather + prior + fix-up + the
    # three reductions, the collectives issued by the library itself through RCCL on its own stream; torch.distributed only carries
    # the 128-byte unique id once).  FFH_BENCH_EXCHANGE=torch keeps the round-2 form (library kernels + torch.distributed
    # collectives on torch's stream) -- also what the gloo rehearsal of several ranks on ONE GPU has to use (RCCL refuses duplicate
    # devices).
    exchange_kind = os.environ.get("FFH_BENCH_EXCHANGE", "torch" if os.environ.get("FFH_BENCH_SAME_GPU") == "1" else "native") if sharded else None
    exch, comm, reduced = None, None, None
    exchange_note = None

    def all_ranks_ok(ok):   # every rank must take the same path: one that failed takes the others with it
        if world <= 1:
            return ok
        flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    if sharded and exchange_kind == "native":
        # the library's own communicator; should it not come up on this node (it has never run on more than one GPU), or fail in its
        # first step, all ranks fall back to the torch.distributed form of the same exchange -- and the line says so
        # Every rank takes every collective of this block, whatever failed where (a rank that skipped the broadcast left the
        # others waiting in it): rank 0 always broadcasts -- the id, or None when it could not make one --, the ranks agree before any
        # of them calls ncclCommInitRank, and again after the first step.  What this cannot cover: a rank that dies INSIDE
        # ncclCommInitRank leaves its peers blocked there (RCCL's own rendezvous).
        err = None
        uid = [None]
        if rank == 0:
            try:
                uid = [capi.comm_unique_id()]
            except Exception as e:   # noqa: BLE001
                err = repr(e)
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        if all_ranks_ok(uid[0] is not None):
            try:
                comm = capi.Comm.rank(ctx, rank, world if world > 1 else 1, uid[0])
            except Exception as e:   # noqa: BLE001
                err = repr(e)
            if all_ranks_ok(comm is not None):
                try:
                    reduced = capi.host_summaries(G)   # page-locked (ffh_host_alloc): the reduced aggregates are copied straight into it
                    comm.discover_device(guides_dev.data_ptr(), int(guides_dev.shape[0]), args.max_mismatch, args.max_offtargets, want_summaries=(rank == 0), out=reduced)
                except Exception as e:   # noqa: BLE001 -- whatever it is, the run goes on with the other exchange
                    err = repr(e)
            elif err is None:
                err = "another rank could not create its communicator"
        elif err is None:
            err = "rank 0 could not make the RCCL unique id"
        if not all_ranks_ok(err is None):
            exchange_note = "ffh_comm failed on a rank (%s); torch.distributed exchange used instead" % (err or "another rank")
            if comm is not None:
                try:
                    comm.close()
                except Exception:   # noqa: BLE001
                    pass
            comm, exchange_kind = None, "torch"
    if sharded and comm is None:
        exch = ffdist.DeviceExchange(G, dev)

    class _Reduced:  # the sharded step's result: the reduced per-guide aggregates, on rank 0
        summaries = None

    # The timed step takes the guide set from HBM (the contract: inputs resident in device memory when the timed region starts); the
    # same step with the guides handed over as a host buffer (800 KB staged through the link every call) is timed after the loop and
    # reported as "ms_per_step_host_guides".
    gptr, G_dev = guides_dev.data_ptr(), int(guides_dev.shape[0])

    def step(host_guides=False):
        if not sharded:
            if host_guides:
                return ctx.discover(guides_np, args.max_mismatch, args.max_offtargets, summaries_only=True)  # ffh_discover = ffh_scan + ffh_finalize
            return ctx.discover_device(gptr, G_dev, args.max_mismatch, args.max_offtargets, summaries_only=True)
        res = _Reduced()
        if comm is not None:   # scan + exchange inside the library; rank 0 takes the reduced aggregates to the host like the single-GPU step does
            comm.discover_device(gptr, G_dev, args.max_mismatch, args.max_offtargets, want_summaries=(rank == 0), out=reduced)
            res.summaries = reduced if rank == 0 else None
            return res
        ctx.scan_device(gptr, G_dev, args.max_mismatch)
        # bin shards: every shard aggregates on its own and reports its totals -> all-gather -> the guides whose ordered cut-off the
        # earlier shards move are aggregated again -> reduction of the aggregates; all on device memory, stream-ordered
        exch.step(ctx, args.max_offtargets)
        if rank == 0:
            res.summaries = exch.summaries_numpy()
        return res

    for _ in range(args.warmup):
        res = step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    tms = []
    for _ in range(args.steps):
        res = step()
        tms.append(ctx.timings().as_dict())
    fence()
    dt = time.perf_counter() - t0
    host_ms = None
    if not sharded:
        fence()
        th = time.perf_counter()
        for _ in range(args.steps):
            res_h = step(host_guides=True)
        fence()
        host_ms = (time.perf_counter() - th) * 1e3 / args.steps
        if res_h.summaries.tobytes() != res.summaries.tobytes():
            raise SystemExit("bench: the step fed from a host buffer differs from the step fed from device memory")
    # what every rank did, for the scaling curve: its shard, its own wall time per step, the compare launch, the library's scan / exchange split
    mine = {"rank": rank, "targets": int(T), "ms_per_step": dt / args.steps * 1e3, "compare_ms": float(np.mean([t["compare_ms"] for t in tms])),
            "prepare_ms": float(np.mean([t["prepare_ms"] for t in tms])), "sort_ms": float(np.mean([t["sort_ms"] for t in tms])),
            "raw_hits": int(np.mean([t["n_raw_hits"] for t in tms]))}
    if comm is not None:
        try:
            mine.update(comm.timings())   # (of the last step: scan_ms = the shard's scan, exchange_ms = epilogue + all-gather + fold (+ rank 0's copy-out))
        except Exception:   # noqa: BLE001
            pass
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        tsum = torch.tensor([T], device=dev, dtype=torch.int64)
        dist.all_reduce(tsum)
        T_total = int(tsum[0])
    else:
        T_total = T

    verified, lists_ms, verify_note = None, None, None
    if verify_db is not None:
        verified, lists_ms, verify_note = verify_step(torch, ctx, verify_db, guides_np, res, args)
        verify_db = None
        torch.cuda.empty_cache()

    skewed, real_genome, c2, pipelined = None, None, None, None
    if rank == 0 and world == 1 and args.pipelined:
        try:
            pipelined = pipelined_leg(torch, capi, synth, ctx, dev, args)
        except SystemExit:
            raise
        except Exception as e:  # a reported extra, never a reason to lose the measurement
            pipelined = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_c2:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_legs
            from flashfry_amd import _build
            c2 = bench_legs.c2_leg(torch, capi, synth, local, args.max_mismatch, args.max_offtargets, cli=_build.CLI)
        except Exception as e:  # a reported extra, never a reason to lose the measurement
            c2 = {"error": repr(e)}
    if rank == 0 and world == 1:
        if not args.no_skewed:
            try:
                skewed = skewed_workload(torch, capi, synth, ctx, dev, local, args)
            except Exception as e:  # a reported extra, never a reason to lose the measurement
                skewed = {"error": repr(e)}
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_legs
            real_genome = bench_legs.fasta_leg(capi, G, args.max_mismatch, args.max_offtargets, device=local)
        except Exception as e:
            real_genome = {"error": repr(e)}

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        cmp_ms = float(np.mean([t["compare_ms"] for t in tms]))
        raw_hits = int(np.mean([t["n_raw_hits"] for t in tms]))
        final = res.summaries
        kept_pos = int(final["ot_count"].sum())
        # algorithmic bytes of ONE compare launch: every resident target once (8 B), every guide once (8 B), one 8-byte record per hit
        b_alg = 8 * T + 8 * G + 8 * raw_hits
        b_survey = 8 * T + 8 * G + 16 * raw_hits + 8 * kept_pos  # SURVEY.md §8d formula for the whole discover
        achieved = b_alg / (cmp_ms * 1e-3) / 1e9
        pairs = float(np.mean([t["pairs_prefix"] + t["pairs_suffix"] for t in tms]))
        traffic, traffic_note = None, None
        if pmc:
            # calibration: k_image_hist streams exactly 8 B per target; scale = true bytes / reported bytes (2.0 on gfx950)
            scale = (8.0 * T) / (pmc["calibration_fetch_kib"] * 1024.0)
            traffic = pmc["fetch_kib"] * 1024.0 * scale + pmc["write_kib"] * 1024.0
            traffic_note = {"fetch_size_kib": pmc["fetch_kib"], "write_size_kib": pmc["write_kib"], "fetch_scale_from_calibration": scale,
                            "calibration": "ffh::k_image_hist reads 8 B per target with the same coalesced 8-byte loads; WRITE_SIZE uncalibrated"}
        sq = {k: pmc[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU") if pmc and k in pmc}
        # 256 CUs x 4 SIMDs issue VALU instructions.  valu_issue_frac: ALL VALU instructions of the launch (PMC) against that peak, priced
        # at the half-rate cost (an upper bound) and at the full-rate cost (a lower bound: most of them are full-rate operations);
        # useful_valu_frac: only the instructions the executed pair tests need when every lane and every slot of a step is used.
        simd_cycles = 1024 * 2.4e9 * cmp_ms * 1e-3
        valu_issue = sq["SQ_INSTS_VALU"] * VALU_CYCLES_HALF_RATE / simd_cycles if "SQ_INSTS_VALU" in sq else None
        valu_issue_lo = sq["SQ_INSTS_VALU"] * VALU_CYCLES_FULL_RATE / simd_cycles if "SQ_INSTS_VALU" in sq else None
        lc = 20
        ops_p, ops_s = pair_step_valu(lc - tms[-1]["prefix_bases"], False), pair_step_valu(tms[-1]["prefix_bases"], True)
        pairs_p = float(np.mean([t["pairs_prefix"] for t in tms]))
        pairs_s = float(np.mean([t["pairs_suffix"] for t in tms]))
        useful_instr = pairs_p / 2048.0 * ops_p + pairs_s / 2048.0 * ops_s
        useful_valu = useful_instr * VALU_CYCLES_FULL_RATE / simd_cycles
        out = {
            "metric": "guide x target comparisons/s, effective = nominal G x T per step (discover, <=%d mismatches, CFD+Hsu2013 aggregate)" % args.max_mismatch,
            "value": G * T_total * args.steps / dt,
            "unit": "comparisons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": ("strong" if strong else "weak") if world > 1 else None, "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: %d random NGG guides vs %d unique targets %s (%d positions on this rank), <=%d mismatches, maximumOffTargets %d, spCas9-NGG; "
                                   "N > 1 is %s" % (args.workload, G, T_total if strong else T, "in ONE database split by bins over the ranks" if strong else "per GPU", P,
                                                    args.max_mismatch, args.max_offtargets,
                                                    "strong scaling (BASELINE.json configs[3]: the same database, bins sharded)" if args.scaling == "strong" else
                                                    "weak scaling (every rank its own hg38-sized shard)"),
                       "guides": G, "targets_per_gpu": T, "targets_total": T_total, "positions_per_gpu": P,
                       "max_mismatch": args.max_mismatch, "max_offtargets": args.max_offtargets, "parallelism": "bin-shard x%d" % world,
                       "exchange": ({"native": "ffh_discover_sharded: RCCL collectives issued by libflashfry_hip (%s)" % (comm.transport if comm else "?"),
                                     "torch": "library kernels + torch.distributed collectives"}[exchange_kind] if sharded else None),
                       "exchange_note": exchange_note},
            # executed full-length comparisons (the pigeonhole candidate generation visits ~1/4300 of the nominal G x T pairs): the figure
            # comparable with the reference's BitEncoding.allComparisons counter
            "executed_pair_tests_per_step": pairs, "executed_pair_tests_per_s": pairs * args.steps / dt,
            # what `value` times, said plainly: the aggregates-only discover (scan + ordered cut-off + CFD / Hsu2013 / closest-hit
            # aggregates of every guide on the host) with the guide set already in HBM; the hit lists themselves are NOT delivered in
            # the timed step -- that call is timed beside it as discover_product_ms (default table / with --positionOutput)
            "step_delivers": "per-guide aggregates only (guides resident in HBM); hit lists: see discover_product_ms",
            # the same step with the guide set handed over as a (pageable) host buffer: PCIe-inclusive, never `value`
            "ms_per_step_host_guides": host_ms,
            # distinct guide batches, two calls in flight against the resident database (ffh_pipe): beside the synchronous step, never `value`
            "pipelined": pipelined, "pipelined_ms_per_step": pipelined.get("pipelined_ms_per_batch") if pipelined else None,
            "verified": verified, "verification": verify_note,
            # the complete discover product: scan + cut-off + aggregates + the retained hits (target long incl. count, mismatches) and
            # their positions on the host -- what ResultsAggregator hands to the writer (CRISPRHit: sequence, count, coordinates)
            "discover_with_lists_ms": lists_ms["lists"] if lists_ms else None,
            # ... without the position arrays: what `discover` asks for unless --positionOutput is given (the reference's default table
            # prints sequence, count and mismatches of every off-target; modules/OffTargetDiscovery.scala:51-53,146): the DEFAULT product
            "discover_with_lists_no_positions_ms": lists_ms["no_positions"] if lists_ms else None,
            "discover_product_ms": {"default_table": lists_ms["no_positions"], "with_positionOutput": lists_ms["lists"]} if lists_ms else None,
            # ... with this library's per-hit pam*cfd array on top (8 more bytes per hit across the link)
            "discover_with_lists_and_hit_scores_ms": lists_ms["lists_and_hit_scores"] if lists_ms else None,
            "roofline": {"bound": "hbm", "kernel": "ffh::k_compare",
                         # what keeps the launch from its HBM roofline: vector issue (valu_issue_frac) and, with four waves per SIMD, the
                         # waves' own LDS round trips (profiles/r04/ab_log.txt)
                         "limiter": "valu-issue",
                         # SURVEY.md section 8d's HBM figure: algorithmic bytes of the launch against the 8 TB/s data-sheet peak
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_detail": traffic_note, "algorithmic_bytes_per_launch": b_alg, "launch_ms": cmp_ms,
                         "hbm_GBps_from_traffic": (traffic / (cmp_ms * 1e-3) / 1e9) if traffic else None,
                         "valu_pairs_per_launch": pairs, "pairs_per_s": pairs / (cmp_ms * 1e-3),
                         "valu_issue_frac": valu_issue, "valu_issue_frac_if_all_full_rate": valu_issue_lo, "useful_valu_frac": useful_valu,
                         "useful_valu_instructions_per_launch": useful_instr,
                         "valu_model": {"cycles_full_rate": VALU_CYCLES_FULL_RATE, "cycles_half_rate": VALU_CYCLES_HALF_RATE,
                                        "instructions_per_2048_pairs_prefix": ops_p, "instructions_per_2048_pairs_suffix": ops_s},
                         "sq_counters_per_launch": sq or None,
                         "device_copy_GBps": stream_gbps, "frac_of_device_copy": achieved / stream_gbps if stream_gbps else None},
            # the second kernel of the step: one random 8-byte read of the target array per raw hit = one 128-byte line each (no cache
            # policy changes that: profiles/r03/gather_policy.txt, a bare gather of as many lines takes 0.22-0.26 ms on this part)
            "roofline_epilogue": epilogue_roofline(raw_hits, float(np.mean([t["finalize_ms"] for t in tms]))),
            "cpu_baseline": _CPU_JVM if (_CPU_JVM and _CPU_JVM.get("value")) else cpu,
            "cpu_baseline_port": cpu if (_CPU_JVM and _CPU_JVM.get("value")) else None,
            "cpu_baseline_all_cores": _CPU_MT,
            "breakdown_ms": {k: float(np.mean([t[k] for t in tms])) for k in ("prepare_ms", "compare_ms", "sort_ms", "finalize_ms", "total_scan_ms")},
            "discover_wall_s": dt / args.steps,
            "db_prepare_ms": info.prepare_ms,
            "plan": {"prefix_bases": tms[-1]["prefix_bases"], "prefix_radius": tms[-1]["prefix_radius"], "suffix_bases": info.suffix_bases,
                     "suffix_radius": tms[-1]["suffix_radius"], "items": tms[-1]["items_prefix"] + tms[-1]["items_suffix"],
                     "tiles": tms[-1]["tiles_prefix"] + tms[-1]["tiles_suffix"]},
            "hits": {"raw": raw_hits, "raw_per_guide": raw_hits / max(G, 1), "kept_positions": kept_pos, "overflowed_guides": int(final["overflow"].sum())},
            "algorithmic_bytes_survey": b_survey,
            # N > 1: every rank's own figures (rank 0's compare launch is the one `roofline` prices); no scaling efficiency is computed here
            "per_rank": per_rank if world > 1 else None,
            "skewed": skewed,
            # BASELINE.json configs[1] / SURVEY.md section 8d: the chr22-scale step and the CLI's wall time from argv to the closed table
            "c2": c2,
            "real_genome": real_genome,
        }
    else:
        out = None
    if comm is not None:
        comm.close()
    ctx.close()
    import ctypes
    ctypes.CDLL(None).fflush(None)  # RCCL prints a version banner into the C stdio buffer of stdout: every rank gets it out first
    sys.stdout.flush()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:  # the one JSON line is the last thing the job writes
        print(json.dumps(out), flush=True)
    if sharded:
        os._exit(0)  # RCCL / torch.distributed teardown may not write after the result line (a profiler needs the normal exit at N = 1)


if __name__ == "__main__":
    main()
