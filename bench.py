#!/usr/bin/env python3
"""bench.py -- packet-steps/sec of the MI355X packet-propagation engine on the synthetic W7-like workload.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole packet population: the reference's update_packets()
(update_packets.cc:530) for one timestep -- cell-cache population for every non-empty cell, then propagation of
every packet to the end of the timestep -- followed, for N>1, by the estimator all-reduce of
radfield::reduce_estimators() (radfield.cc:988) over RCCL. Packets, tables and cell state are resident in HBM
before the timed region starts; every step restarts from the same device-side snapshot of the population.

Workload (BASELINE.json configs[1]): 50^3 Cartesian grid, W7-like exponential ejecta, artisoptions_classic.h
physics (line-by-line Sobolev + macro-atom + k-packets), 1e7 packets per GPU, synthetic atomic data (no network).
A packet-step is one call of do_rpkt_step() (rpkt.cc:542) or of do_kpkt()/do_kpkt_blackbody() (kpkt.cc:425/399);
the count comes from the engine's own event counters and is identical to the CPU oracle's on the same input.

`python bench.py --gpus N` (N > 1) without a torchrun environment starts the N ranks itself: it spawns
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a child process BEFORE anything
touches the GPU and exits with the child's code; under torchrun (RANK/WORLD_SIZE set) it runs as one rank.
`--options <preset>` runs the same workload on the engine built for another options file of the reference (one library
per artisoptions_*.h: kilonova_lte = BASELINE.json configs[3], nltenebular = configs[4], ...); the headline line is the
classic build.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the definitions of roofline and cpu_baseline).
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from artis_amd import abi, synth  # noqa: E402

# Algorithmic bytes per unit of work ("Algorithmic bytes per unit"): what the kernels read and write
# by design for one unit, wherever the cache hierarchy then serves it from.
B_PER_THERMAL_VISIT = 256.0   # hot line (128 B) loaded + stored once per packet per k_thermal launch
B_PER_RPKT_VISIT = 448.0      # hot + flight line (96 of 128 B used) loaded + stored once per packet per k_rpkt launch
B_PER_EMISSION = 120.0        # flight line direction/rest-frame part 56 + em_pos/em_time 28 + trueem 36
B_PER_MA_JUMP = 50.0          # action filter 16 + the first filter lines of both directions 32 (one 64-byte sector, read together) + ~2 for
                              # the searches that go on to a second line; the target comes from LDS (2 + 16 B, not counted) where the
                              # static tables fit, else from HBM (2 + 16). Round 3: 35 (16 + 16 x 0.675 + an 8-byte target in the
                              # line of the sums); round 2: 116
B_PER_KPKT_STEP = 200.0       # ~6 ion sums 48 + ~7 cooling-list sums 56 + ~6 collisional-excitation sums 48 + indices/flags 48
B_PER_RPKT_STEP = 120.0       # cell scalars ~40 + boundary tables ~56 + J, nuJ, ffheating atomics 24
B_PER_LINE = 16.0             # line frequency 8 + the cell's population factor of the line 8
B_PER_CONT = 52.0             # ContPack 32 + {nnlevel, edge part} 16 + cross-section entry 4
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s
# bytes a kernel HAS to write by design (for write_amplification = measured WRITE_SIZE / this)
W_PER_THERMAL_VISIT = 128.0   # hot line
W_PER_RPKT_VISIT = 224.0      # hot line + 96 B of the flight line
W_PER_EMISSION = 120.0
W_PER_ATOMIC = 8.0            # one f64 estimator add
W_PER_LIST_ENTRY = 8.0        # (slot, key) appended to a work list
PROFILE_ROUNDS = ("r06", "r05", "r04")  # the newest round that holds the counters of this command
NCU, NSIMD, CLOCK_GHZ = 256, 1024, 2.4   # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock
GATHER_INSTR_CLOCKS = 40.0             # CU clocks per 64-lane 16-byte gather instruction out of L2 (profiles/r02/gather_microbench.txt)
LINE_FILL_CLOCKS = 150.0 / 64          # CU clocks per 128-byte line filled from L2 (profiles/r03/sector_bench.txt)


class _CudaArrayView:
    """Zero-copy torch view of the engine's estimator block (a raw device pointer from the C-ABI)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def _cpu_worker(args):
    (model, cs, ts, pk) = _cpu_worker.shared
    lo, hi = args
    from oracle import oracle_py

    sub = pk[lo:hi].copy()
    est = abi.estimators_for(model, _cpu_worker.options)
    t0 = time.perf_counter()
    try:
        oracle_py.update_packets(model, cs, ts, sub, est, preset=_cpu_worker.options, fast=_cpu_worker.fast)
    except RuntimeError:
        return None  # (the fast-math build tripped one of the restatement's assertions: reported, not hidden)
    wall = time.perf_counter() - t0
    L = oracle_py.lib(_cpu_worker.options, _cpu_worker.fast)
    L.artis_oracle_last_populate_seconds.restype = __import__("ctypes").c_double
    tpop = L.artis_oracle_last_populate_seconds()
    return int(est.stats[abi.STAT_X_RPKT_STEPS] + est.stats[abi.STAT_X_KPKT_STEPS]), wall, tpop


def _cpu_leg(bounds, cores, fast):
    _cpu_worker.fast = fast
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, bounds)
    wall = time.perf_counter() - t0
    if any(r is None for r in res):
        return None
    steps = sum(r[0] for r in res)
    busy = max(r[1] - r[2] for r in res)
    return steps, busy, max(r[2] for r in res), wall


def cpu_baseline(model, cs, ts, pk, sample: int, cores: int, options: str = "classic", both: bool = False):
    """The CPU oracle (a scalar port of the reference's path) on the first `sample` packets of the same population,
    one process per core. Cell-cache filling is lazy as in the reference's CPU build and its time is excluded
    (at full scale it is amortised over ~150 packets per cell; in a small sample it would dominate).
    `value` always means the build with the reference Makefile's own default flags (-O3 -march=native -flto, fast-math); `--cpu-both` also
    times the build the parity checker uses (-O2 -ffp-contract=off: `value_checker_flags`; round 5 timed both in every run, which doubled the
    bench's set-up). Only when the reference-flags build is not available on the host (no compiler, or it trips an assertion of the
    restatement) is the checker's build timed in its place -- and then `kind` says so ("port-checker-flags"): `value` never silently changes
    its meaning between runs."""
    os.environ.setdefault("ARTIS_ORACLE_CACHE_CAP", "3000")
    sample = min(sample, len(pk))
    bounds = [(sample * i // cores, sample * (i + 1) // cores) for i in range(cores)]
    _cpu_worker.shared = (model, cs, ts, pk)
    _cpu_worker.options = options
    from oracle import oracle_py

    oracle_py.lib(options)  # loaded (and, if it has to be, built) ONCE here: the forked workers inherit it instead of racing to build it
    fast = None
    try:
        oracle_py.lib(options, fast=True)
        fast = _cpu_leg(bounds, cores, True)
    except Exception as exc:  # noqa: BLE001  (no compiler on the box, or the fast-math build failed an assertion)
        print(f"[bench] cpu_baseline: the -O3 -march=native build of the oracle is not available ({exc})", file=sys.stderr)
    checker = _cpu_leg(bounds, cores, False) if (both or fast is None) else None
    if fast is None:
        steps, busy, tpop, wall = checker
        return {"value": steps / busy, "unit": "packet-steps/s", "cores": cores, "kind": "port-checker-flags",
                "flags": "-O2 -ffp-contract=off (the parity checker's build, oracle/Makefile): the build with the reference Makefile's flags was not available on this host",
                "value_checker_flags": steps / busy,
                "sample": f"first {sample} packets of the same population on {cores} processes of the C oracle "
                          f"(oracle/artis_oracle.c); {steps} packet-steps in {busy:.1f} s of propagation "
                          f"(+{tpop:.1f} s lazy cell-cache fill excluded; leg wall {wall:.1f} s)"}
    fsteps, fbusy, ftpop, fwall = fast
    out = {"value": fsteps / fbusy, "unit": "packet-steps/s", "cores": cores, "kind": "port",
           "flags": " ".join(oracle_py.FAST_CFLAGS[:6]) + " (the reference Makefile's defaults: Makefile:38, :236-251; built on this host)",
           "sample": f"first {sample} packets of the same population on {cores} processes of the C oracle (oracle/artis_oracle.c) "
                     f"compiled with the reference's default flags: {fsteps} packet-steps in {fbusy:.1f} s of propagation "
                     f"(+{ftpop:.1f} s lazy cell-cache fill excluded; leg wall {fwall:.1f} s)"}
    if checker is not None:
        steps, busy, tpop, wall = checker
        out["value_checker_flags"] = steps / busy
        out["sample"] += (f"; compiled as the parity checker is (-O2 -ffp-contract=off): {steps} packet-steps in {busy:.1f} s = value_checker_flags")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--packets", type=int, default=10_000_000, help="packets per GPU (weak scaling: per-GPU work fixed as N grows)")
    ap.add_argument("--packets-total", type=int, default=0,
                    help="total packets sharded over the N GPUs with the reference's range rule (BASELINE.json configs[2]: 1e8 over 8); "
                         "overrides --packets, and the line then says scaling: strong")
    ap.add_argument("--ncoord", type=int, default=50)
    ap.add_argument("--preset", default="w7")
    ap.add_argument("--grid", default="3d", choices=("1d", "2d", "3d"),
                    help="grid geometry (GridType of the reference): 3d = the headline workload; 1d / 2d = --ncoord shells / (r, z) cells")
    ap.add_argument("--cpu-sample", type=int, default=160_000)
    ap.add_argument("--cpu-cores", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-both", action="store_true", help="time the CPU port also as the parity checker is compiled (value_checker_flags)")
    ap.add_argument("--options", default="classic", choices=("classic", "kilonova_lte", "nltenebular", "christinenonthermal", "nltephotospheric", "nltewithoutnonthermal",
                                                             "kilonova_expopac", "classic_expopac_therm", "ci_classic_vpkt", "ci_classic_vpkt_expopac"),
                    help="options preset of include/artis_options.h (the reference's artisoptions_*.h; the expansion-opacity builds of "
                         "BASELINE.json configs[3]'s text; the reference's classic CI set-up with virtual packets)")
    ap.add_argument("--t-days", type=float, default=20.0,
                    help="time of the timestep (a VPKT_ON build traces virtual packets inside its spectra window only: synth.vpkt_config 3-8 d)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks ourselves: fresh child processes, nothing in this process has touched the GPU
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"

    t_setup = time.perf_counter()
    gridtype = {"1d": abi.GRID_SPHERICAL1D, "2d": abi.GRID_CYLINDRICAL2D, "3d": abi.GRID_CARTESIAN3D}[args.grid]
    model, cs, ts, aux = synth.build(args.preset, ncoord=args.ncoord, gridtype=gridtype, options=args.options, t_days=args.t_days)
    # packet seeds: the reference's per-rank spacing (input.cc:1912: rank_seed_base = seed + rank * npackets)
    if args.packets_total > 0:  # get_range_chunk (mpi_logging.h:158): nearly equal contiguous shares of one global population
        from artis_amd import dist as adist

        shard_start, args.packets = adist.packet_shard(args.packets_total, world, rank)
        seed_base = (1281360349 + shard_start) & 0xFFFFFFFF
    else:
        seed_base = (1281360349 + rank * args.packets) & 0xFFFFFFFF
    pk = synth.make_packets(model, aux, args.packets, seed_base=seed_base, kpkt_fraction=0.02, seed=99 + rank)

    baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_sample > 0:
        cores = args.cpu_cores or min(os.cpu_count() or 1, 16)
        baseline = cpu_baseline(model, cs, ts, pk, args.cpu_sample, cores, args.options, both=args.cpu_both)  # before any GPU initialisation (fork)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from artis_amd import engine

    eng = engine.Engine(model, device=local_rank, preset=args.options)
    eng.set_cellstate(cs, ts)
    eng.upload_packets(pk)
    eng.snapshot()
    ptr, ndoubles = eng.estimators_devptr()
    est_view = torch.as_tensor(_CudaArrayView(ptr, ndoubles), device=torch.device("cuda", local_rank))
    stream = torch.cuda.current_stream().cuda_stream
    # N > 1: the estimator all-reduce is the library's (artis_amd_allreduce_estimators: one in-place ncclAllReduce over
    # RCCL in the C++ host layer). torch.distributed only carries the 128 bytes of the communicator id, as MPI would in
    # the reference. Should the library's communicator fail to come up, the same reduction is done by torch's RCCL
    # binding on the same device block, and the JSON line says so.
    reduce_via = None
    if world > 1:
        # every rank first proves that its engine library reaches librccl (a local call), so that no rank can be left
        # waiting inside ncclCommInitRank for one that never got there
        my_id = None
        try:
            my_id = eng.comm_unique_id()
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] rank {rank}: librccl not reachable from the engine library ({exc})", file=sys.stderr)
        reachable = [None] * world
        dist.all_gather_object(reachable, my_id is not None)
        ids = [my_id if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        reduce_via = "torch.distributed.all_reduce (RCCL)"
        if all(reachable) and ids[0] is not None:
            try:
                eng.comm_init(world, rank, ids[0])
                reduce_via = "artis_amd_allreduce_estimators (RCCL, C-ABI)"
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] rank {rank}: C-ABI communicator not available ({exc})", file=sys.stderr)
        allv = [None] * world
        dist.all_gather_object(allv, reduce_via)
        if any(v != "artis_amd_allreduce_estimators (RCCL, C-ABI)" for v in allv):
            reduce_via = "torch.distributed.all_reduce (RCCL)"  # every rank must take the same path
    setup_s = time.perf_counter() - t_setup

    kern_ms, kern_launches = 0.0, 0
    reduce_events = []  # (start, end) torch events around the estimator all-reduce of every timed step

    def one_step(timed: bool):
        nonlocal kern_ms, kern_launches
        eng.restore()
        eng.zero_estimators(stream)
        eng.populate_cellcache(stream)
        eng.step(stream)
        if world > 1:  # [J | nuJ | ffheat | colheat | gamma | bfheat | dep_* | scalars], one in-place sum over the ranks
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()  # on torch's current stream = the stream the library's all-reduce is issued on
            if reduce_via.startswith("artis_amd"):
                eng.allreduce_estimators(stream)
            else:
                dist.all_reduce(est_view)
            ev1.record()
            if timed:
                reduce_events.append((ev0, ev1))
        if timed:
            ms, nl = eng.last_kernel_ms()
            kern_ms += ms
            kern_launches += nl

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed_rank = elapsed
    rank_ms = [1e3 * elapsed_rank / args.steps]
    reduce_ms = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        gathered = [None] * world
        # per rank: wall ms per step, propagation-kernel ms per step, all-reduce ms per step (waiting for the slowest rank included)
        mine = (1e3 * elapsed_rank / args.steps, kern_ms / max(args.steps, 1),
                sum(a.elapsed_time(b) for a, b in reduce_events) / max(len(reduce_events), 1))
        dist.all_gather_object(gathered, mine)
        rank_ms = [g[0] for g in gathered]
        reduce_ms = {"per_rank_ms": [round(g[2], 3) for g in gathered], "min_ms": min(g[2] for g in gathered),
                     "bytes": int(ndoubles) * 8,
                     "note": "events around the call on each rank; the minimum over ranks is the collective itself, the rest is waiting for the slowest rank"}
        rank_kernel_ms = [g[1] for g in gathered]

    est = abi.estimators_for(model, args.options)
    # counters of the last step (identical every step: packet histories are deterministic)
    # the all-reduced estimator block is not needed here; only the per-rank event counters
    import ctypes as C

    stats = np.zeros(abi.NSTATS, dtype=np.int64)
    est.c.stats = stats.ctypes.data_as(C.POINTER(C.c_int64))
    eng.download_estimators(est)
    steps_rank = int(stats[abi.STAT_X_RPKT_STEPS] + stats[abi.STAT_X_KPKT_STEPS])
    steps_all = steps_rank
    if world > 1:
        tsum = torch.tensor([steps_rank], dtype=torch.int64, device="cuda")
        dist.all_reduce(tsum)
        steps_all = int(tsum.item())

    if rank == 0:
        geometry = {"3d": f"{args.ncoord}^3 Cartesian", "2d": f"{args.ncoord} x {2 * args.ncoord} cylindrical (r, z)",
                    "1d": f"{args.ncoord}-shell spherical 1D"}[args.grid]
        ms_per_step = 1e3 * elapsed / args.steps
        value = steps_all / (elapsed / args.steps)
        # roofline of the dominant kernel (k_thermal: macro-atom walk + k-packet steps): algorithmic bytes of its work
        # in one timestep / its summed launch durations (HIP events on the launch stream, artis_engine.hip)
        bd = eng.last_kernel_breakdown()
        S = lambda name: float(stats[abi.STAT_NAMES.index(name)])  # noqa: E731
        emissions = (S("K_STAT_TO_R_FF") + S("K_STAT_TO_R_FB") + S("K_STAT_TO_R_BB") + S("MA_STAT_DEACTIVATION_BB") +
                     S("MA_STAT_DEACTIVATION_FB"))
        alg = {
            "k_thermal": (B_PER_MA_JUMP * S("X_MA_JUMPS") + B_PER_KPKT_STEP * S("X_KPKT_STEPS") +
                          B_PER_THERMAL_VISIT * bd["thermal_threads"] + B_PER_EMISSION * emissions),
            "k_rpkt": (B_PER_LINE * S("X_LINES_VISITED") + B_PER_RPKT_STEP * S("X_RPKT_STEPS") +
                       B_PER_CONT * S("X_CONT_VISITED") + B_PER_RPKT_VISIT * bd["rpkt_threads"]),
        }
        # bytes the kernels have to write by design: packet lines at retire, list entries, estimator atomics
        wr = {
            "k_thermal": (W_PER_THERMAL_VISIT + W_PER_LIST_ENTRY) * bd["thermal_threads"] + W_PER_EMISSION * emissions +
                         W_PER_ATOMIC * (S("MA_STAT_DEACTIVATION_COLLDEEXC") + S("MA_STAT_DEACTIVATION_COLLRECOMB")),
            "k_rpkt": (W_PER_RPKT_VISIT + W_PER_LIST_ENTRY) * bd["rpkt_threads"] + 3 * W_PER_ATOMIC * S("X_RPKT_STEPS") +
                      32.0 * S("X_CHI_EVALS"),
        }
        kms = {"k_thermal": bd["thermal_ms"], "k_rpkt": bd["rpkt_ms"]}
        kl = {"k_thermal": bd["thermal_launches"], "k_rpkt": bd["rpkt_launches"]}
        dominant = "k_thermal" if kms["k_thermal"] >= kms["k_rpkt"] else "k_rpkt"
        # HBM traffic per launch: from the committed rocprofv3 --pmc passes of this same command
        # (profiles/<round>/pmc_traffic.json, written by tools/pmc_summary.py; FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for gfx950). None when the workload is not the profiled one.
        tj, traffic_src = {}, None
        # one file per (options build, atomic data set): pmc_traffic[_<options>][_<preset>].json (the headline: pmc_traffic.json)
        tname = "pmc_traffic" + ("" if args.options == "classic" else f"_{args.options}") + ("" if args.preset == "w7" else f"_{args.preset}") + ".json"
        PROFILE_ROUND = next((r for r in PROFILE_ROUNDS if os.path.exists(os.path.join(ROOT, "profiles", r, tname))), PROFILE_ROUNDS[-1])
        tfile = os.path.join(ROOT, "profiles", PROFILE_ROUND, tname)
        if (os.path.exists(tfile) and args.packets == 10_000_000 and args.ncoord == 50 and world == 1 and args.grid == "3d"):
            with open(tfile) as f:
                tj = json.load(f)
            traffic_src = f"profiles/{PROFILE_ROUND}/{tname}"
            if "k_bfest_dense" in tj and "k_rpkt" in tj:
                # DETAILED_BF builds: the engine times k_rpkt together with the k_bfest_dense launch that follows it
                # (kernel_breakdown.rpkt_ms), so their counters are added; "per launch" = per k_rpkt launch
                a, b = tj["k_rpkt"], tj.pop("k_bfest_dense")
                nl_pmc = a["dispatches"]
                m = {"fetch_size_kb": a["fetch_size_kb"] + b["fetch_size_kb"], "write_size_kb": a["write_size_kb"] + b["write_size_kb"],
                     "dispatches": nl_pmc, "seconds_in_fetch_pass": a["seconds_in_fetch_pass"] + b["seconds_in_fetch_pass"],
                     "counters": {k: a["counters"].get(k, 0.) + b["counters"].get(k, 0.) for k in set(a["counters"]) | set(b["counters"])}}
                m["hbm_bytes_per_launch"] = (2.0 * m["fetch_size_kb"] + m["write_size_kb"]) * 1024.0 / nl_pmc
                m["hbm_bytes_per_launch_fetch_undoubled"] = (m["fetch_size_kb"] + m["write_size_kb"]) * 1024.0 / nl_pmc
                tj["k_rpkt"] = m

        def kernel_roofline(k):
            ms = kms[k]
            nl = max(kl[k], 1)
            r = {"kernel_ms_per_step": ms, "launches_per_step": kl[k], "avg_launch_ms": ms / nl,
                 "algorithmic_bytes_per_launch": alg[k] / nl,
                 "algorithmic_gbs": alg[k] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                 "hbm_bytes_per_launch_measured": None, "hbm_gbs_measured": None, "hbm_frac_measured": None,
                 "hbm_gbs_measured_fetch_undoubled": None, "write_amplification": None, "limiter": None}
            if k in tj and ms > 0:
                t = tj[k]
                r["hbm_bytes_per_launch_measured"] = t["hbm_bytes_per_launch"]
                # the PMC passes have their own launch count (one step): scale by bytes per launch
                gbs = t["hbm_bytes_per_launch"] * nl / (ms * 1e-3) / 1e9
                r["hbm_gbs_measured"] = gbs
                r["hbm_frac_measured"] = gbs / HBM_PEAK_GBS
                if "hbm_bytes_per_launch_fetch_undoubled" in t:  # FETCH_SIZE as counted (the guide's x2 is for wide coalesced reads)
                    r["hbm_gbs_measured_fetch_undoubled"] = t["hbm_bytes_per_launch_fetch_undoubled"] * nl / (ms * 1e-3) / 1e9
                # (the designed writes are counted for the classic build only: no counter holds the nebular builds' estimator additions)
                r["write_amplification"] = (t["write_size_kb"] * 1024.0 / t["dispatches"] * nl / wr[k]
                                            if wr[k] > 0 and args.options == "classic" else None)
                c = t.get("counters") or {}
                if c and t.get("seconds_in_fetch_pass"):
                    # what the kernel is held by, from the same committed counters (sums over one step's dispatches);
                    # SQ_* busy counters are in units of 4 clocks
                    clk = t["seconds_in_fetch_pass"] * CLOCK_GHZ * 1e9  # kernel clocks of the profiled step
                    lim = {}
                    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
                        lim["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
                        lim["valu_busy_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (NSIMD * clk)
                        # the ceiling this branchy f64 path is actually under: the fraction of the SIMDs' lane-cycles that do
                        # vector arithmetic (1.0 = every SIMD issues a full 64-lane VALU instruction every cycle)
                        lim["valu_lane_frac"] = lim["valu_busy_frac"] * lim["valu_lane_utilisation"]
                    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
                        lim["wave_wait_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
                    if "SQ_INSTS_VMEM_RD" in c:
                        ach = c["SQ_INSTS_VMEM_RD"] / (NCU * clk)
                        lim["issue"] = {"load_instr_per_clk_cu": ach, "ceiling": 1.0 / GATHER_INSTR_CLOCKS,
                                        "frac": ach * GATHER_INSTR_CLOCKS, "ceiling_source": "tools/gather_bench.hip, 64-lane 16-B gathers out of L2"}
                    if "TCP_TCC_READ_REQ_sum" in c:
                        ach = c["TCP_TCC_READ_REQ_sum"] / (NCU * clk)
                        lim["l1_fill"] = {"read_requests_per_clk_cu": ach, "ceiling": 1.0 / LINE_FILL_CLOCKS,
                                          "frac": ach * LINE_FILL_CLOCKS, "ceiling_source": "tools/sector_bench.hip, 128-B lines out of L2"}
                    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
                        lim["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
                    r["limiter"] = lim
            return r

        per_kernel = {k: kernel_roofline(k) for k in ("k_thermal", "k_rpkt")}
        d = per_kernel[dominant]
        lim = d["limiter"] or {}
        out = {
            "metric": "packet-steps/sec", "value": value, "unit": "packet-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if args.packets_total > 0 else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{geometry} W7-like ejecta, artisoptions_{args.options} physics "
                                   f"(line-by-line Sobolev + macro-atom + k-packets), "
                                   + (f"{args.packets_total} packets over {world} GPU(s)" if args.packets_total > 0 else f"{args.packets} packets per GPU")
                                   + f", synthetic atomic data '{args.preset}' ({model['nlines']} lines, {model['nlevels']} levels, "
                                   f"{model['nions']} ions), one timestep at t={args.t_days:g} d (dt/t=0.05)",
                       "options": args.options,
                       "packets_per_gpu": args.packets, "nonempty_cells": int(model["npts_nonempty"]),
                       "packet_steps_per_step": steps_all, "setup_s": round(setup_s, 1),
                       "parallelism": f"packets sharded over {world} GPU(s); estimator all-reduce: {reduce_via}" if world > 1
                       else "1 GPU"},
            # achieved / frac / traffic: bytes that REACHED HBM by the committed rocprofv3 counters (2 x FETCH_SIZE + WRITE_SIZE
            # per launch) over the kernel's launch time measured live with HIP events -- the number north_star's ">= 30 % of
            # HBM roofline" is about. null when the workload is not the profiled one. `bound` is what the counters say holds the
            # kernel (dependent L2-served gathers: waves wait most of their cycles with idle issue slots), not the priced roof. The bytes the kernel REQUESTS by design
            # (most of them served by L1/L2) are algorithmic_gbs: never to be read as an HBM fraction. The kernel is not
            # HBM-bound: `limiter` holds the measured ratios of what does hold it.
            "roofline": {"bound": "latency", "priced_against": "hbm", "achieved": d["hbm_gbs_measured"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": d["hbm_frac_measured"], "traffic": d["hbm_bytes_per_launch_measured"],
                         # VALU lane-throughput (busy fraction x lane utilisation, from the same committed counters): the yardstick
                         # that applies to this kernel, beside the HBM fraction north_star names; per kernel in `kernels`
                         "valu_lane_frac": (d["limiter"] or {}).get("valu_lane_frac"),
                         "traffic_source": traffic_src, "kernel": dominant,
                         "hbm_gbs_measured_fetch_undoubled": d["hbm_gbs_measured_fetch_undoubled"],
                         "write_amplification": d["write_amplification"],
                         "launches_per_step": d["launches_per_step"], "kernel_ms_per_step": d["kernel_ms_per_step"],
                         "avg_launch_ms": d["avg_launch_ms"],
                         "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                         "algorithmic_gbs": d["algorithmic_gbs"], "algorithmic_over_hbm_peak": d["algorithmic_gbs"] / HBM_PEAK_GBS,
                         "measured_limiter": (
                             "not HBM: instruction issue and the waves' dependent reads together -- "
                             + (f"VALU busy {lim['valu_busy_frac']:.2f} of the SIMDs' cycles at {lim['valu_lane_utilisation']:.2f} lane utilisation "
                                f"(valu_lane_frac {lim['valu_lane_frac']:.2f}), waves waiting {lim['wave_wait_frac']:.2f} of theirs, L2 hit rate "
                                f"{lim['l2_hit_rate']:.2f} (committed counters of this command); "
                                if all(k in lim for k in ("valu_busy_frac", "valu_lane_utilisation", "valu_lane_frac", "wave_wait_frac", "l2_hit_rate")) else "")
                             + "a transition is one sector of the cell's record from L2 and two LDS reads. Round 6 cut the transition loop from 484 to "
                             "164 instructions per wave-round (sorted-entry count, one test for all rare paths: profiles/r06/isa_census.md) and the "
                             "kernel's VALU instructions by 12 %: the loop was ~60 % of them, the rest is the ~2.7e7 wave-level walk ends per step (exit "
                             "process, k-packet step, packet load / store) at 24 of 64 lanes. Filling the loop's lanes (k_thermal_q) still costs in its "
                             "service passes what the fuller rounds save (790 vs 758 ms, profiles/r06/sweep.txt) (DESIGN.md section 7)")
                         if dominant == "k_thermal" else
                         ("k_rpkt (+ k_bfest_dense in DETAILED_BF builds, timed together): divergent per-lane loops over continua and "
                          "lines at 3 waves/SIMD (no spilled register since round 6), their reads requested an iteration ahead, the work list "
                          "sorted by (frequency bin, cell) so that a wave's loops are of similar length; see `limiter` and DESIGN.md section 7"),
                         "limiter": d["limiter"],
                         "kernels": per_kernel},
        }
        if world > 1:
            out["per_rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms), "all": [round(x, 2) for x in rank_ms]}
            out["per_rank_kernel_ms_per_step"] = [round(x, 2) for x in rank_kernel_ms]
            out["allreduce"] = reduce_ms
            try:
                out["rccl_nranks"] = eng.comm_count() if reduce_via.startswith("artis_amd") else dist.get_world_size()
            except Exception as exc:  # noqa: BLE001
                out["rccl_nranks"] = None
                print(f"[bench] ncclCommCount not available ({exc})", file=sys.stderr)
        nt, tcells, bpc = eng.cache_tiles()
        tiers = eng.record_tiers()
        lt = eng.last_tiling()
        out["config"]["cell_cache"] = {"tiles": nt, "cells_per_tile": tcells, "bytes_per_cell": bpc, **(lt if nt > 1 else {})}
        if tiers["ncold"] > 0:  # on-demand macro-atom records: the tiers the engine chose and what the step made of the pool
            out["config"]["cell_cache"].update(record_tiers=tiers, pool_resets=lt["pool_resets"], pool_units_used=lt["pool_units_used"],
                                               pool_units=lt["pool_units"])
        bd.update(ma_transitions=int(S("X_MA_JUMPS")), kpkt_steps=int(S("X_KPKT_STEPS")), rpkt_steps=int(S("X_RPKT_STEPS")))
        out["kernel_breakdown_last_step"] = bd
        out["kernel_ms_by_kind_last_step"] = eng.last_kernel_ms_by_kind()
        if os.environ.get("ARTIS_BENCH_VERBOSE"):
            print({abi.STAT_NAMES[i]: int(stats[i]) for i in range(abi.NSTATS) if stats[i]}, file=sys.stderr)
        if baseline is not None:
            out["cpu_baseline"] = baseline
            out["gpu_over_cpu_baseline"] = value / baseline["value"]
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
