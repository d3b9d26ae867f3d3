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
B_PER_THREAD_LAUNCH = 272.0   # packet state loaded + stored once per packet per launch (2 x 136 B)
B_PER_MA_JUMP = 140.0         # LevelPack 16 + 9 process rates 72 + ~6 cumulative sums 48 + target level 4
B_PER_KPKT_STEP = 200.0       # ~6 ion sums 48 + ~7 cooling-list sums 56 + ~6 collisional-excitation sums 48 + indices/flags 48
B_PER_RPKT_STEP = 120.0       # cell scalars ~40 + boundary tables ~56 + J, nuJ, ffheating atomics 24
B_PER_LINE = 16.0             # line frequency 8 + the cell's population factor of the line 8
B_PER_CONT = 52.0             # ContPack 32 + {nnlevel, edge part} 16 + cross-section entry 4
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s


class _CudaArrayView:
    """Zero-copy torch view of the engine's estimator block (a raw device pointer from the C-ABI)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def _cpu_worker(args):
    (model, cs, ts, pk) = _cpu_worker.shared
    lo, hi = args
    from oracle import oracle_py

    sub = pk[lo:hi].copy()
    est = abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"])
    t0 = time.perf_counter()
    oracle_py.update_packets(model, cs, ts, sub, est)
    wall = time.perf_counter() - t0
    oracle_py.lib().artis_oracle_last_populate_seconds.restype = __import__("ctypes").c_double
    tpop = oracle_py.lib().artis_oracle_last_populate_seconds()
    return int(est.stats[abi.STAT_X_RPKT_STEPS] + est.stats[abi.STAT_X_KPKT_STEPS]), wall, tpop


def cpu_baseline(model, cs, ts, pk, sample: int, cores: int):
    """The CPU oracle (a scalar port of the reference's path) on the first `sample` packets of the same population,
    one process per core. Cell-cache filling is lazy as in the reference's CPU build and its time is excluded
    (at full scale it is amortised over ~150 packets per cell; in a small sample it would dominate)."""
    os.environ.setdefault("ARTIS_ORACLE_CACHE_CAP", "3000")
    sample = min(sample, len(pk))
    bounds = [(sample * i // cores, sample * (i + 1) // cores) for i in range(cores)]
    _cpu_worker.shared = (model, cs, ts, pk)
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, bounds)
    wall = time.perf_counter() - t0
    steps = sum(r[0] for r in res)
    busy = max(r[1] - r[2] for r in res)
    return {"value": steps / busy, "unit": "packet-steps/s", "cores": cores, "kind": "port",
            "sample": f"first {sample} packets of the same population on {cores} processes of the C oracle "
                      f"(oracle/artis_oracle.c); {steps} packet-steps in {busy:.1f} s of propagation "
                      f"(+{max(r[2] for r in res):.1f} s lazy cell-cache fill excluded; leg wall {wall:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--packets", type=int, default=10_000_000, help="packets per GPU")
    ap.add_argument("--ncoord", type=int, default=50)
    ap.add_argument("--preset", default="w7")
    ap.add_argument("--cpu-sample", type=int, default=160_000)
    ap.add_argument("--cpu-cores", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N>1"

    t_setup = time.perf_counter()
    model, cs, ts, aux = synth.build(args.preset, ncoord=args.ncoord)
    # packet seeds: the reference's per-rank spacing (input.cc:1912: rank_seed_base = seed + rank * npackets)
    seed_base = (1281360349 + rank * args.packets) & 0xFFFFFFFF
    pk = synth.make_packets(model, aux, args.packets, seed_base=seed_base, kpkt_fraction=0.02, seed=99 + rank)

    baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = args.cpu_cores or min(os.cpu_count() or 1, 16)
        baseline = cpu_baseline(model, cs, ts, pk, args.cpu_sample, cores)  # before any GPU initialisation (fork)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from artis_amd import engine

    eng = engine.Engine(model, device=local_rank)
    eng.set_cellstate(cs, ts)
    eng.upload_packets(pk)
    eng.snapshot()
    ptr, ndoubles = eng.estimators_devptr()
    est_view = torch.as_tensor(_CudaArrayView(ptr, ndoubles), device=torch.device("cuda", local_rank))
    stream = torch.cuda.current_stream().cuda_stream
    setup_s = time.perf_counter() - t_setup

    kern_ms, kern_launches = 0.0, 0

    def one_step(timed: bool):
        nonlocal kern_ms, kern_launches
        eng.restore()
        eng.zero_estimators(stream)
        eng.populate_cellcache(stream)
        eng.step(stream)
        if world > 1:
            dist.all_reduce(est_view)  # RCCL all-reduce of [J | nuJ | ffheat | colheat | gamma | bfheat]
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
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    est = abi.Estimators(model["npts_nonempty"], model["nbfcontinua_ground"])
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
        ms_per_step = 1e3 * elapsed / args.steps
        value = steps_all / (elapsed / args.steps)
        # roofline of the dominant kernel (k_thermal: macro-atom walk + k-packet steps): algorithmic bytes of its work
        # in one timestep / its summed launch durations (HIP events on the launch stream, artis_engine.hip)
        bd = eng.last_kernel_breakdown()
        alg_thermal = (B_PER_MA_JUMP * stats[abi.STAT_X_MA_JUMPS] + B_PER_KPKT_STEP * stats[abi.STAT_X_KPKT_STEPS] +
                       B_PER_THREAD_LAUNCH * bd["thermal_threads"])
        alg_rpkt = (B_PER_LINE * stats[abi.STAT_X_LINES_VISITED] + B_PER_RPKT_STEP * stats[abi.STAT_X_RPKT_STEPS] +
                    B_PER_CONT * stats[abi.STAT_NAMES.index("X_CONT_VISITED")] + B_PER_THREAD_LAUNCH * bd["rpkt_threads"])
        dominant = "k_thermal" if bd["thermal_ms"] >= bd["rpkt_ms"] else "k_rpkt"
        alg_bytes = alg_thermal if dominant == "k_thermal" else alg_rpkt
        k_ms_per_step = bd["thermal_ms"] if dominant == "k_thermal" else bd["rpkt_ms"]
        launches_per_step = bd["thermal_launches"] if dominant == "k_thermal" else bd["rpkt_launches"]
        achieved = alg_bytes / (k_ms_per_step * 1e-3) / 1e9 if k_ms_per_step > 0 else 0.0
        both = (alg_thermal + alg_rpkt) / ((bd["thermal_ms"] + bd["rpkt_ms"]) * 1e-3) / 1e9
        # HBM traffic of the dominant kernel per launch: from the committed rocprofv3 --pmc passes of this same command
        # (profiles/<round>/pmc_traffic.json, written by tools/pmc_summary.py; FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for gfx950). None when the workload is not the profiled one.
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
        if os.path.exists(tfile) and args.packets == 10_000_000 and args.ncoord == 50 and args.preset == "w7" and world == 1:
            with open(tfile) as f:
                tj = json.load(f)
            if dominant in tj:
                traffic = tj[dominant]["hbm_bytes_per_launch"]
                traffic_src = "profiles/r01/pmc_traffic.json"
        out = {
            "metric": "packet-steps/sec", "value": value, "unit": "packet-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.ncoord}^3 Cartesian W7-like ejecta, artisoptions_classic physics "
                                   f"(line-by-line Sobolev + macro-atom + k-packets), {args.packets} packets per GPU, "
                                   f"synthetic atomic data '{args.preset}' ({model['nlines']} lines, {model['nlevels']} levels, "
                                   f"{model['nions']} ions), one timestep at t=20 d (dt/t=0.05)",
                       "packets_per_gpu": args.packets, "nonempty_cells": int(model["npts_nonempty"]),
                       "packet_steps_per_step": steps_all, "setup_s": round(setup_s, 1),
                       "parallelism": f"packets sharded over {world} GPU(s); estimator all-reduce over RCCL" if world > 1
                       else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": dominant, "launches_per_step": launches_per_step,
                         "kernel_ms_per_step": k_ms_per_step, "avg_launch_ms": k_ms_per_step / max(launches_per_step, 1),
                         "algorithmic_bytes_per_step": float(alg_bytes),
                         "algorithmic_bytes_per_launch": float(alg_bytes) / max(launches_per_step, 1),
                         "achieved_both_propagation_kernels": both},
        }
        out["kernel_breakdown_last_step"] = bd
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
