"""world_size-2 gloo test of the N>1 path (CPU): packets are sharded over ranks with the reference's seed spacing,
each rank propagates its shard, and one all-reduce of the estimator block gives the same estimators and counters as a
single rank owning all packets. The propagation itself runs through the test-only x86 build of the kernel bodies
(there is no GPU in CI); on the GPU the same dist.py calls wrap the engine's device block (bench.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hostemu_binding as emu
from artis_amd import abi, synth
from artis_amd import dist as adist

NPK = 1200
SEED = 4242


def _build_kw(options):
    return {"t_days": 5.0} if "vpkt" in options else {}  # virtual packets: inside the spectra's window of 3-8 days


def _population(model, aux, start, count):
    pk = synth.make_packets(model, aux, NPK, seed_base=adist.rank_seed_base(SEED, 0, NPK), kpkt_fraction=0.2)
    return pk[start:start + count].copy()


def _worker(rank, world, port, outdir, options):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, cs, ts, aux = synth.build("tiny", ncoord=6, options=options, **_build_kw(options))
    start, count = adist.packet_shard(NPK, world, rank)
    pk = _population(model, aux, start, count)
    est = abi.estimators_for(model, options)
    emu.update_packets(model, cs, ts, pk, est, preset=options)
    block = torch.from_numpy(adist.flatten_estimators(est))
    adist.allreduce_estimators(block, dist)
    counters = torch.from_numpy(est.stats.copy())
    dist.all_reduce(counters)
    if rank == 0:
        np.save(os.path.join(outdir, "block.npy"), block.numpy())
        np.save(os.path.join(outdir, "counters.npy"), counters.numpy())
    np.save(os.path.join(outdir, f"pk{rank}.npy"), pk)
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("options", ["classic", "nltenebular", "nltenebular_lineest", "ci_classic_vpkt"])   # nltenebular: the block carries the bin and bound-free estimators too; _lineest: the detailed line estimators
def test_two_ranks_equal_one(tmp_path, options):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), options), nprocs=2, join=True)
    model, cs, ts, aux = synth.build("tiny", ncoord=6, options=options, **_build_kw(options))
    pk = _population(model, aux, 0, NPK)
    est = abi.estimators_for(model, options)
    emu.update_packets(model, cs, ts, pk, est, preset=options)
    block = np.load(tmp_path / "block.npy")
    want = adist.flatten_estimators(est)
    if "vpkt" in options:
        assert est.vspecpol.sum() > 0 and want.size > est.vspecpol.size
    assert np.allclose(block, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    counters = np.load(tmp_path / "counters.npy")
    skip = abi.STAT_NAMES.index("UPDATECELL")  # each rank fills the cell cache once
    mask = np.arange(abi.NSTATS) != skip
    assert np.array_equal(counters[mask], est.stats[mask])
    both = np.concatenate([np.load(tmp_path / "pk0.npy"), np.load(tmp_path / "pk1.npy")])
    for f in abi.PACKET_DTYPE.names:
        assert both[f].tobytes() == pk[f].tobytes(), f  # sharding does not change any packet history


def test_shard_and_seed_rules():
    assert [adist.packet_shard(10, 3, r) for r in range(3)] == [(0, 4), (4, 3), (7, 3)]  # mpi_logging.h:175-177
    assert adist.rank_seed_base(7, 3, 100) == 307
    assert adist.rank_seed_base(2**32 - 1, 1, 5) == 4  # uint32 wrap like the reference's static_cast
