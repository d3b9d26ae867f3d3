"""Multi-GPU plumbing: how packets are sharded over ranks and how the estimators are reduced.

The path shards naturally: every rank owns its own packets (independent histories, per-packet RNG) and accumulates
its own estimator block; the only exchange is one all-reduce(SUM) of that block at the end of the timestep -- the
reference's radfield::reduce_estimators() (radfield.cc:988) and the MPI_Allreduce of the heating/gamma estimators.
Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
from __future__ import annotations

import numpy as np

SEED_DEFAULT = 1281360349  # tests/classicmode_3d_inputfiles/input-newrun.txt line 0 of the reference


def rank_seed_base(seed: int, rank: int, npackets_per_rank: int) -> int:
    """Per-rank packet seed base of the reference's per-packet generator: ranks are spaced by the number of packets they
    own so that seed ranges do not overlap (input.cc:1905-1916)."""
    return (seed + rank * npackets_per_rank) & 0xFFFFFFFF


def packet_shard(npackets_total: int, world: int, rank: int):
    """Contiguous, nearly equal split of a global packet index range (get_range_chunk, mpi_logging.h:158)."""
    base, rem = divmod(npackets_total, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def allreduce_estimators(block, dist=None):
    """In-place SUM all-reduce of an estimator block (a torch tensor: the zero-copy view of the engine's device block on
    the GPU, a CPU tensor under gloo)."""
    if dist is None:
        import torch.distributed as dist  # noqa: PLW0621
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(block)
    return block


def flatten_estimators(est) -> np.ndarray:
    """Host estimators in the engine's block order (artis_engine.hip engine_fill; physics.h Env::est_stride):
    [cell][8]{J, nuJ, ffheating, colheating, dep_gamma, dep_electron, dep_positron, dep_alpha} -- the per-cell sums of a cell in
    one 64-byte record -- | [cell][ground continuum][2]{gamma, bfheating} | scalars,
    followed in nltenebular builds by [cell][bin][2]{radfieldbin_J, radfieldbin_nuJ} | bfrate_raw and, with detailed line
    estimators, by [Jb_lu_raw | Jb_lu_contribcount] (the counts as f64, like the engine's device block: one all-reduce covers it);
    in VPKT_ON builds by [vspecpol | vgrid_flux] (vpkt.cc sums them over the ranks when it writes the spectra)."""
    percell = np.stack([est.J, est.nuJ, est.ffheatingestimator, est.colheatingestimator, est.dep_estimator_gamma,
                        est.dep_estimator_electron, est.dep_estimator_positron, est.dep_estimator_alpha], axis=1).ravel()
    pairs = np.stack([est.gammaestimator, est.bfheatingestimator], axis=1).ravel()
    parts = [percell, pairs, est.scalars]
    if getattr(est, "extended", False):
        parts += [np.stack([est.radfieldbin_J, est.radfieldbin_nuJ], axis=1).ravel(), est.bfrate_raw]
    if getattr(est, "lineest", False):
        parts += [est.Jb_lu_raw, est.Jb_lu_contribcount.astype(np.float64)]
    if getattr(est, "vpkt", False):  # the observers' spectra (and the velocity-grid map when it is on) of a VPKT_ON build
        parts += [est.vspecpol] + ([est.vgrid_flux] if est.c.vgrid_flux else [])
    return np.concatenate(parts)
