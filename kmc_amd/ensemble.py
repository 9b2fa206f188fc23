"""Ensemble driver (row E): the size x concentration x averaging-iteration loops
of hex_LiCoO2_kmc.py:679-777 mapped onto batched replicas, sharded over the GPUs
of a node with ONE exchange step -- an all-reduce of the per-concentration
accumulators that give the reference's mean and np.std (:783-786).

Replicas share nothing but the read-only rate table, so there is no data-path
collective.  torch.distributed is used only as the collective transport
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
import numpy as np

ACC_FIELDS = ('count', 'sum_dstar', 'sum_dstar2', 'sum_dj', 'sum_dj2', 'sum_time', 'sum_hops',
              'h0', 'h1', 'h2', 'h3', 'h4', 'h5')


_M64 = (1 << 64) - 1


def derive_seed(seed, *tags):
    """A 64-bit Philox key for (seed, tags...): splitmix64 over the tags, so that e.g. the seeds of lattice sizes
    d and d + 1 are unrelated (seed + d would make (seed, d + 1) and (seed + 1, d) the same stream)."""
    z = int(seed) & _M64
    for t in tags:
        z = (z + 0x9E3779B97F4A7C15 + ((int(t) & _M64) * 0xD1342543DE82EF95)) & _M64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        z ^= z >> 31
    return z


def shard_replicas(conc_points, replicas_per_point, rank=0, world=1):
    """This rank's share of the ensemble: (conc[R_local], point_index[R_local], replica_offset).

    The ensemble is ordered iteration-major: global replica g = k * n_points + p is averaging iteration k of
    concentration p (the reference's loops hex_LiCoO2_kmc.py:686-693 with the inner one outermost).  Rank r owns
    the iterations [r * K // world, (r + 1) * K // world) of EVERY concentration (hop rates vary by orders of
    magnitude with x, so sharding by concentration would unbalance the GPUs): a contiguous range of global ids
    starting at replica_offset, whatever the number of ranks, and K need not divide by it.  The Philox streams
    are keyed by the global id (kmc_set_replica_offset), so the ensemble is the same at any GPU count.
    """
    conc_points = np.asarray(conc_points, dtype=np.float64)
    if not (0 <= rank < world):
        raise ValueError('rank %d outside world %d' % (rank, world))
    if replicas_per_point < world:
        # checked on EVERY rank before anything else: raised only on the ranks whose share is empty, the others would
        # go on into the all-reduce and wait there for the process-group time-out
        raise ValueError('%d replicas per concentration cannot be shared by %d ranks (some would receive none)'
                         % (replicas_per_point, world))
    k0, k1 = rank * replicas_per_point // world, (rank + 1) * replicas_per_point // world
    idx = np.tile(np.arange(conc_points.size), k1 - k0)
    return conc_points[idx], idx, k0 * conc_points.size


def accumulate(point_index, n_points, dstar, dj, time, hops, hist):
    """Local accumulators [n_points][13] in float64 (ACC_FIELDS)."""
    acc = np.zeros((n_points, len(ACC_FIELDS)), dtype=np.float64)
    for p in range(n_points):
        m = point_index == p
        acc[p, 0] = m.sum()
        acc[p, 1] = dstar[m].sum()
        acc[p, 2] = (dstar[m] ** 2).sum()
        acc[p, 3] = dj[m].sum()
        acc[p, 4] = (dj[m] ** 2).sum()
        acc[p, 5] = time[m].sum()
        acc[p, 6] = hops[m].sum()
        acc[p, 7:13] = hist[m].sum(axis=0)
    return acc


def all_reduce_acc(acc, device=None, rccl=None):
    """Sum the accumulators over all ranks (no-op without a process group).  ``rccl`` = (engine, communicator)
    sends them through the C ABI instead (kmc_allreduce_stats: one ncclAllReduce on the engine's stream)."""
    import sys
    if rccl is not None:
        eng, comm = rccl
        return eng.allreduce_stats(comm, acc).reshape(np.shape(acc))
    if 'torch' not in sys.modules:           # no process group can exist: single rank
        return acc
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return acc
    import torch
    t = torch.from_numpy(np.ascontiguousarray(acc))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def finalize(acc, conc_points):
    """Mean and population std (np.std, ddof 0) per concentration, as
    hex_LiCoO2_kmc.py:783-786; columns of LiCoO2_data.out (:671-673)."""
    out = []
    for p, x in enumerate(conc_points):
        n = acc[p, 0]
        m_s, m_j = acc[p, 1] / n, acc[p, 3] / n
        v_s = max(acc[p, 2] / n - m_s * m_s, 0.0)
        v_j = max(acc[p, 4] / n - m_j * m_j, 0.0)
        out.append(dict(concentration=float(x), replicas=int(n), d_star=m_s, d_j=m_j,
                        error_star=v_s ** 0.5, error_j=v_j ** 0.5, time=acc[p, 5] / n,
                        hops=int(acc[p, 6]), hop_histogram=acc[p, 7:13].astype(np.int64).tolist()))
    return out


def run_ensemble(engine_factory, nx, ny, conc_points, replicas_per_point, presim_steps, sim_steps,
                 seed=0, window_mode='hex_ref', rank=0, world=1, device=0, table=None, reduce_device=None,
                 return_replicas=False):
    """The reference's ensemble loop for this rank's share of the replicas.

    engine_factory(R, nx, ny, window_mode, device) -> KMCEngine (so tests can
    inject one); ``table`` = ('base', vec512, dumbbell) or ('dense', array).
    One "step" = int(nx*ny*x) hops (hex_LiCoO2_kmc.py:484); replicas of different
    concentration take different hop counts per step (kmc_run_counts).
    The replicas are in shard_replicas' order (iteration-major: local replica l has concentration l % n_points).
    With ``return_replicas`` a third value holds THIS rank's per-replica numbers (conc, d_star, d_j,
    hist, barrier statistics) for the per-iteration lines of LiCoO2_kmc.out (kmc_amd.report).
    """
    conc, point, offset = shard_replicas(conc_points, replicas_per_point, rank, world)
    R = conc.size
    eng = engine_factory(R, nx, ny, window_mode, device)
    try:
        eng.set_replica_offset(offset)
        if table[0] == 'base':
            from . import tables as _tables
            eng.set_base_table(table[1], table[2] if len(table) > 2 and table[2] is not None
                               else _tables.dumbbell_rate())
        else:
            eng.set_hex_table(table[1])
        eng.fill_random(conc, seed)                 # ONE seed for all ranks: streams are keyed by the global replica id
        eng.rates_full()
        hops_per_step = (nx * ny * 1 * conc).astype(np.int64)        # int(N*N*1*x), :484
        eng.run_counts(hops_per_step * presim_steps, presim=True, seed=seed + 1)
        eng.init_displacements()
        eng.run_counts(hops_per_step * sim_steps, presim=False, seed=seed + 1)
        st = eng.status()
        if st['status'].any():
            raise RuntimeError('replicas stopped early: status %r' % np.unique(st['status']))
        ds, dj, _ = eng.diffusion(conc)
        sc = eng.scalars()
        acc = accumulate(point, len(conc_points), ds, dj, sc['time'], st['hops_done'].astype(np.float64),
                         sc['hist'].astype(np.float64))
        local = dict(conc=conc, d_star=ds, d_j=dj, hist=sc['hist'], time=sc['time'],
                     barrier=eng.barrier_stats()) if return_replicas else None
    finally:
        eng.close()
    acc = all_reduce_acc(acc, reduce_device)
    if return_replicas:
        return finalize(acc, conc_points), acc, local
    return finalize(acc, conc_points), acc
