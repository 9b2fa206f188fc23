"""The reference's driver script as a function: the dimension x concentration x averaging loops of
hex_LiCoO2_kmc.py:650-818 over batched replicas, writing the same result files.

    python -m kmc_amd.sweep --table normcons_dictionary.p --dimensions 64 \\
        --concentrations 0.1 0.2 0.3 0.4 0.5 0.6 0.7 0.8 0.9 --averaging 512 --presim 250 --sim 750

writes ``LiCoO2_kmc.out`` and ``LiCoO2_data.out`` (kmc_amd.report) into ``--out-dir`` and prints, per
dimension, ``offset_score`` (:760-762) and the barrier summary (:795-796).  Under ``torch.distributed.run``
every rank takes an equal share of each concentration's replicas and rank 0 writes the files (per-iteration
lines then list rank 0's replicas; means and errors are over all ranks).
"""
import argparse
import os
import time

import numpy as np

from . import ensemble, report, tables
from .engine import KMCEngine

KB_T = 0.025                                                 # hex_LiCoO2_kmc.py:33


def run_sweep(table, dimensions, concentrations, averaging_iterations, presimulation_iterations=250,
              simulation_iterations=750, seed=0, out_dir='.', rank=0, world=1, device=0, reduce_device=None,
              engine_factory=None, window_mode='hex_ref', kb_t=KB_T, echo=True):
    """Returns {dimension: results list of ensemble.finalize}.  File layout as the reference's: one
    LiCoO2_kmc.out / LiCoO2_data.out for the whole run; the .out header carries the LAST dimension's size
    only in the reference (it is written before the loop, as [0, 0, 0]); here the first dimension's."""
    factory = engine_factory or (lambda R, nx, ny, wm, dev: KMCEngine(R, nx, ny, wm, dev))
    start_timer = time.time()
    all_results = {}
    out = dat = None
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        out = report.KmcOut(os.path.join(out_dir, 'LiCoO2_kmc.out'), [dimensions[0], dimensions[0], 1], kb_t,
                            presimulation_iterations, simulation_iterations)
        dat = report.DataOut(os.path.join(out_dir, 'LiCoO2_data.out'))
    try:
        for dimension in dimensions:
            results, _, local = ensemble.run_ensemble(
                factory, dimension, dimension, concentrations, averaging_iterations, presimulation_iterations,
                simulation_iterations, seed=ensemble.derive_seed(seed, dimension), window_mode=window_mode, rank=rank, world=world,
                device=device, table=table, reduce_device=reduce_device, return_replicas=True)
            all_results[dimension] = results
            if rank != 0:
                continue
            for r in results:
                x = r['concentration']
                m = local['conc'] == x
                out.point(x, local['d_star'][m], local['d_j'][m],
                          (r['d_star'], r['error_star'], r['d_j'], r['error_j']))
                dat.row(x, dimension, r['d_star'], r['d_j'], r['error_star'], r['error_j'], r['time'])
                if echo:
                    print('concentration is: ' + str(x))                                       # :687
                    print('offset_score = ' + str(report.offset_score(np.array(r['hop_histogram']))))
            if echo:
                try:
                    avg, hi, lo = report.barrier_summary(local['barrier'])
                    print('average barrier = ' + str(avg))                                     # :795
                    print('max, min barrir = ' + str(hi) + ', ' + str(lo))                     # :796
                except ValueError:
                    pass
    finally:
        if rank == 0:
            out.close(time.time() - start_timer)
            dat.close()
    return all_results


def temperature_sweep(kb_ts, dimensions, concentrations, averaging_iterations, presimulation_iterations=250,
                      simulation_iterations=750, prefactor=tables.PREFACTOR, eci=tables.LCE_ECI, seed=0, out_dir='.',
                      **kw):
    """The sweep at several temperatures.  The reference's table "must be rerun for each temperature"
    (hex_LiCoO2_kmc.py:345, minutes of Python per kT); here the 512 rates 1e13 exp(-E / kT) come from the device
    generator (kmc_generate_base_table, :369-403 with :103-199) and the dumbbell rate from
    hex_dict_builder.py:29-30 at the same kT.  Files go to ``out_dir/kbT_<kT>/``.  Returns {kT: run_sweep result}."""
    out = {}
    for kb_t in kb_ts:
        with KMCEngine(1, 8, 8, 'hex_ref', kw.get('device', 0)) as eng:
            base, _ = eng.generate_base_table(eci, prefactor, kb_t)
        table = ('base', base, tables.dumbbell_rate(prefactor, kb_t))
        out[kb_t] = run_sweep(table, dimensions, concentrations, averaging_iterations, presimulation_iterations,
                              simulation_iterations, seed=seed, out_dir=os.path.join(out_dir, 'kbT_' + str(kb_t)),
                              kb_t=kb_t, **kw)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--table', help='normcons_dictionary.p / master_dictionary.p / hex_dict.p')
    ap.add_argument('--kbt', type=float, nargs='+', help='instead of --table: generate the rates at these kT (eV)')
    ap.add_argument('--dimensions', type=int, nargs='+', default=[10])                      # :655
    ap.add_argument('--concentrations', type=float, nargs='+', default=[0.2])               # :659
    ap.add_argument('--averaging', type=int, default=8)
    ap.add_argument('--presim', type=int, default=250)                                      # :657
    ap.add_argument('--sim', type=int, default=750)                                         # :656
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--out-dir', default='.')
    a = ap.parse_args(argv)
    rank, world, local_rank = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('WORLD_SIZE', 1),
                                                                       ('LOCAL_RANK', 0)))
    reduce_device = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl')
        reduce_device = 'cuda:%d' % local_rank
    if (a.table is None) == (a.kbt is None):
        ap.error('give exactly one of --table and --kbt')
    if a.kbt is not None:
        temperature_sweep(a.kbt, a.dimensions, a.concentrations, a.averaging, a.presim, a.sim, seed=a.seed,
                          out_dir=a.out_dir, rank=rank, world=world, device=local_rank, reduce_device=reduce_device)
    else:
        kind, arr = tables.load_any(a.table)
        table = ('base', arr, tables.dumbbell_rate()) if kind == 'base' else ('dense', arr)
        run_sweep(table, a.dimensions, a.concentrations, a.averaging, a.presim, a.sim, a.seed, a.out_dir, rank,
                  world, local_rank, reduce_device)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
