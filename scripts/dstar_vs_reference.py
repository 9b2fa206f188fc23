"""D*(x) of the engine against the reference's own published data files, per concentration.

The files: data_run_4_data (d_star, x, err_d_star, d_j, err_d_j; 45 points) and data_run_3_data (x, D*, err, D_J, err;
35 points), committed as tests/golden/ref_data_run_*_data.  Tables: the fold of normcons_dictionary.p (the table
the hex path is wired to at this revision) and of master_dictionary.p restricted to the builder's 512 keys (the
table BASELINE config 1 names), the latter also with the 1.48 eV dumbbell of the script that reads
master_dictionary.p (LiCoO2_kmc.py:30-32) instead of the hex builder's 0.9 eV (hex_dict_builder.py:29-30).
Shapes: BASELINE config 3 (64 x 64, 512 replicas per point, 250 + 750 KMC steps, hex_LiCoO2_kmc.py:656-657) and the
shape the reference's own LiCoO2_kmc.out header shows for its runs (10 x 10, 100 + 100 cycles).
    python scripts/dstar_vs_reference.py [out_dir] [replicas_per_point]"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kmc_amd
from kmc_amd import ensemble, tables

out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'dstar_vs_reference')
per = int(sys.argv[2]) if len(sys.argv) > 2 else 512
os.makedirs(out_dir, exist_ok=True)
G = os.path.join(ROOT, 'tests', 'golden')
t = np.load(os.path.join(G, 'tables.npz'))
run4 = np.loadtxt(os.path.join(G, 'ref_data_run_4_data'), delimiter=',', skiprows=1)          # d_star, x, err, d_j, err_j
run4 = run4[np.argsort(run4[:, 1])]
run3 = np.loadtxt(os.path.join(G, 'ref_data_run_3_data'), skiprows=1)                          # x, d_star, err, d_j, err_j
ref = {}
for row in run4:
    ref.setdefault(round(float(row[1]), 2), {})['run4'] = (float(row[0]), float(row[2]), float(row[3]), float(row[4]))
for row in run3:
    ref.setdefault(round(float(row[0]), 2), {})['run3'] = (float(row[1]), float(row[2]), float(row[3]), float(row[4]))

variants = [('normcons, dumbbell 0.9 eV', t['base512_normcons'], tables.dumbbell_rate()),
            ('master (restricted), dumbbell 0.9 eV', t['base512_master'], tables.dumbbell_rate()),
            ('master (restricted), dumbbell 1.48 eV', t['base512_master'], tables.PREFACTOR * math.exp(-1.48 / tables.KB_T))]
shapes = [('config 3: 64x64, 250 + 750 steps', 64, 250, 750, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]),
          ('reference run shape: 10x10, 100 + 100 steps', 10, 100, 100, sorted(ref))]
factory = lambda R, nx, ny, wm, dev: kmc_amd.KMCEngine(R, nx, ny, wm, dev)     # noqa: E731
out = {'replicas_per_point': per, 'runs': []}
lines = []
for sname, n, pre, sim, xs in shapes:
    for vname, base, dumb in variants:
        t0 = time.time()
        fin, acc, loc = ensemble.run_ensemble(factory, n, n, xs, per, pre, sim, seed=ensemble.derive_seed(2025, n),
                                              table=('base', base, dumb), return_replicas=True)
        dt = time.time() - t0
        rows = []
        lines.append('\n### %s -- %s (%d replicas per point, %.1f s, %.3g hops)\n' % (sname, vname, per, dt, sum(r['hops'] for r in fin)))
        lines.append('| x | D* here (mean +- std over replicas) | median D* here | run_4 D* +- err | inside run_4 bar | log10(here / run_4) | run_3 D* +- err | inside run_3 bar | log10(here / run_3) |')
        lines.append('|---|---|---|---|---|---|---|---|---|')
        for r in fin:
            x = round(r['concentration'], 2)
            med = float(np.median(loc['d_star'][loc['conc'] == r['concentration']]))
            row = dict(x=x, d_star=r['d_star'], std=r['error_star'], median=med, d_j=r['d_j'], std_j=r['error_j'])
            cells = ['%.2f' % x, '%.3e +- %.1e' % (r['d_star'], r['error_star']), '%.3e' % med]
            for key in ('run4', 'run3'):
                if key in ref.get(x, {}):
                    m, e = ref[x][key][0], ref[x][key][1]
                    inside = abs(r['d_star'] - m) <= e
                    row[key] = dict(d_star=m, err=e, inside=bool(inside), log10_ratio=math.log10(r['d_star'] / m))
                    cells += ['%.3e +- %.1e' % (m, e), 'yes' if inside else 'no', '%+.2f' % row[key]['log10_ratio']]
                else:
                    cells += ['-', '-', '-']
            rows.append(row)
            lines.append('| ' + ' | '.join(cells) + ' |')
        out['runs'].append(dict(shape=sname, table=vname, n=n, presim=pre, sim=sim, seconds=dt, rows=rows))
        print(sname, '|', vname, '| %.1f s' % dt, flush=True)
json.dump(out, open(os.path.join(out_dir, 'dstar_vs_reference.json'), 'w'), indent=1)
open(os.path.join(out_dir, 'dstar_vs_reference.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
