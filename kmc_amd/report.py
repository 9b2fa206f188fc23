"""Result writers and readers of the reference's driver script.

The reference writes two files from its ensemble loop and its plotters read them back:

* ``LiCoO2_kmc.out`` -- free text: a header (hex_LiCoO2_kmc.py:665-668), two lines per
  concentration with the per-iteration D* and D_J (:779-782), the wall time and a final
  table ``d_star variable error_d_star d_j error_d_j`` (:803-814);
* ``LiCoO2_data.out`` -- one space-separated row per (concentration, dimension) "to be opened
  and plotted with pandas" (:671-673, :788-789);
* ``data_run_*_data`` -- the same table as CSV ``d_star,x,err_d_star,d_j,err_d_j`` (files in the
  reference's root; what diffusion_vs_concentration_plotter.py's arrays were pasted from).

Numbers are written with Python-3 ``repr`` precision (the reference ran under Python 2, whose
``str(float)`` keeps 12 digits, and numpy 1.x array printing keeps 8): every reader of the
reference's files parses these unchanged, nothing is lost on a round trip.  The diagnostics of
the loop -- ``offset_score`` (:760-761) and the barrier summary (:795-796) -- are here too.
"""
import re

import numpy as np

DATA_HEADER = 'Concentration Dimension D_star D_j error_star error_j time\n'           # :672-673
TABLE_HEADER = ' d_star  variable error_d_star d_j error_d_j \n'                       # :807
CSV_HEADER = 'd_star,x,err_d_star,d_j,err_d_j\n'                                       # data_run_*_data:1


def _arr(values):
    """``str(np.array)`` of the reference's lists of per-iteration coefficients (:780, :782),
    in numpy 1.x layout: ``[  4.66504294e-20   6.23843623e-20]``."""
    return '[ ' + '  '.join('% .8e' % v for v in np.asarray(values, dtype=np.float64).ravel()) + ']'


def offset_score(hop_histogram):
    """Mean count of directions 1, 4 over mean count of directions 0, 2, 3, 5 (:760-761): 1 for an
    isotropic table.  Accepts (6,) or (R, 6)."""
    h = np.asarray(hop_histogram, dtype=np.float64)
    return ((h[..., 1] + h[..., 4]) / 2) / ((h[..., 0] + h[..., 2] + h[..., 3] + h[..., 5]) / 4)


def barrier_summary(stats):
    """(average, max, min) chosen barrier over all recorded hops of all replicas (:795-796), from
    ``KMCEngine.barrier_stats()``.  The reference appends the LAST barrier of each KMC step to
    ``total_barrier`` (:752); the engine accumulates every hop with idx != 0, a superset."""
    n = stats['count'].sum()
    if n == 0:
        raise ValueError('no barrier recorded (presimulation only, or every hop had idx == 0)')
    return stats['sum'].sum() / n, stats['max'][stats['count'] > 0].max(), stats['min'][stats['count'] > 0].min()


class KmcOut:
    """Writer of ``LiCoO2_kmc.out``.  ``with KmcOut(path, size, kb_t, presim, sim) as out: out.point(...)``."""

    def __init__(self, path, mc_lattice_size, kb_t, presimulation_iterations, simulation_iterations):
        self._f = open(path, 'w')
        self._rows = []
        self._f.write('Size:' + str(list(mc_lattice_size)) + '\n kbT:' + str(kb_t) +
                      '\n Simulation parameters: \n Presimulation cycles: ' + str(presimulation_iterations) +
                      '\n Simulation Cycles: ' + str(simulation_iterations) + '\n')                 # :666-667
        self._f.write('\n RESULTS \n')                                                            # :668

    def point(self, variable, d_star_iterations, d_j_iterations, summary=None):
        """One concentration (or whatever the master loop varies): the per-iteration coefficients
        (:779-782); mean and np.std (ddof 0, :783-786) go to the final table, unless ``summary`` =
        (d_star, error_star, d_j, error_j) brings them (means over more replicas than are listed)."""
        ds = np.asarray(d_star_iterations, dtype=np.float64)
        dj = np.asarray(d_j_iterations, dtype=np.float64)
        self._f.write('Li concentration: ' + str(variable) + '. Diffusion_coefficients at this concentration: ' +
                      _arr(ds) + '\n')
        self._f.write('Li concentration: ' + str(variable) + '. Diffusion_coefficient J at this concentration: ' +
                      _arr(dj) + '\n')
        row = (np.average(ds), float(variable), np.std(ds), np.average(dj), np.std(dj)) if summary is None \
            else (summary[0], float(variable), summary[1], summary[2], summary[3])
        self._rows.append(row)
        return row

    def close(self, wall_seconds=0.0):
        if self._f is None:
            return
        self._f.write('\n\n Total Time: ' + str(wall_seconds) + '\n \n')                          # :803-804
        self._f.write('\n\n\n END\n')                                                             # :805
        self._f.write(TABLE_HEADER)
        self._f.write('\n'.join('  ' + '   '.join('%.8e' % v for v in r[:4]) + ' %.8e' % r[4]
                                for r in self._rows))                                             # :810-814
        self._f.close()
        self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_NUM = r'[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?|nan|inf'


def read_kmc_out(path):
    """Parse a ``LiCoO2_kmc.out`` (the reference's or ours) into a dict: ``size``, ``kb_t``, ``presim``,
    ``sim``, ``points`` = [(x, D* iterations, D_J iterations)], ``total_time``, ``table`` (n, 5)."""
    text = open(path).read()
    head, _, rest = text.partition('RESULTS')
    body, _, tail = rest.partition('END')
    out = dict(size=[int(v) for v in re.search(r'Size:\[(.*?)\]', head).group(1).split(',')],
               kb_t=float(re.search(r'kbT:(\S+)', head).group(1)),
               presim=int(re.search(r'Presimulation cycles:\s*(\d+)', head).group(1)),
               sim=int(re.search(r'Simulation Cycles:\s*(\d+)', head).group(1)))
    pts = {}
    for m in re.finditer(r'Li concentration: (\S+?)\. Diffusion_coefficient(s| J) at this concentration: \[(.*?)\]',
                         body, re.S):
        vals = np.array([float(v) for v in m.group(3).split()])
        pts.setdefault(m.group(1), [None, None])[0 if m.group(2) == 's' else 1] = vals
    out['points'] = [(float(k), v[0], v[1]) for k, v in pts.items()]
    m = re.search(r'Total Time:\s*(\S+)', body)
    out['total_time'] = float(m.group(1)) if m else None
    rows = [[float(v) for v in re.findall(_NUM, ln)] for ln in tail.splitlines()[1:] if re.search(r'\d', ln)]
    out['table'] = np.array([r for r in rows if len(r) == 5]).reshape(-1, 5)
    return out


class DataOut:
    """Writer of ``LiCoO2_data.out`` (:671-673, :788-789)."""

    def __init__(self, path):
        self._f = open(path, 'w')
        self._f.write(DATA_HEADER)

    def row(self, concentration, dimension, d_star, d_j, error_star, error_j, total_time):
        self._f.write(' '.join(str(v) for v in (concentration, dimension, d_star, d_j, error_star, error_j,
                                                 total_time)) + '\n')

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_data_out(path):
    """``LiCoO2_data.out`` as a dict of column -> float array (what ``pandas.read_csv(path, sep=' ')``
    gives, without needing pandas)."""
    with open(path) as f:
        cols = f.readline().split()
        rows = [[float(v) for v in ln.split()] for ln in f if ln.strip()]
    a = np.array(rows, dtype=np.float64).reshape(-1, len(cols))
    return {c: a[:, k] for k, c in enumerate(cols)}


def write_data_run(path, table):
    """``data_run_*_data``: CSV of rows (d_star, x, err_d_star, d_j, err_d_j), %.8e like the reference's files."""
    with open(path, 'w') as f:
        f.write(CSV_HEADER)
        for r in np.asarray(table, dtype=np.float64).reshape(-1, 5):
            f.write(','.join('%.8e' % v for v in r) + '\n')


def read_data_run(path):
    """``data_run_*_data`` -> (n, 5) array (trailing blank lines of the reference's files ignored)."""
    with open(path) as f:
        if f.readline() != CSV_HEADER:
            raise ValueError('%s: not a data_run file (header)' % path)
        rows = [[float(v) for v in ln.split(',')] for ln in f if ln.strip()]
    return np.array(rows, dtype=np.float64).reshape(-1, 5)


def write_sweep(results, dimension, kmc_out_path, data_out_path, kb_t, presimulation_iterations,
                simulation_iterations, per_replica=None, wall_seconds=0.0, data_run_path=None):
    """Both files from ``ensemble.run_ensemble`` results (list of dicts, one per concentration).
    ``per_replica`` = {x: (D* array, D_J array)} fills the per-iteration lines; without it the mean
    stands in as a single entry and the table's errors come from the results' ``error_*``."""
    table = []
    with KmcOut(kmc_out_path, [dimension, dimension, 1], kb_t, presimulation_iterations,
                simulation_iterations) as out, DataOut(data_out_path) as dat:
        for r in results:
            x = r['concentration']
            ds, dj = per_replica[x] if per_replica is not None else ([r['d_star']], [r['d_j']])
            table.append(out.point(x, ds, dj, (r['d_star'], r['error_star'], r['d_j'], r['error_j'])))
            dat.row(x, dimension, r['d_star'], r['d_j'], r['error_star'], r['error_j'], r['time'])
        out.close(wall_seconds)
    if data_run_path is not None:
        write_data_run(data_run_path, table)
    return np.array(table).reshape(-1, 5)
