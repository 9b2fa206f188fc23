"""CPU oracle for the hex-lattice KMC hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch numpy / pure-Python restatement of the algorithm in
the reference (quentinstoyel/KMC).  It is the *checker* for the HIP engine:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product (``kmc_amd``) never does.

Parity status: PINNED.  ``oracle/gen_golden_from_reference.py`` executes the
reference's own function text (in the build container only) and the resulting
fixtures in ``tests/golden/`` are reproduced bit-for-bit by this file
(``tests/test_oracle_golden.py``).

Citations are ``file:line`` in the reference tree.

Conventions
-----------
occ            uint8 (nx, ny), 1 = Li, 0 = vacancy          (reference stores
               float -1 / +1, hex_LiCoO2_kmc.py:25, :703-713)
P              float64 (nx, ny, 6) hop-rate matrix          (:695-696, k axis dropped)
tags           int64 (nx, ny, 3) = [i0, j0, 100*wi + wj]    (:697-698, :521-528)
flat index     (i*ny + j)*6 + dir                           (:494, :500-501)
"""
from __future__ import annotations

import math
import numpy as np

# --------------------------------------------------------------------------
# constants (hex_LiCoO2_kmc.py:25-29, hex_dict_builder.py:26-30)
# --------------------------------------------------------------------------
KB_T = 0.025
PREFACTOR = 10 ** 13
LATTICE_CONSTANT = 2.8334 * 10 ** (-8)
DUMBBELL_BARRIER = 0.900
VACANT_KEY = 0xFFFFFFFF

# 18-site neighbourhood, key char p -> (di, dj)        (hex_LiCoO2_kmc.py:423-424)
NEIGHBOUR_OFFSETS = (
    (0, 2), (-1, 2), (-2, 2),
    (1, 1), (0, 1), (-1, 1), (-2, 1),
    (2, 0), (1, 0), (-1, 0), (-2, 0),
    (2, -1), (1, -1), (0, -1), (-1, -1),
    (2, -2), (1, -2), (0, -2),
)
# hop directions, dir -> (di, dj)                      (hex_LiCoO2_kmc.py:503-504)
HOP_DIRS = ((1, 0), (0, 1), (-1, 1), (-1, 0), (0, -1), (1, -1))
# key position of the end point of each direction      (hex_dict_builder.py:41)
ENDPOINT_LABELS = (8, 4, 5, 9, 13, 12)
# the nine free slots of the 16-char base key           (hex_dict_builder.py:58-62)
BASE_FREE_POSITIONS = (0, 1, 4, 6, 8, 11, 13, 14, 15)
# which hex-key char fills each free slot, per (direction, view)
# view 0 is used when the *previous* ring neighbour is empty, view 1 when the
# *next* one is.                                        (hex_dict_builder.py:56-134)
VIEWS = (
    ((5, 9, 4, 13, 3, 16, 7, 11, 15), (13, 9, 12, 5, 11, 1, 7, 3, 0)),
    ((9, 13, 5, 12, 1, 11, 0, 3, 7), (12, 13, 8, 9, 3, 6, 0, 1, 2)),
    ((13, 12, 9, 8, 6, 3, 2, 1, 0), (8, 12, 4, 13, 1, 14, 2, 6, 10)),
    ((12, 8, 13, 4, 14, 1, 10, 6, 2), (4, 8, 5, 12, 6, 16, 10, 14, 17)),
    ((8, 4, 12, 5, 16, 6, 17, 14, 10), (5, 4, 9, 8, 14, 11, 17, 16, 15)),
    ((4, 5, 8, 9, 11, 14, 15, 16, 17), (9, 5, 13, 4, 16, 3, 15, 11, 7)),
)


def dumbbell_rate(prefactor=PREFACTOR, kb_t=KB_T):
    """hex_dict_builder.py:29-30."""
    return prefactor * math.exp(-(DUMBBELL_BARRIER) / kb_t)


# --------------------------------------------------------------------------
# row G: barrier model -> base table            (hex_LiCoO2_kmc.py:103-199, :369-403)
# --------------------------------------------------------------------------
LCE_ECI = (
    0.5039840998, 0.0231779629, -0.0124155613, 0.0190676809, -0.0240978034,
    0.0359364321, -0.0322688382, 0.0189131728, 0.0059693908, -0.0107495262,
    0.0044709962, 0.0076906411, -0.0135407679, 0.0014011203, -0.0008762663,
    -0.0015711384, -0.0074600309, -0.0049897914, 0.0098191199, 0.0087838401,
    0.0032917204, 0.0075716889, 0.0020836078, -0.0043424798, -0.0107164342,
    -0.0079384991, -0.0047555242, 0.0061737865, 0.0100261837, 0.0016684245,
    0.0040444273, 0.007575093, -0.0146307688, -0.0101701531, 0.0011528037,
    -0.0028328131, -0.0020787791, 0.0038948159, 0.0131163879, 0.0048079312,
    0.0016672859, 0.0047770905, 0.0114784225, 0.0028142975, -0.002387814,
    -0.0109707522, -0.0032730688,
)
# site products of clusters 2..46 (cluster 0 = constant, cluster 1 = sum of all
# nine sites); numbers are the "sites 1-9" of hex_LiCoO2_kmc.py:131-132, in
# the multiplication order of :147-196
LCE_CLUSTERS = (
    (1,), (3,), (4,), (7,), (5,), (8,), (6,), (9,), (2,),
    (6, 9), (4, 1), (3, 7), (4, 5, 6), (7, 8, 9), (1, 2, 3), (5, 2, 8),
    (5, 9), (6, 8), (5, 1), (3, 8), (2, 4), (2, 7), (5, 6), (8, 9), (5, 4),
    (7, 8), (1, 2), (2, 3), (2, 1, 4), (2, 3, 7), (5, 4, 1), (8, 7, 3),
    (5, 6, 9), (8, 9, 6), (1, 3), (4, 6), (7, 9), (1, 4, 5, 6), (3, 7, 8, 9),
    (5, 1, 2, 3, 6), (2, 8), (5, 8), (5, 4, 1, 2), (5, 6, 9, 8), (2, 3, 7, 8),
)
# 16-char key position feeding "site s" (s = 1..9)     (hex_LiCoO2_kmc.py:131)
LCE_SITE_POSITIONS = (None, 0, 4, 8, 1, 6, 11, 13, 14, 15)


def excited_energy(key16: str) -> float:
    """Barrier (eV) of the canonical hop for a 16-char '0'/'1' local key.

    hex_LiCoO2_kmc.py:103-199.  The off-cluster correction (:138-142) counts
    the int 1 in a list of chars and is therefore always 0; it is dropped.
    Guards (:117-122) raise ValueError with the reference's messages.
    """
    if key16[9] != '0':
        raise ValueError("Error: endpoint not empty!")
    if key16[5] != '1':
        raise ValueError("Error: no ion to move!")
    if key16[10] != '0':
        raise ValueError("Error: nearest neighbour not empty!")
    # '1' (Li) -> -1.0, '0' (Vac) -> +1.0                        (:130-132)
    s = [0.0] + [(-1.0 if key16[LCE_SITE_POSITIONS[q]] == '1' else 1.0)
                 for q in range(1, 10)]
    product = [0.0] * 47
    product[0] = LCE_ECI[0]
    tot = 0
    for q in range(1, 10):          # python sum(), left to right   (:146)
        tot = tot + s[q]
    product[1] = LCE_ECI[1] * tot
    for c, sites in enumerate(LCE_CLUSTERS, start=2):
        v = s[sites[0]]
        for q in sites[1:]:
            v = v * s[q]
        product[c] = LCE_ECI[c] * v
    e = 0
    for v in product:               # python sum(), left to right   (:197)
        e = e + v
    return e


def hop_rate_from_key(key16: str, prefactor=PREFACTOR, kb_t=KB_T) -> float:
    """get_hop_probability, live branch: hex_LiCoO2_kmc.py:386, :402-403."""
    return prefactor * math.exp(-(excited_energy(key16)) / kb_t)


def base_key16(index: int) -> str:
    """16-char key of base-table entry ``index`` (nine free chars MSB first).

    Template hex_LiCoO2_kmc.py:353-354 / hex_dict_builder.py:58-62: forced
    '1' at 2,3,5,7,12 and '0' at 9,10.
    """
    chars = ['1', '1', '1', '1', '1', '1', '1', '1', '1', '0', '0', '1', '1', '1', '1', '1']
    for q, pos in enumerate(BASE_FREE_POSITIONS):
        chars[pos] = '1' if (index >> (8 - q)) & 1 else '0'
    return ''.join(chars)


def generate_base512(prefactor=PREFACTOR, kb_t=KB_T) -> np.ndarray:
    """dictionary_builder (hex_LiCoO2_kmc.py:345-365) as a 512-vector."""
    return np.array([hop_rate_from_key(base_key16(i), prefactor, kb_t)
                     for i in range(512)], dtype=np.float64)


def base512_from_normcons(d: dict) -> np.ndarray:
    """512-vector from a normcons_dictionary.p-style dict (str '0'/'1' keys)."""
    out = np.empty(512, dtype=np.float64)
    for i in range(512):
        out[i] = d[base_key16(i)]
    return out


def base512_from_master(d: dict) -> np.ndarray:
    """512-vector from a master_dictionary.p-style dict (int keys of digits
    1 = Li / 3 = Vac, writer = old LiCoO2_kmc.py:170-208): restriction to the
    keys with sites 2,3,7,12 = Li (hex_dict_builder.py:31, :58-62)."""
    out = np.empty(512, dtype=np.float64)
    for i in range(512):
        out[i] = d[int(base_key16(i).replace('0', '3'))]
    return out


# --------------------------------------------------------------------------
# row T: 2^18 x 6 dense table                          (hex_dict_builder.py:38-142)
# --------------------------------------------------------------------------
def build_hex_table(base512: np.ndarray, dumbbell: float | None = None) -> np.ndarray:
    """Dense table T[row, dir], row = int(18-char key, 2) (char 0 = bit 17)."""
    if dumbbell is None:
        dumbbell = dumbbell_rate()
    base512 = np.asarray(base512, dtype=np.float64)
    rows = np.arange(1 << 18, dtype=np.int64)

    def bit(p):
        return (rows >> (17 - p)) & 1

    table = np.zeros((1 << 18, 6), dtype=np.float64)
    for e in range(6):
        end = bit(ENDPOINT_LABELS[e])
        prev = bit(ENDPOINT_LABELS[e - 1])                  # cyclic (:52, :57)
        nxt = bit(ENDPOINT_LABELS[(e + 1) % 6])
        idx = []
        for view in VIEWS[e]:
            v = np.zeros_like(rows)
            for q, p in enumerate(view):
                v |= bit(p) << (8 - q)
            idx.append(v)
        r1 = np.where(prev == 0, base512[idx[0]], 0.0)      # sentinel '3' -> 0 (:21, :46)
        r2 = np.where(nxt == 0, base512[idx[1]], 0.0)
        rate = r1 + r2                                      # (:136-137)
        rate = np.where((prev == 1) & (nxt == 1), dumbbell, rate)   # (:52-53)
        rate = np.where(end == 1, 0.0, rate)                # (:138-139)
        table[:, e] = rate
    return table


# --------------------------------------------------------------------------
# rows R1-R3: keys and rates
# --------------------------------------------------------------------------
def site_key(occ: np.ndarray, i: int, j: int) -> int:
    """18-bit key of site (i, j)                       (hex_LiCoO2_kmc.py:422-427)."""
    nx, ny = occ.shape
    key = 0
    for di, dj in NEIGHBOUR_OFFSETS:
        key = (key << 1) | int(occ[(i + di) % nx, (j + dj) % ny])
    return key


def all_keys(occ: np.ndarray) -> np.ndarray:
    """Fresh keys of every site (VACANT_KEY where there is no Li)."""
    occ = np.asarray(occ).astype(np.int64)
    key = np.zeros(occ.shape, dtype=np.int64)
    for di, dj in NEIGHBOUR_OFFSETS:
        key = (key << 1) | np.roll(np.roll(occ, -di, axis=0), -dj, axis=1)
    return np.where(occ == 1, key, VACANT_KEY).astype(np.uint32)


def rates_full(occ: np.ndarray, table: np.ndarray) -> np.ndarray:
    """kmc_evolve, initialise mode                     (hex_LiCoO2_kmc.py:435-443, :453-465)."""
    keys = all_keys(occ)
    P = np.zeros(occ.shape + (6,), dtype=np.float64)
    li = keys != VACANT_KEY
    P[li] = table[keys[li]]
    return P


def refresh_window(window_mode: str, start, end_unwrapped):
    """Half-open (i_lo, i_hi, j_lo, j_hi) of the sites re-evaluated after a hop.

    hex_ref : hex_LiCoO2_kmc.py:446-449  (misses 6-7 affected sites)
    obj_ref : Mc_Lattice.py:178          (30 sites around the start point)
    full    : inclusive bounding box of both radius-2 hexagons (no stale rates)
    """
    (i_s, j_s), (i_e, j_e) = start, end_unwrapped
    if window_mode == 'hex_ref':
        return (min(i_s, i_e) - 2, max(i_s, i_e) + 2, min(j_s, j_e) - 2, max(j_s, j_e) + 2)
    if window_mode == 'obj_ref':
        return (i_s - 2, i_s + 3, j_s - 2, j_s + 4)
    if window_mode == 'full':
        return (min(i_s, i_e) - 2, max(i_s, i_e) + 3, min(j_s, j_e) - 2, max(j_s, j_e) + 3)
    raise ValueError(window_mode)


def rates_window(occ, P, table, start, end_unwrapped, window_mode='hex_ref'):
    """kmc_evolve, windowed mode, in place             (hex_LiCoO2_kmc.py:444-465)."""
    nx, ny = occ.shape
    i_lo, i_hi, j_lo, j_hi = refresh_window(window_mode, start, end_unwrapped)
    for i in range(i_lo, i_hi):
        for j in range(j_lo, j_hi):
            ii, jj = i % nx, j % ny
            if occ[ii, jj]:
                P[ii, jj] = table[site_key(occ, ii, jj)]
            else:
                P[ii, jj] = 0.0
    return P


# --------------------------------------------------------------------------
# rows S1-S5, U1-U4, K: the hop loop
# --------------------------------------------------------------------------
class SelectionOverflow(IndexError):
    """rho above cumsum[-1]/gamma: the reference raises IndexError."""


def select_hop(P: np.ndarray, rho: float):
    """gamma, normalised CDF, search                   (hex_LiCoO2_kmc.py:486-492).

    Uses numpy's own np.sum / np.cumsum / np.searchsorted, exactly the calls
    the reference makes, so the floating-point association is the reference's.
    Returns (idx, gamma, cdf).
    """
    gamma = np.sum(P)
    with np.errstate(invalid='ignore', divide='ignore'):
        cdf = np.cumsum(P) / gamma
    idx = int(np.searchsorted(cdf, rho))
    return idx, gamma, cdf


def hop_once(occ, P, tags, table, rho, u_time, presim, window_mode='hex_ref'):
    """One iteration of the loop body hex_LiCoO2_kmc.py:485-535 (in place).

    Returns dict(idx, gamma, dt, barrier).  ``barrier`` is None when idx == 0
    (the reference leaves chosen_barrier unassigned, :495).
    """
    nx, ny = occ.shape
    idx, gamma, cdf = select_hop(P, rho)
    if idx >= cdf.size:
        raise SelectionOverflow(idx)
    d = idx % 6                                             # (:494)
    barrier = None
    if idx != 0:                                            # (:495-498)
        with np.errstate(invalid='ignore', divide='ignore'):
            barrier = -math.log((cdf[idx] - cdf[idx - 1]) / 10 ** 13) * 0.025 \
                if cdf[idx] - cdf[idx - 1] > 0 else float('nan')
    j_s = (idx // 6) % ny                                   # (:500-501)
    i_s = (idx // 6) // ny
    i_e = i_s + HOP_DIRS[d][0]                              # unwrapped (:503-506)
    j_e = j_s + HOP_DIRS[d][1]
    occ[i_s, j_s] = 0                                       # (:518)
    occ[i_e % nx, j_e % ny] = 1                             # (:519-520)
    dt = 0.0
    if not presim:                                          # (:521-532)
        tags[i_e % nx, j_e % ny] = tags[i_s, j_s]
        tags[i_e % nx, j_e % ny, 2] += 100 * (i_e // nx) + (j_e // ny)
        tags[i_s, j_s] = (-1, 0, 0)
        dt = (-1. / gamma) * np.log(u_time)
    rates_window(occ, P, table, (i_s, j_s), (i_e, j_e), window_mode)   # (:533-534)
    return dict(idx=idx, gamma=float(gamma), dt=float(dt), barrier=barrier, dir=d)


def kmc_step(occ, tags, P, table, n_hops, draws, presim, window_mode='hex_ref',
             hist=None, trace=None):
    """kmc_step                                         (hex_LiCoO2_kmc.py:470-542).

    ``draws`` is an iterator of uniforms consumed exactly like the reference
    consumes np.random.uniform: one per hop in presimulation mode, two per hop
    (rho, then the clock draw) otherwise (:490, :530).
    Mutates occ, tags, P.  Returns (step_time, last_barrier).
    """
    step_time = 0
    barrier = None
    for _ in range(n_hops):
        rho = next(draws)
        u = None if presim else next(draws)
        r = hop_once(occ, P, tags, table, rho, u, presim, window_mode)
        if r['barrier'] is not None:
            barrier = r['barrier']
        if not presim:
            step_time = step_time + r['dt']                 # (:531-532)
        if hist is not None:
            hist[r['dir']] += 1                             # (:535)
        if trace is not None:
            trace.append((r['idx'], r['gamma']))
    return step_time, barrier


def init_tags(occ):
    """hex_LiCoO2_kmc.py:734-740."""
    nx, ny = occ.shape
    tags = np.zeros((nx, ny, 3), dtype=np.int64)
    tags[..., 0] = -1
    ii, jj = np.nonzero(occ)
    tags[ii, jj, 0] = ii
    tags[ii, jj, 1] = jj
    return tags


def fill_lattice_reference(nx, ny, conc, randint):
    """Random fill exactly as hex_LiCoO2_kmc.py:703-714 (three randint draws
    per attempt).  ``randint(lo, hi)`` has np.random.randint
    semantics."""
    occ = np.zeros((nx, ny), dtype=np.uint8)
    n_li = 0
    while n_li <= conc * (nx * ny) - 1:
        i = randint(0, nx)
        j = randint(0, ny)
        randint(0, 1)
        if occ[i, j] == 0:
            occ[i, j] = 1
            n_li += 1
    return occ


# --------------------------------------------------------------------------
# rows D1, D2: observables
# --------------------------------------------------------------------------
def py2_round(x: float) -> float:
    """Python-2 ``round`` (half away from zero), as the reference ran it."""
    return math.floor(abs(x) + 0.5) * (1.0 if x >= 0 else -1.0)


def get_distance(tags, size, lattice_constant=LATTICE_CONSTANT, rounding=round):
    """get_distance                                    (hex_LiCoO2_kmc.py:558-586).

    ``rounding`` is the ``round`` builtin in use: Python 3's (banker's) by
    default, which is what the golden vectors were generated with; Python 2's
    differs only for |w| = 50 (mod 100), outside the packing's valid range
    (SURVEY B7).
    Returns (matrix*a, list*a, d_j*a).
    """
    nx, ny = size[0], size[1]
    mat = np.zeros((nx, ny))
    lst = []
    s_i = 0.0
    s_j = 0.0
    for i in range(nx):
        for j in range(ny):
            i0, j0, w = (int(v) for v in tags[i, j])
            if i0 >= 0:
                w_i = rounding(float(w) / 100)
                i_index = i + nx * w_i
                j_index = j + ny * (w - w_i * 100)
                di = i0 - i_index
                dj = j0 - j_index
                mat[i, j] = np.sqrt(np.abs(di ** 2 + dj ** 2 + di * dj))
                lst.append(mat[i, j])
                s_i += di
                s_j += dj
    d_j = np.sqrt(np.abs(s_i ** 2 + s_j ** 2 + s_i * s_j))
    return mat * lattice_constant, np.array(lst) * lattice_constant, d_j * lattice_constant


def get_diffusion_coefficient(tags, size, time_diffusing, lattice_constant, conc):
    """get_diffusion_coefficient                       (hex_LiCoO2_kmc.py:589-603)."""
    n_atoms = size[0] * size[1] * (size[2] if len(size) > 2 else 1) * conc   # float (B6)
    if n_atoms == 0:
        raise SystemExit("concentration is zero, no atoms to move!")
    _, lst, d_j = get_distance(tags, size, lattice_constant)
    d_star = 1 / (2 * 2 * time_diffusing / n_atoms) * np.average(np.square(lst))
    d_jump = 1 / (2 * 2 * time_diffusing / n_atoms) * (d_j ** 2) / n_atoms
    return d_star, lst, d_jump


# --------------------------------------------------------------------------
# row E: one replica of the ensemble loop           (hex_LiCoO2_kmc.py:693-768)
# --------------------------------------------------------------------------
def run_replica(occ, table, conc, n_presim_steps, n_sim_steps, draws,
                window_mode='hex_ref', trace=None):
    """fill is done by the caller; this is :716-768 for one replica."""
    nx, ny = occ.shape
    n_hops = int(nx * ny * 1 * conc)                        # (:484)
    P = rates_full(occ, table)
    tags = np.zeros((nx, ny, 3), dtype=np.int64)
    tags[..., 0] = -1
    hist = np.zeros(6)
    for _ in range(n_presim_steps):
        kmc_step(occ, tags, P, table, n_hops, draws, True, window_mode, hist, trace)
    tags = init_tags(occ)
    total_time = 0
    barrier = None
    for _ in range(n_sim_steps):
        t, b = kmc_step(occ, tags, P, table, n_hops, draws, False, window_mode, hist, trace)
        total_time = total_time + t
        barrier = b if b is not None else barrier
    d_star, _, d_j = get_diffusion_coefficient(tags, [nx, ny, 1], total_time,
                                               LATTICE_CONSTANT, conc)
    return dict(occ=occ, P=P, tags=tags, hist=hist, total_time=total_time,
                d_star=d_star, d_j=d_j, barrier=barrier)


# --------------------------------------------------------------------------
# numpy's np.sum association, restated -- used to pin the
# C oracle and the device "exact gamma" kernel against np.sum itself.
# --------------------------------------------------------------------------
def _pairwise(a, lo, n):
    if n < 8:
        res = 0.0
        for k in range(lo, lo + n):
            res += a[k]
        return res
    if n <= 128:
        r = [a[lo + k] for k in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] += a[lo + i + k]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res += a[lo + i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return _pairwise(a, lo, n2) + _pairwise(a, lo + n2, n - n2)


def numpy_sum_model(a, chunk=8192):
    """np.sum of a contiguous float64 vector: pairwise inside each
    ``chunk``-element buffer, buffers accumulated left to right."""
    a = [float(v) for v in np.asarray(a, dtype=np.float64).ravel()]
    res = 0.0
    for lo in range(0, len(a), chunk):
        res = res + _pairwise(a, lo, min(chunk, len(a) - lo))
    return res
