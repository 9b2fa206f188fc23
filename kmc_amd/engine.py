"""KMCEngine -- numpy-facing wrapper of one C-ABI context (one GPU).

A context holds ``n_replicas`` independent nx x ny lattices (the size x
concentration x averaging-iteration loops of hex_LiCoO2_kmc.py:679-693) and
runs the per-hop loop (:484-535) for all of them on the device.
"""
import ctypes as C

import numpy as np

from . import _lib

WINDOW_MODES = {'hex_ref': 0, 'obj_ref': 1, 'full': 2}
VACANT_KEY = 0xFFFFFFFF
LATTICE_CONSTANT = 2.8334 * 10 ** (-8)        # hex_LiCoO2_kmc.py:29


class KMCError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('kmc error %d: %s' % (code, msg))
        self.code = code


class ReplicaStopped(KMCError):
    """A replica hit the reference's IndexError / zero-rate condition."""


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class KMCEngine:
    def __init__(self, n_replicas, nx, ny=None, window_mode='hex_ref', device=0, replica_offset=0):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        ny = nx if ny is None else ny
        mode = WINDOW_MODES[window_mode] if isinstance(window_mode, str) else int(window_mode)
        rc = self._lib.kmc_create(C.byref(self._ctx), int(device), int(n_replicas), int(nx), int(ny), mode)
        if rc != 0:
            msg = self._lib.kmc_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise KMCError(rc, msg)
        self.R, self.nx, self.ny, self.device = int(n_replicas), int(nx), int(ny), int(device)
        self.window_mode = window_mode
        self.replica_offset = 0
        if replica_offset:
            self.set_replica_offset(replica_offset)

    # -- plumbing ----------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            msg = self._lib.kmc_last_error(self._ctx).decode()
            raise (ReplicaStopped if rc == -4 else KMCError)(rc, msg)

    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx.value:
            self._lib.kmc_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._chk(self._lib.kmc_sync(self._ctx))

    def last_kernel_ms(self):
        ms = C.c_float()
        self._chk(self._lib.kmc_last_kernel_ms(self._ctx, C.byref(ms)))
        return ms.value

    def run_kernel_times(self, max_n=64):
        """ms of the last hop-kernel launches (HIP events on the engine's stream)."""
        buf = (C.c_float * max_n)()
        n = C.c_int()
        self._chk(self._lib.kmc_run_kernel_times(self._ctx, buf, max_n, C.byref(n)))
        return [buf[k] for k in range(n.value)]

    def run_span_ms(self, n_runs):
        """(ms from the first kernel start of the n_runs-th last run to the last kernel end, kernels per run)."""
        ms, g = C.c_float(), C.c_int()
        self._chk(self._lib.kmc_run_span_ms(self._ctx, int(n_runs), C.byref(ms), C.byref(g)))
        return ms.value, g.value

    def set_launch_groups(self, mode):
        """0 automatic, 1 one launch per run, 2 two concurrent launch groups (include/kmc_hip.h)."""
        self._chk(self._lib.kmc_set_launch_groups(self._ctx, int(mode)))

    @property
    def uses_lds_path(self):
        return bool(self._lib.kmc_uses_lds_path(self._ctx))

    @property
    def lds_bytes(self):
        return int(self._lib.kmc_lds_bytes(self._ctx))

    # -- tables --------------------------------------------------------------
    def set_base_table(self, base512, dumbbell):
        b = np.ascontiguousarray(base512, dtype=np.float64)
        if b.shape != (512,):
            raise ValueError('base table must have 512 entries')
        self._chk(self._lib.kmc_set_base_table(self._ctx, _ptr(b, C.c_double), float(dumbbell)))

    def set_hex_table(self, table):
        t = np.ascontiguousarray(table, dtype=np.float64)
        if t.shape != (1 << 18, 6):
            raise ValueError('dense table must be (262144, 6)')
        self._chk(self._lib.kmc_set_hex_table(self._ctx, _ptr(t, C.c_double)))

    def get_hex_table(self):
        t = np.empty((1 << 18, 6), dtype=np.float64)
        self._chk(self._lib.kmc_get_hex_table(self._ctx, _ptr(t, C.c_double)))
        return t

    def generate_base_table(self, eci47, prefactor, kb_t):
        e = np.ascontiguousarray(eci47, dtype=np.float64)
        if e.shape != (47,):
            raise ValueError('need 47 ECIs')
        base = np.empty(512)
        en = np.empty(512)
        self._chk(self._lib.kmc_generate_base_table(self._ctx, _ptr(e, C.c_double), float(prefactor),
                                                    float(kb_t), _ptr(base, C.c_double), _ptr(en, C.c_double)))
        return base, en

    # -- lattice -----------------------------------------------------------------
    def _shape(self, *tail):
        return (self.R, self.nx, self.ny) + tail

    def set_lattice(self, occ):
        o = np.ascontiguousarray(occ, dtype=np.int8).reshape(self._shape())
        self._chk(self._lib.kmc_set_lattice(self._ctx, _ptr(o, C.c_int8)))

    def get_lattice(self):
        o = np.empty(self._shape(), dtype=np.int8)
        self._chk(self._lib.kmc_get_lattice(self._ctx, _ptr(o, C.c_int8)))
        return o

    def set_replica_offset(self, offset):
        """This engine's replica 0 is replica ``offset`` of a larger ensemble: the Philox streams of fill_random
        and run are keyed by the global replica id, so the ensemble does not depend on how it is sharded."""
        self._chk(self._lib.kmc_set_replica_offset(self._ctx, int(offset)))
        self.replica_offset = int(offset)

    def fill_random(self, conc, seed):
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(conc, dtype=np.float64), (self.R,)))
        self._chk(self._lib.kmc_fill_random(self._ctx, _ptr(c, C.c_double), int(seed)))

    # -- rates -------------------------------------------------------------------
    def rates_full(self):
        self._chk(self._lib.kmc_rates_full(self._ctx))

    def get_keys(self):
        k = np.empty(self._shape(), dtype=np.uint32)
        self._chk(self._lib.kmc_get_keys(self._ctx, _ptr(k, C.c_uint32)))
        return k

    def set_keys(self, keys):
        k = np.ascontiguousarray(keys, dtype=np.uint32).reshape(self._shape())
        self._chk(self._lib.kmc_set_keys(self._ctx, _ptr(k, C.c_uint32)))

    def get_rates(self):
        p = np.empty(self._shape(6), dtype=np.float64)
        self._chk(self._lib.kmc_get_rates(self._ctx, _ptr(p, C.c_double)))
        return p

    def refresh_window(self, start_end):
        se = np.ascontiguousarray(start_end, dtype=np.int32).reshape(self.R, 4)
        self._chk(self._lib.kmc_refresh_window(self._ctx, _ptr(se, C.c_int32)))

    def site_probability(self, replica, i, j):
        key = C.c_uint32()
        r = np.empty(6)
        self._chk(self._lib.kmc_site_probability(self._ctx, int(replica), int(i), int(j),
                                                 C.byref(key), _ptr(r, C.c_double)))
        return key.value, r

    # -- tags --------------------------------------------------------------------
    def init_displacements(self):
        self._chk(self._lib.kmc_init_displacements(self._ctx))

    def set_displacements(self, tags):
        t = np.ascontiguousarray(tags, dtype=np.int32).reshape(self._shape(4))
        self._chk(self._lib.kmc_set_displacements(self._ctx, _ptr(t, C.c_int32)))

    def get_displacements(self):
        t = np.empty(self._shape(4), dtype=np.int32)
        self._chk(self._lib.kmc_get_displacements(self._ctx, _ptr(t, C.c_int32)))
        return t

    def get_reference_tags(self):
        """Tags in the reference's packing [i0, j0, 100*wi + wj] (:525-526)."""
        t = self.get_displacements().astype(np.int64)
        return np.stack([t[..., 0], t[..., 1], 100 * t[..., 2] + t[..., 3]], axis=-1)

    # -- hop loop ----------------------------------------------------------------
    def run(self, n_hops, presim=False, draws=None, seed=0, want_trace=False):
        """n_hops hops in every replica.  draws: (R, n_hops, 2) injected uniforms
        (rho, clock draw) or None for the Philox stream.  Returns the
        hop_master_index trace (R, n_hops) if want_trace."""
        d = None
        if draws is not None:
            d = np.ascontiguousarray(draws, dtype=np.float64)
            if d.shape != (self.R, n_hops, 2):
                raise ValueError('draws must be (R, n_hops, 2), got %r' % (d.shape,))
        sel = np.empty((self.R, n_hops), dtype=np.int32) if want_trace else None
        self._chk(self._lib.kmc_run(self._ctx, int(n_hops), int(bool(presim)), _ptr(d, C.c_double),
                                    int(seed), _ptr(sel, C.c_int32)))
        return sel

    def run_counts(self, n_hops, presim=False, seed=0):
        """Per-replica hop counts (Philox stream): concentration sweeps."""
        n = np.ascontiguousarray(np.broadcast_to(np.asarray(n_hops, dtype=np.int64), (self.R,)))
        self._chk(self._lib.kmc_run_counts(self._ctx, _ptr(n, C.c_int64), int(bool(presim)), int(seed)))

    def scalars(self, fields=('time', 'step_time', 'gamma', 'hist')):
        """Clock, last run's clock, total rate and direction histogram of every replica; `fields` limits the read-back
        (each is a copy of its own: the per-step drop-in calls ask for what they return)."""
        t = np.empty(self.R) if 'time' in fields else None
        st = np.empty(self.R) if 'step_time' in fields else None
        g = np.empty(self.R) if 'gamma' in fields else None
        h = np.empty((self.R, 6), dtype=np.int64) if 'hist' in fields else None
        self._chk(self._lib.kmc_get_scalars(self._ctx, _ptr(t, C.c_double), _ptr(st, C.c_double),
                                            _ptr(g, C.c_double), _ptr(h, C.c_int64)))
        return {k: v for k, v in dict(time=t, step_time=st, gamma=g, hist=h).items() if v is not None}

    def set_time(self, time):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(time, dtype=np.float64), (self.R,)))
        self._chk(self._lib.kmc_set_time(self._ctx, _ptr(t, C.c_double)))

    def set_counters(self, hops_done=None, hist=None):
        h = None if hops_done is None else np.ascontiguousarray(np.broadcast_to(np.asarray(hops_done, dtype=np.int64), (self.R,)))
        g = None if hist is None else np.ascontiguousarray(hist, dtype=np.int64).reshape(self.R, 6)
        self._chk(self._lib.kmc_set_counters(self._ctx, _ptr(h, C.c_int64), _ptr(g, C.c_int64)))

    # -- checkpoint / resume -------------------------------------------------------
    def checkpoint(self):
        """Everything a run continues from, as numpy arrays (np.savez(path, **eng.checkpoint()) is a checkpoint file):
        the arrays the reference's kmc_step takes and returns (hex_LiCoO2_kmc.py:470, :539) -- lattice, rate matrix
        (as the keys it is a function of), distance_lattice, total_time, hop_histogram -- and each replica's hop count,
        which is its position in the Philox stream -- and the replica offset, because the streams are keyed by the global
        replica id (set_replica_offset).  The chosen-barrier statistics are diagnostics and start afresh."""
        sc, st = self.scalars(), self.status()
        if st['status'].any():
            raise ReplicaStopped(-4, 'checkpoint: replicas have stopped (status %r)' % np.unique(st['status']))
        return dict(nx=np.int64(self.nx), ny=np.int64(self.ny), lattice=self.get_lattice(), keys=self.get_keys(),
                    displacements=self.get_displacements(), time=sc['time'], hist=sc['hist'], hops_done=st['hops_done'],
                    replica_offset=np.int64(self.replica_offset))

    def restore(self, ck):
        """Load a checkpoint() (or the np.load of one) into this engine: same replica count and lattice size;
        the rate table is set separately, as always."""
        if int(ck['nx']) != self.nx or int(ck['ny']) != self.ny or np.asarray(ck['lattice']).shape[0] != self.R:
            raise ValueError('checkpoint is for %d replicas of %d x %d' % (np.asarray(ck['lattice']).shape[0], int(ck['nx']), int(ck['ny'])))
        # the Philox streams are keyed by offset + local replica: a shard restored at another offset would silently
        # continue on other streams (checkpoints written before the field existed were all taken at offset 0)
        self.set_replica_offset(int(ck['replica_offset']) if 'replica_offset' in ck else 0)
        self.set_lattice(ck['lattice'])
        self.set_keys(ck['keys'])
        self.set_displacements(ck['displacements'])
        self.set_time(ck['time'])
        self.set_counters(ck['hops_done'], ck['hist'])

    def barrier_stats(self):
        b = np.empty((self.R, 5))
        self._chk(self._lib.kmc_get_barrier_stats(self._ctx, _ptr(b, C.c_double)))
        return dict(last=b[:, 0], sum=b[:, 1], min=b[:, 2], max=b[:, 3], count=b[:, 4])

    def status(self, fields=('status', 'hops_done', 'n_exact')):
        s = np.empty(self.R, dtype=np.int32) if 'status' in fields else None
        h = np.empty(self.R, dtype=np.int64) if 'hops_done' in fields else None
        e = np.empty(self.R, dtype=np.int64) if 'n_exact' in fields else None
        self._chk(self._lib.kmc_get_status(self._ctx, _ptr(s, C.c_int32), _ptr(h, C.c_int64), _ptr(e, C.c_int64)))
        return {k: v for k, v in dict(status=s, hops_done=h, n_exact=e).items() if v is not None}

    def gamma_exact(self):
        g = np.empty(self.R)
        self._chk(self._lib.kmc_gamma_exact(self._ctx, _ptr(g, C.c_double)))
        return g

    def table_is_compact(self):
        """True if the rate table factors into the compact per-direction table (include/kmc_hip.h)."""
        return bool(self._lib.kmc_table_is_compact(self._ctx))

    def set_hop_kernel(self, generic=False, which=None):
        """Choose the hop kernel (tests): generic=True / which=1 the any-size kernel with block sums in LDS; for
        64 x 64 replicas which=2 the 8-replicas-per-CU kernel (kmc_hop_lds64.h), which=3 the 16-per-CU one
        (kmc_hop_q64.h); which=4 the 16-per-CU layout for 8 ... 64 rows and columns other than 64 x 64; which=5 the
        one-replica-per-CU kernel (kmc_hop_lat.h), which=6 its two-per-CU layout at any replica count; which=0 the default (5 up to 512 replicas of 8 ... 64 rows and
        columns, else 3 at 64 x 64, 4 where it applies).  Lattices beyond 64 x 64: which=1 the global-memory kernel,
        which=2 the large-lattice kernel without its per-site rate words (the round-2 form)."""
        self._chk(self._lib.kmc_set_hop_kernel(self._ctx, int(which) if which is not None else int(bool(generic))))

    @property
    def hop_kernel_name(self):
        """The hop kernel the next run() launches."""
        return self._lib.kmc_hop_kernel_name(self._ctx).decode()

    def set_debug(self, force_exact=False, margin_scale=1.0):
        """force_exact: every hop through the exact replay of the reference's selection (tests).  2 = the same, but the
        large-lattice kernel may use the replay's short form (no np.sum pass), as it does for hops it cannot certify."""
        self._chk(self._lib.kmc_set_debug(self._ctx, 2 if force_exact == 2 else int(bool(force_exact)), float(margin_scale)))

    # -- observables ---------------------------------------------------------------
    def diffusion(self, conc, lattice_constant=LATTICE_CONSTANT):
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(conc, dtype=np.float64), (self.R,)))
        ds = np.empty(self.R)
        dj = np.empty(self.R)
        msd = np.empty((self.R, 4))
        self._chk(self._lib.kmc_diffusion(self._ctx, _ptr(c, C.c_double), float(lattice_constant),
                                          _ptr(ds, C.c_double), _ptr(dj, C.c_double), _ptr(msd, C.c_double)))
        return ds, dj, msd

    def diffusion_noalloc(self, conc_dev, dstar_dev, dj_dev, msd_dev=None, lattice_constant=LATTICE_CONSTANT):
        """kmc_diffusion_noalloc (include/kmc_hip.h): the arguments are DEVICE addresses (ints) on this engine's GPU;
        the call enqueues on the engine's stream and returns at once -- sync() before reading the results."""
        self._chk(self._lib.kmc_diffusion_noalloc(self._ctx, C.c_void_p(int(conc_dev)), float(lattice_constant),
                                                  C.c_void_p(int(dstar_dev)), C.c_void_p(int(dj_dev)),
                                                  C.c_void_p(int(msd_dev)) if msd_dev else None))

    # -- the one collective ----------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId (rank 0 calls this and hands them to the other ranks)."""
        buf = C.create_string_buffer(128)
        lib = _lib.load()
        rc = lib.kmc_comm_unique_id(buf)
        if rc != 0:
            raise KMCError(rc, lib.kmc_last_error(None).decode())
        return buf.raw

    def comm_create(self, unique_id, nranks, rank):
        """RCCL communicator of this engine's GPU (collective over the nranks callers); an opaque handle."""
        if len(unique_id) != 128:
            raise ValueError('unique id must be 128 bytes')
        comm = C.c_void_p()
        self._chk(self._lib.kmc_comm_create(self._ctx, C.c_char_p(bytes(unique_id)), int(nranks), int(rank), C.byref(comm)))
        return comm

    def comm_destroy(self, comm):
        self._chk(self._lib.kmc_comm_destroy(self._ctx, comm))

    def allreduce_stats(self, comm, acc):
        """Sum ``acc`` (float64 array) over the ranks of ``comm`` with one RCCL all-reduce on this engine's stream."""
        a = np.ascontiguousarray(acc, dtype=np.float64).copy()
        self._chk(self._lib.kmc_allreduce_stats(self._ctx, comm, _ptr(a, C.c_double), int(a.size)))
        return a

    def distances(self, lattice_constant=LATTICE_CONSTANT):
        """distances_travelled_matrix * a (hex_LiCoO2_kmc.py:577-578), (R, nx, ny)."""
        m = np.empty(self._shape(), dtype=np.float64)
        self._chk(self._lib.kmc_get_distances(self._ctx, float(lattice_constant), _ptr(m, C.c_double)))
        return m
