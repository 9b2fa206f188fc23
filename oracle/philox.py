"""numpy Philox4x32-10 (Random123) -- TEST INFRASTRUCTURE ONLY.

Checker for the engine's counter-based streams (kmc_device.h): per-hop draws
and the random fill.  Known-answer vectors are Random123's kat_vectors."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
STREAM_HOP, STREAM_FILL = 0x4B4D4300, 0x4B4D4301
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments broadcastable uint32 arrays; returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & MASK for v in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def u53(a, b):
    return ((a.astype(np.uint64) >> np.uint64(5)) * np.uint64(1 << 26) +
            (b.astype(np.uint64) >> np.uint64(6))).astype(np.float64) * (1.0 / 9007199254740992.0)


def u53_open0(a, b):
    return (((a.astype(np.uint64) >> np.uint64(5)) * np.uint64(1 << 26) +
             (b.astype(np.uint64) >> np.uint64(6))) + np.uint64(1)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hop_draws(seed, replica, hop0, n_hops):
    """(n_hops, 2) = rho, clock draw of hops hop0 .. hop0+n_hops-1 of one replica."""
    h = np.arange(hop0, hop0 + n_hops, dtype=np.uint64)
    o = philox4x32_10(h & MASK, h >> np.uint64(32), replica, STREAM_HOP, seed & 0xFFFFFFFF, seed >> 32)
    return np.stack([u53(o[0], o[1]), u53_open0(o[2], o[3])], axis=1)


def fill_lattice(seed, replica, nx, ny, conc):
    """The engine's random fill: int(nx*ny*conc) sites with the smallest 64-bit
    Philox keys (ties by site index) hold Li."""
    n = nx * ny
    want = int(nx * ny * 1 * conc)
    s = np.arange(n, dtype=np.uint64)
    o = philox4x32_10(s, 0, replica, STREAM_FILL, seed & 0xFFFFFFFF, seed >> 32)
    key = (o[0].astype(np.uint64) << np.uint64(32)) | o[1].astype(np.uint64)
    order = np.lexsort((np.arange(n), key))
    occ = np.zeros(n, dtype=np.uint8)
    occ[order[:want]] = 1
    return occ.reshape(nx, ny)
