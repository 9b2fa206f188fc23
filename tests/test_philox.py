import numpy as np

from oracle import philox as PH


def test_known_answers():
    """Random123 kat_vectors, philox4x32 10 rounds."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for c, k, want in kat:
        got = PH.philox4x32_10(*[np.array([v]) for v in c], k[0], k[1])
        assert tuple(int(g[0]) for g in got) == want


def test_uniform_ranges():
    d = PH.hop_draws(1234, 3, 0, 10000)
    assert d[:, 0].min() >= 0 and d[:, 0].max() < 1
    assert d[:, 1].min() > 0 and d[:, 1].max() <= 1
    assert abs(d.mean() - 0.5) < 0.01


def test_fill_count():
    for conc in (0.0, 0.1, 0.5, 0.97, 1.0):
        occ = PH.fill_lattice(99, 5, 30, 30, conc)
        assert occ.sum() == int(30 * 30 * conc)
