"""kmc_diffusion_noalloc (include/kmc_hip.h): the entry point that allocates nothing.

Called through the C ABI on device buffers and compared with kmc_diffusion, the entry point it is a variant
of, on two calls in a row -- the second on a smaller ensemble, so that it finds the first call's sums in the
library's static scratch array.  torch is used only to own the device buffers the caller passes in.
"""
import os

import numpy as np
import pytest

import kmc_amd
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

pytestmark = pytest.mark.gpu


def _run(R, nx, conc, seed, base, steps):
    import torch
    dev = torch.device('cuda', 0)
    conc = np.array(np.broadcast_to(np.asarray(conc, dtype=np.float64), (R,)))
    with kmc_amd.KMCEngine(R, nx) as eng:
        eng.set_base_table(base, kmc_amd.tables.dumbbell_rate())
        eng.fill_random(conc, seed)
        eng.rates_full()
        eng.init_displacements()
        eng.run_counts((nx * nx * conc).astype(np.int64) * steps, presim=False, seed=seed)
        ds, dj, msd = eng.diffusion(conc)                       # the existing entry point (host buffers)
        conc_d = torch.from_numpy(conc).to(dev)
        # poisoned outputs: whatever is not written shows
        ds_d = torch.full((R,), -1.0, dtype=torch.float64, device=dev)
        dj_d = torch.full((R,), -1.0, dtype=torch.float64, device=dev)
        msd_d = torch.full((R, 4), -1.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()                                # the engine runs on a stream of its own
        eng.diffusion_noalloc(conc_d.data_ptr(), ds_d.data_ptr(), dj_d.data_ptr(), msd_d.data_ptr())
        eng.sync()
        a = (ds_d.cpu().numpy(), dj_d.cpu().numpy(), msd_d.cpu().numpy())
        # again without the optional output
        ds_d.fill_(-1.0)
        dj_d.fill_(-1.0)
        torch.cuda.synchronize()
        eng.diffusion_noalloc(conc_d.data_ptr(), ds_d.data_ptr(), dj_d.data_ptr(), None)
        eng.sync()
        b = (ds_d.cpu().numpy(), dj_d.cpu().numpy())
    return (ds, dj, msd), a, b


def test_no_alloc_diffusion_equals_kmc_diffusion():
    base = np.load(os.path.join(GOLDEN, 'tables.npz'))['base512_normcons']
    # first call: 300 replicas of 64 x 64 over a range of concentrations; second call: 37 replicas of 30 x 30
    for R, nx, conc, seed, steps in ((300, 64, np.linspace(0.1, 0.9, 300), 11, 3), (37, 30, 0.5, 12, 5)):
        (ds, dj, msd), (ds2, dj2, msd2), (ds3, dj3) = _run(R, nx, conc, seed, base, steps)
        assert np.all(ds > 0) and np.all(np.isfinite(dj))
        assert np.array_equal(msd, msd2)                        # integer sums: exact
        # the same fp64 expressions in the same order on the device as on the host: 1e-12 relative is the
        # bar for floating point on this path (BASELINE.json north_star); in practice the bits agree
        np.testing.assert_allclose(ds2, ds, rtol=1e-12, atol=0)
        np.testing.assert_allclose(dj2, dj, rtol=1e-12, atol=0)
        assert np.array_equal(ds3, ds2) and np.array_equal(dj3, dj2)


def test_no_alloc_rejects_more_replicas_than_the_scratch_holds():
    # 65537 replicas of 5 x 5: creating the context is cheap, the call must refuse
    import torch
    with kmc_amd.KMCEngine(65537, 5) as eng:
        x = torch.zeros(65537, dtype=torch.float64, device='cuda:0')
        with pytest.raises(kmc_amd.engine.KMCError, match='exceed'):
            eng.diffusion_noalloc(x.data_ptr(), x.data_ptr(), x.data_ptr())
