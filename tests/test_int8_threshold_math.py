"""The integer threshold algebra of the int8 tiles (prefhetch_amd/csrc/pf_flat.hip: tile16_walk<..., I8>), restated in numpy and checked as a property:
with x' = x - 128, y' = y - 128, S = sum x'y', the accumulators start at r0 + c0 and a distance may pass only where S + r0 + c0 >= 0.
That set must contain every pair with dist < tau (conservative) and nothing beyond dist <= tau + 1 (tight).  No GPU involved."""
import numpy as np
import pytest


@pytest.mark.parametrize("d", [32, 64, 96, 128])
def test_int8_threshold_init_is_conservative_and_tight(d):
    rng = np.random.default_rng(d)
    nq, nb = 64, 4000
    x = rng.integers(0, 256, (nq, d)).astype(np.int64)
    y = rng.integers(0, 256, (nb, d)).astype(np.int64)
    x[0] = 0; x[1] = 255; y[:10] = 255; y[10:20] = 0                       # extremes of value - 128
    dist = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)                  # exact integers
    xp, yp = x - 128, y - 128
    S = xp @ yp.T
    assert (x @ y.T == S + 128 * (xp.sum(1)[:, None] + yp.sum(1)[None, :]) + 16384 * d).all()      # what the unfiltered launch adds back
    for tau_kind in ("typical", "tight", "zero", "fractional"):
        if tau_kind == "typical":
            tau = np.sort(dist, axis=1)[:, 200]                            # a k-th distance: ties with it exist
        elif tau_kind == "tight":
            tau = np.sort(dist, axis=1)[:, 1] + 1
        elif tau_kind == "fractional":
            # a threshold that is not an integer (no caller produces one today: every distance on this path is an integer below 2^24;
            # the device rounds UP -- (int)ceilf(tau) -- so that the filter stays a superset whatever tau it is handed)
            tau_f = np.sort(dist, axis=1)[:, 200].astype(np.float64) + 0.5
            assert ((dist <= tau_f[:, None]) == (dist <= np.ceil(tau_f)[:, None])).all()        # integers: ceil keeps the set
            assert ((dist <= tau_f[:, None]) != (dist <= np.trunc(tau_f)[:, None] - 1)).any()
            tau = np.ceil(tau_f).astype(np.int64)
        else:
            tau = np.zeros(nq, np.int64)
        R = (x * x).sum(1) - tau - 256 * xp.sum(1) - 32768 * d             # row half
        C = (y * y).sum(1) - 256 * yp.sum(1)                               # column half
        r0 = -(R >> 1) - 1                                                 # (>> of a negative integer: floor, as on the device)
        c0 = -(C >> 1)
        passes = (S + r0[:, None] + c0[None, :]) >= 0
        must = dist < tau[:, None]
        assert (passes | ~must).all()                                      # nothing below the threshold is lost
        assert (~passes | (dist <= tau[:, None] + 1)).all()                # and at most the boundary value comes along
        assert np.abs(r0).max() < 2 ** 30 and np.abs(c0).max() < 2 ** 30 and np.abs(S).max() < 2 ** 22
