"""Host-side Q-scan geometry (gw_whisper_amd/qscan.py) against the independent CPU restatement oracle/qscan.py,
and the restatement's interpolation against torch's own bicubic.  PARITY UNPINNED w.r.t. ml4gw (see both headers)."""
import numpy as np
import torch

from gw_whisper_amd import qscan as gq
from oracle import qscan as oq


def test_tables_match_the_restatement():
    t = gq.QScanTables(1.0, 2048, (4, 128))
    planes = oq.tiling(1.0, 2048, (4, 128), 0.2)
    assert len(planes) == 5 and len(t.plane_rows) == 5          # qrange [4, 128], mismatch 0.2 -> 5 Q planes
    i = 0
    for p, plane in enumerate(planes):
        assert tuple(t.plane_rows[p]) == (i, len(plane))
        for tile in plane:
            r = t.rows[i]
            assert (r[0], r[1], r[2], r[3]) == (p, tile.ntiles, tile.windowsize, tile.indices[0])
            assert np.array_equal(tile.indices, r[3] + np.arange(r[2]))           # contiguous data indices
            np.testing.assert_allclose(t.window[r[5]:r[5] + r[2]], tile.window, rtol=1e-6)
            i += 1
    assert t.e_total == sum(tile.ntiles for plane in planes for tile in plane)
    nt = t.rows[t.order, 1]
    assert (np.diff(nt) >= 0).all()
    for (a, b), c in zip(t.class_ranges, (128, 256, 512, 1024, 2048)):
        assert (nt[a:b] == c).all()
    assert t.class_ranges[-1][1] == len(t.rows)


def test_rdft_matrix_is_the_forward_rfft_with_doubled_positive_frequencies():
    x = np.random.default_rng(0).standard_normal((3, 2048))
    X = np.fft.rfft(x, axis=-1) / 2048
    X[:, 1:] *= 2
    f = x @ gq.rdft_matrix(2048).astype(np.float64).T
    assert f.shape[1] % 4 == 0
    np.testing.assert_allclose(f[:, 0:2050:2], X.real, atol=1e-8)
    np.testing.assert_allclose(f[:, 1:2050:2], X.imag, atol=1e-8)
    assert np.all(f[:, 2050:] == 0)


def test_restatement_interpolation_is_torch_bicubic():
    import torch.nn.functional as F
    a = np.random.default_rng(1).standard_normal((2, 5, 37))
    ref = F.interpolate(torch.from_numpy(a)[None], (5, 128), mode="bicubic").numpy()[0]
    np.testing.assert_allclose(oq.cubic_resize_last(a, 128), ref, atol=1e-12)
    b = np.random.default_rng(2).standard_normal((2, 24, 128))
    ref2 = F.interpolate(torch.from_numpy(b)[None], (128, 128), mode="bicubic").numpy()[0]
    mine = np.swapaxes(oq.cubic_resize_last(np.swapaxes(b, -1, -2), 128), -1, -2)
    np.testing.assert_allclose(mine, ref2, atol=1e-12)


def test_tile_energy_needs_no_padding_or_shift():
    """The kernels drop the zero padding + ifftshift of the reference (a unit-modulus phase): same energies."""
    tile = oq.tiling()[1][7]
    rng = np.random.default_rng(3)
    X = rng.standard_normal((2, 1025)) + 1j * rng.standard_normal((2, 1025))
    ref = tile.energy(X, norm=False)
    c = X[:, tile.indices] * tile.window
    t = np.arange(tile.ntiles)
    k = np.arange(tile.windowsize)
    z = (c[:, None, :] * np.exp(2j * np.pi * k[None, None, :] * t[None, :, None] / tile.ntiles)).sum(-1) / tile.ntiles
    np.testing.assert_allclose(np.abs(z) ** 2, ref, rtol=1e-9)
