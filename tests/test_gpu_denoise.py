"""GPU parity of PointCloud::denoise (mvr_cloud_denoise, csrc/mvr_denoise.hip) with the oracle: the kept points, their
ORDER (component by component, by smallest index; members ascending) and the number of components, on noisy surfaces,
degenerate inputs and at BASELINE's 200k points per scan."""
import numpy as np
import pytest

from test_oracle import noisy_surface

pytestmark = pytest.mark.gpu


def check(gpu, orc, pts, thr, r):
    gpu.upload(40, pts)
    keep, ncomp = gpu.denoise(40, thr, r)
    okeep, lab, oncomp = orc.denoise(pts, thr, r)
    assert ncomp == oncomp
    assert np.array_equal(keep, okeep)
    out = gpu.download(40)
    assert len(out) == len(okeep) and np.array_equal(out.view(np.uint32), np.ascontiguousarray(pts[okeep]).view(np.uint32))
    return keep


def test_denoise_noisy_surfaces(gpu, orc):
    if gpu.mode != "culled":
        pytest.skip("denoise does not depend on the NN kernel variant")
    rng = np.random.default_rng(41)
    for n, outliers, clusters, thr, r in ((20000, 300, (9, 10, 11, 40), 10, 2.5), (5000, 50, (3, 4), 4, 1.0), (3000, 0, (), 10, 2.5)):
        pts = noisy_surface(rng, n, outliers, clusters)
        keep = check(gpu, orc, pts, thr, r)
        assert 0 < len(keep) <= len(pts)
    # the cloud can be searched afterwards (new point set, new index)
    idx, d2 = gpu.nn(40, 40)
    assert np.array_equal(idx, np.arange(len(idx), dtype=np.uint32)) and np.all(d2 == 0)


def test_denoise_degenerate_inputs(gpu, orc):
    if gpu.mode != "culled":
        pytest.skip("denoise does not depend on the NN kernel variant")
    rng = np.random.default_rng(42)
    one = np.array([[1, 2, 900, 1]], np.float32)
    assert len(check(gpu, orc, one, 1, 2.5)) == 1 and len(check(gpu, orc, one, 2, 2.5)) == 0
    dup = np.tile(np.array([[5, 5, 900, 1]], np.float32), (37, 1))                     # exact duplicates: one component
    assert len(check(gpu, orc, dup, 10, 0.0)) == 37
    far = np.ones((500, 4), np.float32); far[:, :3] = rng.uniform(-1e4, 1e4, (500, 3))   # all isolated: everything is noise
    assert len(check(gpu, orc, far, 2, 2.5)) == 0 and len(check(gpu, orc, far, 1, 2.5)) == 500
    line = np.ones((400, 4), np.float32); line[:, 0] = np.arange(400) * 2.5; line[:, 1:3] = [0, 900]   # spacing == r exactly: linked (<=)
    assert len(check(gpu, orc, line, 400, 2.5)) == 400
    line[:, 0] = np.arange(400) * np.float32(2.5001)
    assert len(check(gpu, orc, line, 2, 2.5)) == 0
    blob = np.ones((3000, 4), np.float32); blob[:, :3] = rng.standard_normal((3000, 3)) * 3 + [0, 0, 900]
    assert len(check(gpu, orc, blob, 10, 1e6)) == 3000                                 # huge radius: one component
    gpu.upload(40, np.zeros((0, 4), np.float32))
    keep, ncomp = gpu.denoise(40, 10, 2.5)
    assert len(keep) == 0 and ncomp == 0


def test_denoise_full_size_scan(gpu, orc, mvr):
    """one 200k-point synthetic scan with 2 % outliers: the surface survives, the outliers go."""
    if gpu.mode != "culled":
        pytest.skip("denoise does not depend on the NN kernel variant")
    rng = np.random.default_rng(43)
    sp = mvr.synth_params(12, 2)
    scan = mvr.synth_view(sp, 3, 200000)
    noise = np.ones((4000, 4), np.float32); noise[:, :3] = rng.uniform(-150, 150, (4000, 3)) + np.array(sp.pivot)
    pts = np.concatenate([scan, noise])[rng.permutation(204000)]
    keep = check(gpu, orc, pts, 10, 2.5)
    assert 195000 < len(keep) <= 204000
