"""Explicit light sampling (rt_params.light_mode 1) has no counterpart in the reference: what is checked on the
CPU is that the oracle's restatement of it is the same integral as the reference's integrator (light_mode 0) --
the two agree within the statistical error of a few hundred thousand unclamped paths -- so that the GPU tests,
which compare the HIP path with this oracle, compare it with something meaningful.  Parity of the mode itself
is unpinned by construction."""
import numpy as np
import pytest

from raytrace_amd import scenes


def camera_rays(po, scene, n, seed):
    rs = np.random.RandomState(seed)
    cam = po.make_camera(scene["camera"]["o"], scene["camera"]["d"], 48, 64)
    camd, cx, cy = po.camera_basis(cam)
    u, v = rs.uniform(-.5, .5, (2, n))
    d = cx[None] * u[:, None] + cy[None] * v[:, None] + camd[None]
    d /= np.linalg.norm(d, axis=1)[:, None]
    o = np.asarray(scene["camera"]["o"], np.float64)[None] + 140 * d
    return np.concatenate([o, d], 1)


@pytest.mark.parametrize("dome", [False, True])
def test_oracle_light_sampling_is_the_same_integral(po, dome):
    sc = scenes.lamp_room(dome=dome)
    orc = po.OracleScene(sc)
    n = 600000
    rays = camera_rays(po, sc, n, 5)
    kw = dict(samps=1, rr_depth=4, rr_mode=1, split_depth=2, hard_depth=32, jitter=1)
    a, st_a = orc.radiance(po.make_params(seed=1, light_mode=0, **kw), rays)
    b, st_b = orc.radiance(po.make_params(seed=2, light_mode=1, **kw), rays)
    ma, mb = a.mean(0), b.mean(0)
    se = np.sqrt(a.var(0) / n + b.var(0) / n)
    assert np.all(np.abs(ma - mb) <= 4.5 * se + 1e-12), (ma, mb, se)
    assert np.all(ma > 0.05)
    # ... and it is worth having: far less noise per path for the same image
    assert b.var(0).sum() < 0.5 * a.var(0).sum()
    assert st_b["rays"] > st_a["rays"]  # the shadow rays are rays like any other in the counters


def test_oracle_light_mode_without_emissive_spheres_changes_nothing(po):
    sc = scenes.lamp_room(dome=False)
    for s in sc["spheres"]:
        s["lig"] = np.zeros(3)
    sc["meshes"][1]["lig"] = np.array([3.0, 2.0, 1.0])  # an emissive mesh: never sampled, always adds
    orc = po.OracleScene(sc)
    rays = camera_rays(po, sc, 3000, 7)
    kw = dict(samps=1, rr_depth=4, rr_mode=1, split_depth=2, hard_depth=32, jitter=1, seed=3)
    a, st_a = orc.radiance(po.make_params(light_mode=0, **kw), rays)
    b, st_b = orc.radiance(po.make_params(light_mode=1, **kw), rays)
    assert np.array_equal(a, b) and st_a == st_b
    assert a.max() > 0
