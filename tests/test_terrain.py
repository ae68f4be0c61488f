"""Terrain generation: the oracle and the product's host generator against the fixture produced by
the reference's own HumanoidTerrain class (tests/golden/make_terrain_fixtures.py), the vectorised product
primitives against the oracle's loop restatement, and the oracle's continuous height function."""
import os

import numpy as np
import pytest

from isaac_amd.envs import terrain as prod
from isaac_amd.envs.configs import HectorCfg
from oracle import terrain as orc

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "terrain_small.npz"))


def small_cfg(rows, cols, curriculum, mesh_type):
    class T(HectorCfg.terrain):
        pass
    T.num_rows, T.num_cols, T.curriculum, T.mesh_type, T.border_size = rows, cols, curriculum, mesh_type, 2.0
    return T


@pytest.mark.parametrize("impl", [orc.HumanoidTerrainOracle, prod.HumanoidTerrain])
def test_randomized_layout_matches_reference_class(impl):
    np.random.seed(5)
    t = impl(small_cfg(4, 7, False, "trimesh"), 64)
    assert t.height_field_raw.dtype == np.int16
    assert np.array_equal(t.height_field_raw, GOLD["rand_heights"])
    assert np.array_equal(t.env_origins, GOLD["rand_origins"])
    assert (t.tot_rows, t.tot_cols, t.border) == (360, 600, 20)


@pytest.mark.parametrize("impl", [orc.HumanoidTerrainOracle, prod.HumanoidTerrain])
def test_curriculum_layout_matches_reference_class(impl):
    np.random.seed(11)
    t = impl(small_cfg(3, 10, True, "heightfield"), 64)
    assert np.array_equal(t.height_field_raw, GOLD["cur_heights"])
    assert np.array_equal(t.env_origins, GOLD["cur_origins"])


def test_trimesh_matches_reference_class_output():
    np.random.seed(5)
    t = prod.HumanoidTerrain(small_cfg(4, 7, False, "trimesh"), 64)
    v, tri = t.vertices, t.triangles
    assert v.dtype == np.float32 and tri.dtype == np.uint32
    assert v.shape == (360 * 600, 3) and tri.shape == (2 * 359 * 599, 3)
    assert np.array_equal(v[::97], GOLD["rand_vertices_sample"])
    assert np.allclose(v.astype(np.float64).sum(0), GOLD["rand_vertex_sum"], rtol=1e-12)
    assert np.array_equal(tri[::1013], GOLD["rand_triangles_sample"])
    # walls: with the slope threshold some vertices moved by exactly one cell, none further
    flat = prod.convert_heightfield_to_trimesh(t.height_field_raw, 0.1, 0.005, None)[0]
    d = np.abs(v[:, :2] - flat[:, :2])
    assert d.max() == pytest.approx(0.1, abs=1e-5) and (d > 1e-5).any()


def _tile():
    return dict(width=80, length=80, vertical_scale=0.005, horizontal_scale=0.1)


@pytest.mark.parametrize("name,kw", [
    ("random_uniform_terrain", dict(min_height=-0.1, max_height=0.1, step=0.005, downsampled_scale=0.2)),
    ("random_uniform_terrain", dict(min_height=-0.03, max_height=0.07, step=0.01)),
    ("pyramid_sloped_terrain", dict(slope=0.3, platform_size=0.1)),
    ("pyramid_sloped_terrain", dict(slope=-0.4, platform_size=3.0)),
    ("discrete_obstacles_terrain", dict(max_height=0.15, min_size=1.0, max_size=2.0, num_rects=20, platform_size=3.0)),
    ("pyramid_stairs_terrain", dict(step_width=0.4, step_height=0.13, platform_size=1.0)),
    ("pyramid_stairs_terrain", dict(step_width=0.31, step_height=-0.2, platform_size=3.0)),
    ("stepping_stones_terrain", dict(stone_size=0.8, stone_distance=0.1, max_height=0.05, platform_size=2.0)),
])
def test_product_primitives_equal_oracle_primitives(name, kw):
    np.random.seed(3)
    a = getattr(orc, name)(orc.SubTerrain(**_tile()), **kw).height_field_raw
    s_a = np.random.get_state()[1][:8].copy()
    np.random.seed(3)
    b = getattr(prod, name)(prod.SubTerrain(**_tile()), **kw).height_field_raw
    s_b = np.random.get_state()[1][:8].copy()
    assert a.dtype == b.dtype == np.int16 and np.array_equal(a, b)
    assert np.array_equal(s_a, s_b), "same number of draws from numpy's global generator"
    assert np.abs(a).max() > 0


def test_trimesh_product_equals_oracle():
    np.random.seed(9)
    hf = prod.discrete_obstacles_terrain(prod.SubTerrain(**_tile()), 0.2, 1.0, 2.0, 20, 3.0).height_field_raw[:40, :33]
    for thr in (None, 0.75):
        va, ta = orc.heightfield_to_trimesh(hf, 0.1, 0.005, thr)
        vb, tb = prod.convert_heightfield_to_trimesh(hf, 0.1, 0.005, thr)
        assert np.array_equal(va, vb) and np.array_equal(ta, tb)


def test_full_size_default_terrain_shape_and_origins():
    np.random.seed(5)
    class T(HectorCfg.terrain):
        mesh_type = "trimesh"
    t = prod.HumanoidTerrain(T, 4096)
    assert t.height_field_raw.shape == (2100, 2100) and t.border == 250
    assert t.env_origins.shape == (20, 20, 3)
    assert np.allclose(t.env_origins[3, 7, :2], [28.0, 60.0])
    assert (t.height_field_raw[:250] == 0).all() and (t.height_field_raw[:, -250:] == 0).all()
    # origin height = highest sample of the 2 m x 2 m centre patch of the tile
    tile = t.height_field_raw[250 + 3 * 80:250 + 4 * 80, 250 + 7 * 80:250 + 8 * 80]
    assert t.env_origins[3, 7, 2] == pytest.approx(tile[30:50, 30:50].max() * 0.005)


def test_height_query_is_the_two_triangle_surface():
    rng = np.random.default_rng(0)
    raw = rng.integers(-40, 40, (12, 9)).astype(np.int16)
    hf = orc.HeightField(raw, 0.1, 0.005, border_size=0.3)
    # grid nodes reproduce the samples
    ii, jj = np.meshgrid(np.arange(12), np.arange(9), indexing="ij")
    z, _ = hf.query(ii.ravel() * 0.1 - 0.3, jj.ravel() * 0.1 - 0.3)
    assert np.allclose(z, raw.ravel() * 0.005, atol=1e-12)
    # inside a cell: barycentric interpolation of the triangle convert_heightfield_to_trimesh emits there
    verts, tris = orc.heightfield_to_trimesh(raw, 0.1, 0.005, None)
    for _ in range(200):
        c = rng.integers(0, len(tris))
        w = rng.dirichlet([1, 1, 1])
        p = (verts[tris[c]].astype(np.float64) * w[:, None]).sum(0)
        z, n = hf.query(np.array([p[0] - 0.3]), np.array([p[1] - 0.3]))
        assert z[0] == pytest.approx(p[2], abs=1e-6)
        a, b, cc = verts[tris[c]].astype(np.float64)
        nn = np.cross(b - a, cc - a)
        nn /= np.linalg.norm(nn)
        nn *= np.sign(nn[2])
        assert np.allclose(n[0], nn, atol=1e-5)
    # outside the grid the border continues flat
    z, n = hf.query(np.array([-5.0, 50.0]), np.array([0.0, 0.0]))
    assert np.isfinite(z).all() and np.allclose(np.linalg.norm(n, axis=1), 1.0)
