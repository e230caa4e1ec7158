"""Committed golden vectors (tests/golden/*.npz, made by make_golden.py).
CPU: the oracle reproduces them.  GPU: the HIP path matches them to 1e-5
relative (fp64, north_star tolerance)."""
import os

import numpy as np
import pytest

from tests import scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL64 = 1e-5


def params_from(vec, keys):
    d = {str(k): float(v) for k, v in zip(keys, vec)}
    d["step"], d["enable_dyn"] = int(d["step"]), int(d["enable_dyn"])
    return d


@pytest.fixture(scope="module")
def small():
    return np.load(os.path.join(GOLD, "small_maps.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def opti():
    return np.load(os.path.join(GOLD, "opti_node_scene.npz"), allow_pickle=False)


# ------------------------------------------------------------------ CPU

def test_oracle_reproduces_small_map_goldens(oracle_mod, small):
    sdf = oracle_mod.Sdf.from_map_size(small["origin"], float(small["resolution"]), small["map_size"])
    assert sdf.grid == tuple(small["grid"])
    sdf.build_from_occupancy(small["occupancy"])
    assert np.array_equal(sdf.dist.reshape(sdf.grid), small["dist"])          # ESDF is exact arithmetic
    for name in small["case_names"]:
        p = params_from(small[f"{name}_params"], small["pkeys"])
        c, g, _ = oracle_mod.eval_batch(small[f"{name}_T"], small[f"{name}_Df"], small[f"{name}_x"], sdf,
                                        oracle_mod.make_params(**p))
        rc, rg = scenes.rel_err(c, g, small[f"{name}_cost"], small[f"{name}_grad"])
        assert rc <= 1e-12 and rg <= 1e-12, (name, rc, rg)


def test_oracle_reproduces_opti_node_golden(oracle_mod, opti):
    sdf = oracle_mod.Sdf.from_map_size(scenes.OPTI_NODE_ORIGIN, scenes.OPTI_NODE_RES, scenes.OPTI_NODE_MAP_SIZE)
    sdf.build_from_points(scenes.opti_node_obstacles())
    d = sdf.dist.reshape(sdf.grid)
    assert np.allclose([d.sum(), (d * d).sum()], opti["dist_sum"], rtol=1e-13)
    assert np.array_equal(d[tuple(opti["probe_idx"].T)], opti["probe_val"])
    for name in opti["set_names"]:
        p = params_from(opti[f"params_{name}"], opti["pkeys"])
        c, g, _ = oracle_mod.eval_batch(opti["T"], opti["Df"], opti["x"], sdf, oracle_mod.make_params(**p))
        rc, rg = scenes.rel_err(c, g, opti[f"cost_{name}"], opti[f"grad_{name}"])
        assert rc <= 1e-12 and rg <= 1e-12, (name, rc, rg)


# ------------------------------------------------------------------ GPU

@pytest.mark.gpu
def test_hip_matches_small_map_goldens(gtop, small):
    ctx = gtop.GtopContext(device=0)
    # the committed distance field goes in through gtop_set_sdf ...
    ctx.set_sdf(small["dist"], small["grid"], small["origin"], float(small["resolution"]), map_size=small["map_size"])
    for name in small["case_names"]:
        p = params_from(small[f"{name}_params"], small["pkeys"])
        ctx.set_params(**p)
        ctx.set_problem(small[f"{name}_T"], small[f"{name}_Df"])
        c, g = ctx.eval_batch(small[f"{name}_x"])
        rc, rg = scenes.rel_err(c, g, small[f"{name}_cost"], small[f"{name}_grad"])
        assert rc <= TOL64 and rg <= TOL64, (name, rc, rg)
    # ... and the one built on the device from the committed occupancy is bit-identical to it
    ctx2 = gtop.GtopContext(device=0)
    ctx2.init_sdf_map(small["map_size"], small["origin"], float(small["resolution"]))
    idx = np.argwhere(small["occupancy"] == 1)
    ctx2.update_sdf_map((idx + 0.5) * float(small["resolution"]) + small["origin"])
    assert np.array_equal(ctx2.get_sdf(), small["dist"])


@pytest.mark.gpu
def test_hip_matches_opti_node_golden(gtop, opti):
    """The reference's own scene (src/opti_node.cpp:61-99) through initSDFMap /
    updateSDFMap / the callback, all on the device."""
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(scenes.OPTI_NODE_MAP_SIZE, scenes.OPTI_NODE_ORIGIN, scenes.OPTI_NODE_RES)
    assert tuple(ctx.grid) == (200, 200, 25)
    ctx.update_sdf_map(scenes.opti_node_obstacles())
    d = ctx.get_sdf()
    assert np.allclose([d.sum(), (d * d).sum()], opti["dist_sum"], rtol=1e-13)
    assert np.array_equal(d[tuple(opti["probe_idx"].T)], opti["probe_val"])
    for name in opti["set_names"]:
        ctx.set_params(**params_from(opti[f"params_{name}"], opti["pkeys"]))
        ctx.set_problem(opti["T"], opti["Df"])
        c, g = ctx.eval_batch(opti["x"])
        rc, rg = scenes.rel_err(c, g, opti[f"cost_{name}"], opti[f"grad_{name}"])
        assert rc <= TOL64 and rg <= TOL64, (name, rc, rg)
