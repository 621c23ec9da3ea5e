"""The level-by-level restatement of the binned-SAH builder (csrc/sah_levels.h — the steps the device kernels of
csrc/sah_device.h run one workgroup per candidate, run here in plain loops) against the recursive host builder
(csrc/bvh_build.h): mi_bvh_build quality 0 must build THE tree of quality 1 — the same BvhNode records in the same breadth-first
numbering, the same triangle sets in the same leaf ranges, the same depth, and the BVH2 heights the 4-wide collapse would compute
itself. Stands where ShapeKDTree::build() stands in the reference (src/librender/scene_native.inl:3-10; its GPU mode builds on
the device, include/mitsuba/render/optix/shapes.h:72-167)."""
import numpy as np
import pytest


@pytest.mark.parametrize("level,max_leaf", [(0, 4), (1, 4), (2, 2), (3, 4), (4, 4), (4, 3)])
def test_same_nodes_as_the_recursive_builder(native, oracle, level, max_leaf):
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(64, 64, 4, diffuse_only=False, ball_level=level, device=-1)
    s = oracle.emu_sah_levels_check(scene.desc(), max_leaf)
    assert not s["need_host"]
    assert s["nodes_levels"] == s["nodes_recursive"] > 0 and s["node_diff"] == 0 and s["leaf_diff"] == 0
    assert s["depth_levels"] == s["depth_recursive"]


def test_fuzz_rooms(native, oracle):
    """the random rooms of tools/fuzz_cpu.py (meshes with shared vertices, degenerate triangles, analytic shapes' bounding triangles)"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_cpu
    from mitsuba2_amd import scenes
    seen = 0
    for seed in range(2000, 2012):
        try:
            scene, sensor, ikw, recipe, keep = fuzz_cpu.make_case(native, scenes, seed)
        except Exception:
            continue
        s = oracle.emu_sah_levels_check(scene.desc(), 4)
        if s["need_host"]:
            continue                    # (coincident centroids: the device hands such a scene to the host builder)
        assert s["node_diff"] == 0 and s["leaf_diff"] == 0 and s["depth_levels"] == s["depth_recursive"], (seed, s)
        seen += 1
    assert seen >= 6


def test_coincident_triangles_go_back_to_the_host_builder(native, oracle):
    """no axis separates the centroids: the recursion takes its median split, which the level sweep does not restate"""
    from mitsuba2_amd import api, scenes
    # (the two halves of a wall quad share their box centre: with one-triangle leaves they cannot stay together)
    scene, _ = scenes.cornell_box(64, 64, 4, diffuse_only=False, ball_level=2, device=-1)
    assert oracle.emu_sah_levels_check(scene.desc(), 1)["need_host"]
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    vs = np.concatenate([v] * 9); fs = np.arange(27, dtype=np.uint32).reshape(-1, 3)
    scene = api.Scene([api.Mesh("stack", vs, fs)]).build(-1)
    assert oracle.emu_sah_levels_check(scene.desc(), 4)["need_host"]
