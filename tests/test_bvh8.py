"""The 8-wide quantised tree of the phase machine (csrc/miw/bvh8.h, collapsed from the BVH2 by csrc/bvh8_build.h — the
dynamic programme of Ylitie et al. 2017 — with the triangles in the tree's own order).

Its contract is the BVH2's (csrc/miw/bvh.h): over all triangles that pass the exact triangle test inside [mint, maxt], the
closest hit, ties to the smaller primitive id; any-hit = some triangle passes — i.e. brute force (the scalar oracle's
orc_trace; reference behaviour: include/mitsuba/render/kdtree.h:2079-2171). The CPU tier walks the host collapse twice: by
the reference walk (bvh8_intersect) and by the two per-lane bodies k_path_phased<..., Wide = 2> itself runs (walk8_node_step /
walk8_tri_step — one definition for the kernel and the checker), every access to the lane's 16-entry stack column checked; the
GPU tier renders through the kernel and compares films with the oracle.
"""
import os

import numpy as np
import pytest

from conftest import has_gpu
from test_bvh4 import _rays, _same


@pytest.mark.parametrize("fan", [8, 6, 3, 2])
def test_bvh8_walk_equals_brute_force(native, oracle, fan):
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(32, 32, 1, diffuse_only=False, ball_level=3, device=-1)    # 2 x 1280-triangle balls + the box
    desc = scene.desc()
    o, d = _rays(desc, 2500, 3)
    for any_hit in (False, True):
        for mint, maxt in ((1e-4, np.inf), (0.0, 150.0)):
            brute = oracle.trace(desc, o, d, mint, maxt, any_hit=any_hit)
            for max_leaf in (1, 4):
                for schedule in (0, 1):
                    w = oracle.emu_trace8(desc, o, d, mint, maxt, any_hit=any_hit, max_leaf=max_leaf, max_fan=fan, schedule=schedule)
                    b = w["bvh8"]
                    assert b["ok"] == 1 and b["stack_seen"] <= b["depth"] <= 16
                    assert fan > 2 or b["nodes8"] == b["nodes2"]                  # fan-out 2 keeps the BVH2's topology
                    assert _same(brute, w, any_hit)
            assert np.isfinite(brute["t"]).sum() > 400


def test_bvh8_programme_beats_the_greedy_collapse_and_the_4_wide_walk(native, oracle):
    """The topology the dynamic programme picks (max_fan = 8) has fewer nodes and needs fewer node steps than the greedy
    "open the largest child" collapse (any fan-out cap below 8 takes that path; 7 is its widest), and clearly fewer steps than
    the 4-wide reference walk over the same rays — the reason the tree exists."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(32, 32, 1, diffuse_only=False, ball_level=4, device=-1)
    desc = scene.desc()
    o, d = _rays(desc, 4000, 7)
    dp = oracle.emu_trace8(desc, o, d, 1e-4, np.inf, compare4=True)["bvh8"]
    greedy = oracle.emu_trace8(desc, o, d, 1e-4, np.inf, max_fan=7)["bvh8"]
    assert dp["nodes8"] < 0.8 * greedy["nodes8"] and dp["node_steps"] < greedy["node_steps"]
    assert dp["node_steps"] < 0.75 * dp["node_steps4"]
    assert dp["nodes8"] < 0.25 * dp["nodes2"]


def test_bvh8_ties_and_degenerate_boxes(native, oracle):
    """Coplanar duplicates (equal t: the smaller primitive id wins) and axis-aligned geometry whose boxes have zero extent on
    one axis (plane spacing of a degenerate axis, rays inside the plane)."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.plugin_box(16, 16, 1, device=-1)
    desc = scene.desc()
    o, d = _rays(desc, 4000, 9)
    d_ = desc.contents
    v = np.ctypeslib.as_array(d_.vertex_positions, (d_.vertex_count * 3,)).reshape(-1, 3)
    o[:200, 1] = v[:, 1].min(); d[:200, 1] = 0.0                              # rays inside the floor plane
    d[:200] /= np.maximum(np.linalg.norm(d[:200], axis=1, keepdims=True), 1e-9)
    for any_hit in (False, True):
        brute = oracle.trace(desc, o, d, 0.0, np.inf, any_hit=any_hit)
        for max_leaf in (1, 2, 4):
            for schedule in (0, 1):
                assert _same(brute, oracle.emu_trace8(desc, o, d, 0.0, np.inf, any_hit=any_hit, max_leaf=max_leaf, schedule=schedule), any_hit)


def test_bvh8_speculating_bodies_under_random_schedules(native, oracle):
    """The speculating variant of the two bodies (MIW_W8_SPEC: a lane holding an untested triangle group keeps descending, the next
    node's triangles wait in a second group): whenever a lane could take either body a coin decides — brute force's answers
    under every schedule tried, no access outside the column."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(32, 32, 1, diffuse_only=False, ball_level=3, device=-1)
    desc = scene.desc()
    o, d = _rays(desc, 2500, 11)
    for any_hit in (False, True):
        for mint, maxt in ((1e-4, np.inf), (0.0, 150.0)):
            brute = oracle.trace(desc, o, d, mint, maxt, any_hit=any_hit)
            for max_leaf, schedule in ((4, 1), (4, 77), (1, 5), (2, 9)):
                w = oracle.emu_trace8(desc, o, d, mint, maxt, any_hit=any_hit, max_leaf=max_leaf, schedule=schedule, spec=True)
                assert w["bvh8"]["stack_seen"] <= w["bvh8"]["depth"] and _same(brute, w, any_hit)


def test_bvh8_refuses_leaves_it_cannot_name(native, oracle):
    """A leaf slot names its run in one byte (count <= 4): a BVH2 built with larger leaves is refused (the device then keeps
    the 4-wide tree)."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.plugin_box(16, 16, 1, device=-1)             # coplanar duplicates: the SAH sweep cannot separate them, they share a leaf
    desc = scene.desc()
    o, d = _rays(desc, 100, 5)
    assert oracle.emu_trace8(desc, o, d, 1e-4, np.inf, max_leaf=4)["bvh8"]["ok"] == 1
    with pytest.raises(RuntimeError):
        oracle.emu_trace8(desc, o, d, 1e-4, np.inf, max_leaf=8)


def test_bvh8_big_tree(native, oracle):
    """A 348 k-triangle tree (the interior scene at a coarser grid): depth inside the 16-entry column, every triangle in the
    tree's own order exactly once (the collapse checks the permutation's size; a wrong permutation would change the answers),
    brute force's answers from both walks."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.interior_scene(32, 32, 1, grid=96, device=-1, env_size=(32, 16))
    desc = scene.desc()
    o, d = _rays(desc, 200, 17)
    for any_hit in (False, True):
        brute = oracle.trace(desc, o, d, 1e-4, np.inf, any_hit=any_hit)
        for schedule in (0, 1):
            w = oracle.emu_trace8(desc, o, d, 1e-4, np.inf, any_hit=any_hit, schedule=schedule, compare4=(schedule == 0))
            b = w["bvh8"]
            assert b["ok"] == 1 and b["stack_seen"] <= b["depth"] <= 16
            assert _same(brute, w, any_hit)
        assert b["nodes8"] < 0.25 * b["nodes2"]
    assert np.isfinite(brute["t"]).sum() > 50


def test_emulated_render_walks_trees_with_the_8_wide_bodies(native, oracle, monkeypatch):
    """emu_render sends the queries of a tree scene through walk8_* over the permuted triangle / vertex-normal arrays (the hit's
    triangle index is a position in THAT order; shading reads the same arrays): the film is the scalar oracle's, and the same
    as with the 4-wide bodies (MIW_EMU_WALK=bvh4) and the BVH2 walk."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(24, 20, 4, diffuse_only=False, ball_level=2, device=-1)      # 2 x 320-triangle balls with shading normals
    job = native.PathIntegrator().render_job(sensor, n_threads=4)
    job.cfg.plan = 2
    o32, o64, ost = oracle.render(scene.desc(), job, threads=4)
    e64, e32, est = oracle.emu_render(scene.desc(), job)
    assert est[0] == ost.samples and est[1] == ost.segments
    assert np.array_equal(e32, o32)
    for walk in ("bvh4", "bvh2"):
        monkeypatch.setenv("MIW_EMU_WALK", walk)
        b64, b32, bst = oracle.emu_render(scene.desc(), job)
        assert np.array_equal(b32, e32) and list(bst)[:3] == list(est)[:3]


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("quality", [0, 1])
def test_phase_machine_over_bvh8_equals_oracle(native, oracle, quality):
    """The film of the material-ball scene rendered by k_path_phased over the 8-wide tree — collapsed on the device from the
    device-built SAH tree (quality 0) and on the host from the host-built one (quality 1) — is the oracle's bit for bit, and so
    is the film of the same context rendered over the 4-wide tree (debug_tree_width = 4 / option MIW_BVH8=0 at render time); the two builders agree on the
    8-wide tree's node count and depth (same programme, same BVH2)."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(48, 40, 8, diffuse_only=False, ball_level=3, device=-1)
    job = native.PathIntegrator().render_job(sensor, n_threads=8)
    o32, _, ost = oracle.render(scene.desc(), job, threads=os.cpu_count() or 8, want_f64=False)
    dev = native.Device(0)
    try:
        dev.upload(scene.desc(), bvh_quality=quality)
        b = dev.counters()
        assert b.bvh8_nodes > 0 and b.bvh8_depth <= 16 and b.bvh8_on_device == (1 if quality == 0 else 0)
        g, st = dev.render(job)
        c = dev.counters()
        assert st == 0 and c.path_kernel == 1 and c.tree_width == 8
        assert c.samples == ost.samples and c.segments == ost.segments
        assert np.array_equal(g, o32)
        g4, st = dev.render(job, tree_width=4)                  # mi_render_cfg::debug_tree_width (the library reads the environment in mi_create only)
        assert st == 0 and dev.counters().tree_width == 4 and np.array_equal(g4, o32)
        dev.set_option("MIW_BVH8", "0")                           # ... and the context's own switch does the same
        g4b, st = dev.render(job)
        dev.set_option("MIW_BVH8", None)
        assert st == 0 and dev.counters().tree_width == 4 and np.array_equal(g4b, o32)
        want = oracle.emu_trace8(scene.desc(), np.zeros((1, 3), np.float32), np.array([[0, 0, 1]], np.float32))["bvh8"]
        assert (b.bvh8_nodes, b.bvh8_depth) == (want["nodes8"], want["depth"])
    finally:
        dev.close()
