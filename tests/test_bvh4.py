"""The 4-wide quantised tree of the phase machine (csrc/miw/bvh4.h, collapsed from the BVH2 by csrc/bvh4_build.h).

Its contract is the BVH2's (csrc/miw/bvh.h): over all triangles that pass the exact triangle test inside [mint, maxt], the
closest hit, ties to the smaller primitive id; any-hit = some triangle passes — i.e. brute force (the scalar oracle's
orc_trace). The CPU tier walks the host collapse twice: by the reference walk (bvh4_intersect) and by the two per-lane bodies
k_path_phased itself runs (walk4_node_step / walk4_tri_step of csrc/miw/bvh4.h — one definition for the kernel and the checker)
under pseudo-random body schedules, with every stack access checked against the lane's column; the GPU tier renders through
the kernel and compares films with the oracle.
"""
import os

import numpy as np
import pytest

from conftest import has_gpu


def _rays(desc, n, seed):
    d_ = desc.contents
    v = np.ctypeslib.as_array(d_.vertex_positions, (d_.vertex_count * 3,)).reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    g = np.random.default_rng(seed)
    o = (lo - 0.1 * (hi - lo) + 1.2 * (hi - lo) * g.random((n, 3))).astype(np.float32)
    d = g.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n // 10                                                   # axis-parallel rays: zero components in the slab test
    d[:k] = np.eye(3)[g.integers(0, 3, k)] * np.sign(g.normal(size=(k, 1)))
    return o, d.astype(np.float32)


def _same(a, b, any_hit):
    ok = np.array_equal(np.asarray(a["t"]).view(np.uint32), np.asarray(b["t"]).view(np.uint32))
    return ok and (any_hit or np.array_equal(a["prim"], b["prim"]))


@pytest.mark.parametrize("fan,budget", [(4, 32), (3, 32), (2, 32), (4, 20)])
def test_bvh4_walk_equals_brute_force(native, oracle, fan, budget):
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(32, 32, 1, diffuse_only=False, ball_level=3, device=-1)    # 2 x 1280-triangle balls + the box
    desc = scene.desc()
    o, d = _rays(desc, 6000, 3)
    for any_hit in (False, True):
        for mint, maxt in ((1e-4, np.inf), (0.0, 150.0)):
            brute = oracle.trace(desc, o, d, mint, maxt, any_hit=any_hit)
            for max_leaf in (1, 4):
                w = oracle.emu_trace4(desc, o, d, mint, maxt, any_hit=any_hit, max_leaf=max_leaf, stack_budget=budget, max_fan=fan)
                b = w["bvh4"]
                assert b["ok"] == 1 and b["stack_seen"] <= b["stack_bound"] <= budget
                assert fan > 2 or b["nodes4"] == b["nodes2"]                      # fan-out 2 keeps the BVH2's topology
                assert _same(brute, w, any_hit)
            assert np.isfinite(brute["t"]).sum() > 1000


def test_bvh4_collapse_respects_the_stack_budget(native, oracle):
    """A budget below the BVH2's height cannot be met by any fan-out: the collapse refuses (the device then keeps the BVH2
    kernels); a budget that just fits degrades the fan-out instead of overflowing the per-lane LDS stack."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(32, 32, 1, diffuse_only=False, ball_level=3, device=-1)
    desc = scene.desc()
    o, d = _rays(desc, 2000, 5)
    full = oracle.emu_trace4(desc, o, d, 1e-4, np.inf)["bvh4"]
    two = oracle.emu_trace4(desc, o, d, 1e-4, np.inf, max_fan=2)["bvh4"]
    height = two["stack_bound"]                                               # fan-out 2: the BVH2's own worst case
    assert full["nodes4"] < 0.6 * full["nodes2"] and full["stack_bound"] > height
    tight = oracle.emu_trace4(desc, o, d, 1e-4, np.inf, stack_budget=height)
    assert tight["bvh4"]["ok"] == 1 and tight["bvh4"]["stack_bound"] <= height
    assert _same(oracle.trace(desc, o, d, 1e-4, np.inf), tight, False)
    with pytest.raises(RuntimeError):
        oracle.emu_trace4(desc, o, d, 1e-4, np.inf, stack_budget=height - 1)


def test_bvh4_ties_and_degenerate_boxes(native, oracle):
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
            assert _same(brute, oracle.emu_trace4(desc, o, d, 0.0, np.inf, any_hit=any_hit, max_leaf=max_leaf), any_hit)


def test_bvh4_big_tree_with_a_binding_stack_budget(native, oracle):
    """A 348 k-triangle tree (the interior scene at a coarser grid) whose full collapse would need more stack than the device
    has: the budget the device uses (MIW_STACK_ENTRIES - 1 = 31) binds, the fan-out degrades locally, the walk never goes
    deeper than the bound, and the answers are brute force's."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.interior_scene(32, 32, 1, grid=96, device=-1, env_size=(32, 16))
    desc = scene.desc()
    o, d = _rays(desc, 400, 17)
    loose = oracle.emu_trace4(desc, o, d, 1e-4, np.inf, stack_budget=64)["bvh4"]
    for any_hit in (False, True):
        brute = oracle.trace(desc, o, d, 1e-4, np.inf, any_hit=any_hit)
        w = oracle.emu_trace4(desc, o, d, 1e-4, np.inf, any_hit=any_hit, stack_budget=31)
        b = w["bvh4"]
        assert b["ok"] == 1 and b["stack_seen"] <= b["stack_bound"] <= 31 < loose["stack_bound"]
        assert b["nodes4"] > loose["nodes4"]                          # the budget cost some fan-out ...
        assert b["nodes4"] < 0.62 * b["nodes2"]                       # ... but the tree is still about half the BVH2
        assert _same(brute, w, any_hit)
    assert np.isfinite(brute["t"]).sum() > 100


@pytest.mark.parametrize("spec", [True, False])
def test_phase_machine_bodies_under_random_schedules(native, oracle, spec):
    """The node step and the triangle-pair step of the device's phase machine, run per ray under several pseudo-random
    schedules (whenever a lane could take either body, a coin decides — every interleaving the wave votes can produce is
    reachable), both the speculating variant (the device default: a lane holding an untested leaf range keeps descending) and
    the plain one: brute force's answers, no access outside the lane's stack column (budget + 1 slots: the unconditional
    stores of the node step need the one slot of slack the collapse leaves), never deeper than the collapse's bound + 1."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(32, 32, 1, diffuse_only=False, ball_level=3, device=-1)
    desc = scene.desc()
    o, d = _rays(desc, 5000, 11)
    for any_hit in (False, True):
        for mint, maxt in ((1e-4, np.inf), (0.0, 150.0)):
            brute = oracle.trace(desc, o, d, mint, maxt, any_hit=any_hit)
            for max_leaf, fan, budget, schedule in ((4, 4, 31, 1), (4, 4, 31, 77), (1, 4, 31, 5), (2, 3, 31, 9), (4, 4, 20, 3), (4, 2, 31, 21)):
                w = oracle.emu_trace4(desc, o, d, mint, maxt, any_hit=any_hit, max_leaf=max_leaf, stack_budget=budget, max_fan=fan,
                                      schedule=schedule, spec=spec)       # raises if a slot outside the column was touched
                b = w["bvh4"]
                assert b["ok"] == 1 and b["stack_bound"] <= budget and b["stack_seen"] <= b["stack_bound"] + 1
                assert _same(brute, w, any_hit)
    # the scenes of the degenerate-box / tie test, and a tree whose collapse the device's budget binds
    scene, _ = scenes.plugin_box(16, 16, 1, device=-1)
    desc = scene.desc()
    o, d = _rays(desc, 3000, 9)
    for any_hit in (False, True):
        brute = oracle.trace(desc, o, d, 0.0, np.inf, any_hit=any_hit)
        for max_leaf in (1, 2, 4):
            assert _same(brute, oracle.emu_trace4(desc, o, d, 0.0, np.inf, any_hit=any_hit, max_leaf=max_leaf, stack_budget=31, schedule=13, spec=spec), any_hit)
    scene, _ = scenes.interior_scene(32, 32, 1, grid=96, device=-1, env_size=(32, 16))
    desc = scene.desc()
    o, d = _rays(desc, 300, 17)
    for any_hit in (False, True):
        brute = oracle.trace(desc, o, d, 1e-4, np.inf, any_hit=any_hit)
        w = oracle.emu_trace4(desc, o, d, 1e-4, np.inf, any_hit=any_hit, stack_budget=31, schedule=2, spec=spec)
        assert w["bvh4"]["stack_seen"] <= 32 and _same(brute, w, any_hit)


def test_emulated_render_walks_trees_with_the_phase_machine_bodies(native, oracle, monkeypatch):
    """emu_render sends the queries of a tree scene through those bodies (an E walk, then the S walk that must leave the hit
    record alone): the film is the scalar oracle's, and the same as with the BVH2 walk (MIW_EMU_WALK=bvh2)."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(24, 20, 4, diffuse_only=False, ball_level=2, device=-1)      # 2 x 320-triangle balls: a tree scene
    job = native.PathIntegrator().render_job(sensor, n_threads=4)
    job.cfg.plan = 2
    o32, o64, ost = oracle.render(scene.desc(), job, threads=4)
    e64, e32, est = oracle.emu_render(scene.desc(), job)
    assert est[0] == ost.samples and est[1] == ost.segments
    assert np.array_equal(e32, o32)
    monkeypatch.setenv("MIW_EMU_WALK", "bvh2")
    b64, b32, bst = oracle.emu_render(scene.desc(), job)
    assert np.array_equal(b32, e32) and list(bst)[:3] == list(est)[:3]


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("fan", ["4", "3", "2", "off"])
def test_phase_machine_over_bvh4_equals_oracle(native, oracle, fan):
    """The film of the material-ball scene rendered by k_path_phased over the collapsed tree, for every fan-out the
    collapse can fall back to (MIW_BVH4_FAN, read per mi_bvh_build) and over the BVH2 (MIW_BVH4=0, the A/B switch), is
    the oracle's bit for bit."""
    from mitsuba2_amd import scenes
    old = os.environ.get("MIW_BVH4_FAN"), os.environ.get("MIW_BVH4")
    if fan == "off":
        os.environ["MIW_BVH4"] = "0"
    else:
        os.environ["MIW_BVH4_FAN"] = fan
    try:
        scene, sensor = scenes.cornell_box(48, 40, 8, diffuse_only=False, ball_level=3, device=-1)
        job = native.PathIntegrator().render_job(sensor, n_threads=8)
        o32, _, ost = oracle.render(scene.desc(), job, threads=os.cpu_count() or 8, want_f64=False)
        dev = native.Device(0)
        dev.upload(scene.desc(), bvh_quality=1)
        g, st = dev.render(job)
        c = dev.counters()
        dev.close()
        assert st == 0 and c.path_kernel == (3 if fan == "off" else 1)
        assert c.samples == ost.samples and c.segments == ost.segments
        assert np.array_equal(g, o32)
    finally:
        for k, v in zip(("MIW_BVH4_FAN", "MIW_BVH4"), old):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
