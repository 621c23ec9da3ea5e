"""GPU parity tests proper: the HIP path (through the C ABI / host classes) against
the CPU oracle on the same seeded inputs. Bars:
  * integer / index work (hit primitive ids, any-hit flags, sample counts): bit-exact
  * float32 leaf functions evaluated on the device (mi_eval): bit-exact vs the oracle
  * hit records (t, u, v): bit-exact (same Moeller-Trumbore arithmetic)
  * film, default mode (sample log + ordered gather): BIT-IDENTICAL to the oracle's
    reference-semantics float32 film (same float32 additions in the same order).
  * film, float64-atomics mode: equals the oracle's exact-sum film after rounding
    to float32, and is within 1e-5 relative L2 of the float32 film (the tolerance
    BASELINE.json's north_star states).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_L2_TOL = 1e-5     # north_star: image L2 vs scalar_rgb < 1e-5


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.fixture(scope="module")
def dev(native):
    d = native.Device(0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def cbox(native):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(128, 96, 16, device=-1)
    return scene, sensor


def _rays_for(sensor, w, h, n, seed=1):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    rays = np.array([sensor.sample_ray(x, y) for x, y in xy])
    return rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7]


def test_rcp_exhaustive(native, dev):
    """miw::rcp on the device (v_rcp_f32 + one Newton step inside [2^-126, 2^126), IEEE division elsewhere)
    must be the correctly rounded 1/x for all 2^32 float bit patterns (base.h)."""
    assert dev.selftest(0) == 0


def test_fp_semantics_match_host(native, oracle, dev):
    """+,*,/,sqrt,fma,rcp,min,max incl. denormal / inf / signed-zero inputs: identical bits."""
    rng = np.random.default_rng(3)
    specials = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-38, -1e-38, 1e-40, 3e-39, 1.17549435e-38,
                         3.4e38, 1e-20, 1e20, 0.5, 1.5, 2.0 ** -126, 2.0 ** -127, 2.0 ** -149], np.float32)
    a = np.concatenate([rng.normal(0, 1, 4096), rng.choice(specials, 4096), np.exp(rng.uniform(-88, 88, 4096))]).astype(np.float32)
    b = np.concatenate([rng.normal(0, 1, 4096), rng.choice(specials, 4096), np.exp(rng.uniform(-88, 88, 4096))]).astype(np.float32)
    c = np.concatenate([rng.normal(0, 1, 4096), rng.choice(specials, 4096), rng.normal(0, 1e-30, 4096)]).astype(np.float32)
    rng.shuffle(b); rng.shuffle(c)
    x = np.stack([a, b, c], 1)
    g = dev.eval(7, x); o = oracle.eval(7, x)
    gv, ov = g.view(np.uint32), o.view(np.uint32)
    nan_both = np.isnan(g) & np.isnan(o)
    bad = (gv != ov) & ~nan_both
    assert bad.sum() == 0, "first mismatches: %s" % [(x[i].tolist(), g[i, j], o[i, j]) for i, j in np.argwhere(bad)[:8]]


@pytest.mark.parametrize("op,gen", [
    (0, lambda r: r.integers(0, 2 ** 32, (4096, 2), dtype=np.uint64).astype(np.uint32).view(np.float32)),   # PCG32
    (1, lambda r: r.uniform(-7, 7, (8192, 1))),                                                             # sincos
    (2, lambda r: r.uniform(0, 1, (8192, 2))),                                                              # cosine hemisphere
    (4, lambda r: np.stack([r.uniform(-1, 1, 8192), r.choice([1.5, 1.0 / 1.5, 1.5046 / 1.000277, 1.0], 8192)], 1)),  # fresnel
    (8, lambda r: np.concatenate([r.uniform(-1, 1, (8192, 1)), r.uniform(-100, 100, (4096, 1)),               # exp/log/erf/erfinv
                                  10.0 ** r.uniform(-44, 38, (4096, 1)), [[0.0], [1.0], [-1.0], [88.8], [-104.0], [np.inf]]])),
    (10, lambda r: np.concatenate([r.uniform(-1, 1, (8192, 2)), r.uniform(-50, 50, (2048, 2)),               # atan2 / acos / asin
                                   [[0.0, -1.0], [-0.0, -1.0], [0.0, 1.0], [1.0, 0.0], [-1.0, 0.0], [0.0, 0.0]]])),
])
def test_leaf_functions_bit_exact(native, oracle, dev, op, gen):
    x = np.ascontiguousarray(gen(np.random.default_rng(op + 10)), np.float32)
    g = dev.eval(op, x); o = oracle.eval(op, x)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))


def test_bsdf_and_emitter_bit_exact(native, oracle, dev):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(64, 64, 4, diffuse_only=False, device=-1, ball_level=1)
    dev.upload(scene.desc())
    rng = np.random.default_rng(5)
    n = 4096
    nb = scene.desc().contents.bsdf_count
    wi = rng.normal(0, 1, (n, 3)); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wo = rng.normal(0, 1, (n, 3)); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    idx = rng.integers(0, nb, n).astype(np.uint32).view(np.float32)
    x = np.concatenate([idx[:, None], wi, rng.uniform(0, 1, (n, 3)), wo], 1).astype(np.float32)
    x[:, 0] = idx
    g = dev.eval(3, x); o = oracle.eval(3, x, desc=scene.desc())
    same = (g.view(np.uint32) == o.view(np.uint32)) | (np.isnan(g) & np.isnan(o))
    assert same.all(), np.argwhere(~same)[:5]
    ref = np.concatenate([rng.uniform(0, 555, (n, 3)), rng.uniform(0, 1, (n, 2))], 1).astype(np.float32)
    g = dev.eval(6, ref); o = oracle.eval(6, ref, desc=scene.desc())
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))


def test_camera_rays_bit_exact(native, oracle, dev, cbox):
    scene, sensor = cbox
    job = native.PathIntegrator().render_job(sensor)
    xy = np.random.default_rng(2).uniform(0, [128, 96], (4096, 2)).astype(np.float32)
    g = dev.eval(5, xy, cfg=job.cfg); o = oracle.eval(5, xy, cfg=job.cfg)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))


def test_trace_cornell_matches_brute_force(native, oracle, dev, cbox):
    """Scene::ray_intersect / ray_test on the device == brute-force definition (t,u,v bits, prim id)."""
    scene, sensor = cbox
    dev.upload(scene.desc())
    o, d, mint, maxt = _rays_for(sensor, 128, 96, 20000)
    g = dev.trace(o, d, mint, maxt)
    r = oracle.trace(scene.desc(), o, d, mint, maxt)
    assert np.array_equal(g["prim"], r["prim"])
    hit = r["prim"] != 0xffffffff
    assert hit.mean() > 0.9
    for k in ("t", "u", "v"):
        assert np.array_equal(g[k][hit].view(np.uint32), r[k][hit].view(np.uint32)), k
    assert np.array_equal(g["shape"], r["shape"])
    # secondary rays from the hit points (random directions), both closest and any-hit
    rng = np.random.default_rng(9)
    p = o[hit] + d[hit] * r["t"][hit][:, None]
    dd = rng.normal(0, 1, p.shape).astype(np.float32); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    g2 = dev.trace(p, dd, 1e-2, np.inf); r2 = oracle.trace(scene.desc(), p, dd, 1e-2, np.inf)
    assert np.array_equal(g2["prim"], r2["prim"])
    h2 = r2["prim"] != 0xffffffff
    assert np.array_equal(g2["t"][h2].view(np.uint32), r2["t"][h2].view(np.uint32))
    ga = dev.trace(p, dd, 1e-2, 300.0, any_hit=True); ra = oracle.trace(scene.desc(), p, dd, 1e-2, 300.0, any_hit=True)
    assert np.array_equal(np.isfinite(ga["t"]), np.isfinite(ra["t"]))


def test_trace_stairs_known_answer(native, dev):
    """src/librender/tests/test_kdtrees.py:26-59: t = 2 - floor(y*n)/n on the stairs mesh."""
    from mitsuba2_amd import scenes, api
    n_steps = 20
    v, f = scenes.stairs(n_steps)
    scene = api.Scene([api.Mesh("stairs", v, f)]).build(-1)
    dev.upload(scene.desc())
    n = 128; inv_n = 1.0 / (n - 1)
    xs, ys = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    o = np.stack([xs.ravel() * inv_n, ys.ravel() * inv_n, np.full(xs.size, 2.0)], 1).astype(np.float32)
    d = np.tile(np.array([0, 0, -1], np.float32), (len(o), 1))
    g = dev.trace(o, d, 0.0, 100.0)
    expected = 2.0 - np.floor((ys.ravel() * inv_n) * n_steps) / n_steps
    assert np.all(np.isfinite(g["t"]))
    assert np.allclose(g["t"], expected, atol=1e-6)
    assert dev.trace(o, d, 0.0, 100.0, any_hit=True)["t"].max() == 0.0


def test_trace_big_mesh_matches_brute_force(native, oracle, dev):
    """BVH with global-memory nodes (scene larger than the LDS budget) vs brute force."""
    from mitsuba2_amd import scenes, api
    v, f = scenes.random_triangles(3000, seed=11)
    scene = api.Scene([api.Mesh("soup", v, f)]).build(-1)
    dev.upload(scene.desc())
    rng = np.random.default_rng(12)
    n = 6000
    o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    g = dev.trace(o, d, 1e-4, np.inf); r = oracle.trace(scene.desc(), o, d, 1e-4, np.inf)
    assert np.array_equal(g["prim"], r["prim"])
    hit = r["prim"] != 0xffffffff
    assert np.array_equal(g["t"][hit].view(np.uint32), r["t"][hit].view(np.uint32))
    ga = dev.trace(o, d, 1e-4, 0.7, any_hit=True); ra = oracle.trace(scene.desc(), o, d, 1e-4, 0.7, any_hit=True)
    assert np.array_equal(np.isfinite(ga["t"]), np.isfinite(ra["t"]))


@pytest.mark.parametrize("n_tri", [3000, 40000])
def test_device_lbvh_matches_brute_force(native, oracle, n_tri):
    """mi_bvh_build(quality = 0): Morton sort + Karras radix tree + bottom-up fit on the device; the walk over
    that tree (stackless trail or LDS stack, whichever its depth admits) must equal brute force exactly."""
    from mitsuba2_amd import scenes, api
    v, f = scenes.random_triangles(n_tri, seed=21, size=0.03)
    v[3:12] = v[0:9]                                    # coincident triangles: identical Morton codes, t ties
    scene = api.Scene([api.Mesh("soup", v, f)]).build(-1)
    d = native.Device(0)
    d.upload(scene.desc(), bvh_quality=0x40)            # MI_BVH_RADIX_TREE: the radix tree over the Morton codes
    c = d.counters()
    assert c.bvh_on_device == 1 and c.bvh_builder == 1 and c.bvh_tris == n_tri and c.bvh_nodes == n_tri - 1 and 10 < c.bvh_depth <= 62
    rng = np.random.default_rng(22)
    n = 4000
    o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    dd = rng.normal(0, 1, (n, 3)).astype(np.float32); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    g = d.trace(o, dd, 1e-4, np.inf); r = oracle.trace(scene.desc(), o, dd, 1e-4, np.inf)
    assert np.array_equal(g["prim"], r["prim"]) and (r["prim"] != 0xffffffff).sum() > 100
    hit = r["prim"] != 0xffffffff
    for k in ("t", "u", "v"):
        assert np.array_equal(g[k][hit].view(np.uint32), r[k][hit].view(np.uint32))
    ga = d.trace(o, dd, 1e-4, 0.7, any_hit=True); ra = oracle.trace(scene.desc(), o, dd, 1e-4, 0.7, any_hit=True)
    assert np.array_equal(np.isfinite(ga["t"]), np.isfinite(ra["t"]))
    d.close()


def test_render_on_device_lbvh(native, oracle):
    """Whole-image parity does not depend on the builder: LBVH tree, material balls, both plans."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(64, 48, 8, diffuse_only=False, device=-1, ball_level=3)
    job = native.PathIntegrator().render_job(sensor)
    o32, _, ost = oracle.render(scene.desc(), job, threads=8, want_f64=False)
    d = native.Device(0)
    d.upload(scene.desc(), bvh_quality=0)
    assert d.counters().bvh_on_device == 1
    for plan in (1, 2):
        g32, st = d.render(job, plan=plan)
        assert st == 0 and d.counters().segments == ost.segments and np.array_equal(g32, o32)
    d.close()


def _render_both(native, oracle, dev, scene, sensor, **integ_kw):
    """-> device film (float64 atomics), oracle f32 film, oracle exact-sum film, counters, oracle stats.
    Also renders in the default mode (ordered gather) and requires that film to be bit-identical."""
    integ = native.PathIntegrator(**integ_kw)
    job = integ.render_job(sensor)
    dev.upload(scene.desc())
    o32, o64, ost = oracle.render(scene.desc(), job, threads=8)
    g64 = cnt = None
    for plan, spl in ((1, 0), (2, 0), (2, 3)):     # wavefront (HBM queues) / resident / resident in 3-sample passes
        g32, st = dev.render(job, plan=plan, samples_per_launch=spl)   # film auto -> sample log + ordered gather
        c = dev.counters()
        assert st == 0 and c.film_mode == 1 and c.plan == plan
        assert c.samples == ost.samples and c.segments == ost.segments
        assert np.array_equal(g32, o32), "plan %d: ordered-gather film is not bit-identical: rel L2 %g" % (plan, rel_l2(g32, o32))
        g64p, st = dev.render(job, f64=True, film_mode=2, plan=plan, samples_per_launch=spl)
        assert st == 0 and dev.counters().film_mode == 2 and dev.counters().plan == plan
        if g64 is not None:                        # float64 atomics: order-free to float32 precision
            assert np.array_equal(g64p.astype(np.float32), g64.astype(np.float32))
        g64, cnt = g64p, dev.counters()
    return g64, o32, o64, cnt, ost


def test_render_cornell_diffuse_parity(native, oracle, dev, cbox):
    """Config C1-class: Cornell box, path integrator, gaussian filter — film parity."""
    scene, sensor = cbox
    g64, o32, o64, cnt, ost = _render_both(native, oracle, dev, scene, sensor)
    # same work, exactly: every sample took the same number of path segments
    assert cnt.samples == ost.samples == 128 * 96 * 16
    assert cnt.segments == ost.segments
    # exact-sum film: identical samples, identical filter weights
    assert np.array_equal(g64.astype(np.float32), o64.astype(np.float32))
    assert np.abs(g64 - o64).max() <= 1e-9 * np.abs(o64).max()
    # reference-semantics float32 film within the north-star tolerance
    assert rel_l2(g64, o32) < REL_L2_TOL
    assert o32[..., 4].min() > 0 and np.isfinite(g64).all()


def test_render_float32_film_and_host_classes(native, oracle, cbox):
    """The drop-in route: Scene/PerspectiveCamera/PathIntegrator classes -> mi_render -> Film storage."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(96, 64, 8, device=0)
    integ = native.PathIntegrator()
    assert integ.render(scene, sensor) is True
    film = sensor.film.data((64, 96, 5))
    job = integ.render_job(sensor)
    o32, o64, _ = oracle.render(scene.desc(), job, threads=8)
    assert np.array_equal(film, o32)
    c = integ.counters()
    assert c.samples == 96 * 64 * 8 and c.bvh_tris == 32


@pytest.mark.parametrize("rfilter", ["gaussian", "tent", "box", "mitchell", "catmullrom"])
def test_film_replay_kernels_agree(native, oracle, rfilter, monkeypatch):
    """The ordered film (film_mode 1) through k_film_lanes (the default: a 4 x 4 texel block per lane, the sample loop specialised per
    footprint), k_film_quads (2 x 4 texel groups inside DPP quads; and its 4 x 2, 2 x 8, 4 x 4 shapes), k_film_columns<4,2> and k_film_groups: the same float32 additions per texel, so the same film bit for bit, and the oracle's. A ragged film with a crop
    window, 11 spp (trips of 16 records end inside a run)."""
    from mitsuba2_amd import scenes
    films = {}
    for name, env in (("lanes", {"MIW_FILM_LANES": "1"}), ("lanes_plain_log", {"MIW_FILM_LANES": "2"}), ("default", {}), ("quads", {"MIW_FILM_QUADS": "24"}), ("quads42", {"MIW_FILM_QUADS": "42"}), ("quads28", {"MIW_FILM_QUADS": "28"}), ("quads44", {"MIW_FILM_QUADS": "44"}),
                      ("columns", {"MIW_FILM_QUADS": "0"}), ("groups", {"MIW_FILM_COLUMNS": "0"})):
        for k in ("MIW_FILM_LANES", "MIW_FILM_QUADS", "MIW_FILM_COLUMNS", "MIW_FILM_GROUP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        scene, sensor = scenes.cornell_box(150, 83, 11, device=0, seed=77, rfilter=rfilter,
                                           crop_offset_x=7, crop_offset_y=2, crop_width=131, crop_height=70)
        integ = native.PathIntegrator()
        assert integ.render(scene, sensor) is True
        films[name] = sensor.film.data((70, 131, 5)).copy()
        c = integ.counters()
        # (15 tiles: below 448 the default is the group kernel; MIW_FILM_LANES = 1 asks for the block-per-lane kernel over its own log layout)
        assert (c.film_kernel, c.log_interleaved) == {"lanes": (4, 1), "lanes_plain_log": (4, 0), "default": (3, 0), "quads": (3, 0), "quads42": (3, 0),
                                                      "quads28": (3, 0), "quads44": (3, 0), "columns": (2, 0), "groups": (1, 0)}[name], (name, c.film_kernel, c.log_interleaved)
        if name == "lanes":
            o32, _, _ = oracle.render(scene.desc(), integ.render_job(sensor), threads=8)
    for name in films:
        assert np.array_equal(films[name], o32), name


@pytest.mark.parametrize("crop,n_threads,bs,rfilter", [((48, 48), 576, 2, "gaussian"), ((32, 32), 1024, 1, "tent"), ((100, 76), 450, 4, "mitchell"), ((197, 157), 480, 8, "box")])
def test_film_replay_by_texel_blocks_on_tiny_blocks(native, oracle, crop, n_threads, bs, rfilter):
    """integrator.cpp:88-97 halves the block size until there are as many blocks as worker threads. From 448 tiles on the film is
    replayed by k_film_lanes (a 4 x 4 texel block per lane over the tile-interleaved log) whatever the block size: tiles of 2 x 2 and
    1 x 1 pixels (bordered: 6 x 6 / 5 x 5 texels — a tile narrower than two texel blocks, windows clipped on every side), ragged
    last tiles, more than one group of 64 tiles with a partial last group; every filter reach."""
    from mitsuba2_amd import scenes
    w, h = crop
    scene, _ = scenes.cornell_box(384, 216, 5, device=-1, seed=9)
    sensor = scenes.cornell_sensor(384, 216, 5, rfilter=rfilter, crop_offset_x=40, crop_offset_y=20, crop_width=w, crop_height=h)
    job = native.PathIntegrator().render_job(sensor, n_threads=n_threads)
    assert job.cfg.block_size == bs and job.cfg.block_count >= 448, (job.cfg.block_size, job.cfg.block_count)
    o32, _, ost = oracle.render(scene.desc(), job, threads=8, want_f64=False)
    dev = native.Device(0)
    dev.upload(scene.desc())
    film, st = dev.render(job)
    c = dev.counters()
    assert st == 0 and (c.film_kernel, c.log_interleaved, c.samples) == (4, 1, ost.samples)
    assert np.array_equal(film, o32)
    # round 6, opt-in (MIW_FILM_OVERLAP=1): three quarters of that replay queued BESIDE the path kernel — sets of four 64-tile groups on three more streams,
    # each behind the flags its groups' last finished pixels raise (miwave.hip: overlap_enqueue), the rest in one launch after it — is the same film
    assert (c.film_overlapped, c.film_groups) == (0, 0)
    dev.set_option("MIW_FILM_OVERLAP", "1")
    launches = (((job.cfg.block_count + 63) // 64) * 3 // 4 + 3) // 4
    for _ in range(2):                                            # (twice: the flags and counters of the frame before are reset)
        beside, st = dev.render(job)
        c1 = dev.counters()
        assert st == 0 and np.array_equal(beside, o32) and c1.film_kernel == 4
        assert (c1.film_overlapped, c1.film_groups) in ((1, launches), (0, 0)), (c1.film_overlapped, c1.film_groups)   # (0, 0): the runtime refuses stream memory operations
    dev.set_option("MIW_FILM_OVERLAP", None)
    dev.close()


def test_render_samples_per_pass(native, oracle, dev):
    """samples_per_pass < sample_count (integrator.cpp:75-86) through the host classes: three passes of 2 spp, each
    seeded from its own block ids and accumulated onto the film (mi_render_cfg::accumulate) == the oracle run the
    same way, bit for bit, in both film modes and both plans."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(96, 64, 6, device=0)
    integ = native.PathIntegrator(samples_per_pass=2)
    assert integ.pass_count(sensor) == 3 and integ.render(scene, sensor) is True
    film = sensor.film.data((64, 96, 5)).copy()
    assert integ.counters().samples == 96 * 64 * 6
    dev.upload(scene.desc())
    o32 = o64 = None
    g = {(plan, fm): None for plan in (1, 2) for fm in (1, 2)}
    for p in range(3):
        job = integ.render_job(sensor, pass_index=p)
        o32, o64, _ = oracle.render(scene.desc(), job, threads=8, onto=(o32, o64))
        for (plan, fm) in g:
            g[(plan, fm)], st = dev.render(job, plan=plan, film_mode=fm, f64=(fm == 2), onto=g[(plan, fm)])
            assert st == 0
    assert np.array_equal(film, o32)
    for plan in (1, 2):
        assert np.array_equal(g[(plan, 1)], o32)
        assert np.allclose(g[(plan, 2)], o64, rtol=1e-12, atol=0)


def test_render_materials_parity(native, oracle, dev):
    """Config C3-class: GGX rough conductor + dielectric balls with shading normals."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(64, 48, 8, diffuse_only=False, device=-1, ball_level=2)
    g64, o32, o64, cnt, ost = _render_both(native, oracle, dev, scene, sensor)
    assert cnt.samples == ost.samples and cnt.segments == ost.segments
    assert np.array_equal(g64.astype(np.float32), o64.astype(np.float32))
    assert rel_l2(g64, o32) < REL_L2_TOL


@pytest.mark.parametrize("metal", [dict(distribution="beckmann"), dict(distribution="beckmann", sample_visible=False),
                                   dict(distribution="beckmann", alpha_u=0.05, alpha_v=0.3)])
def test_render_beckmann_parity(native, oracle, dev, metal):
    """roughconductor's default distribution (Beckmann, roughconductor.cpp:167-169): exp/log/erf/erfinv
    of miw/special.h agree bit for bit between gfx950 and the host, so the film does too."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(64, 48, 8, diffuse_only=False, device=-1, ball_level=2, metal=metal)
    g64, o32, o64, cnt, ost = _render_both(native, oracle, dev, scene, sensor)
    assert cnt.samples == ost.samples and cnt.segments == ost.segments
    assert np.array_equal(g64.astype(np.float32), o64.astype(np.float32))
    assert rel_l2(g64, o32) < REL_L2_TOL


def test_envmap_leaf_functions_bit_exact(native, oracle):
    """EnvironmentMapEmitter::eval / pdf_direction / sample_direction (hierarchical warp, lat-long lookup) on the
    device == the host, bit for bit."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.open_box(32, 32, 1, device=-1)
    d = native.Device(0); d.upload(scene.desc())
    rng = np.random.default_rng(11)
    v = rng.normal(size=(4096, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    x = np.concatenate([v, rng.uniform(0, 555, (4096, 3)), rng.uniform(0, 1, (4096, 2))], 1).astype(np.float32)
    g = d.eval(9, x); o = oracle.eval(9, x, desc=scene.desc())
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
    assert (g[:, 3] > 0).all() and (g[:, 8] > 0).all() and np.isfinite(g).all()
    d.close()


@pytest.mark.parametrize("kw", [dict(), dict(with_area_light=False), dict(envmap_after=0)])
def test_render_environment_map_parity(native, oracle, dev, kw):
    """Config C4-class emitter mix: area light + environment map (or the map alone), MIS on both."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.open_box(64, 48, 8, device=-1, ball_level=2, **kw)
    g64, o32, o64, cnt, ost = _render_both(native, oracle, dev, scene, sensor)
    assert cnt.samples == ost.samples and cnt.segments == ost.segments
    assert np.array_equal(g64.astype(np.float32), o64.astype(np.float32))
    assert rel_l2(g64, o32) < REL_L2_TOL


@pytest.mark.parametrize("kw", [dict(max_depth=1), dict(max_depth=2), dict(max_depth=4, rr_depth=2), dict(rr_depth=1)])
def test_render_depth_and_rr_variants(native, oracle, dev, kw):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(48, 40, 4, device=-1)
    g64, o32, o64, cnt, ost = _render_both(native, oracle, dev, scene, sensor, **kw)
    assert cnt.segments == ost.segments
    assert np.array_equal(g64.astype(np.float32), o64.astype(np.float32))


def test_render_ragged_film_crop_and_box_filter(native, oracle, dev):
    """Edge cases: film not a multiple of the block size, crop window, box filter, odd seed."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(77, 45, 4, device=-1, seed=1234, rfilter="box",
                                       crop_offset_x=5, crop_offset_y=3, crop_width=50, crop_height=37)
    g64, o32, o64, cnt, ost = _render_both(native, oracle, dev, scene, sensor)
    assert cnt.samples == 50 * 37 * 4 == ost.samples
    assert np.array_equal(g64.astype(np.float32), o64.astype(np.float32))
    assert rel_l2(g64, o32) < REL_L2_TOL


def test_tile_shards_sum_to_full_film(native, oracle, dev, cbox):
    """Multi-GPU partition: rank shards (interleaved spiral blocks) add up to the 1-GPU film."""
    scene, sensor = cbox
    dev.upload(scene.desc())
    for mode in (2, 1):
        full, _ = dev.render(native.PathIntegrator().render_job(sensor), f64=True, film_mode=mode)
        acc = np.zeros_like(full)
        for rank in range(3):
            integ = native.PathIntegrator(); integ.set_shard(rank, 3)
            part, _ = dev.render(integ.render_job(sensor), f64=True, film_mode=mode)
            acc += part
        if mode == 2:
            assert np.array_equal(acc.astype(np.float32), full.astype(np.float32))
        else:
            # float32 block partials: texels covered by one block are exact, texels under a block
            # border differ at most by the association of <= 4 float32 partials (independent.cpp:36-40)
            assert rel_l2(acc, full) < 1e-7
            assert (acc.astype(np.float32) == full.astype(np.float32)).mean() > 0.7


def test_empty_tile_shard_renders_nothing(native, dev):
    """world > block count (64x48 = 4 blocks, 8 ranks): ranks 4..7 render nothing — a sharded job with zero tiles is
    not an unsharded job — and the eight partial films still add up to the 1-rank film."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(64, 48, 4, device=-1)
    dev.upload(scene.desc())
    for mode in (1, 2):
        full, _ = dev.render(native.PathIntegrator().render_job(sensor), f64=True, film_mode=mode)
        acc = np.zeros_like(full); samples = 0
        for rank in range(8):
            integ = native.PathIntegrator(); integ.set_shard(rank, 8)
            part, st = dev.render(integ.render_job(sensor), f64=True, film_mode=mode)
            c = dev.counters()
            assert st == 0 and (c.samples > 0) == (rank < 4)
            if rank >= 4:
                assert not part.any()
            acc += part; samples += c.samples
        assert samples == 64 * 48 * 4 and rel_l2(acc, full) < 1e-7


def test_errors_are_loud(native, dev):
    import ctypes as C
    from mitsuba2_amd import _capi
    d2 = native.Device(0)
    job_cfg = _capi.mi_render_cfg()
    film = np.zeros(5, np.float32)
    st = d2.L.mi_render(d2.ctx, C.byref(job_cfg), film.ctypes.data_as(C.c_void_p))
    assert st == _capi.MI_ERR_STATE and b"mi_scene_upload" in d2.L.mi_last_error(d2.ctx)
    d2.close()
    with pytest.raises(RuntimeError):
        native.PathIntegrator(rr_depth=0)
    with pytest.raises(RuntimeError):
        native.BSDF("blendbsdf")
    # malformed descriptions of the newer record types are refused by mi_scene_upload with a message
    from mitsuba2_amd import scenes
    d3 = native.Device(0)
    for breaker, msg in ((lambda d: setattr(d.bsdfs[0], "type", 99), b"unknown type"),
                         (lambda d: (setattr(d.bsdfs[0], "flags", 0x100), setattr(d.bsdfs[0], "back", 77)), b"back-side record"),
                         (lambda d: setattr(d, "rectangle_count", 0), b"no mi_rectangle record"),
                         (lambda d: setattr(d.rectangles[0], "shape", 7), b"is not a (single) MI_SHAPE_RECTANGLE shape")):
        scene = native.Scene(scenes.rect_box_meshes()).build(-1)
        desc = scene.desc().contents
        breaker(desc)
        st = d3.L.mi_scene_upload(d3.ctx, scene.desc())
        assert st == _capi.MI_ERR_INVALID and msg in d3.L.mi_last_error(d3.ctx), (msg, d3.L.mi_last_error(d3.ctx))
    d3.close()


def test_forced_tree_walk_on_cornell(native, oracle, cbox):
    """The Cornell box normally takes the LDS brute-force sweep (<= 64 triangles); force the
    stackless BVH walk (LDS-resident nodes + triangles) and require the same bits."""
    from mitsuba2_amd import _capi
    scene, sensor = cbox
    d = native.Device(0)
    d.upload(scene.desc(), bvh_quality=1 | 0x10)
    o, dd, mint, maxt = _rays_for(sensor, 128, 96, 20000, seed=4)
    g = d.trace(o, dd, mint, maxt); r = oracle.trace(scene.desc(), o, dd, mint, maxt)
    assert np.array_equal(g["prim"], r["prim"])
    hit = r["prim"] != 0xffffffff
    for k in ("t", "u", "v"):
        assert np.array_equal(g[k][hit].view(np.uint32), r[k][hit].view(np.uint32))
    job = native.PathIntegrator().render_job(sensor)
    o32, _, ost = oracle.render(scene.desc(), job, threads=8, want_f64=False)
    g32, st = d.render(job)
    assert st == 0 and np.array_equal(g32, o32)
    d.close()


def test_resident_leaf_filter_equals_full_sweep(native, oracle):
    """The resident plan's two-phase scene query (leaf-box candidate filter + per-lane exact tests) must
    return exactly what sweeping every triangle returns — checked on ~40 M rays (films bit-identical,
    identical segment / shadow-ray counts), and against the scalar oracle on a slice."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(512, 288, 64, diffuse_only=True, device=-1)
    job = native.PathIntegrator().render_job(sensor)
    films, counts = [], []
    for flags in (1, 1 | 0x20):                       # SAH build; with and without MI_BVH_NO_LEAF_FILTER
        d = native.Device(0)
        d.upload(scene.desc(), bvh_quality=flags)
        f, st = d.render(job, plan=2, samples_per_launch=64)
        c = d.counters()
        assert st == 0 and c.plan == 2 and c.film_mode == 1
        films.append(f); counts.append((c.samples, c.segments, c.shadow_rays))
        d.close()
    assert counts[0] == counts[1] and counts[0][0] == 512 * 288 * 64
    assert np.array_equal(films[0], films[1])
    o32, _, ost = oracle.render(scene.desc(), job, threads=os.cpu_count() or 8, want_f64=False,
                                only_blocks=np.arange(12, dtype=np.uint32))
    assert ost.samples == 12 * 1024 * 64
    # the 12 centre-most spiral blocks: texels not under another block's border must match the oracle bit for bit
    touched = o32[..., 4] > 0
    inner = np.zeros_like(touched)
    ids = np.asarray(job.block_ids[:job.cfg.block_count]).reshape(-1, (512 + 31) // 32)
    for by, bx in zip(*np.nonzero(ids < 12)):
        inner[by * 32 + 2: by * 32 + 30, bx * 32 + 2: bx * 32 + 30] = True
    assert inner.sum() == 12 * 28 * 28 and np.array_equal(films[0][inner], o32[inner])


@pytest.mark.parametrize("quality", [0, 0x40])
def test_interior_scene_crop_parity(native, oracle, quality):
    """BASELINE config 4 class at full geometric scale (911 362 triangles, area light + environment map, all three
    BSDFs with shading normals): a 24 x 16 crop window rendered on the device (SAH and device-LBVH trees, LDS-stack
    or stackless walk) against the oracle's brute-force scene queries — film bit-identical."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.interior_scene(192, 128, 2, device=-1, crop_offset_x=90, crop_offset_y=70, crop_width=24, crop_height=16)
    job = native.PathIntegrator().render_job(sensor)
    o32, _, ost = oracle.render(scene.desc(), job, threads=os.cpu_count() or 8, want_f64=False)
    d = native.Device(0)
    d.upload(scene.desc(), bvh_quality=quality)
    c = d.counters()
    assert c.bvh_tris == 911362 and c.bvh_builder == (3 if quality == 0 else 1)
    g32, st = d.render(job)
    c = d.counters()
    assert st == 0 and c.samples == ost.samples == 24 * 16 * 2 and c.segments == ost.segments
    assert np.array_equal(g32, o32)
    d.close()


def test_timeout_and_cancel_return_partial_films(native, cbox):
    """Integrator `timeout` / cancel() (integrator.cpp:34,43-45,143-146,178): render() returns false, the film holds
    the samples finished so far, and a later render on the same objects is complete again."""
    import threading
    import time
    from mitsuba2_amd import scenes, _capi
    scene, sensor = scenes.cornell_box(256, 256, 4096, device=-1)
    d = native.Device(0)
    d.upload(scene.desc())
    job = native.PathIntegrator().render_job(sensor)
    total = 256 * 256 * 4096
    for plan in (2, 1):
        job.cfg.timeout_s = 0.05
        film, st = d.render(job, plan=plan, samples_per_launch=64)
        c = d.counters()
        assert st == _capi.MI_ERR_CANCELLED and 0 < c.samples < total and np.isfinite(film).all() and film[..., 4].max() > 0
        assert b"cancel" in d.L.mi_last_error(d.ctx)
    job.cfg.timeout_s = 0.0
    t = threading.Timer(0.05, lambda: d.L.mi_cancel(d.ctx))
    t0 = time.time(); t.start()
    film, st = d.render(job, plan=2, samples_per_launch=64)
    t.join()
    assert st == _capi.MI_ERR_CANCELLED and 0 < d.counters().samples < total and time.time() - t0 < 5.0
    small_scene, small_sensor = cbox
    d.upload(small_scene.desc())
    film, st = d.render(native.PathIntegrator().render_job(small_sensor))
    assert st == 0 and d.counters().samples == 128 * 96 * 16
    d.close()


def test_full_size_c2_properties(native, oracle):
    """BASELINE config 2 at its full size (Cornell box, 1920x1080 @ 512 spp: 1.06 G samples — hours for the oracle),
    checked through size-independent properties: sample accounting, run-to-run determinism, the two execution plans
    and the 8-rank partition all produce the same float32 film (the partition to <= 1 ulp under block borders), the
    exact-sum film agrees within the north-star tolerance, and a 64x64 crop window rendered at the full 512 spp is
    bit-identical to the oracle's."""
    from mitsuba2_amd import scenes
    W, H, SPP = 1920, 1080, 512
    scene, sensor = scenes.cornell_box(W, H, SPP, device=-1)
    dev = native.Device(0)
    dev.upload(scene.desc())
    job = native.PathIntegrator().render_job(sensor)
    a, st = dev.render(job)
    c = dev.counters()
    assert st == 0 and c.samples == W * H * SPP and c.film_mode == 1 and c.plan == 2
    assert 3.0 < c.segments / c.samples < 4.0 and np.isfinite(a).all() and a[..., 4].min() > 0
    b, _ = dev.render(job)
    assert np.array_equal(a, b)                                     # idempotent
    p1, _ = dev.render(job, plan=1)                                 # HBM-queue wavefront plan: same bits
    assert dev.counters().plan == 1 and np.array_equal(p1, a)
    f64, _ = dev.render(job, f64=True, film_mode=2)                 # order-free float64 sums
    assert rel_l2(a, f64) < REL_L2_TOL
    acc = np.zeros((H, W, 5), np.float64)
    for rank in range(8):                                           # the 8-GPU partition of the frame
        integ = native.PathIntegrator(); integ.set_shard(rank, 8)
        part, _ = dev.render(integ.render_job(sensor))
        assert dev.counters().samples < W * H * SPP / 7
        acc += part
    assert rel_l2(acc, a) < 1e-7 and (acc.astype(np.float32) == a).mean() > 0.7
    # alpha / weight: every camera ray of this closed view but the open front hits geometry
    alpha = a[..., 3] / a[..., 4]
    assert 0.99 < alpha[H // 2 - 100:H // 2 + 100, W // 2 - 100:W // 2 + 100].mean() <= 1.0 + 1e-6
    # full sample count on a crop window: bit-identical to the oracle
    crop_sensor = scenes.cornell_sensor(W, H, SPP, crop_offset_x=928, crop_offset_y=508, crop_width=64, crop_height=64)
    cjob = native.PathIntegrator().render_job(crop_sensor)
    g, _ = dev.render(cjob)
    o32, _, ost = oracle.render(scene.desc(), cjob, threads=16, want_f64=False)
    assert dev.counters().samples == ost.samples == 64 * 64 * SPP and np.array_equal(g, o32)
    dev.close()


def test_full_size_c3_structures_agree(native, oracle):
    """BASELINE config 3 geometry (material balls: GGX conductor + dielectric, 40 972 triangles) at 1920x1080, 128 spp
    (2.7e8 samples): the SAH tree, the device LBVH (both through the LDS-stack walk of the resident plan) and the
    HBM-queue wavefront plan (stackless walk) produce the same float32 film bit for bit — three traversals over two
    different trees. (The brute-force oracle needs hours at 41 k triangles; it checks this scene class at low
    tessellation in test_render_materials_parity.)"""
    from mitsuba2_amd import scenes
    W, H, SPP = 1920, 1080, 128
    scene, sensor = scenes.cornell_box(W, H, SPP, diffuse_only=False, device=-1)
    job = native.PathIntegrator().render_job(sensor)
    dev = native.Device(0)
    dev.upload(scene.desc())                                       # the binned-SAH tree (built on the device)
    a, st = dev.render(job)
    ca = dev.counters()
    assert st == 0 and ca.samples == W * H * SPP and ca.plan == 2 and ca.bvh_builder == 3 and ca.bvh_tris == 40972
    p1, _ = dev.render(job, plan=1)
    assert dev.counters().plan == 1 and np.array_equal(p1, a)
    dev.upload(scene.desc(), bvh_quality=0x40)                     # MI_BVH_RADIX_TREE: another tree, same answers
    b, _ = dev.render(job)
    cb = dev.counters()
    assert cb.bvh_builder == 1 and cb.segments == ca.segments and np.array_equal(b, a)
    assert np.isfinite(a).all() and 3.0 < ca.segments / ca.samples < 5.0
    dev.close()


def test_device_against_committed_golden_films(native, dev):
    """The device against tests/golden/*.npz directly — no oracle in the loop on the GPU box."""
    import os
    from mitsuba2_amd import scenes
    from test_cpu_pipeline import _round1_plugin_jobs, GOLDEN
    g = np.load(os.path.join(GOLDEN, "cornell_48x32_4spp.npz"))
    scene, sensor = scenes.cornell_box(48, 32, 4, device=-1)
    dev.upload(scene.desc())
    film, st = dev.render(native.PathIntegrator().render_job(sensor))
    assert st == 0 and np.array_equal(film, g["film"]) and dev.counters().segments == int(g["segments"])
    g = np.load(os.path.join(GOLDEN, "round1_plugins.npz"))
    for key, scene, job in _round1_plugin_jobs(native):
        dev.upload(scene.desc())
        film, st = dev.render(job)
        assert st == 0 and np.array_equal(film, g["film_" + key]) and dev.counters().segments == int(g["segments_" + key]), key


@pytest.mark.parametrize("which", ["packets", "tree", "tree_lbvh"])
def test_placed_queues_and_priorities_do_not_change_the_film(native, which):
    """What one rank of an 8-GPU frame renders (1/8 of the 1080p tiles: one pixel per resident lane) goes through a measuring
    launch + a device sort of the pixels by cost + per-SIMD pixel queues + least-progress-first wave priorities
    (csrc/device/resident_kernel.h: QueueWork). All of it is scheduling: the samples and their log slots are the same, so the film
    must be the plain launch's bit for bit — for the packet kernel (Cornell box) and for the phase machine's Placed instantiation
    (material balls: SAH tree and device-built tree)."""
    import os
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(1920, 1080, 128, diffuse_only=(which == "packets"), device=-1)
    dev = native.Device(0)
    dev.upload(scene.desc(), bvh_quality=0x40 if which == "tree_lbvh" else 0)
    assert dev.counters().bvh_tris == (32 if which == "packets" else 40972)
    integ = native.PathIntegrator(); integ.set_shard(0, 8)
    job = integ.render_job(sensor)
    films = {}
    for name, env in (("default", {}), ("plain", {"MIW_PLACE": "0", "MIW_TAIL_PRIO": "0"})):
        for k, v in env.items():
            dev.set_option(k, v)                                  # (the context's own switches: the environment is read in mi_create only)
        try:
            films[name], st = dev.render(job, samples_per_launch=128)
        finally:
            for k in env:
                dev.set_option(k, None)
        c = dev.counters()
        assert st == 0 and c.samples > 0
        assert c.placed == (1 if name == "default" else 0) and c.n_path == (2 if name == "default" else 1)
    assert np.array_equal(films["default"], films["plain"]) and films["default"][..., 4].max() > 0
    dev.close()


@pytest.mark.parametrize("which", ["matball", "interior"])
def test_device_builder_builds_the_host_builders_tree(native, which):
    """mi_bvh_build quality 0 (csrc/sah_device.h: the binned-SAH builder run level by level on the GPU) against quality 1 (the host
    recursion, csrc/bvh_build.h): same inner-node count, same depth, the 4-wide tree collapsed on the device, and the same film —
    on the 41 k-triangle material balls and the 0.9 M-triangle interior (ShapeKDTree::build's place, src/librender/scene_native.inl:3-10;
    the reference's GPU mode builds on the device, include/mitsuba/render/optix/shapes.h:72-167)."""
    import sys
    from mitsuba2_amd import scenes
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_r3 as G
    if which == "matball":
        scene, _ = scenes.cornell_box(1920, 1080, 16, diffuse_only=False, device=-1)
        x, y, w, h = G.C3_WINDOW
    else:
        scene, _ = scenes.interior_scene(1920, 1080, 16, device=-1)
        x, y, w, h = G.C4_WINDOWS["clutter"]
    job = G.crop_job(native, scenes, 16, x, y, w, h, n_threads=64)
    dev = native.Device(0)
    got = {}
    for quality in (1, 0, 0):                                          # (the second quality-0 build: a warm one, for the time)
        dev.upload(scene.desc(), bvh_quality=quality)
        c = dev.counters()
        film, st = dev.render(job)
        assert st == 0
        got[quality] = (c.bvh_nodes, c.bvh_tris, c.bvh_depth, film, c.bvh_builder, c.bvh_on_device, c.bvh4_on_device, c.ms_bvh_build, dev.counters().segments)
    host, device = got[1], got[0]
    assert host[4] == 0 and host[5] == 0 and device[4] == 3 and device[5] == 1 and device[6] == 1
    assert device[:3] == host[:3] and device[8] == host[8]
    assert np.array_equal(device[3], host[3])
    print("device build of %s: %.1f ms (host recursion %.1f ms)" % (which, device[7], host[7]))
    if os.environ.get("MIW_TEST_TIMING"):                              # a wall-clock bound only on request (a shared or cold GPU must not fail the parity tier)
        assert device[7] < (60.0 if which == "interior" else 30.0), "device build took %.1f ms" % device[7]
    dev.close()


def test_scene_tables_that_do_not_fit_lds_take_the_lock_step_kernels(native, oracle):
    """round 4: the phase machine and the packet kernels read the scene's small tables (shape / BSDF / emitter records) from LDS
    (device/trace.h: stage_tables). A scene whose tables exceed what the stack leaves of a workgroup's 40 KB — here 330 meshes
    with a BSDF each, 47 KB of records — must still render the oracle's film: through the lock-step tree kernel, which reads
    the tables from global memory (counters().path_kernel 0 instead of 1)."""
    from mitsuba2_amd import scenes
    rng = np.random.default_rng(41)
    meshes = [m for m in scenes.cornell_box_meshes(True, 1)]
    for i in range(330):                                            # confetti: one small triangle and one diffuse BSDF each
        c = np.array([rng.uniform(60, 490), rng.uniform(20, 520), rng.uniform(60, 490)])
        v = (c + rng.normal(0, 12, (3, 3))).astype(np.float32)
        meshes.append(native.Mesh("confetti%d" % i, v, np.array([[0, 1, 2]], np.uint32),
                                  bsdf=native.BSDF("diffuse", reflectance=tuple(float(x) for x in rng.uniform(0.1, 0.9, 3)))))
    scene = native.Scene(meshes).build(-1)
    assert scene.desc().contents.bsdf_count >= 330
    sensor = scenes.cornell_sensor(48, 40, 6)
    job = native.PathIntegrator().render_job(sensor)
    o32, _, ost = oracle.render(scene.desc(), job, threads=8, want_f64=False)
    d = native.Device(0)
    d.upload(scene.desc())
    g32, st = d.render(job)
    c = d.counters()
    assert st == 0 and c.plan == 2 and c.path_kernel == 0 and (c.samples, c.segments) == (ost.samples, ost.segments)
    assert np.array_equal(g32, o32)
    d.close()
    # the same room without the confetti (forced onto the tree kernels): tables of a few hundred bytes -> the phase machine
    scene2, _ = scenes.cornell_box(48, 40, 6, diffuse_only=False, device=-1, ball_level=2)
    d = native.Device(0)
    d.upload(scene2.desc())
    g, st = d.render(job)
    assert st == 0 and d.counters().path_kernel == 1
    d.close()
