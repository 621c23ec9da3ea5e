"""Randomised CPU cross-check: the staged emulator of the device kernels (oracle/wavefront_emu.cpp — the product's per-lane
stage functions in plain loops) against the scalar restatement of the reference (oracle/miw_oracle.cpp) on random scenes.

    python tools/fuzz_cpu.py [--seeds 200] [--first 0] [--verbose]

Every seed draws a scene (closed or open room, random blocks / spheres / rectangles / triangle soups with or without shading
normals and texture coordinates, a random BSDF of every plugin on each of them, one to three area lights among meshes, spheres
and rectangles, an optional environment map at a random position in the emitter order), a sensor (resolution, crop window,
reconstruction filter, spp, seed) and an integrator (path: max_depth, rr_depth; direct: any split of emitter / BSDF samples,
hide_emitters; samples_per_pass; wavefront stages or the resident sample loop in launches of a few samples), renders it both
ways and requires
the float32 films to be bit-identical and the segment counts equal. The two sides share the leaf headers but not their control
flow (queues, stage cuts, regeneration, film replay vs. one scalar loop), which is what this exercises; the tree walks (the
stackless BVH2 walk and the 4-wide quantised tree of the phase machine) are checked against brute force on random rays. A failing seed is reported with its recipe. Test infrastructure only.
"""
import argparse
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def random_bsdf(api, g, textured_ok):
    kind = g.choice(["diffuse", "diffuse", "dielectric", "roughconductor", "conductor", "plastic", "roughdielectric", "roughplastic", "twosided"])
    col = lambda lo=0.05, hi=0.9: tuple(float(x) for x in g.uniform(lo, hi, 3))
    dist = lambda: dict(distribution=str(g.choice(["ggx", "beckmann"])), sample_visible=bool(g.random() < 0.7))
    rough = lambda: (dict(alpha=float(g.uniform(0.03, 0.6))) if g.random() < 0.6 else
                     dict(alpha_u=float(g.uniform(0.03, 0.6)), alpha_v=float(g.uniform(0.03, 0.6))))
    if kind == "diffuse":
        if textured_ok and g.random() < 0.4:
            tex = api.BitmapTexture(g.uniform(0.05, 0.9, (int(g.integers(2, 9)), int(g.integers(2, 9)), 3)).astype(np.float32),
                                    filter_type=str(g.choice(["bilinear", "nearest"])), wrap_mode=str(g.choice(["repeat", "mirror", "clamp"])))
            return api.BSDF("diffuse", reflectance=tex), "diffuse(bitmap)"
        return api.BSDF("diffuse", reflectance=col()), "diffuse"
    if kind == "dielectric":
        return api.BSDF("dielectric", int_ior=float(g.uniform(1.1, 2.2)), ext_ior=float(g.uniform(1.0, 1.3))), kind
    if kind == "roughconductor":
        kw = dict(eta=col(0.1, 2.0), k=col(1.0, 4.0)); kw.update(dist()); kw.update(rough())
        return api.BSDF("roughconductor", **kw), "roughconductor %s" % kw
    if kind == "conductor":
        return api.BSDF("conductor", eta=col(0.1, 2.0), k=col(1.0, 4.0)), kind
    if kind == "plastic":
        return api.BSDF("plastic", diffuse_reflectance=col(), int_ior=float(g.uniform(1.2, 1.9)), nonlinear=bool(g.random() < 0.5)), kind
    if kind == "roughdielectric":
        kw = dict(int_ior=float(g.uniform(1.2, 2.0)), ext_ior=1.0); kw.update(dist()); kw.update(rough())
        return api.BSDF("roughdielectric", **kw), "roughdielectric %s" % kw
    if kind == "roughplastic":
        kw = dict(diffuse_reflectance=col(), int_ior=float(g.uniform(1.2, 1.9)), alpha=float(g.uniform(0.05, 0.5)),
                  distribution=str(g.choice(["ggx", "beckmann"])), nonlinear=bool(g.random() < 0.5))
        return api.BSDF("roughplastic", **kw), "roughplastic %s" % kw
    front, fn = random_bsdf(api, np.random.default_rng(int(g.integers(1 << 30))), False)
    while fn.startswith(("dielectric", "roughdielectric", "twosided")):                  # twosided takes reflective BSDFs only
        front, fn = random_bsdf(api, np.random.default_rng(int(g.integers(1 << 30))), False)
    if g.random() < 0.5:
        return api.TwoSided(front), "twosided(%s)" % fn
    back, bn = random_bsdf(api, np.random.default_rng(int(g.integers(1 << 30))), False)
    while bn.startswith(("dielectric", "roughdielectric", "twosided")):
        back, bn = random_bsdf(api, np.random.default_rng(int(g.integers(1 << 30))), False)
    return api.TwoSided(front, back), "twosided(%s | %s)" % (fn, bn)


def box_mesh(lo, hi):
    x0, y0, z0 = lo; x1, y1, z1 = hi
    v = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]], np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [3, 6, 2], [3, 7, 6], [0, 4, 7], [0, 7, 3], [1, 2, 6], [1, 6, 5]], np.uint32)
    return v, f


def make_case(api, scenes, seed):
    """-> scene, sensor, integrator kwargs, recipe lines, objects to keep alive. Twelve seeds of this draw are pinned as film digests
    in tests/golden/round2.json: after changing the draw, rerun tests/golden/make_golden_r2.py."""
    g = np.random.default_rng(seed)
    recipe = []
    shapes = []
    room = 10.0
    v, f = box_mesh((0, 0, 0), (room, room, room))
    f = f[:, ::-1].copy()                                                    # inward-facing walls
    open_top = g.random() < 0.4
    if open_top:
        f = np.array([t for t in f if not all(v[i][1] == room for i in t)], np.uint32)
    wall, wn = random_bsdf(api, g, False)
    if wn.startswith(("dielectric", "roughdielectric")):
        wall, wn = api.BSDF("diffuse", reflectance=(0.6, 0.6, 0.6)), "diffuse"
    shapes.append(api.Mesh("room", v, f, bsdf=wall)); recipe.append("room %s%s" % (wn, " open" if open_top else ""))
    n_lights = 0
    for k in range(int(g.integers(1, 6))):
        kind = g.choice(["block", "sphere", "rect", "soup", "ball"])
        c = g.uniform(2.0, room - 2.0, 3); s = g.uniform(0.4, 1.6)
        emit = g.random() < 0.35 and n_lights < 3
        em = api.AreaLight(tuple(float(x) for x in g.uniform(2.0, 30.0, 3))) if emit else None
        n_lights += int(emit)
        if kind == "sphere":
            b, bn = (None, "emitter") if emit else random_bsdf(api, g, False)
            shapes.append(api.Mesh.sphere(tuple(float(x) for x in c), float(s), bsdf=b, emitter=em, name="s%d" % k))
        elif kind == "rect":
            a = g.uniform(0, 2 * np.pi); m = np.eye(4, dtype=np.float32)
            m[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.diag([s, s, 1.0]) @ \
                np.array([[1, 0, 0], [0, np.cos(a / 2), -np.sin(a / 2)], [0, np.sin(a / 2), np.cos(a / 2)]])
            m[:3, 3] = c
            b, bn = (None, "emitter") if emit else random_bsdf(api, g, True)
            shapes.append(api.Mesh.rectangle(to_world=m, flip_normals=bool(g.random() < 0.5), bsdf=b, emitter=em, name="r%d" % k))
        elif kind == "ball":
            vv, ff, nn = scenes.icosphere(tuple(float(x) for x in c), float(s), int(g.integers(0, 3)))
            uv = None
            if g.random() < 0.5:
                uv = np.stack([np.arctan2(vv[:, 0] - c[0], vv[:, 2] - c[2]) / (2 * np.pi) + 0.5, (vv[:, 1] - c[1]) / (2 * s) + 0.5], 1)
            b, bn = (None, "emitter") if emit else random_bsdf(api, g, uv is not None)
            shapes.append(api.Mesh("b%d" % k, vv, ff, normals=nn if g.random() < 0.7 else None, bsdf=b, emitter=em, texcoords=uv))
        elif kind == "soup":
            n = int(g.integers(3, 30))
            p = c + g.normal(0, s, (n, 1, 3)) + g.normal(0, 0.5 * s, (n, 3, 3))
            vv = np.clip(p.reshape(-1, 3), 0.2, room - 0.2).astype(np.float32); ff = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
            b, bn = (None, "emitter") if emit else random_bsdf(api, g, False)
            shapes.append(api.Mesh("t%d" % k, vv, ff, bsdf=b, emitter=em))
        else:
            vv, ff = box_mesh(c - s, c + s)
            b, bn = (None, "emitter") if emit else random_bsdf(api, g, False)
            shapes.append(api.Mesh("k%d" % k, vv, ff, bsdf=b, emitter=em))
        recipe.append("%s %s" % (kind, bn))
    env = None; env_after = None
    if open_top or g.random() < 0.25:
        env = api.EnvMap(scenes.sky_envmap(int(g.integers(8, 40)), int(g.integers(6, 20)), seed=int(g.integers(100))),
                         scale=float(g.uniform(0.2, 2.0)),
                         to_world=dict(origin=(0, 0, 0), target=tuple(float(x) for x in g.normal(size=3)), up=(0.1, 1, 0.05)))
        env_after = int(g.integers(0, len(shapes) + 1)) if g.random() < 0.5 else None
        recipe.append("envmap after %s" % env_after)
    if n_lights == 0 and env is None:
        vv, ff = box_mesh((4, room - 0.3, 4), (6, room - 0.2, 6))
        shapes.append(api.Mesh("lamp", vv, ff, emitter=api.AreaLight((20.0, 18.0, 15.0)))); recipe.append("lamp")
    scene = api.Scene(shapes, envmap=env, envmap_after=env_after).build(-1)

    W, H = int(g.integers(8, 41)), int(g.integers(8, 33))
    film_kw = {}
    if g.random() < 0.3:
        cw, ch = int(g.integers(3, W + 1)), int(g.integers(3, H + 1))
        film_kw = dict(crop_offset_x=int(g.integers(0, W - cw + 1)), crop_offset_y=int(g.integers(0, H - ch + 1)), crop_width=cw, crop_height=ch)
    rfilter = str(g.choice(["gaussian", "gaussian", "box", "tent", "mitchell", "catmullrom", "lanczos"]))
    spp = int(g.choice([1, 2, 3, 4, 6, 8]))
    film = api.Film(rfilter=rfilter, width=W, height=H, **film_kw)
    sampler = api.Sampler(sample_count=spp, seed=int(g.integers(0, 1000)))
    eye = g.uniform(1.0, room - 1.0, 3); tgt = g.uniform(3.0, room - 3.0, 3)
    sensor = api.Sensor(film, sampler, fov=float(g.uniform(25, 95)),
                        to_world=dict(origin=tuple(float(x) for x in eye), target=tuple(float(x) for x in tgt), up=(0, 1, 0)))
    if g.random() < 0.2:                                                          # src/integrators/direct.cpp on the same stages
        ikw = dict(integrator="direct", hide_emitters=bool(g.random() < 0.3))
        if g.random() < 0.5:
            ikw["shading_samples"] = int(g.integers(1, 4))
        else:
            ikw["emitter_samples"] = int(g.integers(0, 4)); ikw["bsdf_samples"] = int(g.integers(0 if ikw["emitter_samples"] else 1, 4))
    else:
        ikw = dict(max_depth=int(g.choice([-1, -1, 1, 2, 3, 6])), rr_depth=int(g.choice([5, 5, 1, 2, 8])))
    if spp % 2 == 0 and g.random() < 0.2:
        ikw["samples_per_pass"] = spp // 2
    recipe.append("film %dx%d %s %s spp %d; %s" % (W, H, film_kw, rfilter, spp, ikw))
    return scene, sensor, ikw, recipe, (scene, shapes, env, film, sampler)


def run_case(api, scenes, orc, seed, resident_only=False):
    """resident_only: the scalar_spectral libraries have no HBM-queue plan (path.h: RGB path state) — its recipes take the resident plan"""
    scene, sensor, ikw, recipe, keep = make_case(api, scenes, seed)
    ikw = dict(ikw)
    integ = (api.DirectIntegrator if ikw.pop("integrator", "path") == "direct" else api.PathIntegrator)(**ikw)
    passes = integ.pass_count(sensor)
    gp = np.random.default_rng(seed + 99)
    plan2, per_launch = bool(gp.random() < 0.5), int(gp.integers(1, 9))
    plan2 = plan2 or resident_only
    recipe.append("plan %s" % ("2, %d samples per launch" % per_launch if plan2 else "1"))
    o_acc, e_acc = (None, None), (None, None)                    # (f32, f64) of the oracle, (f64, f32) of the emulator
    segs = [0, 0]
    for p in range(passes):
        job = integ.render_job(sensor, pass_index=p)
        if plan2:                                                  # the resident plan's per-pixel sample loop, advanced in launches
            job.cfg.plan = 2; job.cfg.samples_per_launch = per_launch
        o32, o64, st = orc.render(scene.desc(), job, threads=2, onto=o_acc)
        e64, e32, est = orc.emu_render(scene.desc(), job, onto=e_acc)
        o_acc, e_acc = (o32, o64), (e64, e32)
        segs[0] += st.segments; segs[1] += est[1]
        o_last, e_last = o32, e32
    ok = np.array_equal(np.asarray(o_last).view(np.uint32), np.asarray(e_last).view(np.uint32)) and segs[0] == segs[1]
    # the tree walk against brute force on a few camera rays
    g = np.random.default_rng(seed + 7)
    d_ = scene.desc().contents
    v = np.ctypeslib.as_array(d_.vertex_positions, (d_.vertex_count * 3,)).reshape(-1, 3)
    o = g.uniform(v.min(0), v.max(0), (64, 3)).astype(np.float32)
    d = g.normal(size=(64, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    for any_hit in (False, True):
        a = orc.trace(scene.desc(), o, d, 1e-4, np.inf, any_hit=any_hit)
        b = orc.emu_trace(scene.desc(), o, d, 1e-4, np.inf, any_hit=any_hit)
        ok = ok and np.array_equal(np.asarray(a["t"]).view(np.uint32), np.asarray(b["t"]).view(np.uint32))
        w = orc.emu_trace4(scene.desc(), o, d, 1e-4, np.inf, any_hit=any_hit, stack_budget=31)      # the 4-wide quantised tree
        ok = ok and np.array_equal(np.asarray(a["t"]).view(np.uint32), np.asarray(w["t"]).view(np.uint32))
        ok = ok and (any_hit or np.array_equal(a["prim"], w["prim"])) and w["bvh4"]["stack_seen"] <= w["bvh4"]["stack_bound"] <= 31
    return ok, recipe, segs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    from mitsuba2_amd import api, scenes, build
    build.build_all(oracle=True)
    api.host_lib()
    import oracle_py
    orc = oracle_py.load()
    bad = 0
    for seed in range(a.first, a.first + a.seeds):
        try:
            ok, recipe, segs = run_case(api, scenes, orc, seed)
        except Exception as e:                                     # a scene the host layer rejects is a recipe problem, not a parity failure
            print("seed %d: %s: %s" % (seed, type(e).__name__, str(e)[:200]))
            if a.verbose:
                traceback.print_exc()
            continue
        if not ok:
            bad += 1
            print("seed %d MISMATCH (segments %s)\n    %s" % (seed, segs, "\n    ".join(recipe)))
        elif a.verbose:
            print("seed %d ok  %s" % (seed, recipe[-1]))
    print("%d seeds, %d mismatches" % (a.seeds, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
