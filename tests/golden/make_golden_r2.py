"""Regenerates tests/golden/round2.json (run from the repo root: python tests/golden/make_golden_r2.py).

Digests instead of films: sha256 of the float32 film bytes + the segment count of small renders by the CPU oracle — the scene
classes and plugins the first two fixture files do not cover (environment map + area light, analytic spheres and rectangles,
the conductor / plastic / twosided box, rough dielectric and rough plastic, Beckmann lobes, the interior-class scene with its
two emitters, the spectral variant's glass block) and twelve recipes of tools/fuzz_cpu.py. The oracle and the device share
their leaf headers, so an accidental edit of a leaf moves both and no parity test notices: these digests do."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def digest(film):
    return hashlib.sha256(np.ascontiguousarray(film, np.float32).tobytes()).hexdigest()


def rgb_cases(api, scenes):
    """-> [(key, scene, job)] of the scalar_rgb digests"""
    import fuzz_cpu
    out = []
    s, c = scenes.open_box(28, 20, 3, device=-1); out.append(("open_box", s, api.PathIntegrator().render_job(c)))
    s, c = scenes.open_box(24, 16, 2, device=-1, with_area_light=False, envmap_after=0, env_scale=0.5)
    out.append(("open_box_env_only", s, api.PathIntegrator(max_depth=4).render_job(c)))
    s, c = scenes.sphere_box(24, 20, 3, device=-1); out.append(("sphere_box", s, api.PathIntegrator().render_job(c)))
    s, c = scenes.rect_box(24, 20, 3, device=-1); out.append(("rect_box", s, api.PathIntegrator().render_job(c)))
    s, c = scenes.plugin_box(24, 20, 3, device=-1); out.append(("plugin_box", s, api.PathIntegrator().render_job(c)))
    s, c = scenes.cornell_box(24, 20, 3, diffuse_only=False, device=-1, ball_level=1, metal=dict(distribution="beckmann", alpha_u=0.08, alpha_v=0.3),
                              glass=dict(plugin="roughdielectric", alpha=0.2, distribution="ggx"))
    out.append(("beckmann_roughdielectric", s, api.PathIntegrator().render_job(c)))
    s, c = scenes.cornell_box(24, 20, 3, diffuse_only=False, device=-1, ball_level=1,
                              glass=dict(plugin="roughplastic", alpha=0.15, distribution="beckmann", diffuse_reflectance=(0.2, 0.5, 0.3), nonlinear=True))
    out.append(("roughplastic", s, api.PathIntegrator().render_job(c)))
    s, c = scenes.interior_scene(24, 16, 2, grid=12, n_clutter=6, clutter_level=1, device=-1, env_size=(32, 16))
    out.append(("interior_small", s, api.PathIntegrator().render_job(c)))
    s, c = scenes.cornell_box(24, 20, 2, device=-1, rfilter="mitchell", crop_offset_x=3, crop_offset_y=2, crop_width=17, crop_height=13)
    out.append(("crop_mitchell", s, api.PathIntegrator(rr_depth=2).render_job(c)))
    for seed in (3, 11, 42, 77, 105, 230, 1001, 1002, 1003, 1004, 1005, 1006):
        scene, sensor, ikw, recipe, keep = fuzz_cpu.make_case(api, scenes, seed)
        ikw = dict(ikw); ikw.pop("samples_per_pass", None)
        integ = api.DirectIntegrator if ikw.pop("integrator", "path") == "direct" else api.PathIntegrator
        out.append(("fuzz_%d" % seed, (scene, keep), integ(**ikw).render_job(sensor)))
    return out


def spectral_cases(api, scenes):
    s, c = scenes.cornell_box(20, 16, 2, diffuse_only=True, glass_block=True, device=-1)
    return [("spectral_glassblock", s, api.PathIntegrator().render_job(c))]


def compute(api, scenes, orc, cases):
    res = {}
    for key, scene, job in cases:
        sc = scene[0] if isinstance(scene, tuple) else scene
        film, _, st = orc.render(sc.desc(), job, threads=2, want_f64=False)
        res[key] = dict(sha256=digest(film), segments=int(st.segments), samples=int(st.samples), mean_y=float(np.asarray(film)[..., 1].mean()))
    return res


if __name__ == "__main__":
    from mitsuba2_amd import api, scenes, build
    build.build_all(oracle=True)
    api.host_lib()
    import oracle_py
    res = compute(api, scenes, oracle_py.load(), rgb_cases(api, scenes))
    api.set_variant("scalar_spectral"); api.set_srgb_model(api.default_srgb_coeff())
    res.update(compute(api, scenes, oracle_py.load("scalar_spectral"), spectral_cases(api, scenes)))
    api.set_variant("scalar_rgb")
    json.dump(res, open(os.path.join(ROOT, "tests", "golden", "round2.json"), "w"), indent=1, sort_keys=True)
    print("wrote round2.json:", len(res), "digests")
