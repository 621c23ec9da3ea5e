"""Regenerates tests/golden/*.npz from the CPU oracle (run from the repo root:
python tests/golden/make_golden.py). The reference itself cannot be built or imported
here (ext/enoki, ext/tbb, ... are empty submodules), so these fixtures pin the
*restatement's* output: any later change to the leaf arithmetic, the RNG draw ledger or
the film order shows up as a diff against them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from mitsuba2_amd import api, scenes, build  # noqa: E402

build.build_all(oracle=True)
import oracle_py  # noqa: E402

O = oracle_py.load()
scene, sensor = scenes.cornell_box(48, 32, 4, device=-1)
film, _, st = O.render(scene.desc(), api.PathIntegrator().render_job(sensor), threads=2, want_f64=False)
scene, sensor = scenes.cornell_box(32, 24, 4, diffuse_only=False, device=-1, ball_level=1)
film_m, _, st_m = O.render(scene.desc(), api.PathIntegrator().render_job(sensor), threads=2, want_f64=False)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cornell_48x32_4spp.npz"), film=film, segments=st.segments,
                    film_materials=film_m, segments_materials=st_m.segments)
print("wrote golden fixtures:", film.shape, st.segments, film_m.shape, st_m.segments)

# round-1 additions: direct integrator, textured scene (texture coordinates + bitmap), moment squares pass.
# A second file, so that the first one stays byte-identical to what the earlier sessions wrote.
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_textures import textured_quad, quad_sensor, checker  # noqa: E402

scene, sensor = scenes.cornell_box(32, 24, 4, diffuse_only=False, device=-1, ball_level=1)
film_d, _, st_d = O.render(scene.desc(), api.DirectIntegrator(emitter_samples=2, bsdf_samples=1).render_job(sensor), threads=2, want_f64=False)
film_q, _, st_q = O.render(scene.desc(), api.MomentIntegrator(api.PathIntegrator(max_depth=3)).render_job(sensor, moment_pass=2),
                           threads=2, want_f64=False)
tscene = api.Scene(textured_quad(api, api.BitmapTexture(checker(), wrap_mode="mirror"))).build(-1)
tsensor = quad_sensor(api, 32, 24, 4)
film_t, _, st_t = O.render(tscene.desc(), api.PathIntegrator(max_depth=4).render_job(tsensor), threads=2, want_f64=False)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "round1_plugins.npz"), film_direct=film_d, segments_direct=st_d.segments,
                    film_squares=film_q, segments_squares=st_q.segments, film_textured=film_t, segments_textured=st_t.segments)
print("wrote round1_plugins.npz:", st_d.segments, st_q.segments, st_t.segments)
