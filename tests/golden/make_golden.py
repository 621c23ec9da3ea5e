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
