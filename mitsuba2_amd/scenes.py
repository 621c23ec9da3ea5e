"""Synthetic scenes for parity tests and the benchmark (the reference's data
submodule is absent): a classic Cornell box as triangles, a
material-ball scene (rough conductor + dielectric), and the `stairs` mesh of the
reference's kd-tree tests (src/librender/tests/mesh_generation.py:27-59).
"""
import numpy as np

from . import api

WHITE = (0.725, 0.71, 0.68)
RED = (0.63, 0.065, 0.05)
GREEN = (0.14, 0.45, 0.091)
LIGHT_RADIANCE = (17.0, 12.0, 4.0)


def _quad(pts, inward_point=None, outward_point=None):
    """4 corner points -> (vertices[4,3], faces[2,3]); winding chosen so that the geometric
    normal cross(p1-p0, p2-p0) faces `inward_point` (or away from `outward_point`)."""
    p = np.asarray(pts, np.float32)
    n = np.cross(p[1] - p[0], p[2] - p[0])
    c = p.mean(0)
    flip = False
    if inward_point is not None:
        flip = np.dot(n, np.asarray(inward_point, np.float32) - c) < 0
    if outward_point is not None:
        flip = np.dot(n, c - np.asarray(outward_point, np.float32)) < 0
    if flip:
        p = p[::-1].copy()
    return p, np.array([[0, 1, 2], [0, 2, 3]], np.uint32)


def _merge(parts):
    vs, fs, base = [], [], 0
    for v, f in parts:
        vs.append(v); fs.append(f + base); base += len(v)
    return np.concatenate(vs), np.concatenate(fs)


_ROOM_CENTER = (278.0, 274.4, 279.6)
_CBOX = dict(
    floor=[(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)],
    ceiling=[(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)],
    back=[(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)],
    right=[(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)],
    left=[(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)],
    # 0.8 below the ceiling so that no light/ceiling triangles are coplanar (closest-hit ties)
    light=[(343, 548.0, 227), (343, 548.0, 332), (213, 548.0, 332), (213, 548.0, 227)],
)
_SHORT = [
    [(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)],
    [(290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)],
    [(130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)],
    [(82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)],
    [(240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)],
]
_TALL = [
    [(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)],
    [(423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)],
    [(472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)],
    [(314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)],
    [(265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)],
]


def _block(quads):
    c = np.mean([np.mean(q, 0) for q in quads], 0)
    return _merge([_quad(q, outward_point=c) for q in quads])


def icosphere(center, radius, level):
    """Subdivided icosahedron with exact unit vertex normals; 20 * 4**level faces."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    verts = [tuple(x) for x in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (np.asarray(verts[a]) + np.asarray(verts[b])) * 0.5
                m /= np.linalg.norm(m)
                cache[key] = len(verts); verts.append(tuple(m))
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = np.array(nf, np.int64)
    n = np.asarray(verts, np.float64)
    p = n * radius + np.asarray(center, np.float64)
    return p.astype(np.float32), f.astype(np.uint32), n.astype(np.float32)


def cornell_box_meshes(diffuse_only=True, ball_level=5, metal=None, glass=None, glass_block=False):
    """-> list of api.Mesh. diffuse_only: the classic box (36 triangles, config C2).
    Otherwise the short block becomes a GGX rough-conductor ball with shading normals
    and the tall block a dielectric (bk7) ball — the material-ball configuration (C3).
    glass_block (with diffuse_only): the tall block is a smooth dielectric (bk7, constant IOR — Mitsuba 2's
    dielectric has no dispersion) with a bottom face, so that it is a closed solid — config C5's geometry."""
    white = api.BSDF("diffuse", reflectance=WHITE)
    red = api.BSDF("diffuse", reflectance=RED)
    green = api.BSDF("diffuse", reflectance=GREEN)
    meshes = []
    for name, bsdf in (("floor", white), ("ceiling", white), ("back", white), ("right", green), ("left", red)):
        v, f = _quad(_CBOX[name], inward_point=_ROOM_CENTER)
        meshes.append(api.Mesh(name, v, f, bsdf=bsdf))
    v, f = _quad(_CBOX["light"], inward_point=_ROOM_CENTER)
    meshes.append(api.Mesh("light", v, f, emitter=api.AreaLight(LIGHT_RADIANCE)))   # default BSDF: diffuse 0
    if diffuse_only:
        v, f = _block(_SHORT); meshes.append(api.Mesh("short_block", v, f, bsdf=white))
        if glass_block:
            bottom = [(423, 0.05, 247), (472, 0.05, 406), (314, 0.05, 456), (265, 0.05, 296)]   # just above the floor: no coplanar ties
            v, f = _merge([_quad(q, outward_point=(368, 165, 351)) for q in _TALL + [bottom]])
            meshes.append(api.Mesh("tall_block", v, f, bsdf=api.BSDF("dielectric", int_ior=1.5046, ext_ior=1.000277)))
        else:
            v, f = _block(_TALL); meshes.append(api.Mesh("tall_block", v, f, bsdf=white))
    else:
        mkw = dict(distribution="ggx", alpha=0.1, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14))
        mkw.update(metal or {})
        if "alpha_u" in mkw:
            mkw.pop("alpha", None)
        metal = api.BSDF("roughconductor", **mkw)
        if glass is None:
            glass = api.BSDF("dielectric", int_ior=1.5046, ext_ior=1.000277)
        else:                                             # e.g. dict(plugin="roughdielectric", alpha=0.2, distribution="ggx")
            gkw = dict(int_ior=1.5046, ext_ior=1.000277); gkw.update(glass)
            glass = api.BSDF(gkw.pop("plugin", "roughdielectric"), **gkw)
        v, f, n = icosphere((185.0, 82.5, 169.0), 82.5, ball_level)
        meshes.append(api.Mesh("metal_ball", v, f, normals=n, bsdf=metal))
        v, f, n = icosphere((368.0, 110.0, 351.0), 110.0, ball_level)
        meshes.append(api.Mesh("glass_ball", v, f, normals=n, bsdf=glass))
    return meshes


def cornell_sensor(width, height, spp, seed=0, rfilter="gaussian", **film_kw):
    film = api.Film(rfilter=rfilter, width=width, height=height, **film_kw)
    sampler = api.Sampler(sample_count=spp, seed=seed)
    sensor = api.Sensor(film, sampler, fov=39.3,
                        to_world=dict(origin=(278, 273, -800), target=(278, 273, 0), up=(0, 1, 0)))
    return sensor


def cornell_box(width, height, spp, diffuse_only=True, seed=0, device=0, ball_level=5, rfilter="gaussian",
                metal=None, glass=None, glass_block=False, **film_kw):
    """-> (scene, sensor). device < 0 builds only the host-side description.
    `metal`: property overrides of the rough-conductor ball (e.g. dict(distribution="beckmann"))."""
    scene = api.Scene(cornell_box_meshes(diffuse_only, ball_level, metal, glass, glass_block)).build(device)
    return scene, cornell_sensor(width, height, spp, seed, rfilter, **film_kw)


def plugin_box_meshes():
    """Cornell box dressed with the smooth conductor, smooth plastic and twosided plugins: plastic floor
    (nonlinear) and back wall, a copper-like smooth conductor short block, the tall block replaced by a
    free-standing panel (one quad, seen from both sides) with a twosided material — red diffuse front, rough
    conductor back — and a second panel whose single twosided diffuse serves both sides."""
    white = api.BSDF("diffuse", reflectance=WHITE)
    red = api.BSDF("diffuse", reflectance=RED)
    green = api.BSDF("diffuse", reflectance=GREEN)
    floor = api.BSDF("plastic", diffuse_reflectance=(0.3, 0.45, 0.7), int_ior=1.49, ext_ior=1.000277, nonlinear=True)
    back = api.BSDF("plastic", diffuse_reflectance=WHITE, specular_reflectance=(0.9, 0.8, 0.7))
    copper = api.BSDF("conductor", eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), specular_reflectance=(0.95, 0.95, 0.95))
    rough = api.BSDF("roughconductor", distribution="ggx", alpha=0.25, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14))
    panel = api.TwoSided(red, rough)
    both = api.TwoSided(api.BSDF("diffuse", reflectance=(0.2, 0.7, 0.3)))
    meshes = []
    for name, bsdf in (("floor", floor), ("ceiling", white), ("back", back), ("right", green), ("left", red)):
        v, f = _quad(_CBOX[name], inward_point=_ROOM_CENTER)
        meshes.append(api.Mesh(name, v, f, bsdf=bsdf))
    v, f = _quad(_CBOX["light"], inward_point=_ROOM_CENTER)
    meshes.append(api.Mesh("light", v, f, emitter=api.AreaLight(LIGHT_RADIANCE)))
    v, f = _block(_SHORT); meshes.append(api.Mesh("short_block", v, f, bsdf=copper))
    quad = np.array([(300.0, 0.0, 420.0), (470.0, 0.0, 300.0), (470.0, 330.0, 300.0), (300.0, 330.0, 420.0)], np.float32)
    meshes.append(api.Mesh("panel", quad, np.array([(0, 1, 2), (0, 2, 3)], np.uint32), bsdf=panel))
    quad2 = np.array([(60.0, 0.0, 330.0), (150.0, 0.0, 460.0), (150.0, 220.0, 460.0), (60.0, 220.0, 330.0)], np.float32)
    meshes.append(api.Mesh("panel2", quad2, np.array([(0, 1, 2), (0, 2, 3)], np.uint32), bsdf=both))
    return meshes


def plugin_box(width, height, spp, seed=0, device=0, rfilter="gaussian", **film_kw):
    """-> (scene, sensor): the conductor / plastic / twosided test scene (plugin_box_meshes)"""
    scene = api.Scene(plugin_box_meshes()).build(device)
    return scene, cornell_sensor(width, height, spp, seed, rfilter, **film_kw)


def _rect(name, corners, inward_point, **kw):
    """api.Mesh.rectangle over the parallelogram corners[0] + s * (corners[1] - corners[0]) + t * (corners[3] - corners[0]),
    normal towards `inward_point`"""
    c = np.asarray(corners, np.float64)
    u, v = c[1] - c[0], c[3] - c[0]
    if np.dot(np.cross(u, v), np.asarray(inward_point, np.float64) - c.mean(0)) < 0:
        u, v = v, u
    return api.Mesh.rectangle(api.quad_to_world(c[0], u, v), name=name, **kw)


def rect_box_meshes():
    """The Cornell box with analytic rectangles (src/shapes/rectangle.cpp) for the walls and for the area light,
    triangle meshes for the two blocks, and a free-standing twosided rectangle seen from both sides."""
    white = api.BSDF("diffuse", reflectance=WHITE)
    red = api.BSDF("diffuse", reflectance=RED)
    green = api.BSDF("diffuse", reflectance=GREEN)
    shapes = []
    for name, bsdf in (("floor", white), ("ceiling", white), ("back", white), ("right", green), ("left", red)):
        shapes.append(_rect(name, _CBOX[name], _ROOM_CENTER, bsdf=bsdf))
    shapes.append(_rect("light", _CBOX["light"], _ROOM_CENTER, emitter=api.AreaLight(LIGHT_RADIANCE)))
    v, f = _block(_SHORT); shapes.append(api.Mesh("short_block", v, f, bsdf=white))
    v, f = _block(_TALL); shapes.append(api.Mesh("tall_block", v, f, bsdf=white))
    panel = api.TwoSided(api.BSDF("diffuse", reflectance=(0.2, 0.3, 0.8)), api.BSDF("diffuse", reflectance=(0.8, 0.7, 0.2)))
    shapes.append(_rect("panel", [(60, 0, 330), (150, 0, 460), (150, 220, 460), (60, 220, 330)], (0, 100, 600), bsdf=panel))
    return shapes


def rect_box(width, height, spp, seed=0, device=0, rfilter="gaussian", **film_kw):
    """-> (scene, sensor): analytic rectangles + meshes (rect_box_meshes)"""
    scene = api.Scene(rect_box_meshes()).build(device)
    return scene, cornell_sensor(width, height, spp, seed, rfilter, **film_kw)


def sphere_box_meshes():
    """Cornell box (mesh walls) lit by an analytic sphere light hanging under the ceiling, with an analytic glass sphere,
    a rough-conductor sphere (rotated, to exercise to_world) and the short block."""
    white = api.BSDF("diffuse", reflectance=WHITE)
    red = api.BSDF("diffuse", reflectance=RED)
    green = api.BSDF("diffuse", reflectance=GREEN)
    shapes = []
    for name, bsdf in (("floor", white), ("ceiling", white), ("back", white), ("right", green), ("left", red)):
        v, f = _quad(_CBOX[name], inward_point=_ROOM_CENTER)
        shapes.append(api.Mesh(name, v, f, bsdf=bsdf))
    shapes.append(api.Mesh.sphere((278, 470, 280), 40.0, emitter=api.AreaLight((30.0, 24.0, 12.0)), name="bulb"))
    v, f = _block(_SHORT); shapes.append(api.Mesh("short_block", v, f, bsdf=white))
    shapes.append(api.Mesh.sphere((368, 110, 351), 110.0, bsdf=api.BSDF("dielectric", int_ior=1.5046, ext_ior=1.000277), name="glass"))
    a = np.deg2rad(30.0)
    rot = np.array([[np.cos(a), 0, np.sin(a), 150.0], [0, 1, 0, 260.0], [-np.sin(a), 0, np.cos(a), 150.0], [0, 0, 0, 1]], np.float32)
    metal = api.BSDF("roughconductor", distribution="ggx", alpha=0.15, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14))
    shapes.append(api.Mesh.sphere((0, 0, 0), 60.0, to_world=rot, bsdf=metal, name="metal"))
    return shapes


def sphere_box(width, height, spp, seed=0, device=0, rfilter="gaussian", **film_kw):
    """-> (scene, sensor): analytic spheres (one of them the light) in the Cornell room (sphere_box_meshes)"""
    scene = api.Scene(sphere_box_meshes()).build(device)
    return scene, cornell_sensor(width, height, spp, seed, rfilter, **film_kw)


def sky_envmap(width=64, height=32, seed=1):
    """Synthetic lat-long HDR sky (the reference's data submodule is absent): vertical sky
    gradient, warm horizon band, dark ground, a sun blob and a little seeded noise. -> H x W x 3 float32."""
    rng = np.random.default_rng(seed)
    v = (np.arange(height, dtype=np.float64) + 0.0) / (height - 1)          # 0 = +Y pole
    u = np.arange(width, dtype=np.float64) / (width - 1)
    U, V = np.meshgrid(u, v)
    sky = np.stack([0.25 + 0.3 * V, 0.45 + 0.3 * V, 0.9 - 0.2 * V], 2)
    horizon = np.exp(-((V - 0.5) / 0.06) ** 2)[..., None] * np.array([0.9, 0.6, 0.3])
    ground = np.array([0.12, 0.10, 0.08])
    img = np.where((V < 0.5)[..., None], sky + horizon, ground + 0.3 * horizon)
    sun = np.exp(-(((U - 0.3) / 0.03) ** 2 + ((V - 0.28) / 0.04) ** 2))[..., None] * np.array([60.0, 52.0, 40.0])
    img = img + sun + 0.02 * rng.random((height, width, 3))
    return img.astype(np.float32)


def open_box(width, height, spp, seed=0, device=0, with_area_light=True, envmap_after=None, env_scale=1.0,
             env_size=(64, 32), ball_level=1, rfilter="gaussian", **film_kw):
    """Cornell box without its ceiling under the synthetic sky: area light + environment map (the emitter mix
    of config C4), rough-conductor and dielectric balls. -> (scene, sensor)."""
    meshes = [m for m in cornell_box_meshes(False, ball_level) if m.name not in ("ceiling",) and
              (with_area_light or m.name != "light")]
    env = api.EnvMap(sky_envmap(env_size[0], env_size[1]), scale=env_scale,
                     to_world=dict(origin=(0, 0, 0), target=(0.3, 0.1, 1.0), up=(0, 1, 0)))
    scene = api.Scene(meshes, envmap=env, envmap_after=envmap_after).build(device)
    return scene, cornell_sensor(width, height, spp, seed, rfilter, **film_kw)


def _displaced_grid(corners, n, amp, rng, inward_point):
    """n x n quad grid over the bilinear patch `corners` (4 points), vertices displaced along the patch normal by
    seeded noise of amplitude `amp`; winding facing `inward_point`. -> (vertices, faces)"""
    c = np.asarray(corners, np.float64)
    u = np.linspace(0, 1, n + 1)
    U, V = np.meshgrid(u, u)
    P = ((1 - U)[..., None] * (1 - V)[..., None] * c[0] + U[..., None] * (1 - V)[..., None] * c[1] +
         U[..., None] * V[..., None] * c[2] + (1 - U)[..., None] * V[..., None] * c[3])
    nrm = np.cross(c[1] - c[0], c[3] - c[0]); nrm /= np.linalg.norm(nrm)
    if np.dot(nrm, np.asarray(inward_point, np.float64) - c.mean(0)) < 0:
        nrm = -nrm
    bump = amp * (0.5 * np.sin(9 * np.pi * U) * np.cos(7 * np.pi * V) + 0.5 * (rng.random(U.shape) - 0.5))
    bump[0, :] = bump[-1, :] = 0; bump[:, 0] = bump[:, -1] = 0              # keep the seams closed
    P = P + bump[..., None] * nrm
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a, b, cc, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    f = np.concatenate([np.stack([a, b, cc], 1), np.stack([a, cc, d], 1)])
    v = P.reshape(-1, 3)
    tn = np.cross(v[f[0, 1]] - v[f[0, 0]], v[f[0, 2]] - v[f[0, 0]])
    if np.dot(tn, nrm) < 0:
        f = f[:, ::-1]
    return v.astype(np.float32), f.astype(np.uint32)


def interior_scene(width, height, spp, grid=256, n_clutter=200, clutter_level=3, seed=1234, device=0, env_size=(1024, 512),
                   rfilter="gaussian", **film_kw):
    """BASELINE config 4 class: a procedurally generated ~1 M-triangle interior — the Cornell
    room without its ceiling, every wall a displaced `grid` x `grid` mesh, `n_clutter` icospheres (diffuse / GGX
    conductor / dielectric, shading normals) scattered by a fixed-seed generator, one area light and the synthetic
    lat-long sky. grid=256, n_clutter=200, clutter_level=3 -> 655 360 + 256 000 + 2 = 911 362 triangles."""
    rng = np.random.default_rng(seed)
    white = api.BSDF("diffuse", reflectance=WHITE); red = api.BSDF("diffuse", reflectance=RED)
    green = api.BSDF("diffuse", reflectance=GREEN)
    meshes = []
    for name, bsdf in (("floor", white), ("back", white), ("right", green), ("left", red)):
        v, f = _displaced_grid(_CBOX[name], grid, 6.0, rng, _ROOM_CENTER)
        meshes.append(api.Mesh(name, v, f, bsdf=bsdf))
    v, f = _displaced_grid([(556, 548.8, -200), (0, 548.8, -200), (0, 548.8, 0), (556, 548.8, 0)], grid, 4.0, rng, (278, 0, -100))
    meshes.append(api.Mesh("awning", v, f, bsdf=white))
    v, f = _quad(_CBOX["light"], inward_point=_ROOM_CENTER)
    meshes.append(api.Mesh("light", v, f, emitter=api.AreaLight(LIGHT_RADIANCE)))
    mats = [white, red, green,
            api.BSDF("roughconductor", distribution="ggx", alpha=0.15, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14)),
            api.BSDF("roughconductor", distribution="beckmann", alpha=0.3, eta=(1.66, 0.88, 0.52), k=(9.2, 6.3, 4.8)),
            api.BSDF("dielectric", int_ior=1.5046, ext_ior=1.000277)]
    for i in range(n_clutter):
        r = float(rng.uniform(8, 28))
        c = (float(rng.uniform(40, 510)), r + float(rng.uniform(0, 260)) * float(rng.random() < 0.3), float(rng.uniform(40, 520)))
        v, f, n = icosphere(c, r, clutter_level)
        meshes.append(api.Mesh("clutter%d" % i, v, f, normals=n, bsdf=mats[i % len(mats)]))
    env = api.EnvMap(sky_envmap(env_size[0], env_size[1]), scale=1.0,
                     to_world=dict(origin=(0, 0, 0), target=(0.3, 0.1, 1.0), up=(0, 1, 0)))
    scene = api.Scene(meshes, envmap=env).build(device)
    return scene, cornell_sensor(width, height, spp, seed=0, rfilter=rfilter, **film_kw)


def stairs(num_steps):
    """src/librender/tests/mesh_generation.py:27-59"""
    size_step = 1.0 / num_steps
    v = np.zeros((4 * num_steps, 3)); f = np.zeros((4 * num_steps - 2, 3), np.uint32)
    for i in range(num_steps):
        h = i * size_step; s1 = i * size_step; s2 = (i + 1) * size_step; k = 4 * i
        v[k + 0] = [0.0, s1, h]; v[k + 1] = [1.0, s1, h]; v[k + 2] = [0.0, s2, h]; v[k + 3] = [1.0, s2, h]
        f[k] = [k, k + 1, k + 2]; f[k + 1] = [k + 1, k + 3, k + 2]
        if i < num_steps - 1:
            f[k + 2] = [k + 2, k + 3, k + 5]; f[k + 3] = [k + 5, k + 4, k + 2]
    return v.astype(np.float32), f


def random_triangles(n, seed=7, extent=1.0, size=0.05):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n, 1, 3))
    v = (c + rng.normal(0, size, (n, 3, 3))).reshape(-1, 3).astype(np.float32)
    f = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    return v, f
