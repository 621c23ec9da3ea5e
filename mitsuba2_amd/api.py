"""Thin Python objects over the host facade — test / benchmark plumbing.

The classes carry the reference's plugin names and property names
(`diffuse.reflectance`, `perspective.fov`, `hdrfilm.width`, `path.max_depth` …)
so that scripts written against Mitsuba 2's Python API read the same way; all
logic lives in libmiwave_host.so (C++) and libmiwave.so (HIP).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (mi_counters, mi_hits_soa, mi_rays_soa, mi_render_cfg, mi_scene_desc, c_float_p, c_u32_p)

_variant = "scalar_rgb"
_host = {}
_dev = {}


def set_variant(name):
    """mitsuba.set_variant(): 'scalar_rgb' (default) or 'scalar_spectral'. Like in the reference, objects
    belong to the variant they were created under (each variant is its own pair of native libraries)."""
    global _variant
    if name not in _capi.VARIANT_SUFFIX:
        raise ValueError("unknown variant %r (scalar_rgb, scalar_spectral)" % (name,))
    _variant = name


def variant():
    return _variant


def host_lib():
    if _variant not in _host:
        _host[_variant] = _capi.load_host_lib(_variant)
    return _host[_variant]


def device_lib():
    if _variant not in _dev:
        _dev[_variant] = _capi.load_device_lib(_variant)
    return _dev[_variant]


def set_srgb_model(path):
    """scalar_spectral: where the reference's data/srgb.coeff lives (src/librender/srgb.cpp:20-27)"""
    if host_lib().mih_set_srgb_model(str(path).encode()) != 0:
        raise RuntimeError(_err())


def _err():
    return host_lib().mih_last_error().decode()


def _fp(a):
    return a.ctypes.data_as(c_float_p)


class Properties:
    """Properties(plugin_name, **values): float / int / bool / str / 3-tuple (rgb) values;
    `to_world=dict(origin=, target=, up=)` becomes a look-at transform, a 4x4 numpy array a <matrix>."""

    def __init__(self, plugin, **kw):
        L = host_lib()
        self.h = L.mih_props_create(plugin.encode())
        for k, v in kw.items():
            n = k.encode()
            if isinstance(v, bool):
                L.mih_props_set_bool(self.h, n, int(v))
            elif isinstance(v, int):
                L.mih_props_set_int(self.h, n, v)
            elif isinstance(v, float):
                L.mih_props_set_float(self.h, n, v)
            elif isinstance(v, str):
                L.mih_props_set_string(self.h, n, v.encode())
            elif isinstance(v, np.ndarray) and v.shape == (4, 4):
                m = np.ascontiguousarray(v, np.float32)
                L.mih_props_set_matrix(self.h, n, _fp(m))
            elif isinstance(v, BitmapTexture):
                L.mih_props_set_texture(self.h, n, v.h)
                self._keep = getattr(self, "_keep", []) + [v]
            elif isinstance(v, dict):
                o = np.asarray(v["origin"], np.float32); t = np.asarray(v["target"], np.float32)
                u = np.asarray(v.get("up", (0, 1, 0)), np.float32)
                L.mih_props_set_lookat(self.h, n, _fp(o), _fp(t), _fp(u))
            else:
                r, g, b = [float(x) for x in v]
                L.mih_props_set_color(self.h, n, r, g, b)

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_props_destroy(self.h)
            self.h = None


class BitmapTexture:
    """<texture type="bitmap"> (src/textures/bitmap.cpp): BitmapTexture(pixels, filter_type="bilinear" | "nearest",
    wrap_mode="repeat" | "mirror" | "clamp", raw=False, to_uv=4x4) with `pixels` a (h, w) or (h, w, 3) array of linear
    floats, or BitmapTexture(filename="x.pfm", ...). Pass it where a BSDF takes a colour: BSDF("diffuse", reflectance=tex).
    Needs texture coordinates on the mesh (Mesh(..., texcoords=)), or an analytic shape's own uv."""

    def __init__(self, pixels=None, **kw):
        self._p = Properties("bitmap", **kw)
        w = h = c = 0; data = None
        if pixels is not None:
            a = np.ascontiguousarray(pixels, np.float32)
            if a.ndim == 2:
                a = a[..., None]
            if a.ndim != 3 or a.shape[2] not in (1, 3):
                raise ValueError("BitmapTexture: expected an (h, w) or (h, w, 3) array")
            h, w, c = a.shape; data = _fp(a); self._pixels = a
        self.h = host_lib().mih_bitmap_create(self._p.h, w, h, c, data)
        if not self.h:
            raise RuntimeError(_err())

    def info(self):
        """-> (width, height, channels), per-channel mean"""
        whc = np.zeros(3, np.uint32); mean = np.zeros(3, np.float32)
        if host_lib().mih_bitmap_info(self.h, whc.ctypes.data_as(c_u32_p), _fp(mean)) != 0:
            raise RuntimeError(_err())
        return tuple(int(x) for x in whc), mean

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_bitmap_destroy(self.h); self.h = None


class BSDF:
    def __init__(self, plugin, **kw):
        self._p = Properties(plugin, **kw)
        self.h = host_lib().mih_bsdf_create(self._p.h)
        if not self.h:
            raise RuntimeError(_err())

    def record(self):
        r = _capi.mi_bsdf()
        host_lib().mih_bsdf_record(self.h, C.byref(r))
        return r

    def flags(self):
        return host_lib().mih_bsdf_flags(self.h)

    def table(self):
        """the float table the plugin precomputes (roughplastic: external transmittance over mu = i / 63), or empty"""
        n = host_lib().mih_bsdf_table(self.h, None, 0)
        out = np.zeros(n, np.float32)
        if n:
            host_lib().mih_bsdf_table(self.h, _fp(out), n)
        return out

    def sample(self, wi, sample1, sample2):
        """-> dict(wo, pdf, eta, sampled_type, weight)  (BSDF::sample, local frame)"""
        wi = np.asarray(wi, np.float32); s2 = np.asarray(sample2, np.float32); out = np.zeros(9, np.float32)
        if host_lib().mih_bsdf_sample(self.h, _fp(wi), float(sample1), _fp(s2), _fp(out)) != 0:
            raise RuntimeError(_err())
        return dict(wo=out[0:3].copy(), pdf=out[3], eta=out[4], sampled_type=int(out[5:6].view(np.uint32)[0]),
                    weight=out[6:9].copy())

    def eval_pdf(self, wi, wo):
        wi = np.asarray(wi, np.float32); wo = np.asarray(wo, np.float32); out = np.zeros(4, np.float32)
        if host_lib().mih_bsdf_eval_pdf(self.h, _fp(wi), _fp(wo), _fp(out)) != 0:
            raise RuntimeError(_err())
        return out[0:3].copy(), out[3]

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_bsdf_destroy(self.h); self.h = None


class TwoSided(BSDF):
    """<bsdf type="twosided">: `front` on both sides, or `front` / `back` (src/bsdfs/twosided.cpp)"""
    def __init__(self, front, back=None):
        self._front, self._back = front, back            # keep the nested handles alive
        self.h = host_lib().mih_bsdf_create_twosided(front.h, back.h if back is not None else None)
        if not self.h:
            raise RuntimeError(_err())


def fresnel_diffuse_reflectance(eta):
    return float(host_lib().mih_fresnel_diffuse_reflectance(float(eta)))


def quad_to_world(corner, edge_u, edge_v):
    """4x4 to_world that maps the rectangle plugin's [-1, 1]^2 onto the parallelogram corner + s * edge_u + t * edge_v
    (s, t in [0, 1]); +z maps to normalize(edge_u x edge_v)."""
    c = np.asarray(corner, np.float64); u = np.asarray(edge_u, np.float64) / 2; v = np.asarray(edge_v, np.float64) / 2
    n = np.cross(u, v); n /= np.linalg.norm(n)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = u, v, n, c + u + v
    return m.astype(np.float32)


class AreaLight:
    def __init__(self, radiance):
        self._p = Properties("area", radiance=tuple(radiance))
        self.h = host_lib().mih_emitter_create(self._p.h)
        if not self.h:
            raise RuntimeError(_err())

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_emitter_destroy(self.h); self.h = None


class Mesh:
    def __init__(self, name, vertices, faces, normals=None, bsdf=None, emitter=None, texcoords=None):
        self.name = name
        self.vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        self.faces = np.ascontiguousarray(faces, np.uint32).reshape(-1, 3)
        self.normals = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        self.texcoords = None if texcoords is None else np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
        self.h = host_lib().mih_mesh_create(name.encode(), _fp(self.vertices), len(self.vertices),
                                            self.faces.ctypes.data_as(c_u32_p), len(self.faces),
                                            None if self.normals is None else _fp(self.normals),
                                            None if self.texcoords is None else _fp(self.texcoords))
        if not self.h:
            raise RuntimeError(_err())
        self.bsdf, self.emitter = bsdf, emitter
        if bsdf is not None:
            host_lib().mih_mesh_set_bsdf(self.h, bsdf.h)
        if emitter is not None:
            host_lib().mih_mesh_set_emitter(self.h, emitter.h)

    @classmethod
    def load(cls, filename, bsdf=None, emitter=None, **kw):
        """<shape type="obj"|"ply"> (src/shapes/obj.cpp, ply.cpp): properties face_normals, flip_tex_coords, to_world"""
        kind = {".obj": 0, ".ply": 1}.get(str(filename)[-4:].lower())
        if kind is None:
            raise ValueError("Mesh.load: expected a .obj or .ply file")
        props = Properties("obj" if kind == 0 else "ply", filename=str(filename), **kw)
        h = host_lib().mih_mesh_load(kind, props.h)
        if not h:
            raise RuntimeError(_err())
        self = cls.__new__(cls)
        self.h, self.name = h, str(filename)
        self._sync()
        self.bsdf, self.emitter = bsdf, emitter
        if bsdf is not None:
            host_lib().mih_mesh_set_bsdf(self.h, bsdf.h)
        if emitter is not None:
            host_lib().mih_mesh_set_emitter(self.h, emitter.h)
        return self

    @classmethod
    def rectangle(cls, to_world=None, flip_normals=False, bsdf=None, emitter=None, name="rectangle"):
        """<shape type="rectangle"> (src/shapes/rectangle.cpp): the analytic [-1, 1]^2 quad in z = 0 placed by the 4x4
        `to_world` — one primitive, intersected and sampled analytically (not two triangles)."""
        kw = dict(flip_normals=bool(flip_normals))
        if to_world is not None:
            kw["to_world"] = np.asarray(to_world, np.float32).reshape(4, 4)
        props = Properties("rectangle", **kw)
        h = host_lib().mih_rectangle_create(props.h)
        if not h:
            raise RuntimeError(_err())
        self = cls.__new__(cls)
        self.h, self.name = h, name
        self._sync()
        self.bsdf, self.emitter = bsdf, emitter
        if bsdf is not None:
            host_lib().mih_mesh_set_bsdf(self.h, bsdf.h)
        if emitter is not None:
            host_lib().mih_mesh_set_emitter(self.h, emitter.h)
        return self

    @classmethod
    def sphere(cls, center=(0, 0, 0), radius=1.0, to_world=None, flip_normals=False, bsdf=None, emitter=None, name="sphere"):
        """<shape type="sphere"> (src/shapes/sphere.cpp): the analytic sphere — one primitive"""
        kw = dict(center=tuple(float(x) for x in center), radius=float(radius), flip_normals=bool(flip_normals))
        if to_world is not None:
            kw["to_world"] = np.asarray(to_world, np.float32).reshape(4, 4)
        props = Properties("sphere", **kw)
        h = host_lib().mih_sphere_create(props.h)
        if not h:
            raise RuntimeError(_err())
        self = cls.__new__(cls)
        self.h, self.name = h, name
        self._sync()
        self.bsdf, self.emitter = bsdf, emitter
        if bsdf is not None:
            host_lib().mih_mesh_set_bsdf(self.h, bsdf.h)
        if emitter is not None:
            host_lib().mih_mesh_set_emitter(self.h, emitter.h)
        return self

    def _sync(self):
        nv, nf, hn = C.c_uint32(), C.c_uint32(), C.c_int32()
        host_lib().mih_mesh_counts(self.h, C.byref(nv), C.byref(nf), C.byref(hn))
        self.vertices = np.zeros((nv.value, 3), np.float32); self.faces = np.zeros((nf.value, 3), np.uint32)
        self.normals = np.zeros((nv.value, 3), np.float32) if hn.value & 1 else None
        self.texcoords = np.zeros((nv.value, 2), np.float32) if hn.value & 2 else None
        host_lib().mih_mesh_copy(self.h, _fp(self.vertices), self.faces.ctypes.data_as(c_u32_p),
                                 None if self.normals is None else _fp(self.normals))
        if self.texcoords is not None:
            host_lib().mih_mesh_copy_texcoords(self.h, _fp(self.texcoords))

    def bbox(self):
        """Shape::bbox() -> (min xyz, max xyz)"""
        out = np.zeros(7, np.float32); host_lib().mih_mesh_bbox_area(self.h, _fp(out))
        return out[0:3].copy(), out[3:6].copy()

    def surface_area(self):
        """Shape::surface_area()"""
        out = np.zeros(7, np.float32); host_lib().mih_mesh_bbox_area(self.h, _fp(out))
        return float(out[6])

    def recompute_vertex_normals(self):
        """Mesh::recompute_vertex_normals (src/librender/mesh.cpp:200-246)"""
        if host_lib().mih_mesh_recompute_normals(self.h) != 0:
            raise RuntimeError(_err())
        self._sync()

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_mesh_destroy(self.h); self.h = None


def _rays_struct(o, d, mint, maxt):
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    n = len(o)
    cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
    cols.append(np.ascontiguousarray(np.broadcast_to(np.asarray(mint, np.float32), (n,))))
    cols.append(np.ascontiguousarray(np.broadcast_to(np.asarray(maxt, np.float32), (n,))))
    r = mi_rays_soa(*[_fp(c) for c in cols])
    return r, cols, n


def _hits_struct(n):
    t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    prim = np.zeros(n, np.uint32); shape = np.zeros(n, np.uint32)
    h = mi_hits_soa(_fp(t), _fp(u), _fp(v), prim.ctypes.data_as(c_u32_p), shape.ctypes.data_as(c_u32_p))
    return h, dict(t=t, u=u, v=v, prim=prim, shape=shape)


class EnvMap:
    """<emitter type="envmap"> (src/emitters/envmap.cpp): `rgb` = H x W x 3 (or x 4) linear float pixels in
    lat-long layout, row 0 = +Y pole; properties `scale`, `to_world`."""

    def __init__(self, rgb, **kw):
        a = np.asarray(rgb, np.float32)
        if a.ndim != 3 or a.shape[2] not in (3, 4):
            raise ValueError("EnvMap: expected an H x W x 3|4 array")
        if a.shape[2] == 3:
            a = np.concatenate([a, np.ones(a.shape[:2] + (1,), np.float32)], 2)
        self.rgba = np.ascontiguousarray(a, np.float32)
        self._p = Properties("envmap", **kw)
        self.h = host_lib().mih_envmap_create(self._p.h, a.shape[1], a.shape[0], _fp(self.rgba))
        if not self.h:
            raise RuntimeError(_err())

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_envmap_destroy(self.h); self.h = None


class Scene:
    def bbox(self):
        """Scene::bbox() -> (min xyz, max xyz)"""
        out = np.zeros(6, np.float32); host_lib().mih_scene_bbox(self.h, _fp(out))
        return out[0:3].copy(), out[3:6].copy()

    def __init__(self, shapes, envmap=None, envmap_after=None):
        """`envmap`: an EnvMap child; `envmap_after` = number of shapes listed before it in the scene
        (default: after all shapes) — fixes its position in the emitter order (scene.cpp:38-60)."""
        self.shapes = list(shapes)
        self.envmap = envmap
        self.h = host_lib().mih_scene_create()
        pos = len(self.shapes) if envmap_after is None else envmap_after
        for i, s in enumerate(self.shapes):
            if envmap is not None and i == pos and host_lib().mih_scene_add_envmap(self.h, envmap.h) != 0:
                raise RuntimeError(_err())
            if host_lib().mih_scene_add_shape(self.h, s.h) != 0:
                raise RuntimeError(_err())
        if envmap is not None and pos >= len(self.shapes) and host_lib().mih_scene_add_envmap(self.h, envmap.h) != 0:
            raise RuntimeError(_err())
        self.device = None

    def build(self, device=0, bvh_quality=1):
        """device < 0: flatten only (host-side description, no GPU)."""
        if host_lib().mih_scene_build(self.h, device, bvh_quality) != 0:
            raise RuntimeError(_err())
        self.device = device
        return self

    def desc(self):
        return host_lib().mih_scene_desc(self.h)

    def ctx(self):
        return host_lib().mih_scene_ctx(self.h)

    def ray_intersect_preliminary(self, o, d, mint=0.0, maxt=np.inf):
        """Scene::ray_intersect_preliminary for a batch -> dict(t,u,v,prim,shape)"""
        r, keep, n = _rays_struct(o, d, mint, maxt)
        h, out = _hits_struct(n)
        if host_lib().mih_scene_ray_intersect(self.h, C.byref(r), C.byref(h), n) != 0:
            raise RuntimeError(_err())
        return out

    def ray_intersect(self, o, d, mint=0.0, maxt=np.inf):
        """Scene::ray_intersect for a batch -> structured array of SurfaceInteraction3f records (_capi.SI_DTYPE:
        t, p, n, sh_s, sh_t, sh_n, uv, wi, prim_index, shape_index, emitter_index)"""
        r, keep, n = _rays_struct(o, d, mint, maxt)
        si = np.zeros(n, _capi.SI_DTYPE)
        if host_lib().mih_scene_ray_intersect_si(self.h, C.byref(r), si.ctypes.data_as(C.POINTER(_capi.mi_surface_interaction)), n) != 0:
            raise RuntimeError(_err())
        return si

    def ray_intersect_one(self, ray8):
        """The one-ray C++ overload Scene::ray_intersect(Ray3f) -> (valid, record, has_bsdf, emitter index of si.emitter(scene))"""
        r = np.ascontiguousarray(ray8, np.float32)
        si = np.zeros(1, _capi.SI_DTYPE); hb = C.c_int32(0); ei = C.c_int32(0)
        rc = host_lib().mih_scene_ray_intersect_one(self.h, _fp(r), si.ctypes.data_as(C.POINTER(_capi.mi_surface_interaction)), C.byref(hb), C.byref(ei))
        if rc < 0:
            raise RuntimeError(_err())
        return bool(rc), si[0], bool(hb.value), int(ei.value)

    def emitter_count(self):
        return host_lib().mih_scene_emitter_count(self.h)

    def sample_emitter_direction(self, ref_p, sample, test_visibility=True, emitter=-1, wavelengths=None):
        """Scene::sample_emitter_direction(ref, sample, test_visibility) (emitter = -1) or
        Scene::emitters()[emitter].sample_direction(ref, sample) for ONE reference point -> (record, spectrum)"""
        ds = np.zeros(1, _capi.DS_DTYPE); spec = np.zeros(4, np.float32)
        wl = None if wavelengths is None else np.ascontiguousarray(wavelengths, np.float32)
        rc = host_lib().mih_scene_sample_emitter_direction(self.h, int(emitter), _fp(np.ascontiguousarray(ref_p, np.float32)),
                                                           None if wl is None else _fp(wl), _fp(np.ascontiguousarray(sample, np.float32)),
                                                           int(bool(test_visibility)), ds.ctypes.data_as(C.POINTER(_capi.mi_direction_sample)), _fp(spec))
        if rc != 0:
            raise RuntimeError(_err())
        return ds[0], spec[:host_lib().mih_spectrum_channels()].copy()

    def pdf_emitter_direction(self, ref_p, ds, emitter=-1):
        d = np.array([ds], _capi.DS_DTYPE); pdf = C.c_float(0)
        if host_lib().mih_scene_pdf_emitter_direction(self.h, int(emitter), _fp(np.ascontiguousarray(ref_p, np.float32)),
                                                      d.ctypes.data_as(C.POINTER(_capi.mi_direction_sample)), C.byref(pdf)) != 0:
            raise RuntimeError(_err())
        return float(pdf.value)

    def emitter_eval(self, si, emitter=-1, wavelengths=None):
        """si.emitter(scene).eval(si) (emitter = -1; zero spectrum and found = False when the interaction sees none) or
        Scene::emitters()[emitter].eval(si) -> (found, spectrum)"""
        x = np.array([si], _capi.SI_DTYPE); spec = np.zeros(4, np.float32)
        wl = None if wavelengths is None else np.ascontiguousarray(wavelengths, np.float32)
        rc = host_lib().mih_scene_emitter_eval(self.h, int(emitter), x.ctypes.data_as(C.POINTER(_capi.mi_surface_interaction)),
                                               None if wl is None else _fp(wl), _fp(spec))
        if rc < 0:
            raise RuntimeError(_err())
        return bool(rc), spec[:host_lib().mih_spectrum_channels()].copy()

    def ray_test(self, o, d, mint=0.0, maxt=np.inf):
        r, keep, n = _rays_struct(o, d, mint, maxt)
        t = np.zeros(n, np.float32)
        if host_lib().mih_scene_ray_test(self.h, C.byref(r), _fp(t), n) != 0:
            raise RuntimeError(_err())
        return np.isfinite(t)

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_scene_destroy(self.h); self.h = None


class Film:
    def __init__(self, rfilter="gaussian", **kw):
        self._p = Properties("hdrfilm", **kw)
        self.h = host_lib().mih_film_create(self._p.h)
        if not self.h:
            raise RuntimeError(_err())
        if rfilter != "gaussian":
            if host_lib().mih_film_set_filter(self.h, rfilter.encode(), None) != 0:
                raise RuntimeError(_err())

    def crop_size(self):
        w, h = C.c_int32(), C.c_int32()
        host_lib().mih_film_crop_size(self.h, C.byref(w), C.byref(h))
        return w.value, h.value

    def rfilter(self, x):
        """the film's reconstruction filter at x -> dict(eval, eval_discretized, radius, border_size)"""
        out = np.zeros(4, np.float32)
        if host_lib().mih_film_filter_eval(self.h, float(x), _fp(out)) != 0:
            raise RuntimeError(_err())
        return dict(eval=float(out[0]), eval_discretized=float(out[1]), radius=float(out[2]), border_size=int(out[3]))

    def set_data(self, xyzaw):
        """fill the film's X, Y, Z, A, W storage (what mi_render writes)"""
        a = np.ascontiguousarray(xyzaw, np.float32)
        if host_lib().mih_film_set_data(self.h, _fp(a), a.size) != 0:
            raise RuntimeError(_err())

    def develop(self):
        """HDRFilm::bitmap(): W-normalised XYZ -> linear sRGB, H x W x 3 (hdrfilm.cpp:251-322)"""
        w, h = self.crop_size()
        out = np.zeros((h, w, 3), np.float32)
        if host_lib().mih_film_develop_rgb(self.h, _fp(out)) != 0:
            raise RuntimeError(_err())
        return out

    def develop_to(self, filename):
        """HDRFilm::set_destination_file + develop (hdrfilm.cpp:213-217,327-345) -> path of the file written"""
        r = host_lib().mih_film_develop(self.h, str(filename).encode())
        if not r:
            raise RuntimeError(_err())
        return r.decode()

    def data(self, shape):
        n = C.c_uint64()
        p = host_lib().mih_film_data(self.h, C.byref(n))
        a = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        return a.reshape(shape)

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_film_destroy(self.h); self.h = None


class Sampler:
    def __init__(self, **kw):
        self._p = Properties("independent", **kw)
        self.h = host_lib().mih_sampler_create(self._p.h)

    def seed(self, off):
        host_lib().mih_sampler_seed(self.h, off)

    def next_1d(self):
        return host_lib().mih_sampler_next_1d(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_sampler_destroy(self.h); self.h = None


class Sensor:
    def __init__(self, film, sampler, **kw):
        self.film, self.sampler = film, sampler
        self._p = Properties("perspective", **kw)
        self.h = host_lib().mih_sensor_create(self._p.h, film.h, sampler.h)
        if not self.h:
            raise RuntimeError(_err())

    def sample_ray(self, x, y):
        out = np.zeros(8, np.float32)
        if host_lib().mih_sensor_sample_ray(self.h, float(x), float(y), _fp(out)) != 0:
            raise RuntimeError(_err())
        return out

    def x_fov(self):
        return host_lib().mih_sensor_x_fov(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_sensor_destroy(self.h); self.h = None


def _wrap(cls, handle, **attrs):
    obj = cls.__new__(cls)
    obj.h = handle
    for k, v in attrs.items():
        setattr(obj, k, v)
    return obj


def _load_xml(text_or_path, is_file, params):
    hs = [C.c_void_p() for _ in range(5)]
    ps = "\n".join("%s=%s" % (k, v) for k, v in params.items())
    if host_lib().mih_load_xml(str(text_or_path).encode(), int(is_file), ps.encode(), *[C.byref(h) for h in hs]) != 0:
        raise RuntimeError(_err())
    scene = _wrap(Scene, hs[0], shapes=[], envmap=None, device=None)
    sensor = None
    if hs[1]:
        sensor = _wrap(Sensor, hs[1], film=_wrap(Film, hs[2]), sampler=_wrap(Sampler, hs[3]))
    return scene, sensor, _wrap(SamplingIntegrator, hs[4])


def load_string(xml, **params):
    """mitsuba.core.xml.load_string for the subset of host/miwave_host.h (scene, shape obj/ply/rectangle, bsdf, ref,
    area emitter, perspective sensor + hdrfilm + independent sampler + rfilter, path integrator, $parameters).
    -> (scene, sensor or None, integrator); call scene.build(device) before rendering."""
    return _load_xml(xml, False, params)


def load_file(path, **params):
    """mitsuba.core.xml.load_file (same subset); relative mesh file names resolve against the scene file's directory."""
    return _load_xml(path, True, params)


class RenderJob:
    """The host-side job description (mi_render_cfg + its tables), kept alive together."""

    def __init__(self, cfg, block_ids, tiles):
        self.cfg, self.block_ids, self.tiles = cfg, block_ids, tiles


class SamplingIntegrator:
    """include/mitsuba/render/integrator.h: what every sampling integrator shares; `plugin` names the sample() routine"""
    plugin = None

    def __init__(self, **kw):
        self._p = Properties(self.plugin, **kw)
        self.h = host_lib().mih_integrator_create(self._p.h)
        if not self.h:
            raise RuntimeError(_err())

    def set_shard(self, rank, world):
        host_lib().mih_integrator_set_shard(self.h, rank, world)

    def set_profile(self, on=True):
        host_lib().mih_integrator_set_profile(self.h, int(on))

    def set_plan(self, plan):
        """0 auto / 1 wavefront (HBM queues) / 2 resident (registers + LDS)"""
        host_lib().mih_integrator_set_plan(self.h, int(plan))

    def render(self, scene, sensor):
        """SamplingIntegrator::render -> True (finished) / False (cancelled, timeout)"""
        r = host_lib().mih_integrator_render(self.h, scene.h, sensor.h)
        if r < 0:
            raise RuntimeError(_err())
        return bool(r)

    def counters(self):
        c = mi_counters()
        host_lib().mih_integrator_counters(self.h, C.byref(c))
        return c

    def aov_names(self):
        buf = C.create_string_buffer(1024)
        n = host_lib().mih_integrator_aov_names(self.h, buf, 1024)
        if n < 0:
            raise RuntimeError(_err())
        return [x for x in buf.value.decode().split(",") if x]

    def pass_count(self, sensor):
        """sample_count / samples_per_pass (integrator.cpp:75-86)"""
        n = host_lib().mih_integrator_pass_count(self.h, sensor.h)
        if n < 0:
            raise RuntimeError(_err())
        return n

    def render_job(self, sensor, n_threads=1, capacity=1 << 16, pass_index=0):
        """The job of pass `pass_index` (execution order; the first pass carries the highest block-id offset, spiral.cpp:41; passes after the first accumulate onto the film)"""
        cfg = mi_render_cfg()
        block_ids = np.zeros(capacity, np.uint32); tiles = np.zeros(capacity, np.uint32)
        if host_lib().mih_make_render_cfg_pass(self.h, sensor.h, C.byref(cfg), block_ids.ctypes.data_as(c_u32_p),
                                               tiles.ctypes.data_as(c_u32_p), capacity, n_threads, pass_index) != 0:
            raise RuntimeError(_err())
        return RenderJob(cfg, block_ids, tiles)

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().mih_integrator_destroy(self.h); self.h = None


class PathIntegrator(SamplingIntegrator):
    """src/integrators/path.cpp: max_depth, rr_depth (+ block_size, samples_per_pass, timeout)"""
    plugin = "path"


class DirectIntegrator(SamplingIntegrator):
    """src/integrators/direct.cpp: shading_samples | emitter_samples + bsdf_samples, hide_emitters"""
    plugin = "direct"


class MomentIntegrator(SamplingIntegrator):
    """src/integrators/moment.cpp around one nested integrator (scalar_rgb): render() fills a film with the channels
    X Y Z A W <name>.X <name>.Y <name>.Z m2_<name>.X m2_<name>.Y m2_<name>.Z (Film.data(11)). Through the raw C ABI
    the two halves are two mi_render calls: render_job() is the nested integrator's job, with cfg.moment_pass = 1 / 2."""
    plugin = "moment"

    def __init__(self, nested, name="integrator", **kw):
        self._p = Properties("moment", **kw)
        self.nested = nested
        self.h = host_lib().mih_integrator_create_moment(self._p.h, nested.h, name.encode())
        if not self.h:
            raise RuntimeError(_err())

    def render_job(self, sensor, n_threads=1, capacity=1 << 16, pass_index=0, moment_pass=1):
        job = self.nested.render_job(sensor, n_threads, capacity, pass_index)
        job.cfg.moment_pass = moment_pass
        return job


def gauss_legendre(n):
    """quad::gauss_legendre (src/libcore/quad.cpp:7-64) -> nodes, weights"""
    x = np.zeros(n, np.float32); w = np.zeros(n, np.float32)
    host_lib().mih_gauss_legendre(n, _fp(x), _fp(w))
    return x, w


def spiral(w, h, block_size, offset=(0, 0)):
    n = ((w + block_size - 1) // block_size) * ((h + block_size - 1) // block_size)
    out = np.zeros((n, 5), np.int32)
    host_lib().mih_spiral(w, h, offset[0], offset[1], block_size, out.ctypes.data_as(_capi.c_i32_p), n)
    return out


# ---- direct C-ABI helpers (tests call the boundary itself) -----------------------------------------
class Device:
    """One mi_ctx on one GPU, driven through the raw C ABI."""

    def __init__(self, device=0):
        self.L = device_lib()
        self.ctx = C.c_void_p()
        st = self.L.mi_create(device, C.byref(self.ctx))
        if st != 0:
            raise RuntimeError("mi_create failed (%d): %s" % (st, self.L.mi_last_error(None).decode()))

    def check(self, st):
        if st != 0:
            raise RuntimeError("miwave error %d: %s" % (st, self.L.mi_last_error(self.ctx).decode()))

    def upload(self, desc, bvh_quality=1):
        self.check(self.L.mi_scene_upload(self.ctx, desc))
        self.check(self.L.mi_bvh_build(self.ctx, bvh_quality))

    def trace(self, o, d, mint=0.0, maxt=np.inf, any_hit=False):
        r, keep, n = _rays_struct(o, d, mint, maxt)
        h, out = _hits_struct(n)
        self.check(self.L.mi_trace(self.ctx, C.byref(r), C.byref(h), n, int(any_hit)))
        return out

    # ---- the Scene query surface through the raw C ABI (batches) ----
    def ray_intersect(self, o, d, mint=0.0, maxt=np.inf):
        """mi_ray_intersect -> structured array of _capi.SI_DTYPE records"""
        r, keep, n = _rays_struct(o, d, mint, maxt)
        si = np.zeros(n, _capi.SI_DTYPE)
        self.check(self.L.mi_ray_intersect(self.ctx, C.byref(r), si.ctypes.data_as(C.POINTER(_capi.mi_surface_interaction)), n))
        return si

    def sample_emitter_direction(self, ref_p, sample, test_visibility=True, emitter=-1, wavelengths=None):
        """mi_sample_emitter_direction -> (records (_capi.DS_DTYPE), spectra [n, N])"""
        ref = np.ascontiguousarray(ref_p, np.float32).reshape(-1, 3); smp = np.ascontiguousarray(sample, np.float32).reshape(-1, 2)
        n = len(ref); nch = self.L.mi_spectrum_channels()
        wl = None if wavelengths is None else np.ascontiguousarray(wavelengths, np.float32).reshape(n, 4)
        ds = np.zeros(n, _capi.DS_DTYPE); spec = np.zeros((n, nch), np.float32)
        self.check(self.L.mi_sample_emitter_direction(self.ctx, int(emitter), _fp(ref), _fp(smp), None if wl is None else _fp(wl),
                                                      int(bool(test_visibility)), ds.ctypes.data_as(C.POINTER(_capi.mi_direction_sample)), _fp(spec), n))
        return ds, spec

    def pdf_emitter_direction(self, ref_p, ds, emitter=-1):
        ref = np.ascontiguousarray(ref_p, np.float32).reshape(-1, 3); d = np.ascontiguousarray(ds, _capi.DS_DTYPE)
        pdf = np.zeros(len(ref), np.float32)
        self.check(self.L.mi_pdf_emitter_direction(self.ctx, int(emitter), _fp(ref), d.ctypes.data_as(C.POINTER(_capi.mi_direction_sample)), _fp(pdf), len(ref)))
        return pdf

    def emitter_eval(self, si, wavelengths=None):
        x = np.ascontiguousarray(si, _capi.SI_DTYPE); n = len(x); nch = self.L.mi_spectrum_channels()
        wl = None if wavelengths is None else np.ascontiguousarray(wavelengths, np.float32).reshape(n, 4)
        spec = np.zeros((n, nch), np.float32)
        self.check(self.L.mi_emitter_eval(self.ctx, x.ctypes.data_as(C.POINTER(_capi.mi_surface_interaction)), None if wl is None else _fp(wl), _fp(spec), n))
        return spec

    def render(self, job, f64=False, profile=False, film_mode=0, plan=None, samples_per_launch=None, onto=None):
        """film_mode 0 auto / 1 sample log + ordered gather (float32, reference order) / 2 float64 atomics;
        plan 0 auto / 1 wavefront (HBM queues) / 2 resident (registers + LDS);
        onto: the film of the earlier passes when job.cfg.accumulate is set (a copy is accumulated onto)"""
        cfg = job.cfg
        cfg.film_on_device = 0; cfg.film_f64 = int(f64); cfg.profile = int(profile); cfg.film_mode = film_mode
        if plan is not None:
            cfg.plan = int(plan)
        if samples_per_launch is not None:
            cfg.samples_per_launch = int(samples_per_launch)
        n = cfg.crop_w * cfg.crop_h * 5
        film = np.zeros(n, np.float64 if f64 else np.float32)
        if cfg.accumulate:
            film[:] = np.asarray(onto).reshape(-1)
        st = self.L.mi_render(self.ctx, C.byref(cfg), film.ctypes.data_as(C.c_void_p))
        if st not in (0, _capi.MI_ERR_CANCELLED):
            self.check(st)
        return film.reshape(cfg.crop_h, cfg.crop_w, 5), st

    def counters(self):
        c = mi_counters()
        self.L.mi_get_counters(self.ctx, C.byref(c))
        return c

    def eval(self, op, inputs, cfg=None):
        i_s, o_s = _capi.eval_strides(op, self.L.mi_spectrum_channels())
        a = np.ascontiguousarray(inputs, np.float32).reshape(-1, i_s)
        out = np.zeros((len(a), o_s), np.float32)
        self.check(self.L.mi_eval(self.ctx, op, C.byref(cfg) if cfg is not None else None, _fp(a), i_s, _fp(out), o_s,
                                  len(a)))
        return out

    def selftest(self, which=0):
        """mi_selftest: number of inputs on which a device-specific leaf form differs from the IEEE form"""
        bad = C.c_uint64(0)
        self.check(self.L.mi_selftest(self.ctx, int(which), C.byref(bad)))
        return int(bad.value)

    def close(self):
        if self.ctx:
            self.L.mi_destroy(self.ctx); self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
