"""ctypes binding of the CPU checker (oracle/_build/libmiw_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the mitsuba2_amd package.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from mitsuba2_amd._capi import (mi_hits_soa, mi_rays_soa, mi_render_cfg, mi_scene_desc, c_float_p, c_double_p,  # noqa: E402
                                c_u32_p, c_i32_p, MI_EVAL_STRIDES, VARIANT_SUFFIX, eval_strides)

_libs = {}


class orc_stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("shadow_rays", C.c_uint64), ("seconds", C.c_double)]


def _fp(a):
    return a.ctypes.data_as(c_float_p)


class Oracle:
    def __init__(self, lib, channels=3):
        self.L = lib
        self.channels = channels
        L = lib
        vp = C.c_void_p
        L.orc_render.argtypes = [C.POINTER(mi_scene_desc), C.POINTER(mi_render_cfg), c_float_p, c_double_p, C.c_int,
                                 c_u32_p, C.c_uint32, C.POINTER(orc_stats)]
        L.orc_render.restype = C.c_int
        L.orc_set_accel.argtypes = [C.c_int]; L.orc_set_accel.restype = None
        L.orc_trace.argtypes = [C.POINTER(mi_scene_desc), C.POINTER(mi_rays_soa), C.POINTER(mi_hits_soa), C.c_uint64, C.c_int]
        L.orc_trace.restype = C.c_int
        L.orc_ray_intersect_full.argtypes = [C.POINTER(mi_scene_desc), c_float_p, c_float_p]
        L.orc_ray_intersect_full.restype = C.c_int
        L.emu_trace.argtypes = [C.POINTER(mi_scene_desc), C.POINTER(mi_rays_soa), C.POINTER(mi_hits_soa), C.c_uint64,
                                C.c_int, C.c_int, c_u32_p]
        L.emu_trace.restype = C.c_int
        L.emu_trace4.argtypes = [C.POINTER(mi_scene_desc), C.POINTER(mi_rays_soa), C.POINTER(mi_hits_soa), C.c_uint64,
                                 C.c_int, C.c_int, C.c_int, C.c_int, c_u32_p, C.c_uint32]
        L.emu_trace4.restype = C.c_int
        L.emu_trace8.argtypes = [C.POINTER(mi_scene_desc), C.POINTER(mi_rays_soa), C.POINTER(mi_hits_soa), C.c_uint64,
                                 C.c_int, C.c_int, C.c_int, c_u32_p, C.c_uint32, C.c_int]
        L.emu_trace8.restype = C.c_int
        L.emu_sah_levels_check.argtypes = [C.POINTER(mi_scene_desc), C.c_int, c_u32_p]; L.emu_sah_levels_check.restype = C.c_int
        L.emu_render.argtypes = [C.POINTER(mi_scene_desc), C.POINTER(mi_render_cfg), c_double_p, c_float_p, C.POINTER(C.c_uint64)]
        L.emu_render.restype = C.c_int
        L.orc_tea_float32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]; L.orc_tea_float32.restype = C.c_float
        L.orc_tea_float64.argtypes = [C.c_uint32, C.c_uint32, C.c_int]; L.orc_tea_float64.restype = C.c_double
        L.orc_tea_32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]; L.orc_tea_32.restype = C.c_uint32
        L.orc_pcg32_u32.argtypes = [C.c_uint64, C.c_uint64, c_u32_p, C.c_int]
        L.orc_pcg32_f32.argtypes = [C.c_uint64, C.c_uint64, c_float_p, C.c_int]
        L.orc_morton_decode.argtypes = [C.c_uint32, c_u32_p]
        L.orc_fresnel.argtypes = [C.c_float, C.c_float, c_float_p]
        L.orc_fresnel_conductor.argtypes = [C.c_float, C.c_float, C.c_float]; L.orc_fresnel_conductor.restype = C.c_float
        L.orc_microfacet.argtypes = [C.c_int, C.c_uint32, C.c_float, C.c_float, C.c_int, c_float_p, c_float_p, c_float_p]
        L.orc_ray_triangle.argtypes = [c_float_p, c_float_p, c_float_p]
        L.orc_special.argtypes = [C.c_float, c_float_p]
        if channels == 4:
            L.orc_spectral.argtypes = [C.c_int, c_float_p, c_float_p]
        L.orc_hier2d.argtypes = [c_float_p, C.c_uint32, C.c_uint32, C.c_int, c_float_p, c_float_p]; L.orc_hier2d.restype = C.c_int
        L.orc_imageblock_put.argtypes = [C.POINTER(mi_render_cfg), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p,
                                         c_float_p, C.c_int, c_float_p]
        L.orc_imageblock_put.restype = C.c_int
        L.orc_film_splat_shared.argtypes = [C.POINTER(mi_render_cfg), c_float_p, c_float_p, C.c_int, c_double_p]
        L.orc_spiral.argtypes = [C.c_int] * 5 + [c_i32_p, C.c_int]; L.orc_spiral.restype = C.c_int
        L.orc_coordinate_system.argtypes = [c_float_p, c_float_p]
        L.orc_warp.argtypes = [C.c_int, c_float_p, c_float_p]
        L.orc_eval.argtypes = [C.c_int, C.POINTER(mi_scene_desc), C.POINTER(mi_render_cfg), c_float_p, C.c_int, c_float_p,
                               C.c_int, C.c_uint64]
        L.orc_eval.restype = C.c_int
        from mitsuba2_amd._capi import mi_surface_interaction, mi_direction_sample
        sip, dsp = C.POINTER(mi_surface_interaction), C.POINTER(mi_direction_sample)
        L.orc_ray_intersect.argtypes = [C.POINTER(mi_scene_desc), C.POINTER(mi_rays_soa), sip, C.c_uint64]; L.orc_ray_intersect.restype = C.c_int
        L.orc_sample_emitter_direction.argtypes = [C.POINTER(mi_scene_desc), C.c_int32, c_float_p, c_float_p, c_float_p, C.c_int32, dsp, c_float_p, C.c_uint64]
        L.orc_sample_emitter_direction.restype = C.c_int
        L.orc_pdf_emitter_direction.argtypes = [C.POINTER(mi_scene_desc), C.c_int32, c_float_p, dsp, c_float_p, C.c_uint64]; L.orc_pdf_emitter_direction.restype = C.c_int
        L.orc_emitter_eval.argtypes = [C.POINTER(mi_scene_desc), sip, c_float_p, c_float_p, C.c_uint64]; L.orc_emitter_eval.restype = C.c_int

    def set_accel(self, on):
        """0: brute-force scene queries (the definition, default); 1: the checker's own spatial index (same answers)"""
        self.L.orc_set_accel(int(bool(on)))

    # ---- renders ----
    def render(self, desc, job, threads=1, want_f64=True, only_blocks=None, onto=None):
        """onto = (f32, f64) films of the earlier passes when job.cfg.accumulate is set"""
        cfg = job.cfg
        n = cfg.crop_w * cfg.crop_h * 5
        f32 = np.zeros(n, np.float32)
        f64 = np.zeros(n, np.float64) if want_f64 else None
        if cfg.accumulate:
            f32[:] = np.asarray(onto[0]).reshape(-1)
            if f64 is not None:
                f64[:] = np.asarray(onto[1]).reshape(-1)
        st = orc_stats()
        ob = None if only_blocks is None else np.ascontiguousarray(only_blocks, np.uint32)
        rc = self.L.orc_render(desc, C.byref(cfg), _fp(f32), None if f64 is None else f64.ctypes.data_as(c_double_p),
                               threads, None if ob is None else ob.ctypes.data_as(c_u32_p), 0 if ob is None else len(ob),
                               C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render failed: %d" % rc)
        shape = (cfg.crop_h, cfg.crop_w, 5)
        return f32.reshape(shape), None if f64 is None else f64.reshape(shape), st

    def emu_render(self, desc, job, onto=None):
        """onto = (f64, f32) films of the earlier passes when job.cfg.accumulate is set"""
        cfg = job.cfg
        n = cfg.crop_w * cfg.crop_h * 5
        f64 = np.zeros(n, np.float64); f32 = np.zeros(n, np.float32)
        if cfg.accumulate:
            f64[:] = np.asarray(onto[0]).reshape(-1); f32[:] = np.asarray(onto[1]).reshape(-1)
        stats = (C.c_uint64 * 4)()
        rc = self.L.emu_render(desc, C.byref(cfg), f64.ctypes.data_as(c_double_p), _fp(f32), stats)
        if rc != 0:
            raise RuntimeError("emu_render failed: %d" % rc)
        shape = (cfg.crop_h, cfg.crop_w, 5)
        return f64.reshape(shape), f32.reshape(shape), list(stats)

    def _trace(self, fn, desc, o, d, mint, maxt, any_hit, *extra):
        from mitsuba2_amd.api import _rays_struct, _hits_struct
        r, keep, n = _rays_struct(o, d, mint, maxt)
        h, out = _hits_struct(n)
        rc = fn(desc, C.byref(r), C.byref(h), n, int(any_hit), *extra)
        if rc != 0:
            raise RuntimeError("trace failed: %d" % rc)
        return out

    def trace(self, desc, o, d, mint=0.0, maxt=np.inf, any_hit=False):
        return self._trace(self.L.orc_trace, desc, o, d, mint, maxt, any_hit)

    def emu_trace(self, desc, o, d, mint=0.0, maxt=np.inf, any_hit=False, max_leaf=4):
        stats = (C.c_uint32 * 3)()
        out = self._trace(self.L.emu_trace, desc, o, d, mint, maxt, any_hit, max_leaf, stats)
        out["bvh"] = list(stats)
        return out

    def emu_sah_levels_check(self, desc, max_leaf=4):
        """the level-by-level binned-SAH builder (csrc/sah_levels.h, what the device runs) against the recursive host builder"""
        stats = (C.c_uint32 * 6)()
        rc = self.L.emu_sah_levels_check(desc, int(max_leaf), stats)
        if rc < 0:
            raise RuntimeError("emu_sah_levels_check failed")
        d = dict(zip(("nodes_recursive", "nodes_levels", "node_diff", "depth_recursive", "depth_levels", "leaf_diff"), list(stats)))
        d["need_host"] = rc == 1
        return d

    def emu_trace4(self, desc, o, d, mint=0.0, maxt=np.inf, any_hit=False, max_leaf=4, stack_budget=32, max_fan=4, schedule=0, spec=True):
        """The 4-wide quantised tree (csrc/miw/bvh4.h, collapsed by csrc/bvh4_build.h) walked on the CPU: by the reference walk
        bvh4_intersect (schedule = 0) or by the per-lane bodies of the device's phase machine (walk4_node_step / walk4_tri_step)
        under a pseudo-random schedule seeded with `schedule` > 0 (spec: the speculating variant the device runs by default)."""
        stats = (C.c_uint32 * 6)()
        sched = (int(schedule) & 0x7fffffff) | (0 if spec or not schedule else 0x80000000)
        out = self._trace(self.L.emu_trace4, desc, o, d, mint, maxt, any_hit, max_leaf, stack_budget, max_fan, stats, sched)
        out["bvh4"] = dict(zip(("nodes2", "nodes4", "depth", "stack_bound", "stack_seen", "ok"), list(stats)))
        return out

    def emu_trace8(self, desc, o, d, mint=0.0, maxt=np.inf, any_hit=False, max_leaf=4, max_fan=8, schedule=0, compare4=False, spec=False):
        """The 8-wide quantised tree (csrc/miw/bvh8.h, collapsed by csrc/bvh8_build.h, triangles in the tree's own order) walked on
        the CPU: by the reference walk bvh8_intersect (schedule = 0) or by the per-lane bodies of the device's phase machine
        (walk8_node_step / walk8_tri_step; schedule > 0), every stack-column access checked. out["bvh8"]: node / depth / step
        counts (+ the 4-wide reference walk's counts over the same rays with compare4)."""
        stats = (C.c_uint32 * 15)()
        sched = (int(schedule) & 0x7fffffff) | (0x80000000 if spec and schedule else 0)
        out = self._trace(self.L.emu_trace8, desc, o, d, mint, maxt, any_hit, max_leaf, max_fan, stats, sched, int(bool(compare4)))
        st = list(stats)
        out["bvh8"] = dict(nodes2=st[0], nodes8=st[1], depth=st[2], stack_seen=st[3], ok=st[4], node_steps=st[5] | st[6] << 32,
                           tri_tests=st[7] | st[8] << 32, tri_steps=st[9] | st[10] << 32, node_steps4=st[11] | st[12] << 32,
                           tri_tests4=st[13] | st[14] << 32)
        return out

    def ray_intersect_full(self, desc, ray8):
        r = np.ascontiguousarray(ray8, np.float32); out = np.zeros(21, np.float32)
        ok = self.L.orc_ray_intersect_full(desc, _fp(r), _fp(out))
        return ok, out

    # ---- the Scene query surface (checkers of mi_ray_intersect, mi_sample_emitter_direction, ...) ----
    def ray_intersect(self, desc, o, d, mint=0.0, maxt=np.inf):
        from mitsuba2_amd.api import _rays_struct
        from mitsuba2_amd import _capi
        r, keep, n = _rays_struct(o, d, mint, maxt)
        si = np.zeros(n, _capi.SI_DTYPE)
        if self.L.orc_ray_intersect(desc, C.byref(r), si.ctypes.data_as(C.POINTER(_capi.mi_surface_interaction)), n) != 0:
            raise RuntimeError("orc_ray_intersect failed")
        return si

    def sample_emitter_direction(self, desc, ref_p, sample, test_visibility=True, emitter=-1, wavelengths=None):
        from mitsuba2_amd import _capi
        ref = np.ascontiguousarray(ref_p, np.float32).reshape(-1, 3); smp = np.ascontiguousarray(sample, np.float32).reshape(-1, 2)
        n = len(ref)
        wl = None if wavelengths is None else np.ascontiguousarray(wavelengths, np.float32).reshape(n, 4)
        ds = np.zeros(n, _capi.DS_DTYPE); spec = np.zeros((n, self.channels), np.float32)
        if self.L.orc_sample_emitter_direction(desc, int(emitter), _fp(ref), _fp(smp), None if wl is None else _fp(wl), int(bool(test_visibility)),
                                               ds.ctypes.data_as(C.POINTER(_capi.mi_direction_sample)), _fp(spec), n) != 0:
            raise RuntimeError("orc_sample_emitter_direction failed")
        return ds, spec

    def pdf_emitter_direction(self, desc, ref_p, ds, emitter=-1):
        from mitsuba2_amd import _capi
        ref = np.ascontiguousarray(ref_p, np.float32).reshape(-1, 3); d = np.ascontiguousarray(ds, _capi.DS_DTYPE)
        pdf = np.zeros(len(ref), np.float32)
        if self.L.orc_pdf_emitter_direction(desc, int(emitter), _fp(ref), d.ctypes.data_as(C.POINTER(_capi.mi_direction_sample)), _fp(pdf), len(ref)) != 0:
            raise RuntimeError("orc_pdf_emitter_direction failed")
        return pdf

    def emitter_eval(self, desc, si, wavelengths=None):
        from mitsuba2_amd import _capi
        x = np.ascontiguousarray(si, _capi.SI_DTYPE); n = len(x)
        wl = None if wavelengths is None else np.ascontiguousarray(wavelengths, np.float32).reshape(n, 4)
        spec = np.zeros((n, self.channels), np.float32)
        if self.L.orc_emitter_eval(desc, x.ctypes.data_as(C.POINTER(_capi.mi_surface_interaction)), None if wl is None else _fp(wl), _fp(spec), n) != 0:
            raise RuntimeError("orc_emitter_eval failed")
        return spec

    def eval(self, op, inputs, desc=None, cfg=None):
        i_s, o_s = eval_strides(op, self.channels)
        a = np.ascontiguousarray(inputs, np.float32).reshape(-1, i_s)
        out = np.zeros((len(a), o_s), np.float32)
        rc = self.L.orc_eval(op, desc, C.byref(cfg) if cfg is not None else None, _fp(a), i_s, _fp(out), o_s, len(a))
        if rc != 0:
            raise RuntimeError("orc_eval failed")
        return out


def load(variant="scalar_rgb"):
    if variant not in _libs:
        # MIW_ORACLE_DIR: another build of the checker (tools/sanitize_cpu.sh: the AddressSanitizer + UBSan build of the same sources)
        path = os.path.join(os.environ.get("MIW_ORACLE_DIR") or os.path.join(ROOT, "oracle", "_build"), "libmiw_oracle%s.so" % VARIANT_SUFFIX[variant])
        if not os.path.exists(path):
            raise ImportError(path + " missing: python -m mitsuba2_amd.build --oracle")
        _libs[variant] = Oracle(C.CDLL(path), 4 if variant == "scalar_spectral" else 3)
    return _libs[variant]
