"""ctypes declarations of the C ABI (include/miwave.h) and of the host facade (mih_*).

Loading fails loudly when the native libraries are missing: there is no Python
or CPU fallback for any entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIWAVE_LIB_DIR: another directory holding the same libraries (tools/build_variants.py writes experiment builds of the
# kernels — other -D switches — next to a copy of the host library); the default is the in-tree product build.
LIB_DIR = os.environ.get("MIWAVE_LIB_DIR") or os.path.join(_HERE, "lib")

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_u32_p = C.POINTER(C.c_uint32)
c_i32_p = C.POINTER(C.c_int32)


class mi_texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("v", C.c_float * 4)]


class mi_bsdf(C.Structure):
    _fields_ = [("type", C.c_uint32), ("flags", C.c_uint32), ("params", C.c_float * 14),
                ("tex", mi_texture * 3), ("back", C.c_uint32)]


class mi_shape(C.Structure):
    _fields_ = [("bsdf", C.c_uint32), ("emitter", C.c_int32), ("flags", C.c_uint32),
                ("first_face", C.c_uint32), ("face_count", C.c_uint32)]


class mi_emitter(C.Structure):
    _fields_ = [("shape", C.c_uint32), ("radiance", C.c_float * 3), ("radiance_tex", mi_texture)]


class mi_envmap(C.Structure):
    _fields_ = [("rgba", c_float_p), ("width", C.c_uint32), ("height", C.c_uint32), ("scale", C.c_float),
                ("to_world", C.c_float * 16), ("bsphere_radius", C.c_float), ("emitter_index", C.c_uint32),
                ("density", c_float_p)]


class mi_rectangle(C.Structure):
    _fields_ = [("shape", C.c_uint32), ("to_world", C.c_float * 16), ("to_object", C.c_float * 16)]


class mi_sphere(C.Structure):
    _fields_ = [("shape", C.c_uint32), ("center", C.c_float * 3), ("radius", C.c_float), ("flip_normals", C.c_uint32),
                ("to_world", C.c_float * 16), ("to_object", C.c_float * 16)]


class mi_bitmap(C.Structure):
    _fields_ = [("data", c_float_p), ("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32),
                ("filter_type", C.c_uint32), ("wrap_mode", C.c_uint32), ("to_uv", C.c_float * 6)]


class mi_scene_desc(C.Structure):
    _fields_ = [("vertex_positions", c_float_p), ("vertex_normals", c_float_p), ("vertex_count", C.c_uint32),
                ("faces", c_u32_p), ("face_count", C.c_uint32),
                ("shapes", C.POINTER(mi_shape)), ("shape_count", C.c_uint32),
                ("bsdfs", C.POINTER(mi_bsdf)), ("bsdf_count", C.c_uint32),
                ("emitters", C.POINTER(mi_emitter)), ("emitter_count", C.c_uint32),
                ("envmap", C.POINTER(mi_envmap)),
                ("rectangles", C.POINTER(mi_rectangle)), ("rectangle_count", C.c_uint32),
                ("spheres", C.POINTER(mi_sphere)), ("sphere_count", C.c_uint32),
                ("vertex_texcoords", c_float_p), ("bitmaps", C.POINTER(mi_bitmap)), ("bitmap_count", C.c_uint32),
                ("bsdf_tables", c_float_p), ("bsdf_table_floats", C.c_uint32)]


class mi_rays_soa(C.Structure):
    _fields_ = [(n, c_float_p) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "mint", "maxt")]


class mi_hits_soa(C.Structure):
    _fields_ = [("t", c_float_p), ("u", c_float_p), ("v", c_float_p), ("prim", c_u32_p), ("shape", c_u32_p)]


class mi_surface_interaction(C.Structure):
    _fields_ = [("t", C.c_float), ("p", C.c_float * 3), ("n", C.c_float * 3), ("sh_s", C.c_float * 3), ("sh_t", C.c_float * 3),
                ("sh_n", C.c_float * 3), ("uv", C.c_float * 2), ("wi", C.c_float * 3), ("prim_index", C.c_uint32),
                ("shape_index", C.c_uint32), ("emitter_index", C.c_int32)]


class mi_direction_sample(C.Structure):
    _fields_ = [("p", C.c_float * 3), ("n", C.c_float * 3), ("d", C.c_float * 3), ("dist", C.c_float), ("pdf", C.c_float),
                ("emitter_index", C.c_int32)]


# numpy views of the two records (same layout: 24 and 12 four-byte words)
import numpy as _np
SI_DTYPE = _np.dtype([("t", "f4"), ("p", "f4", 3), ("n", "f4", 3), ("sh_s", "f4", 3), ("sh_t", "f4", 3), ("sh_n", "f4", 3),
                      ("uv", "f4", 2), ("wi", "f4", 3), ("prim_index", "u4"), ("shape_index", "u4"), ("emitter_index", "i4")])
DS_DTYPE = _np.dtype([("p", "f4", 3), ("n", "f4", 3), ("d", "f4", 3), ("dist", "f4"), ("pdf", "f4"), ("emitter_index", "i4")])
assert SI_DTYPE.itemsize == C.sizeof(mi_surface_interaction) == 96 and DS_DTYPE.itemsize == C.sizeof(mi_direction_sample) == 48


class mi_render_cfg(C.Structure):
    _fields_ = [("crop_x", C.c_int32), ("crop_y", C.c_int32), ("crop_w", C.c_int32), ("crop_h", C.c_int32),
                ("spp", C.c_uint32), ("max_depth", C.c_int32), ("rr_depth", C.c_int32),
                ("base_seed", C.c_uint64), ("block_size", C.c_int32),
                ("block_ids", c_u32_p), ("block_count", C.c_uint32),
                ("tile_list", c_u32_p), ("tile_count", C.c_uint32),
                ("sample_to_camera", C.c_float * 16), ("to_world", C.c_float * 16),
                ("near_clip", C.c_float), ("far_clip", C.c_float), ("principal_point_offset", C.c_float * 2),
                ("filter_lut", C.c_float * 32), ("filter_radius", C.c_float), ("filter_border", C.c_int32),
                ("film_on_device", C.c_int32), ("film_f64", C.c_int32), ("film_mode", C.c_int32),
                ("profile", C.c_int32),
                ("timeout_s", C.c_float), ("plan", C.c_int32), ("samples_per_launch", C.c_int32), ("accumulate", C.c_int32),
                ("integrator", C.c_int32), ("emitter_samples", C.c_uint32), ("bsdf_samples", C.c_uint32),
                ("hide_emitters", C.c_int32), ("moment_pass", C.c_int32),
                ("debug_film_replay", C.c_int32), ("debug_tree_width", C.c_int32), ("debug_path_kernel", C.c_int32)]


class mi_counters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("iterations", C.c_uint64), ("lanes", C.c_uint64), ("ms_render", C.c_double),
                ("ms_trace_closest", C.c_double), ("ms_trace_any", C.c_double), ("ms_shade", C.c_double),
                ("ms_init", C.c_double), ("ms_resolve", C.c_double),
                ("n_trace_closest", C.c_uint64), ("n_trace_any", C.c_uint64), ("n_shade", C.c_uint64),
                ("ms_bvh_build", C.c_double), ("bvh_nodes", C.c_uint32), ("bvh_tris", C.c_uint32),
                ("bvh_depth", C.c_uint32), ("film_mode", C.c_uint32), ("plan", C.c_uint32),
                ("ms_path", C.c_double), ("n_path", C.c_uint64),
                ("ms_film_blocks", C.c_double), ("ms_film_merge", C.c_double),
                ("bvh_on_device", C.c_uint32), ("path_kernel", C.c_uint32), ("ms_film_pack", C.c_double),
                ("log_bytes", C.c_uint64), ("log_record_bytes", C.c_uint32), ("bvh4_on_device", C.c_uint32), ("ms_bvh4", C.c_double),
                ("placed", C.c_uint32), ("bvh_builder", C.c_uint32),
                ("tree_width", C.c_uint32), ("bvh8_nodes", C.c_uint32), ("bvh8_depth", C.c_uint32), ("bvh8_on_device", C.c_uint32), ("ms_bvh8", C.c_double),
                ("place_cost_max", C.c_uint32), ("place_cost_unit", C.c_uint32), ("place_max_pixel", C.c_uint32), ("place_measure_spp", C.c_uint32),
                ("place_cost_mean", C.c_double), ("film_kernel", C.c_uint32), ("log_interleaved", C.c_uint32),
                ("pooled", C.c_uint32), ("pool_waves", C.c_uint32), ("film_overlapped", C.c_uint32), ("film_groups", C.c_uint32),
                ("job_chunk", C.c_uint32), ("job_chunks", C.c_uint32)]


MI_INTEGRATOR_PATH, MI_INTEGRATOR_DIRECT = 0, 1
MI_BVH_FORCE_TREE, MI_BVH_NO_LEAF_FILTER, MI_BVH_RADIX_TREE = 0x10, 0x20, 0x40      # flags of mi_bvh_build's quality argument
MI_OK, MI_ERR_INVALID, MI_ERR_DEVICE, MI_ERR_STATE, MI_ERR_CANCELLED = 0, -1, -2, -3, -4
MI_EVAL = dict(PCG32=0, SINCOS=1, COSINE_HEMISPHERE=2, BSDF=3, FRESNEL=4, CAMERA_RAY=5, EMITTER_SAMPLE=6,
               FP_SEMANTICS=7, SPECIAL=8, ENVMAP=9, INVTRIG=10, SPECTRUM=11, TEXTURE=12)
MI_EVAL_STRIDES = {0: (2, 8), 1: (1, 2), 2: (2, 4), 3: (10, 13), 4: (2, 4), 5: (2, 8), 6: (5, 14), 7: (3, 8), 8: (1, 4), 9: (8, 12), 10: (2, 3)}

# every symbol include/miwave.h declares (tests check that the library exports all of them)
MI_SYMBOLS = ["mi_spectrum_channels", "mi_device_count", "mi_create", "mi_destroy", "mi_set_stream", "mi_scene_upload", "mi_bvh_build",
              "mi_trace", "mi_render", "mi_cancel", "mi_get_counters", "mi_last_error", "mi_eval", "mi_selftest",
              "mi_ray_intersect", "mi_sample_emitter_direction", "mi_pdf_emitter_direction", "mi_emitter_eval",
              "mi_film_alloc", "mi_film_free", "mi_film_download", "mi_film_reduce",
              "mi_set_option", "mi_get_option", "mi_option_count", "mi_option_name", "mi_option_help"]


VARIANT_SUFFIX = {"scalar_rgb": "", "scalar_spectral": "_spectral"}


def eval_strides(op, channels=3):
    """(input stride, output stride) of a mi_eval op for a library with `channels` spectrum channels"""
    extra = 4 if channels == 4 else 0
    if op == 3:
        return 10 + extra, 7 + 2 * channels
    if op == 6:
        return 5 + extra, 11 + channels
    if op == 9:
        return 8 + extra, 6 + 2 * channels
    if op == 11:
        return 5, 19
    if op == 12:
        return 4, channels
    return MI_EVAL_STRIDES[op]


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build the native libraries first (python -m mitsuba2_amd.build). "
            "mitsuba2_amd has no CPU fallback." % path)
    return C.CDLL(path)          # RTLD_LOCAL: both variants export the same symbol names


def load_device_lib(variant="scalar_rgb"):
    lib = _load("libmiwave%s.so" % VARIANT_SUFFIX[variant])
    lib.mi_spectrum_channels.argtypes = []; lib.mi_spectrum_channels.restype = C.c_int32
    vp = C.c_void_p
    lib.mi_device_count.argtypes = [c_i32_p]; lib.mi_device_count.restype = C.c_int32
    lib.mi_create.argtypes = [C.c_int32, C.POINTER(vp)]; lib.mi_create.restype = C.c_int32
    lib.mi_destroy.argtypes = [vp]; lib.mi_destroy.restype = None
    lib.mi_set_stream.argtypes = [vp, vp]; lib.mi_set_stream.restype = C.c_int32
    lib.mi_scene_upload.argtypes = [vp, C.POINTER(mi_scene_desc)]; lib.mi_scene_upload.restype = C.c_int32
    lib.mi_bvh_build.argtypes = [vp, C.c_int32]; lib.mi_bvh_build.restype = C.c_int32
    lib.mi_trace.argtypes = [vp, C.POINTER(mi_rays_soa), C.POINTER(mi_hits_soa), C.c_uint64, C.c_int32]
    lib.mi_trace.restype = C.c_int32
    lib.mi_render.argtypes = [vp, C.POINTER(mi_render_cfg), vp]; lib.mi_render.restype = C.c_int32
    lib.mi_cancel.argtypes = [vp]; lib.mi_cancel.restype = C.c_int32
    lib.mi_get_counters.argtypes = [vp, C.POINTER(mi_counters)]; lib.mi_get_counters.restype = C.c_int32
    lib.mi_last_error.argtypes = [vp]; lib.mi_last_error.restype = C.c_char_p
    lib.mi_eval.argtypes = [vp, C.c_int32, C.POINTER(mi_render_cfg), c_float_p, C.c_int32, c_float_p, C.c_int32,
                            C.c_uint64]
    lib.mi_eval.restype = C.c_int32
    lib.mi_selftest.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint64)]; lib.mi_selftest.restype = C.c_int32
    sip, dsp = C.POINTER(mi_surface_interaction), C.POINTER(mi_direction_sample)
    lib.mi_ray_intersect.argtypes = [vp, C.POINTER(mi_rays_soa), sip, C.c_uint64]; lib.mi_ray_intersect.restype = C.c_int32
    lib.mi_sample_emitter_direction.argtypes = [vp, C.c_int32, c_float_p, c_float_p, c_float_p, C.c_int32, dsp, c_float_p, C.c_uint64]
    lib.mi_sample_emitter_direction.restype = C.c_int32
    lib.mi_pdf_emitter_direction.argtypes = [vp, C.c_int32, c_float_p, dsp, c_float_p, C.c_uint64]; lib.mi_pdf_emitter_direction.restype = C.c_int32
    lib.mi_emitter_eval.argtypes = [vp, sip, c_float_p, c_float_p, C.c_uint64]; lib.mi_emitter_eval.restype = C.c_int32
    return lib


def load_host_lib(variant="scalar_rgb"):
    lib = _load("libmiwave_host%s.so" % VARIANT_SUFFIX[variant])
    vp, cp, f, i32, u32, u64 = C.c_void_p, C.c_char_p, C.c_float, C.c_int32, C.c_uint32, C.c_uint64
    sig = {
        "mih_last_error": (cp, []), "mih_spectrum_channels": (i32, []), "mih_set_srgb_model": (i32, [cp]),
        "mih_props_create": (vp, [cp]), "mih_props_destroy": (None, [vp]),
        "mih_props_set_float": (None, [vp, cp, f]), "mih_props_set_int": (None, [vp, cp, C.c_int64]),
        "mih_props_set_bool": (None, [vp, cp, i32]), "mih_props_set_string": (None, [vp, cp, cp]),
        "mih_props_set_color": (None, [vp, cp, f, f, f]), "mih_props_set_texture": (None, [vp, cp, vp]),
        "mih_bitmap_create": (vp, [vp, u32, u32, u32, c_float_p]), "mih_bitmap_destroy": (None, [vp]),
        "mih_bitmap_info": (C.c_int, [vp, c_u32_p, c_float_p]),
        "mih_props_set_lookat": (None, [vp, cp, c_float_p, c_float_p, c_float_p]),
        "mih_props_set_matrix": (None, [vp, cp, c_float_p]), "mih_rectangle_create": (vp, [vp]), "mih_sphere_create": (vp, [vp]),
        "mih_bsdf_create": (vp, [vp]), "mih_bsdf_destroy": (None, [vp]), "mih_bsdf_create_twosided": (vp, [vp, vp]),
        "mih_fresnel_diffuse_reflectance": (C.c_float, [C.c_float]),
        "mih_bsdf_record": (i32, [vp, C.POINTER(mi_bsdf)]), "mih_bsdf_flags": (u32, [vp]),
        "mih_bsdf_table": (i32, [vp, c_float_p, u32]), "mih_gauss_legendre": (None, [i32, c_float_p, c_float_p]),
        "mih_bsdf_sample": (i32, [vp, c_float_p, f, c_float_p, c_float_p]),
        "mih_bsdf_eval_pdf": (i32, [vp, c_float_p, c_float_p, c_float_p]),
        "mih_emitter_create": (vp, [vp]), "mih_emitter_destroy": (None, [vp]),
        "mih_mesh_create": (vp, [cp, c_float_p, u32, c_u32_p, u32, c_float_p, c_float_p]), "mih_mesh_copy_texcoords": (None, [vp, c_float_p]),
        "mih_mesh_bbox_area": (None, [vp, c_float_p]), "mih_scene_bbox": (None, [vp, c_float_p]), "mih_mesh_destroy": (None, [vp]),
        "mih_mesh_load": (vp, [i32, vp]), "mih_mesh_recompute_normals": (i32, [vp]),
        "mih_mesh_counts": (None, [vp, c_u32_p, c_u32_p, c_i32_p]), "mih_mesh_copy": (None, [vp, c_float_p, c_u32_p, c_float_p]),
        "mih_mesh_set_bsdf": (None, [vp, vp]), "mih_mesh_set_emitter": (None, [vp, vp]),
        "mih_envmap_create": (vp, [vp, u32, u32, c_float_p]), "mih_envmap_destroy": (None, [vp]),
        "mih_scene_add_envmap": (i32, [vp, vp]),
        "mih_load_xml": (i32, [cp, i32, cp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "mih_scene_create": (vp, []), "mih_scene_destroy": (None, [vp]),
        "mih_scene_add_shape": (i32, [vp, vp]), "mih_scene_build": (i32, [vp, i32, i32]),
        "mih_scene_build_multi": (i32, [vp, C.POINTER(C.c_int), i32, i32]), "mih_scene_device_count": (i32, [vp]),
        "mih_render_multi": (i32, [vp, vp, vp, C.POINTER(C.c_int)]), "mih_integrator_last_reduce": (i32, [vp]),
        "mih_scene_desc": (C.POINTER(mi_scene_desc), [vp]), "mih_scene_ctx": (vp, [vp]),
        "mih_scene_ray_intersect": (i32, [vp, C.POINTER(mi_rays_soa), C.POINTER(mi_hits_soa), u64]),
        "mih_scene_ray_test": (i32, [vp, C.POINTER(mi_rays_soa), c_float_p, u64]),
        "mih_scene_ray_intersect_si": (i32, [vp, C.POINTER(mi_rays_soa), C.POINTER(mi_surface_interaction), u64]),
        "mih_scene_ray_intersect_one": (i32, [vp, c_float_p, C.POINTER(mi_surface_interaction), c_i32_p, c_i32_p]),
        "mih_scene_sample_emitter_direction": (i32, [vp, i32, c_float_p, c_float_p, c_float_p, i32, C.POINTER(mi_direction_sample), c_float_p]),
        "mih_scene_pdf_emitter_direction": (i32, [vp, i32, c_float_p, C.POINTER(mi_direction_sample), c_float_p]),
        "mih_scene_emitter_eval": (i32, [vp, i32, C.POINTER(mi_surface_interaction), c_float_p, c_float_p]),
        "mih_scene_emitter_count": (i32, [vp]),
        "mih_bsdf_sample_ctx": (i32, [vp, u32, u32, u32, c_float_p, f, c_float_p, c_float_p]),
        "mih_film_create": (vp, [vp]), "mih_film_destroy": (None, [vp]),
        "mih_film_set_filter": (i32, [vp, cp, vp]), "mih_film_filter_eval": (i32, [vp, f, c_float_p]),
        "mih_film_develop": (cp, [vp, cp]), "mih_film_set_data": (i32, [vp, c_float_p, u64]),
        "mih_film_crop_size": (None, [vp, c_i32_p, c_i32_p]),
        "mih_film_data": (c_float_p, [vp, C.POINTER(u64)]), "mih_film_develop_rgb": (i32, [vp, c_float_p]),
        "mih_sampler_create": (vp, [vp]), "mih_sampler_destroy": (None, [vp]),
        "mih_sampler_seed": (None, [vp, u64]), "mih_sampler_next_1d": (f, [vp]),
        "mih_sensor_create": (vp, [vp, vp, vp]), "mih_sensor_destroy": (None, [vp]),
        "mih_sensor_sample_ray": (i32, [vp, f, f, c_float_p]), "mih_sensor_x_fov": (f, [vp]),
        "mih_integrator_create": (vp, [vp]), "mih_integrator_destroy": (None, [vp]),
        "mih_integrator_create_moment": (vp, [vp, vp, cp]), "mih_integrator_aov_names": (C.c_int, [vp, C.c_char_p, u32]),
        "mih_integrator_set_shard": (None, [vp, u32, u32]), "mih_integrator_set_profile": (None, [vp, i32]),
        "mih_integrator_set_plan": (None, [vp, i32]),
        "mih_integrator_cancel": (None, [vp]), "mih_integrator_render": (i32, [vp, vp, vp]),
        "mih_integrator_counters": (i32, [vp, C.POINTER(mi_counters)]),
        "mih_make_render_cfg": (i32, [vp, vp, C.POINTER(mi_render_cfg), c_u32_p, c_u32_p, u32, u32]),
        "mih_make_render_cfg_pass": (i32, [vp, vp, C.POINTER(mi_render_cfg), c_u32_p, c_u32_p, u32, u32, u32]),
        "mih_integrator_pass_count": (i32, [vp, vp]),
        "mih_spiral": (i32, [i32, i32, i32, i32, i32, c_i32_p, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
