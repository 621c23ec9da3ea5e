/* miwave — C ABI of the MI355X wavefront path-tracing core.
 *
 * This is the drop-in boundary: everything below Mitsuba 2's
 * SamplingIntegrator::render() / Scene::ray_intersect() for the `path`
 * integrator runs behind these entry points on one gfx950 device. The host
 * side (mitsuba2_amd/host, C++17) mirrors the reference's plugin classes and
 * calls only what is declared here: plain pointers and sizes, int status codes,
 * no exceptions, no C++ or torch types.
 *
 * Conventions: every function returns MI_OK (0) or a negative mi_status; the
 * text of the last failure is mi_last_error(ctx). The caller owns every host
 * buffer it passes; a ctx owns all device memory it allocates; one ctx per GPU;
 * a ctx is thread-compatible (one host thread at a time). Pointers are host
 * addresses unless a field says "device".
 */
#ifndef MIWAVE_H
#define MIWAVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_ctx mi_ctx;
typedef int32_t mi_status;

enum {
    MI_OK = 0,
    MI_ERR_INVALID = -1,      /* bad argument / unsupported plugin parameter          */
    MI_ERR_DEVICE = -2,       /* HIP runtime failure (text in mi_last_error)           */
    MI_ERR_STATE = -3,        /* call out of order (e.g. render before bvh_build)      */
    MI_ERR_CANCELLED = -4     /* mi_cancel() or timeout: render() returned false       */
};

/* ---- scene description (flat POD) -------------------------------------------------
 * Replaces what Scene(props) collects (src/librender/scene.cpp:22-104): shapes
 * (triangle meshes, include/mitsuba/render/mesh.h:80-104 buffers), their BSDFs
 * and area emitters. Vertex indices in `faces` are global (into the pooled
 * vertex arrays). Faces of shape s are the contiguous range
 * [first_face, first_face + face_count): global primitive id == face index,
 * like ShapeKDTree::m_primitive_map (kdtree.h:2335-2353). */

enum { MI_BSDF_DIFFUSE = 0, MI_BSDF_DIELECTRIC = 1, MI_BSDF_ROUGHCONDUCTOR = 2, MI_BSDF_CONDUCTOR = 3, MI_BSDF_PLASTIC = 4,
       MI_BSDF_ROUGHDIELECTRIC = 5, MI_BSDF_ROUGHPLASTIC = 6 };
enum { MI_BSDF_FLAG_GGX = 1, MI_BSDF_FLAG_SAMPLE_VISIBLE = 2,            /* roughconductor */
       MI_BSDF_FLAG_NONLINEAR = 1, MI_BSDF_FLAG_HAS_SPECULAR = 2,          /* plastic */
       MI_BSDF_FLAG_HAS_SPEC_REFLECTANCE = 4, MI_BSDF_FLAG_HAS_SPEC_TRANSMITTANCE = 8,   /* roughdielectric (+ GGX, SAMPLE_VISIBLE) */
       MI_BSDF_FLAG_RP_NONLINEAR = 0x10,                                   /* roughplastic (+ GGX, SAMPLE_VISIBLE, HAS_SPEC_REFLECTANCE) */
       MI_BSDF_FLAG_TWOSIDED = 0x100 };                                    /* any type: wrapped by <bsdf type="twosided"> */
enum { MI_SHAPE_HAS_TEXCOORDS = 8 };   /* the shape's vertices carry texture coordinates (mi_scene_desc::vertex_texcoords) */
enum { MI_SHAPE_HAS_NORMALS = 1,
       MI_SHAPE_RECTANGLE = 2,
       MI_SHAPE_SPHERE = 4 };      /* analytic sphere (src/shapes/sphere.cpp): like MI_SHAPE_RECTANGLE, geometry in `spheres` */   /* analytic rectangle (src/shapes/rectangle.cpp): face_count == 1 — the shape's single
                                      primitive, id first_face; its faces[] entry is ignored — geometry in `rectangles` */

/* A spectrum-valued plugin parameter (what src/libcore/xml.cpp:1073-1170 turns an <rgb> / <spectrum>
 * tag into). scalar_rgb library: only MI_TEX_RGB. scalar_spectral library (libmiwave_spectral.so):
 *   MI_TEX_UNIFORM   v[0] = value                              (src/spectra/uniform.cpp)
 *   MI_TEX_SRGB      v[0..2] = srgb_model_fetch coefficients   (src/spectra/srgb.cpp)
 *   MI_TEX_D65       v[0] = scale / 10568                      (src/spectra/d65.cpp)
 *   MI_TEX_SRGB_D65  v[0..2] = coefficients, v[3] = d65 scale / 10568 (src/spectra/srgb_d65.cpp)
 *   MI_TEX_BITMAP    v[0] = (float) index into mi_scene_desc::bitmaps   (src/textures/bitmap.cpp; either library) */
enum { MI_TEX_RGB = 0, MI_TEX_UNIFORM = 1, MI_TEX_SRGB = 2, MI_TEX_D65 = 3, MI_TEX_SRGB_D65 = 4, MI_TEX_BITMAP = 5 };
typedef struct { uint32_t type; float v[4]; } mi_texture;

/* BitmapTexture (src/textures/bitmap.cpp:85-262) after its constructor ran on the host: the image converted to the
 * working float representation, looked up at si.uv (mesh texture coordinates, or the analytic shapes' own uv).
 *   channels 1: a scalar image (Y);  channels 3: linear RGB in the scalar_rgb library, the per-texel coefficients of
 *   the sRGB upsampling model (srgb_model_fetch, bitmap.cpp:156-165) in the scalar_spectral library.
 *   to_uv: the affine 2 x 3 part of the `to_uv` transform, column by column (m00 m10  m01 m11  m03 m13 of the 4 x 4). */
enum { MI_BITMAP_NEAREST = 0, MI_BITMAP_BILINEAR = 1 };
enum { MI_BITMAP_REPEAT = 0, MI_BITMAP_MIRROR = 1, MI_BITMAP_CLAMP = 2 };
typedef struct {
    const float *data;                 /* width * height * channels, row-major                   */
    uint32_t width, height, channels;  /* >= 2 x 2 (bitmap.cpp:137-143 up-samples smaller images) */
    uint32_t filter_type, wrap_mode;   /* MI_BITMAP_*                                             */
    float to_uv[6];
} mi_bitmap;

typedef struct {
    uint32_t type;        /* MI_BSDF_*                                                   */
    uint32_t flags;       /* roughconductor: MI_BSDF_FLAG_*                              */
    /* diffuse  (src/bsdfs/diffuse.cpp:72):         [0..2] reflectance
     * dielectric (src/bsdfs/dielectric.cpp:174-199): [0] eta = int_ior/ext_ior,
     *                                              [1..3] specular_reflectance, [4..6] specular_transmittance
     * roughconductor (src/bsdfs/roughconductor.cpp:146-194): [0] alpha_u, [1] alpha_v,
     *                                              [2..4] eta, [5..7] k, [8..10] specular_reflectance
     * conductor (src/bsdfs/conductor.cpp:201-215):  [2..4] eta, [5..7] k, [8..10] specular_reflectance
     * plastic (src/bsdfs/plastic.cpp:135-174):      [0] eta = int_ior/ext_ior, [1] 1/eta^2, [2] fdr_int,
     *                                              [3] specular_sampling_weight, [4..6] diffuse_reflectance,
     *                                              [7..9] specular_reflectance (MI_BSDF_FLAG_HAS_SPECULAR)
     * roughdielectric (src/bsdfs/roughdielectric.cpp:146-201): [0] alpha_u, [1] alpha_v, [2] eta, [3] 1/eta,
     *                                              [4..6] specular_reflectance, [7..9] specular_transmittance
     * roughplastic (src/bsdfs/roughplastic.cpp:146-181,336-371): [0] alpha, [1] eta = int_ior/ext_ior, [2] 1/eta^2,
     *                                              [3] internal reflectance, [4] specular_sampling_weight, [5] (float) offset
     *                                              of its MI_ROUGH_TRANSMITTANCE_RES-entry external-transmittance table in
     *                                              mi_scene_desc::bsdf_tables, [6..8] diffuse_reflectance, [9..11] specular_reflectance */
    float params[14];
    /* scalar_spectral (and optionally scalar_rgb: used when tex[0].type != MI_TEX_RGB or params carry no colour):
     * diffuse: tex[0] reflectance; dielectric: tex[0] specular_reflectance, tex[1] specular_transmittance;
     * roughconductor / conductor: tex[0] eta, tex[1] k, tex[2] specular_reflectance; plastic: tex[0]
     * diffuse_reflectance, tex[1] specular_reflectance; roughdielectric: tex[0] specular_reflectance, tex[1]
     * specular_transmittance. The scalar_rgb library derives these from params[] itself. */
    mi_texture tex[3];
    uint32_t back;        /* MI_BSDF_FLAG_TWOSIDED: index of the back side's record (its own index: same BSDF on both
                             sides, src/bsdfs/twosided.cpp:72-73); else 0 */
} mi_bsdf;

typedef struct {
    uint32_t bsdf;        /* index into bsdfs                                            */
    int32_t  emitter;     /* index into emitters, or -1                                  */
    uint32_t flags;       /* MI_SHAPE_HAS_NORMALS                                        */
    uint32_t first_face, face_count;
} mi_shape;

typedef struct {          /* area light, src/emitters/area.cpp                            */
    uint32_t shape;       /* the shape it is attached to                                 */
    float radiance[3];    /* scalar_rgb                                                  */
    mi_texture radiance_tex;   /* scalar_spectral: MI_TEX_SRGB_D65 (<rgb>) or MI_TEX_D65 (<spectrum value>, default) */
} mi_emitter;

typedef struct {          /* environment map, src/emitters/envmap.cpp (one per scene)     */
    const float *rgba;    /* width * height * 4 linear RGBA floats, row 0 = +Y pole (bitmap->convert(RGBA, Float32)) */
    uint32_t width, height;                /* >= 2 x 2                                   */
    float scale;                           /* `scale` property                            */
    float to_world[16];                    /* `to_world`, column-major 4x4 (rotation part is used) */
    float bsphere_radius;                  /* scene->bbox().bounding_sphere().radius (set_scene, envmap.cpp:128-132
                                              applies the (1 + RayEpsilon) enlargement itself)  */
    uint32_t emitter_index;                /* position among the scene's emitters (Scene::m_emitters order,
                                              scene.cpp:38-60): area emitters at or after it shift up by one */
    const float *density;                  /* NULL, or width * height floats: the array the sampling warp is built from
                                              (envmap.cpp:81-116: luminance(rgb) * sin(theta) per texel). The scalar_spectral
                                              library REQUIRES it: there `rgba` holds what the reference's constructor stores
                                              instead of colours — per texel the three coefficients of the sRGB upsampling model
                                              of rgb / max(1e-8, scale) and scale = 2 * hmax(rgb) (envmap.cpp:101-110) — from
                                              which the luminance cannot be recovered. scalar_rgb: IGNORED (computed from `rgba`;
                                              the member was appended in round 4 — the struct grew by one pointer: callers built
                                              against the older header must be recompiled for the scalar_spectral library only) */
} mi_envmap;

/* Rectangle(props): [-1, 1]^2 in the z = 0 plane of object space, normal +z, placed by to_world
 * (flip_normals already folded in, rectangle.cpp:78-80). Matrices 4x4 column-major: m_to_world and its inverse. */
typedef struct {
    uint32_t shape;                        /* index into shapes (which carries MI_SHAPE_RECTANGLE) */
    float to_world[16], to_object[16];
} mi_rectangle;

/* Sphere(props) after update() (src/shapes/sphere.cpp:96-131): center and radius extracted from to_world, to_world
 * rebuilt from them (uniform scale, rotation, translation), its inverse, flip_normals. */
typedef struct {
    uint32_t shape;                        /* index into shapes (which carries MI_SHAPE_SPHERE) */
    float center[3], radius;
    uint32_t flip_normals;
    float to_world[16], to_object[16];
} mi_sphere;

typedef struct {
    const float    *vertex_positions;  /* 3 * vertex_count                               */
    const float    *vertex_normals;    /* 3 * vertex_count, or NULL                      */
    uint32_t        vertex_count;
    const uint32_t *faces;             /* 3 * face_count                                 */
    uint32_t        face_count;
    const mi_shape   *shapes;   uint32_t shape_count;
    const mi_bsdf    *bsdfs;    uint32_t bsdf_count;
    const mi_emitter *emitters; uint32_t emitter_count;
    const mi_envmap  *envmap;          /* or NULL                                        */
    const mi_rectangle *rectangles; uint32_t rectangle_count;   /* one per MI_SHAPE_RECTANGLE shape, or NULL / 0 */
    const mi_sphere *spheres; uint32_t sphere_count;            /* one per MI_SHAPE_SPHERE shape, or NULL / 0    */
    /* Mesh::vertex_texcoord (include/mitsuba/render/mesh.h:100-104): 2 * vertex_count, or NULL. Read for the shapes
     * that carry MI_SHAPE_HAS_TEXCOORDS: si.uv is interpolated from them and dp_du / dp_dv — hence the shading
     * frame's tangent — follow the uv parameterisation (src/librender/mesh.cpp:492-511). */
    const float    *vertex_texcoords;
    /* bitmap textures referenced by MI_TEX_BITMAP records of the bsdfs (emitter radiances stay constant) */
    const mi_bitmap *bitmaps; uint32_t bitmap_count;
    /* float tables plugins precompute in their constructors (roughplastic's eval_transmittance over
     * mu = i / 63, microfacet.h:504-552): one buffer, addressed by offsets stored in mi_bsdf::params */
    const float *bsdf_tables; uint32_t bsdf_table_floats;
} mi_scene_desc;
enum { MI_ROUGH_TRANSMITTANCE_RES = 64 };

/* ---- rays / hits for the Scene::ray_intersect surface ------------------------------- */
typedef struct {          /* SoA, n entries each (Ray3f: o, d, mint, maxt)               */
    const float *ox, *oy, *oz, *dx, *dy, *dz, *mint, *maxt;
} mi_rays_soa;

typedef struct {          /* PreliminaryIntersection3f: t, prim_uv, prim_index, shape    */
    float *t, *u, *v;     /* t = +inf on a miss; any-hit: t = 0 (hit) or +inf            */
    uint32_t *prim;       /* global primitive id, 0xffffffff on a miss (may be NULL)     */
    uint32_t *shape;      /* shape index, 0xffffffff on a miss (may be NULL)             */
} mi_hits_soa;

/* ---- render job ---------------------------------------------------------------------
 * Everything SamplingIntegrator::render() derives on the host
 * (src/librender/integrator.cpp:51-139) is passed in precomputed. */
typedef struct {
    int32_t crop_x, crop_y, crop_w, crop_h;   /* film->crop_offset() / crop_size()       */
    uint32_t spp;                             /* sampler->sample_count()                 */
    int32_t max_depth, rr_depth;              /* integrator.cpp:305-314                  */
    uint64_t base_seed;                       /* sampler `seed` property                 */
    int32_t block_size;                       /* integrator.cpp:88-97                    */
    /* spiral visitation id of every block, row-major over the block grid
     * (ceil(crop_w/bs) x ceil(crop_h/bs)); seeds are block_id*bs^2 + morton_i
     * (integrator.cpp:198, spiral.cpp:41) */
    const uint32_t *block_ids;
    uint32_t block_count;
    /* pixel-tile shard: row-major block indices this ctx renders; NULL = all      */
    const uint32_t *tile_list;
    uint32_t tile_count;
    /* perspective sensor (src/sensors/perspective.cpp:118-141), column-major 4x4  */
    float sample_to_camera[16];
    float to_world[16];
    float near_clip, far_clip;
    float principal_point_offset[2];
    /* reconstruction filter (include/mitsuba/core/rfilter.h:62-65, rfilter.cpp:9-20) */
    float filter_lut[32];
    float filter_radius;
    int32_t filter_border;
    /* output: crop_w*crop_h*5 values X,Y,Z,A,W (integrator.cpp:71-72)             */
    int32_t film_on_device;                   /* 0: host pointer, 1: device pointer      */
    int32_t film_f64;                         /* 0: float32 film, 1: float64 values      */
    /* how ImageBlock::put is realised:
     * 1 = sample log + ordered gather: float32 sums in the reference's order
     *     (bit-identical film; needs spp * lanes * 24 B of HBM),
     * 2 = float64 sums, order-free to float32 precision (resident plan: per-workgroup
     *     LDS tiles flushed once per launch; wavefront plan: atomics straight into the film),
     * 0 = auto: 1 if the log fits in free device memory, else 2               */
    int32_t film_mode;
    int32_t profile;                          /* 1: time every launch with HIP events    */
    float timeout_s;                          /* <= 0: none (integrator.cpp:34)          */
    /* execution plan of the sample loop (same arithmetic, same film, either way):
     * 1 = wavefront: SoA ray / hit / shadow queues in HBM, one kernel per stage
     *     (trace closest, shade, trace any) per depth-loop iteration,
     * 2 = resident: the whole depth loop of a pixel runs in registers, geometry in LDS
     *     (or read through L2), the pixel is advanced `samples_per_launch` samples per launch,
     * 0 = auto: 2 when the geometry is LDS-resident or the tree fits the LDS-stack walk, else 1 */
    int32_t plan;
    int32_t samples_per_launch;               /* plan 2: <= 0 = default (128)            */
    /* 1: `film` already holds earlier passes (samples_per_pass < sample_count, integrator.cpp:75-86: the blocks
     * of the p-th pass rendered carry ids (n_passes - 1 - p) * block_count + counter, spiral.cpp:41); this pass's block tiles are added onto it,
     * after the ids already in it. 0: the film is overwritten. */
    int32_t accumulate;
    /* which SamplingIntegrator::sample runs per camera sample:
     * MI_INTEGRATOR_PATH   PathIntegrator (src/integrators/path.cpp:100-211): max_depth, rr_depth above,
     * MI_INTEGRATOR_DIRECT DirectIntegrator (src/integrators/direct.cpp:78-198): emitter_samples + bsdf_samples
     *                      (>= 1 in total; `shading_samples` sets both, :89-96) and hide_emitters
     *                      (integrator.cpp:37); max_depth / rr_depth are ignored. Resident plan only. */
    int32_t integrator;
    uint32_t emitter_samples, bsdf_samples;
    int32_t hide_emitters;
    /* MomentIntegrator (src/integrators/moment.cpp) around the integrator above, scalar_rgb library: its film carries
     * X Y Z A W, nested.X nested.Y nested.Z and their squares m2_nested.* (:44-53, :83-88). Every channel is the
     * same filtered sum over the same samples, so the host renders the job twice with the same seeds:
     *   MI_MOMENT_VALUES   this call delivers X Y Z A W (nested.XYZ are the same numbers),
     *   MI_MOMENT_SQUARES  this call delivers X^2 Y^2 Z^2 A W (the m2_ channels);
     * in both a sample is dropped when any of the eleven values fails ImageBlock::put's test (imageblock.cpp:85-109). */
    int32_t moment_pass;
    /* ---- round 6 (appended; zero = the library chooses, and the context's options apply — mi_set_option below) --------------------
     * Per-render overrides of the library's own kernel choices: what the parity tests use to put ONE job through several kernels
     * (every choice produces the same film bits). Not tuning knobs: the defaults are the measured best. */
    int32_t debug_film_replay;  /* MI_FILM_REPLAY_*: which block replay assembles the film of a film_mode-1 render (and the log format it needs) */
    int32_t debug_tree_width;   /* 8 | 4: the tree the phase machine walks, when mi_bvh_build produced both                                       */
    int32_t debug_path_kernel;  /* MI_PATH_KERNEL_*: lock-step kernel / k_path_phased / k_path_pooled for a tree scene                          */
} mi_render_cfg;
enum { MI_MOMENT_OFF = 0, MI_MOMENT_VALUES = 1, MI_MOMENT_SQUARES = 2 };
enum { MI_FILM_REPLAY_AUTO = 0, MI_FILM_REPLAY_BLOCKS = 1 /* 24-byte position log + k_film_blocks */, MI_FILM_REPLAY_GROUPS = 2, MI_FILM_REPLAY_COLUMNS = 3,
       MI_FILM_REPLAY_QUADS = 4, MI_FILM_REPLAY_LANES = 5 /* tile-interleaved log */, MI_FILM_REPLAY_LANES_PLAIN_LOG = 6 };
enum { MI_PATH_KERNEL_AUTO = 0, MI_PATH_KERNEL_LOCKSTEP = 1, MI_PATH_KERNEL_PHASED = 2, MI_PATH_KERNEL_POOLED = 3 };
enum { MI_INTEGRATOR_PATH = 0, MI_INTEGRATOR_DIRECT = 1 };

typedef struct {
    uint64_t samples;          /* camera samples finished                               */
    uint64_t segments;         /* depth-loop iterations that reached shading            */
    uint64_t shadow_rays;
    uint64_t iterations;       /* wavefront iterations launched                         */
    uint64_t lanes;            /* pixels (lanes) in the job                             */
    double ms_render;          /* wall time of the last mi_render (host clock)          */
    /* HIP-event time per kernel class, summed over launches (profile=1 only)     */
    double ms_trace_closest, ms_trace_any, ms_shade, ms_init, ms_resolve;
    uint64_t n_trace_closest, n_trace_any, n_shade;
    double ms_bvh_build;
    uint32_t bvh_nodes, bvh_tris, bvh_depth;
    uint32_t film_mode;        /* 1 = ordered gather, 2 = float64 atomics (last render)   */
    uint32_t plan;             /* 1 = wavefront, 2 = resident (last render)               */
    double ms_path;            /* plan 2: HIP-event time of the k_path_resident launches  */
    uint64_t n_path;
    double ms_film_blocks, ms_film_merge;   /* split of ms_resolve for film_mode 1       */
    uint32_t bvh_on_device;    /* 1: the last mi_bvh_build ran a device builder (bvh_builder says which) */
    uint32_t path_kernel;      /* last render. plan 2: 0 = k_path_resident (lock-step lanes: packet scenes, direct integrator,
                                  float64 film, trees the 4-wide collapse refuses, scenes whose shape / BSDF / emitter tables do not fit a workgroup's LDS), 1 = k_path_phased (wave-level phase machine over the 4-wide
                                  quantised tree, per-lane LDS stack), 3 = k_path_phased over the BVH2 (MIW_BVH4=0, A/B switch).
                                  plan 1: 0 = k_trace<closest|any> per list slice, 2 = k_trace_stream (persistent walk kernel
                                  with dynamic ray fetch; ms_trace_closest = its time, ms_trace_any = k_sort_hits) */
    double ms_film_pack;       /* always 0 since round 3 (k_film_pack is gone: the render kernels log finished records); kept for layout */
    uint64_t log_bytes;        /* film_mode 1: bytes of the sample log of the last render (lanes x spp x log_record_bytes)          */
    uint32_t log_record_bytes; /* 16 = X Y Z + phase classes (filters film_classes.h covers), 24 = position + X Y Z alpha; 0: no log */
    uint32_t bvh4_on_device;   /* 1: the 4-wide tree of the last mi_bvh_build was collapsed on the device (quality 0)                */
    double ms_bvh4;            /* part of ms_bvh_build spent producing the 4-wide tree                                               */
    uint32_t placed;           /* 1: the last render ran a measuring launch + a launch with per-SIMD pixel queues (a shard of at most
                                  one pixel per resident lane: what one rank of an N-GPU frame renders at N >= 8)                     */
    uint32_t bvh_builder;      /* the last mi_bvh_build: 0 = host binned SAH (quality 1, or a scene the device sweep hands back), 3 = the same
                                  tree built on the device level by level (csrc/sah_device.h: quality 0 since round 4), 1 = device LBVH
                                  (radix tree over Morton codes: MIW_DEVICE_BUILDER=lbvh, A/B runs)                                      */
    /* ---- round 5 (appended: the fields above keep their offsets) ---- */
    uint32_t tree_width;       /* last render through k_path_phased: 8, 4 or 2 = the tree its node body walked (8: csrc/miw/bvh8.h,
                                  the default wherever mi_bvh_build produced it); 0: another kernel                                       */
    uint32_t bvh8_nodes;       /* the last mi_bvh_build: nodes of the 8-wide tree; 0: none (packet scene, radix tree, refused, MIW_BVH8=0) */
    uint32_t bvh8_depth;       /* its levels = the walk's worst-case stack entries + 1 (<= 16)                                           */
    uint32_t bvh8_on_device;   /* 1: programme + collapse + triangle gather ran on the device (quality 0)                                */
    double ms_bvh8;            /* part of ms_bvh_build spent on it                                                                       */
    /* placed launches (`placed` above): what the measuring launch found — the cost of the dearest pixel and the mean over the
     * shard's pixels, in wavefront iterations (place_cost_unit 0: packet kernel) or units of 256 shader clocks (1: phase machine),
     * over place_measure_spp samples per pixel; place_max_pixel = x | y << 16 of that pixel. 0 when the last render was not placed. */
    uint32_t place_cost_max, place_cost_unit, place_max_pixel, place_measure_spp;
    double place_cost_mean;
    /* the block replay of the last render with film_mode 1 (device/film_kernels.h): 0 = k_film_blocks (24-byte log), 1 = k_film_groups,
     * 2 = k_film_columns, 3 = k_film_quads (the default for shards of fewer than 448 tiles), 4 = k_film_lanes (the default from 448 tiles on),
     * and whether the render kernels wrote the 16-byte log interleaved over groups of 64 tiles for it (miw/film.h: log_index) */
    uint32_t film_kernel, log_interleaved;
    /* ---- round 6 (appended) ---- */
    uint32_t pooled;           /* 1: the last render's phase machine was k_path_pooled (device/pooled_kernel.h: walk jobs pooled across the
                                  workgroup through LDS; path_kernel stays 1, tree_width 8); 0: k_path_phased or another kernel             */
    uint32_t pool_waves;       /* wavefronts per workgroup of that launch (= jobs per LDS column); 0: not pooled                          */
    uint32_t film_overlapped;  /* 1: the last render's film replay (k_film_lanes) was queued beside the path kernel on three more streams, one launch per
                                  set of 64-tile groups behind the flags the groups' last finished pixels raise (full frames of the packet / lock-step
                                  kernels, one launch for all samples; MIW_FILM_OVERLAP=0 switches it off); ms_film_blocks then overlaps ms_path */
    uint32_t film_groups;      /* launches of that replay; 0: one launch after the path kernel                                            */
    uint32_t job_chunk;        /* chunk jobs of the last render's path kernel: 0 = a job is all the samples of a pixel; else the pixels' sample streams were cut into
                                  halving chunks down to this many samples (512 spp, 64: 256 + 128 + 64 + 64), drawn chunk-major from one queue, a pixel changing
                                  lanes between chunks (full frames: the launch's tail is as long as the smallest chunk, not as a pixel; MIW_JOB_CHUNK=0 switches it off) */
    uint32_t job_chunks;       /* chunks per pixel of that render (0: none)                                                                */
} mi_counters;

/* ---- entry points -------------------------------------------------------------------- */

/* 3 for the scalar_rgb library, 4 for the scalar_spectral library (channels of a Spectrum) */
int32_t mi_spectrum_channels(void);

/* Options: the switches INTEGRATION.md section 5 lists (A/B runs, fallbacks, test hooks; names MIW_*). mi_create copies the ones set
 * in the ENVIRONMENT into the context — the only time the library reads the environment; afterwards a context's behaviour does not
 * depend on its caller's environment. mi_set_option changes one for this context (value NULL: unset; unknown name: MI_ERR_INVALID),
 * mi_get_option reads it back (NULL: unset), mi_option_count / _name / _help enumerate them. Options consulted by mi_bvh_build
 * take effect at the next build, the others at the next render. */
mi_status mi_set_option(mi_ctx *ctx, const char *name, const char *value);
const char *mi_get_option(mi_ctx *ctx, const char *name);
int32_t mi_option_count(void);
const char *mi_option_name(int32_t i);
const char *mi_option_help(int32_t i);

/* number of visible HIP devices */
mi_status mi_device_count(int32_t *count);
/* create a context on `device`; *out is NULL on failure */
mi_status mi_create(int32_t device, mi_ctx **out);
void      mi_destroy(mi_ctx *ctx);
/* run all work on this hipStream_t (e.g. torch's current stream); NULL = default */
mi_status mi_set_stream(mi_ctx *ctx, void *hip_stream);

/* Scene(props): copy the flat scene to HBM; builds emitter sampling tables
 * (Mesh::build_pmf, src/librender/mesh.cpp:285-312) */
mi_status mi_scene_upload(mi_ctx *ctx, const mi_scene_desc *scene);
/* Scene::accel_init_cpu (src/librender/scene_native.inl:3-10): build the BVH.
 * quality 1 = binned SAH built on the host (csrc/bvh_build.h); quality 0 = THE SAME TREE built on the device level by level
 * (csrc/sah_device.h; MIW_DEVICE_BUILDER=lbvh: the radix tree of rounds 2 - 3, A/B runs) and collapsed on the device into the 4-wide
 * tree the render kernels walk. A scene the level sweep hands back (coincident centroids) gets the host builder.
 * Scenes of <= 64 triangles are traced by a brute-force sweep over LDS-resident
 * triangle packets instead of the tree; OR in MI_BVH_FORCE_TREE to walk the tree
 * anyway (tests). */
enum { MI_BVH_FORCE_TREE = 0x10,       /* walk the tree even for <= 64 triangles                  */
       MI_BVH_NO_LEAF_FILTER = 0x20,    /* resident plan: sweep every triangle, no leaf-box filter (tests) */
       MI_BVH_RADIX_TREE = 0x40 };      /* quality 0 only: the radix-tree builder over Morton codes (csrc/lbvh_device.h) instead of the
                                           SAH level sweep — ANOTHER tree, same answers (tests, A/B runs)   */
mi_status mi_bvh_build(mi_ctx *ctx, int32_t quality);

/* Scene::ray_intersect_preliminary (any_hit = 0) / Scene::ray_test (any_hit = 1),
 * include/mitsuba/render/scene.h:38-128, for n rays. */
mi_status mi_trace(mi_ctx *ctx, const mi_rays_soa *rays, const mi_hits_soa *hits,
                   uint64_t n, int32_t any_hit);

/* ---- the rest of the Scene query surface (include/mitsuba/render/scene.h:38-128) -------------------
 * The reference's integrators, BSDF tests and Python scripts call these on the Scene; the path kernels run the same
 * device functions in place. Records are arrays of structs, one per query. */
typedef struct {            /* SurfaceInteraction3f (include/mitsuba/render/interaction.h:104-199), the fields the hot path fills */
    float t;                /* +inf: the ray hit nothing (is_valid() false); then only wi = -ray.d (world) is meaningful  */
    float p[3];             /* position, interpolated from the vertices (mesh.cpp:484)                                    */
    float n[3];             /* geometric normal                                                                          */
    float sh_s[3], sh_t[3], sh_n[3];   /* shading frame (initialize_sh_frame, interaction.h:153-156)                      */
    float uv[2];            /* mesh texture coordinates if present, else the barycentrics (mesh.cpp:490-497)              */
    float wi[3];            /* incident direction in the shading frame (interaction.h:591)                               */
    uint32_t prim_index;    /* global primitive id, 0xffffffff on a miss                                                  */
    uint32_t shape_index;   /* 0xffffffff on a miss                                                                      */
    int32_t emitter_index;  /* si.emitter(scene), scene.h:243-253: the shape's area light, the environment map on a miss,
                               -1 for none; indexes the scene's emitter list (mi_scene_desc order, envmap slot included) */
} mi_surface_interaction;   /* 24 words */

typedef struct {            /* DirectionSample3f (include/mitsuba/render/records.h:120-214) without uv / time (no consumer on this path) */
    float p[3], n[3];       /* sampled position on the emitter, its normal                                               */
    float d[3], dist;       /* unit direction ref -> p, distance                                                          */
    float pdf;              /* solid-angle density; 0: no sample                                                          */
    int32_t emitter_index;  /* `object`: the emitter that was sampled                                                     */
} mi_direction_sample;      /* 12 words */

/* Scene::ray_intersect(ray) (scene.cpp:113-121 -> scene_native.inl:23-41): closest hit + full surface interaction */
mi_status mi_ray_intersect(mi_ctx *ctx, const mi_rays_soa *rays, mi_surface_interaction *si, uint64_t n);

/* Scene::sample_emitter_direction(ref, sample, test_visibility) (scene.cpp:164-214) for n reference points.
 *   emitter < 0: the scene's own choice among all emitters (scene.cpp:180-197, pdf and value rescaled);
 *   emitter >= 0: Endpoint::sample_direction of that emitter alone (endpoint.h:119-139; area.cpp:121-166, envmap.cpp:157-190).
 *   ref_p: 3 n floats; sample: 2 n floats; wavelengths: 4 n floats (scalar_spectral library), else NULL.
 *   test_visibility != 0: the shadow ray of scene.cpp:203-207 is traced; occluded samples come back with a zero spectrum.
 *   spec: N n floats (N = mi_spectrum_channels()): emitted radiance / pdf. */
mi_status mi_sample_emitter_direction(mi_ctx *ctx, int32_t emitter, const float *ref_p, const float *sample, const float *wavelengths,
                                      int32_t test_visibility, mi_direction_sample *ds, float *spec, uint64_t n);
/* Scene::pdf_emitter_direction(ref, ds) (scene.cpp:216-231; emitter < 0: ds[i].emitter_index, times 1 / emitter count) or
 * Endpoint::pdf_direction of one emitter (emitter >= 0; area.cpp:168-187, envmap.cpp:192-208). Reads ds[i].d / dist / n. */
mi_status mi_pdf_emitter_direction(mi_ctx *ctx, int32_t emitter, const float *ref_p, const mi_direction_sample *ds, float *pdf, uint64_t n);
/* Endpoint::eval(si) of the emitter each surface interaction sees, si.emitter(scene)->eval(si) (area.cpp:63-71,
 * envmap.cpp:134-147); zero where si[i].emitter_index < 0. spec: N n floats. */
mi_status mi_emitter_eval(mi_ctx *ctx, const mi_surface_interaction *si, const float *wavelengths, float *spec, uint64_t n);

/* SamplingIntegrator::render (integrator.cpp:51-179) with PathIntegrator::sample
 * (src/integrators/path.cpp:100-211): renders the ctx's tile shard into `film`.
 * Returns MI_ERR_CANCELLED if mi_cancel()/timeout stopped it (film holds the
 * partial result, like the reference's `return !m_stop`). */
mi_status mi_render(mi_ctx *ctx, const mi_render_cfg *cfg, void *film);
/* Integrator::cancel() — may be called from another thread */
mi_status mi_cancel(mi_ctx *ctx);

/* ---- N-GPU frames inside ONE process (round 5; SURVEY.md section 8e: pixel-tile shards + one film reduce) ----------------
 * The reference renders a frame with one call in one process (include/mitsuba/render/integrator.h:42). Over N GPUs that is:
 * one context per GPU (the scene uploaded to each), every context renders its shard (mi_render_cfg::tile_list) with
 * film_on_device = 1 into a device film of its own, then ONE reduce sums the films onto the root's.
 *   mi_film_alloc     `count` zeroed floats on ctx's device (a film = crop_w * crop_h * 5)
 *   mi_film_reduce    films[i] (resident on ctxs[i]'s device) summed onto films[root], in place. All devices distinct: RCCL
 *                     (ncclCommInitAll communicators cached per device list, one grouped ncclReduce over xGMI; librccl is
 *                     dlopen'ed on first use); contexts sharing a device, or no RCCL (MIW_RCCL=0): added in rank order by a
 *                     device kernel. *how (may be NULL) says which. Synchronous: waits for the contexts' streams before and after.
 *   mi_film_download  a device film to host memory. */
enum { MI_REDUCE_NONE = 0, MI_REDUCE_DEVICE_ADD = 1, MI_REDUCE_RCCL = 2 };
mi_status mi_film_alloc(mi_ctx *ctx, uint64_t count, void **device_film);
void      mi_film_free(mi_ctx *ctx, void *device_film);
mi_status mi_film_download(mi_ctx *ctx, const void *device_film, float *host, uint64_t count);
mi_status mi_film_reduce(mi_ctx *const *ctxs, void *const *films, int32_t n, uint64_t count, int32_t root, int32_t *how);

mi_status   mi_get_counters(mi_ctx *ctx, mi_counters *out);
const char *mi_last_error(mi_ctx *ctx);

/* Device-side evaluation of the leaf functions, for known-answer parity tests
 * (one work item per entry; `in`/`out` are host arrays of n * stride floats). */
enum {
    MI_EVAL_PCG32 = 0,            /* in: seed lo, seed hi (as bits)      out: 8 floats   */
    MI_EVAL_SINCOS = 1,           /* in: x                               out: sin, cos   */
    MI_EVAL_COSINE_HEMISPHERE = 2,/* in: u1,u2                           out: wo.xyz,pdf */
    MI_EVAL_BSDF = 3,             /* in: bsdf idx, wi.xyz, s1, s2x, s2y, wo.xyz (10) [+ wavelengths[4] in the spectral library]
                                     out: sample{wo.xyz,pdf,eta,type}, weight[N], eval[N], pdf (7 + 2N; N = mi_spectrum_channels) */
    MI_EVAL_FRESNEL = 4,          /* in: cos_theta_i, eta                out: r,cos_t,eta_it,eta_ti */
    MI_EVAL_CAMERA_RAY = 5,       /* in: x,y (film sample)               out: o.xyz,d.xyz,mint,maxt */
    MI_EVAL_EMITTER_SAMPLE = 6,   /* in: ref.xyz, u1, u2 [+ wavelengths[4]] out: d.xyz,dist,pdf,p.xyz,n.xyz,value[N] (11 + N) */
    MI_EVAL_FP_SEMANTICS = 7,     /* in: a,b,c                           out: a+b,a*b,a/b,sqrt|a|,fma(a,b,c),1/a,min,max (8) */
    MI_EVAL_SPECIAL = 8,          /* in: x                               out: exp, log, erf, erfinv (miw/special.h)   */
    MI_EVAL_ENVMAP = 9,           /* in: d.xyz (world), ref.xyz, u1, u2 (8) [+ wavelengths[4] in the spectral library]
                                     out: eval[N], pdf_direction, sample{d.xyz, dist, pdf, spec[N]} (6 + 2N)         */
    MI_EVAL_INVTRIG = 10,         /* in: y, x                            out: atan2(y,x), acos(x), asin(x) (3)       */
    MI_EVAL_SPECTRUM = 11,        /* spectral library only. in: wavelength sample, c0,c1,c2, d65 scale (5)
                                     out: wavelengths[4], weights[4], srgb[4], srgb_d65[4], xyz of weight*srgb_d65 (19) */
    MI_EVAL_TEXTURE = 12          /* BitmapTexture::eval at si.uv. in: u, v, (float) bitmap index, wavelength sample (4)
                                     out: the texture's Spectrum (3 / 4)                                             */
};
mi_status mi_eval(mi_ctx *ctx, int32_t op, const mi_render_cfg *cfg,
                  const float *in, int32_t in_stride, float *out, int32_t out_stride, uint64_t n);

/* Exhaustive device self-checks of leaf arithmetic whose device form differs from the host form.
 * MI_SELFTEST_RCP: miw::rcp (v_rcp_f32 + one Newton step inside [2^-126, 2^126), IEEE division elsewhere)
 * against the correctly rounded 1.f / x for all 2^32 float bit patterns. *mismatches = inputs whose bits differ
 * (NaN results count as equal). */
enum { MI_SELFTEST_RCP = 0 };
mi_status mi_selftest(mi_ctx *ctx, int32_t which, uint64_t *mismatches);

#ifdef __cplusplus
}
#endif
#endif /* MIWAVE_H */
