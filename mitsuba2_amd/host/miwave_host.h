// miwave host layer — C++17 classes with the shape of Mitsuba 2's plugin API
// for the path-integrator hot path, sitting on the C ABI of include/miwave.h.
//
// Class / method names, argument meaning and error behaviour follow the
// reference (file:line cited per class) so that a maintainer can put these
// where the scalar_rgb plugins are today: Scene::ray_intersect / ray_test,
// PathIntegrator::render, Sampler, Sensor, Film, BSDF and Emitter property
// parsing. Everything heavy is delegated to libmiwave.so (one mi_ctx per Scene);
// what stays here is exactly what the reference also does once on the host:
// Properties parsing, the spiral tile order, camera matrices, filter tables.
//
// Errors: std::runtime_error like the reference's Throw() (render() returns
// false only on cancel/timeout, src/librender/integrator.cpp:178).
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <limits>
#include <string>
#include <variant>
#include <vector>
#include <array>
#include <atomic>
#include <mutex>
#include <thread>

#include "../../include/miwave.h"

namespace miwave {

using Color3f = std::array<float, 3>;
using Point3f = std::array<float, 3>;
using Vector3f = std::array<float, 3>;

// ---- Transform4f (include/mitsuba/core/transform.h) ---------------------------------
// column-major 4x4 + its inverse, composed like the reference (analytic inverses).
struct Transform4f {
    float m[16];      // m[c*4 + r]
    float inv[16];
    Transform4f();
    static Transform4f translate(const Vector3f &v);                            // :166-174
    static Transform4f scale(const Vector3f &v);                                // :176-184
    static Transform4f perspective(float fov, float near_, float far_);         // :203-219
    static Transform4f look_at(const Point3f &origin, const Point3f &target, const Vector3f &up); // :241-269
    static Transform4f rotate(const Vector3f &axis, float angle_degrees);       // :175-178
    static Transform4f from_matrix(const float *row_major16);                   // <matrix value="..."/>
    Transform4f operator*(const Transform4f &o) const;                          // :58-62
    Transform4f inverse() const;                                                // :64-72
    bool has_scale() const;
};

class BitmapTexture;

// ---- Properties (include/mitsuba/core/properties.h) ------------------------------------
class Properties {
public:
    using Value = std::variant<bool, int64_t, float, std::string, Color3f, Transform4f, std::shared_ptr<BitmapTexture>>;
    Properties() = default;
    explicit Properties(std::string plugin_name) : m_plugin_name(std::move(plugin_name)) {}
    const std::string &plugin_name() const { return m_plugin_name; }
    bool has_property(const std::string &name) const { return m_values.count(name) != 0; }
    void set_bool(const std::string &n, bool v) { m_values[n] = v; }
    void set_int(const std::string &n, int64_t v) { m_values[n] = v; }
    void set_float(const std::string &n, float v) { m_values[n] = v; }
    void set_string(const std::string &n, const std::string &v) { m_values[n] = v; }
    void set_color(const std::string &n, const Color3f &v) { m_values[n] = v; }
    void set_transform(const std::string &n, const Transform4f &v) { m_values[n] = v; }
    // a nested <texture type="bitmap" name=...> object (Properties::set_object / texture<>(), properties.h:307-315)
    void set_texture(const std::string &n, std::shared_ptr<BitmapTexture> t) { m_values[n] = std::move(t); }
    std::shared_ptr<BitmapTexture> bitmap(const std::string &n) const;     // nullptr: a constant (or absent)
    bool bool_(const std::string &n) const;
    bool bool_(const std::string &n, bool def) const;
    int64_t int_(const std::string &n) const;
    int64_t int_(const std::string &n, int64_t def) const;
    float float_(const std::string &n) const;
    float float_(const std::string &n, float def) const;
    std::string string(const std::string &n) const;
    std::string string(const std::string &n, const std::string &def) const;
    // Properties::texture(name, default) for constant textures only
    // (properties.h:307-315; <rgb>/<spectrum value> -> srgb/uniform, xml.cpp:1073-1170)
    Color3f texture(const std::string &n, float def) const;
    Color3f texture(const std::string &n) const;
    float texture_mean(const std::string &n, float def) const;   // Texture::mean() of that constant texture
    Transform4f transform(const std::string &n, const Transform4f &def) const;
    // The texture object an <rgb> / <spectrum value> property expands to in the compiled variant
    // (src/libcore/xml.cpp:1073-1100): srgb / srgb_d65 / uniform / d65, as a C-ABI record.
    // `within_emitter`: the property belongs to an emitter; `unbounded`: xml.cpp's is_unbounded_spectrum
    // names (eta, k, int_ior, ext_ior) skip the [0, 1] reflectance check (src/spectra/srgb.cpp:30-31).
    mi_texture texture_record(const std::string &n, float def, bool within_emitter, bool unbounded) const;
private:
    std::string m_plugin_name;
    std::map<std::string, Value> m_values;
};

// ---- Bitmap texture (src/textures/bitmap.cpp) -------------------------------------------------------
// Properties: filter_type ("bilinear" | "nearest"), wrap_mode ("repeat" | "mirror" | "clamp"), raw (false),
// to_uv (identity), filename (a PFM file; other image formats are out of scope — hand the pixels over with
// set_bitmap instead). Values are linear floats: 1 channel (Y) or 3 (RGB). What the constructor of the reference
// does with them happens in record(): in the scalar_spectral build an RGB image that is not `raw` is replaced by
// its per-texel sRGB-model coefficients (bitmap.cpp:156-165).
class BitmapTexture {
public:
    explicit BitmapTexture(const Properties &props);
    void set_bitmap(uint32_t width, uint32_t height, uint32_t channels, const float *data);
    uint32_t width() const { return m_width; }
    uint32_t height() const { return m_height; }
    uint32_t channels() const { return m_channels; }
    Color3f mean() const;                                      // per-channel mean (Texture::mean() is its luminance)
    mi_bitmap record() const;                                  // pointers stay valid while this object lives
private:
    void finish();
    uint32_t m_width = 0, m_height = 0, m_channels = 0, m_filter = MI_BITMAP_BILINEAR, m_wrap = MI_BITMAP_REPEAT;
    bool m_raw = false; Transform4f m_to_uv; std::string m_name;
    std::vector<float> m_data, m_device_data;
};
// Bitmap(path) for Portable Float Maps ("PF" = RGB, "Pf" = Y; bottom-to-top scanlines; src/libcore/bitmap.cpp read_pfm)
void read_pfm(const std::string &path, uint32_t &width, uint32_t &height, uint32_t &channels, std::vector<float> &data);

// ---- sRGB -> spectrum upsampling model (src/librender/srgb.cpp:14-42, ext/rgb2spec/rgb2spec.c) ----
// scalar_spectral only. The coefficient table `data/srgb.coeff` is an artefact of the reference's
// build (rgb2spec_opt 64 srgb.coeff, ext/rgb2spec/CMakeLists.txt:47-52); point the layer at it with
// set_srgb_model_path() or the MIWAVE_SRGB_COEFF environment variable.
void set_srgb_model_path(const std::string &path);
std::array<float, 3> srgb_model_fetch(const Color3f &c);
// 3 = scalar_rgb, 4 = scalar_spectral build of this layer
int spectrum_channels();

// ---- ReconstructionFilter (include/mitsuba/core/rfilter.h, src/libcore/rfilter.cpp) ----
class ReconstructionFilter {
public:
    virtual ~ReconstructionFilter() = default;
    float radius() const { return m_radius; }
    uint32_t border_size() const { return m_border_size; }
    virtual float eval(float x) const = 0;
    float eval_discretized(float x) const;                     // rfilter.h:62-65
    const std::vector<float> &values() const { return m_values; }
protected:
    void init_discretization();                                // rfilter.cpp:9-20
    float m_radius = 0.f, m_scale_factor = 0.f;
    uint32_t m_border_size = 0;
    std::vector<float> m_values;
};
class GaussianFilter final : public ReconstructionFilter {    // src/rfilters/gaussian.cpp:32-47
public:
    explicit GaussianFilter(const Properties &props = Properties("gaussian"));
    float eval(float x) const override;
private:
    float m_stddev, m_alpha, m_bias;
};
class BoxFilter final : public ReconstructionFilter {         // src/rfilters/box.cpp
public:
    explicit BoxFilter(const Properties &props = Properties("box"));
    float eval(float x) const override;
};

// src/rfilters/{tent,mitchell,catmullrom,lanczos}.cpp. The device takes any filter as its 32-entry table + radius
// (mi_render_cfg::filter_lut): footprints up to 4 x 4 texels replay through k_film_groups, wider ones (lanczos)
// through k_film_blocks. On-device parity has been verified for box and gaussian; the others share their code paths.
class TentFilter final : public ReconstructionFilter {
public:
    explicit TentFilter(const Properties &props = Properties("tent"));          // tent.cpp:25-31: radius 1
    float eval(float x) const override;
private:
    float m_inv_radius;
};
class MitchellNetravaliFilter final : public ReconstructionFilter {
public:
    explicit MitchellNetravaliFilter(const Properties &props = Properties("mitchell"));   // mitchell.cpp:28-36: radius 2, B = C = 1/3
    MitchellNetravaliFilter(float b, float c);                                  // catmullrom.cpp: B = 0, C = 1/2
    float eval(float x) const override;
private:
    float m_b, m_c;
};
class LanczosSincFilter final : public ReconstructionFilter {
public:
    explicit LanczosSincFilter(const Properties &props = Properties("lanczos")); // lanczos.cpp:33-38: radius = lobes (3)
    float eval(float x) const override;
};
// PluginManager::create_object<ReconstructionFilter>(props): box, tent, gaussian, mitchell, catmullrom, lanczos
std::shared_ptr<ReconstructionFilter> make_rfilter(const Properties &props);

// ---- Film / HDRFilm (src/librender/film.cpp:7-50, src/films/hdrfilm.cpp) ---------------
class Film {
public:
    explicit Film(const Properties &props = Properties("hdrfilm"));
    int width() const { return m_size[0]; }
    int height() const { return m_size[1]; }
    std::array<int, 2> size() const { return m_size; }
    std::array<int, 2> crop_size() const { return m_crop_size; }
    std::array<int, 2> crop_offset() const { return m_crop_offset; }
    const ReconstructionFilter *reconstruction_filter() const { return m_filter.get(); }
    void set_reconstruction_filter(std::shared_ptr<ReconstructionFilter> f) { m_filter = std::move(f); }
    void prepare(const std::vector<std::string> &channels);    // hdrfilm.cpp:190-205
    // raw XYZAW storage, crop_w*crop_h*5 float32 (hdrfilm.cpp:200-204)
    std::vector<float> &storage() { return m_storage; }
    const std::vector<float> &storage() const { return m_storage; }
    // develop: divide by W, XYZ -> linear sRGB (hdrfilm.cpp:251-322, bitmap.cpp:187-188)
    std::vector<float> bitmap_rgb() const;
    // HDRFilm::set_destination_file / develop (hdrfilm.cpp:213-217,327-345): writes bitmap() in `file_format`
    // ("openexr" (default) / "pfm"; "rgbe" is not provided), `pixel_format` rgb or rgba, `component_format`
    // float16 (default for OpenEXR) / float32; the extension is replaced by the format's proper one.
    void set_destination_file(const std::string &filename) { m_dest_file = filename; }
    std::string develop() const;                               // -> the path written
private:
    std::array<int, 2> m_size, m_crop_size, m_crop_offset;
    std::string m_dest_file, m_file_format = "openexr", m_pixel_format = "rgb", m_component_format = "float16";
    std::shared_ptr<ReconstructionFilter> m_filter;
    std::vector<float> m_storage;
    std::vector<std::string> m_channels;
};

// ---- Spiral (src/librender/spiral.cpp) ---------------------------------------------------
class Spiral {
public:
    Spiral(std::array<int, 2> size, std::array<int, 2> offset, size_t block_size, size_t passes = 1);
    size_t block_count() const { return m_cells.size(); }
    size_t max_block_size() const { return m_block_size; }
    void reset();
    // (offset, size, block_id); size == 0 when exhausted (spiral.cpp:27-72)
    struct Block { std::array<int, 2> offset, size; size_t block_id; };
    Block next_block();
private:
    std::vector<std::array<int, 2>> m_cells;                   // block positions in visiting order (film_sensor.inl: spiral_cells)
    size_t m_cursor, m_block_size;
    std::array<int, 2> m_size, m_offset;
    size_t m_passes_left;
};

// ---- Sampler (include/mitsuba/render/sampler.h, src/samplers/independent.cpp) -------------
class IndependentSampler {
public:
    explicit IndependentSampler(const Properties &props = Properties("independent"));
    std::shared_ptr<IndependentSampler> clone() const;
    void seed(uint64_t seed_offset);                           // sampler.cpp:83-96
    void advance() {}
    float next_1d();
    std::array<float, 2> next_2d();
    size_t sample_count() const { return m_sample_count; }
    uint64_t base_seed() const { return m_base_seed; }
private:
    size_t m_sample_count; uint64_t m_base_seed; uint64_t m_state, m_inc;
};

// ---- Sensor (src/sensors/perspective.cpp, src/librender/sensor.cpp) ------------------------
struct Ray3f { Point3f o; Vector3f d; float mint = 0.f, maxt = 0.f; };

class PerspectiveCamera {
public:
    explicit PerspectiveCamera(const Properties &props, std::shared_ptr<Film> film,
                               std::shared_ptr<IndependentSampler> sampler);
    const std::shared_ptr<Film> &film() const { return m_film; }
    const std::shared_ptr<IndependentSampler> &sampler() const { return m_sampler; }
    float near_clip() const { return m_near_clip; }
    float far_clip() const { return m_far_clip; }
    float x_fov() const { return m_x_fov; }
    const Transform4f &world_transform() const { return m_to_world; }
    const Transform4f &sample_to_camera() const { return m_sample_to_camera; }
    std::array<float, 2> principal_point_offset() const { return m_pp_offset; }
    // perspective.cpp:182-216 (position_sample already divided by the crop size)
    Ray3f sample_ray(const std::array<float, 2> &position_sample) const;
private:
    void update_camera_transforms();                           // :118-141
    std::shared_ptr<Film> m_film;
    std::shared_ptr<IndependentSampler> m_sampler;
    Transform4f m_to_world, m_camera_to_sample, m_sample_to_camera;
    float m_near_clip, m_far_clip, m_x_fov;
    std::array<float, 2> m_pp_offset;
};
float parse_fov(const Properties &props, float aspect);       // src/librender/sensor.cpp:113-167

// ---- BSDF plugins ----------------------------------------------------------------------------
struct BSDFSample3f { Vector3f wo; float pdf, eta; uint32_t sampled_type; };
struct SurfaceInteraction3f;
// BSDFContext (bsdf.h:146-190). The device loop and the host helpers implement the context PathIntegrator uses —
// radiance transport, every lobe enabled; any other context is refused (NotImplementedError in the reference's terms).
enum class TransportMode : uint32_t { Radiance = 0, Importance = 1 };
struct BSDFContext {
    TransportMode mode = TransportMode::Radiance;
    uint32_t type_mask = 0x1ffu;                               // BSDFFlags::All
    uint32_t component = (uint32_t) -1;
    bool is_full() const { return mode == TransportMode::Radiance && (type_mask & 0x1ffu) == 0x1ffu && component == (uint32_t) -1; }
};
class BSDF {
public:
    virtual ~BSDF() = default;
    uint32_t flags() const;                                    // bsdf.h:417-428
    // The reference's signatures (bsdf.h:328-394): sample(ctx, si, sample1, sample2), eval(ctx, si, wo), pdf(ctx, si, wo);
    // si.wi is the incident direction in the shading frame. Full contexts only (see BSDFContext).
    std::pair<BSDFSample3f, Color3f> sample(const BSDFContext &ctx, const SurfaceInteraction3f &si, float sample1, const std::array<float, 2> &sample2) const;
    Color3f eval(const BSDFContext &ctx, const SurfaceInteraction3f &si, const Vector3f &wo) const;
    float pdf(const BSDFContext &ctx, const SurfaceInteraction3f &si, const Vector3f &wo) const;
    // BSDF::sample / eval / pdf in local coordinates (bsdf.h:328-394), scalar semantics
    std::pair<BSDFSample3f, Color3f> sample(const Vector3f &wi, float sample1, const std::array<float, 2> &sample2) const;
    Color3f eval(const Vector3f &wi, const Vector3f &wo) const;
    float pdf(const Vector3f &wi, const Vector3f &wo) const;
    const mi_bsdf &record() const { return m_rec; }             // bitmap parameters appear as their mean here;
    // the bitmap bound to texture slot k (nullptr: constant). Scene::build turns it into a MI_TEX_BITMAP record.
    const std::shared_ptr<BitmapTexture> &bitmap(int k) const { return m_bitmaps[k]; }
    // the float table the plugin's record addresses with params[5] (roughplastic: 64 entries; empty otherwise).
    // Scene::build appends it to mi_scene_desc::bsdf_tables and stores the offset in the record.
    const std::vector<float> &table() const { return m_table; }
    // twosided adapter: the nested back-side BSDF (nullptr: not twosided; == this: same BSDF on both sides)
    const std::shared_ptr<BSDF> &back() const { return m_back; }
    bool twosided() const { return (m_rec.flags & MI_BSDF_FLAG_TWOSIDED) != 0; }
protected:
    // m_rec.tex[slot] = props.texture_record(...) and remember the bitmap, if the property holds one
    void bind_texture(int slot, const Properties &props, const std::string &name, float def, bool unbounded);
    mi_bsdf m_rec{};
    std::shared_ptr<BSDF> m_back;
    std::shared_ptr<BitmapTexture> m_bitmaps[3];
    std::vector<float> m_table;
};
class SmoothDiffuse final : public BSDF { public: explicit SmoothDiffuse(const Properties &props); };        // diffuse.cpp:72-76
class SmoothDielectric final : public BSDF { public: explicit SmoothDielectric(const Properties &props); };  // dielectric.cpp:174-199
class RoughConductor final : public BSDF { public: explicit RoughConductor(const Properties &props); };      // roughconductor.cpp:146-194
class SmoothConductor final : public BSDF { public: explicit SmoothConductor(const Properties &props); };    // conductor.cpp:201-215
class SmoothPlastic final : public BSDF { public: explicit SmoothPlastic(const Properties &props); };        // plastic.cpp:135-174
class RoughDielectric final : public BSDF { public: explicit RoughDielectric(const Properties &props); };    // roughdielectric.cpp:146-201
// roughplastic.cpp:146-181 + parameters_changed :336-371: isotropic Beckmann / GGX coating over a diffuse base; its
// constructor integrates the rough transmittance table (eval_transmittance, microfacet.h:504-552) and the internal
// reflectance (eval_reflectance, :454-502) with Gauss-Legendre quadrature (src/libcore/quad.cpp:7-64)
class RoughPlastic final : public BSDF { public: explicit RoughPlastic(const Properties &props); };
// nodes and weights of the n-point Gauss-Legendre rule on [-1, 1] (quad::gauss_legendre)
void gauss_legendre(int n, std::vector<float> &nodes, std::vector<float> &weights);
// twosided.cpp:62-92: wraps one nested BRDF (both sides) or two (front, back); nested BSDFs must not transmit
class TwoSidedBRDF final : public BSDF { public: explicit TwoSidedBRDF(std::shared_ptr<BSDF> front, std::shared_ptr<BSDF> back = nullptr); };
float fresnel_diffuse_reflectance(float eta);                                                                 // fresnel.h:327-361
float lookup_ior(const Properties &props, const std::string &name, const std::string &def);                   // include/mitsuba/render/ior.h

// ---- Emitter / Shape ----------------------------------------------------------------------------
#if MIW_SPECTRAL
using Spectrum = std::array<float, 4>;
#else
using Spectrum = std::array<float, 3>;
#endif
class Scene; class Mesh; class Emitter;
struct Frame3f { Vector3f s, t, n; };
// SurfaceInteraction3f (interaction.h:104-199): what Scene::ray_intersect returns (mi_surface_interaction + the
// pointers the reference keeps in it)
struct SurfaceInteraction3f {
    float t = std::numeric_limits<float>::infinity();
    Point3f p{}; Vector3f n{}; Frame3f sh_frame{}; std::array<float, 2> uv{}; Vector3f wi{};
    std::array<float, 4> wavelengths{};                       // scalar_spectral: set by the caller before eval calls
    uint32_t prim_index = 0xffffffffu, shape_index = 0xffffffffu; int32_t emitter_index = -1;
    const Mesh *shape = nullptr;
    bool is_valid() const { return t != std::numeric_limits<float>::infinity(); }
    const Emitter *emitter(const Scene *scene) const;          // scene.h:243-253
    const BSDF *bsdf() const;                                  // bsdf.h:485-500
    Vector3f to_world(const Vector3f &v) const;                // interaction.h:58-61
    Vector3f to_local(const Vector3f &v) const;
};
// the reference point of an emitter query: Interaction3f (interaction.h:27-101) reduced to what is read
struct Interaction3f { Point3f p{}; std::array<float, 4> wavelengths{}; };
// DirectionSample3f (records.h:120-214)
struct DirectionSample3f {
    Point3f p{}; Vector3f n{}; Vector3f d{}; float dist = 0.f, pdf = 0.f; bool delta = false;
    int32_t emitter_index = -1; const Emitter *object = nullptr;
};
// Endpoint / Emitter (endpoint.h:86-163): evaluated on the device of the scene the emitter was built into
class Emitter {
public:
    virtual ~Emitter() = default;
    Spectrum eval(const SurfaceInteraction3f &si) const;                                                        // endpoint.h:155-163
    std::pair<DirectionSample3f, Spectrum> sample_direction(const Interaction3f &it, const std::array<float, 2> &sample) const;   // :119-139
    float pdf_direction(const Interaction3f &it, const DirectionSample3f &ds) const;                            // :141-153
    bool is_environment() const { return m_is_env; }
    int32_t index() const { return m_index; }                  // position in Scene::emitters(), -1 before Scene::build
protected:
    friend class Scene;
    const Scene *m_scene = nullptr; int32_t m_index = -1; bool m_is_env = false;
};
class AreaLight final : public Emitter {                      // src/emitters/area.cpp:52-60
public:
    explicit AreaLight(const Properties &props);
    Color3f radiance() const { return m_radiance; }
    const mi_texture &radiance_texture() const { return m_radiance_tex; }   // srgb_d65 / d65 in spectral builds
private:
    Color3f m_radiance; mi_texture m_radiance_tex{};
};

// src/emitters/envmap.cpp. The reference loads `filename` through Bitmap (out of scope here: no
// image I/O on the path); the linear RGBA float32 pixels it would hold after
// bitmap->convert(RGBA, Float32) (:71-75) are handed over with set_bitmap().
class EnvironmentMapEmitter final : public Emitter {
public:
    explicit EnvironmentMapEmitter(const Properties &props);  // `scale` (:124), `to_world` (endpoint.cpp)
    void set_bitmap(uint32_t width, uint32_t height, const float *rgba);
    uint32_t width() const { return m_width; }
    uint32_t height() const { return m_height; }
    float scale() const { return m_scale; }
    const Transform4f &world_transform() const { return m_to_world; }
    const std::vector<float> &data() const { return m_data; }        // m_data: RGBA, or (scalar_spectral) c0 c1 c2 scale per texel, :101-110
    const std::vector<float> &density() const { return m_density; }  // scalar_spectral: luminance(rgb) * sin(theta), :113 (empty in scalar_rgb)
private:
    float m_scale; Transform4f m_to_world;
    uint32_t m_width = 0, m_height = 0;
    std::vector<float> m_data, m_density;
};

class Mesh {                                                  // include/mitsuba/render/mesh.h
public:
    Mesh(std::string name, std::vector<float> vertex_positions, std::vector<uint32_t> faces,
         std::vector<float> vertex_normals = {}, std::vector<float> vertex_texcoords = {});
    uint32_t vertex_count() const { return (uint32_t) (m_positions.size() / 3); }
    uint32_t face_count() const { return (uint32_t) (m_faces.size() / 3); }
    uint32_t primitive_count() const { return face_count(); }
    // Shape::bbox (mesh.cpp:98-112) as {min xyz, max xyz}; Mesh::surface_area (mesh.cpp:285-312, 417-420: the face
    // areas summed in double, as build_pmf feeds them to the DiscreteDistribution)
    std::array<float, 6> bbox() const;
    float surface_area() const;
    bool has_vertex_normals() const { return !m_normals.empty(); }
    const std::vector<float> &vertex_positions_buffer() const { return m_positions; }
    const std::vector<float> &vertex_normals_buffer() const { return m_normals; }
    // Mesh::has_vertex_texcoords / vertex_texcoord (mesh.h:100-104): 2 floats per vertex; they parameterise si.uv
    // and the tangents of the shading frame (mesh.cpp:492-511)
    bool has_vertex_texcoords() const { return !m_texcoords.empty(); }
    const std::vector<float> &vertex_texcoords_buffer() const { return m_texcoords; }
    const std::vector<uint32_t> &faces_buffer() const { return m_faces; }
    // Mesh::recompute_vertex_normals (src/librender/mesh.cpp:200-246): angle-weighted face normals
    // (Thuermer & Wuethrich 1998); creates the normal buffer if the mesh has none
    void recompute_vertex_normals();
    void discard_vertex_normals() { m_normals.clear(); }       // `face_normals = true` (m_disable_vertex_normals)
    void set_bsdf(std::shared_ptr<BSDF> b) { m_bsdf = std::move(b); }
    void set_emitter(std::shared_ptr<AreaLight> e) { m_emitter = std::move(e); }
    const std::shared_ptr<BSDF> &bsdf() const { return m_bsdf; }
    const std::shared_ptr<AreaLight> &emitter() const { return m_emitter; }
    const std::string &name() const { return m_name; }
    // the analytic `rectangle` shape (make_rectangle): one primitive, intersected / sampled analytically on the
    // device; its vertex buffer holds the four corners (scene bounds), its single face entry is a placeholder
    bool is_rectangle() const { return m_rectangle; }
    const Transform4f &rectangle_to_world() const { return m_rect_to_world; }
    // the analytic `sphere` shape (make_sphere): one primitive; vertex buffer = the corners of its bounding box
    bool is_sphere() const { return m_sphere; }
    const mi_sphere &sphere_record() const { return m_sphere_rec; }
private:
    friend std::shared_ptr<Mesh> make_rectangle(const Properties &props);
    friend std::shared_ptr<Mesh> make_sphere(const Properties &props);
    std::string m_name;
    std::vector<float> m_positions, m_normals, m_texcoords;
    std::vector<uint32_t> m_faces;
    std::shared_ptr<BSDF> m_bsdf;
    std::shared_ptr<AreaLight> m_emitter;
    bool m_rectangle = false; Transform4f m_rect_to_world;
    bool m_sphere = false; mi_sphere m_sphere_rec{};
};
// The `rectangle` shape plugin (src/shapes/rectangle.cpp:76-84): [-1, 1]^2 in z = 0, normal +z, properties
// to_world (identity) and flip_normals (false). An analytic primitive — not two triangles.
std::shared_ptr<Mesh> make_rectangle(const Properties &props);
// The `sphere` shape plugin (src/shapes/sphere.cpp:96-131): properties center (0), radius (1), to_world (identity:
// rotation, translation, uniform scale only), flip_normals (false). An analytic primitive.
std::shared_ptr<Mesh> make_sphere(const Properties &props);

// Mesh file loaders: the `obj` and `ply` shape plugins.
// Properties: filename, face_normals (false), to_world (identity); obj: flip_tex_coords (true).
std::shared_ptr<Mesh> load_obj(const Properties &props);      // src/shapes/obj.cpp:95-345
std::shared_ptr<Mesh> load_ply(const Properties &props);      // src/shapes/ply.cpp (ascii / binary_little_endian / big_endian)

// ---- Scene (include/mitsuba/render/scene.h:38-161, src/librender/scene.cpp) --------------------
struct PreliminaryIntersection3f { float t; float u, v; uint32_t prim_index, shape_index; bool is_valid() const; };

class Scene {
public:
    Scene();
    ~Scene();
    Scene(const Scene &) = delete;
    void add_shape(std::shared_ptr<Mesh> mesh);               // scene.cpp:33-61
    // an <emitter type="envmap"> child: takes its place in the emitter order after the shapes added
    // so far; "Only one environment emitter can be specified per scene." (scene.cpp:47-50)
    void add_emitter(std::shared_ptr<EnvironmentMapEmitter> env);
    const EnvironmentMapEmitter *environment() const { return m_env.get(); }   // scene.h:150-151
    // finishes construction: default BSDFs (shape.cpp:75-81), flatten, upload, build accel (scene.cpp:94-97)
    void build(int device = 0, int bvh_quality = 0);   // 0: the binned-SAH tree built on the device (csrc/sah_device.h), 1: by the host recursion
    // The same scene resident on SEVERAL GPUs of the node (round 5): one context per entry of `devices` (devices[0] = the primary,
    // ctx(); an index may repeat — several contexts on one GPU, which is how the single-GPU test tier runs the path), uploads and
    // BVH builds in parallel. SamplingIntegrator::render(scene, sensor) then shards the frame's tiles over the contexts and
    // closes it with one film reduce (mi_film_reduce: RCCL over xGMI between distinct GPUs) — still ONE call in ONE process,
    // like the reference's Integrator::render (include/mitsuba/render/integrator.h:42).
    void build(const std::vector<int> &devices, int bvh_quality = 0);
    size_t device_count() const { return m_ctx ? 1 + m_replicas.size() : 0; }
    mi_ctx *ctx(size_t i) const { return i == 0 ? m_ctx : m_replicas.at(i - 1); }
    const std::vector<std::shared_ptr<Mesh>> &shapes() const { return m_shapes; }
    size_t emitter_count() const { return m_emitters.size(); }
    std::array<float, 6> bbox() const;                         // Scene::bbox(): union of the shapes' boxes
    // Scene::ray_intersect_preliminary / ray_test for one ray or a batch
    PreliminaryIntersection3f ray_intersect_preliminary(const Ray3f &ray) const;
    bool ray_test(const Ray3f &ray) const;
    // Scene::ray_intersect (scene.h:38-60, scene.cpp:113-121): closest hit + full SurfaceInteraction3f
    SurfaceInteraction3f ray_intersect(const Ray3f &ray) const;
    void ray_intersect(const mi_rays_soa &rays, mi_surface_interaction *si, uint64_t n) const;
    // Scene::sample_emitter_direction / pdf_emitter_direction (scene.h:98-128, scene.cpp:164-231)
    std::pair<DirectionSample3f, Spectrum> sample_emitter_direction(const Interaction3f &ref, const std::array<float, 2> &sample,
                                                                     bool test_visibility = true) const;
    float pdf_emitter_direction(const Interaction3f &ref, const DirectionSample3f &ds) const;
    // Scene::emitters() (scene.h:139-141): area lights and the environment map in the scene's emitter order
    const std::vector<const Emitter *> &emitters() const { return m_emitter_objs; }
    void ray_intersect_preliminary(const mi_rays_soa &rays, const mi_hits_soa &hits, uint64_t n) const;
    void ray_test(const mi_rays_soa &rays, float *t_out, uint64_t n) const;
    mi_ctx *ctx() const { return m_ctx; }
    const mi_scene_desc &desc() const { return m_desc; }
private:
    std::vector<std::shared_ptr<Mesh>> m_shapes;
    std::vector<float> m_positions, m_normals, m_texcoords;
    std::vector<uint32_t> m_faces;
    std::vector<mi_shape> m_shape_recs;
    std::vector<mi_bsdf> m_bsdf_recs;
    std::vector<mi_emitter> m_emitters;
    std::vector<const Emitter *> m_emitter_objs;
    std::vector<mi_rectangle> m_rect_recs; std::vector<mi_sphere> m_sphere_recs;
    std::vector<mi_bitmap> m_bitmap_recs; std::vector<std::shared_ptr<BitmapTexture>> m_bitmap_objs;
    std::vector<float> m_bsdf_tables;
    std::shared_ptr<EnvironmentMapEmitter> m_env; size_t m_env_after_shapes = 0; mi_envmap m_env_rec{};
    mi_scene_desc m_desc{};
    mi_ctx *m_ctx = nullptr;
    std::vector<mi_ctx *> m_replicas;                          // the contexts of devices[1..] of a multi-GPU build
    bool m_built = false;
};

// ---- Integrator (include/mitsuba/render/integrator.h, src/librender/integrator.cpp) -------------
// SamplingIntegrator: the block / seed / film machinery every sampling integrator shares (integrator.cpp:23-179);
// the plugin picks which `sample()` the device loop runs per camera sample (fill_integrator).
class SamplingIntegrator {
public:
    explicit SamplingIntegrator(const Properties &props);      // integrator.cpp:23-38
    virtual ~SamplingIntegrator() = default;
    // SamplingIntegrator::render (integrator.cpp:51-179). Returns !m_stop.
    virtual bool render(Scene *scene, PerspectiveCamera *sensor);
    // Film channels render() produces (integrator.cpp:67-73: X Y Z A W + aov_names())
    virtual std::vector<std::string> aov_names() const { return {}; }
    virtual void cancel();                                     // integrator.cpp:43-45
    uint32_t block_size() const { return m_block_size; }
    // pixel-tile shard for multi-GPU: this process renders blocks with
    // (spiral index % world_size) == rank; the film holds the partial sum.
    void set_shard(uint32_t rank, uint32_t world_size) { m_rank = rank; m_world = world_size; }
    // how the last render() over a multi-GPU scene (Scene::build(devices)) summed its partial films: MI_REDUCE_* of include/miwave.h
    int last_reduce() const { return m_last_reduce; }
    // fills everything SamplingIntegrator::render derives on the host
    // for pass `pass` of pass_count(sensor) (samples_per_pass < sample_count, integrator.cpp:75-86). Passes are
    // in execution order; like spiral.cpp:41 the first pass rendered carries the highest block-id offset: pass p's blocks carry ids (n_passes - 1 - p) * block_count + spiral counter
    // (spiral.cpp:41), the film adds block tiles in ascending id, so render() runs p = 0, 1, ... and every pass
    // after the first accumulates onto the film (mi_render_cfg::accumulate).
    void make_render_cfg(const PerspectiveCamera *sensor, mi_render_cfg &cfg,
                         std::vector<uint32_t> &block_ids, std::vector<uint32_t> &tiles,
                         uint32_t n_threads_hint = 1, uint32_t pass = 0) const;
    // the same for an explicit (rank, world): context r of a multi-GPU scene renders shard (m_rank * n + r, m_world * n)
    void make_render_cfg(const PerspectiveCamera *sensor, mi_render_cfg &cfg,
                         std::vector<uint32_t> &block_ids, std::vector<uint32_t> &tiles,
                         uint32_t n_threads_hint, uint32_t pass, uint32_t rank, uint32_t world) const;
    uint32_t pass_count(const PerspectiveCamera *sensor) const;
    const mi_counters &counters() const { return m_counters; }
    void set_profile(bool p) { m_profile = p; }
    // execution plan of the device sample loop (mi_render_cfg::plan): 0 auto, 1 wavefront, 2 resident
    void set_plan(int plan) { m_plan = plan; }
    bool hide_emitters() const { return m_hide_emitters; }
    // mi_render_cfg::integrator and the plugin's own parameters
    virtual void fill_integrator(mi_render_cfg &cfg) const = 0;
    // all passes of one job into `film5` (crop_w * crop_h * 5 floats); moment_pass = mi_render_cfg::moment_pass
    bool render_passes(Scene *scene, PerspectiveCamera *sensor, float *film5, int moment_pass);
protected:
    // one pass of a frame over the contexts of a multi-GPU scene: every context its shard on its own host thread, then the film reduce
    bool render_pass_multi(Scene *scene, PerspectiveCamera *sensor, float *film5, int moment_pass, uint32_t pass);
    int m_last_reduce = 0;
    std::mutex m_active_mutex; std::vector<mi_ctx *> m_active_multi;
    uint32_t m_block_size; uint32_t m_samples_per_pass; float m_timeout; bool m_hide_emitters;
    uint32_t m_rank = 0, m_world = 1;
    bool m_profile = false;
    int m_plan = 0;
    std::atomic<mi_ctx *> m_active_ctx{nullptr};
    mi_counters m_counters{};
};

// src/integrators/path.cpp (MonteCarloIntegrator parameters, integrator.cpp:305-314)
class PathIntegrator final : public SamplingIntegrator {
public:
    explicit PathIntegrator(const Properties &props = Properties("path"));
    int max_depth() const { return m_max_depth; }
    int rr_depth() const { return m_rr_depth; }
protected:
    void fill_integrator(mi_render_cfg &cfg) const override;
private:
    int m_max_depth, m_rr_depth;
};

// src/integrators/direct.cpp:78-104: shading_samples | emitter_samples + bsdf_samples, hide_emitters
class DirectIntegrator final : public SamplingIntegrator {
public:
    explicit DirectIntegrator(const Properties &props = Properties("direct"));
    size_t emitter_samples() const { return m_emitter_samples; }
    size_t bsdf_samples() const { return m_bsdf_samples; }
protected:
    void fill_integrator(mi_render_cfg &cfg) const override;
private:
    size_t m_emitter_samples, m_bsdf_samples;
};

// src/integrators/moment.cpp around ONE nested sampling integrator (scalar_rgb build): the film gets the channels
// X Y Z A W <name>.X <name>.Y <name>.Z m2_<name>.X m2_<name>.Y m2_<name>.Z — what the reference's z-test harness
// (src/python/python/test/test_renders.py) estimates per-pixel variances from. The device renders the job twice with
// the same seeds (values, then squares: mi_render_cfg::moment_pass); block_size / samples_per_pass / timeout are the
// nested integrator's.
class MomentIntegrator final : public SamplingIntegrator {
public:
    MomentIntegrator(const Properties &props, std::shared_ptr<SamplingIntegrator> nested, std::string nested_name = "integrator");
    bool render(Scene *scene, PerspectiveCamera *sensor) override;
    void cancel() override { m_nested->cancel(); }
    std::vector<std::string> aov_names() const override;
    const std::shared_ptr<SamplingIntegrator> &nested() const { return m_nested; }
    void fill_integrator(mi_render_cfg &cfg) const override { m_nested->fill_integrator(cfg); }
private:
    std::shared_ptr<SamplingIntegrator> m_nested; std::string m_name;
};

// PluginManager::create_object<Integrator>(props) for the integrators built here ("path", "direct")
std::shared_ptr<SamplingIntegrator> make_integrator(const Properties &props);

// ---- XML scene front-end, a subset (src/libcore/xml.cpp) --------------------------
// <scene>, <default>, <include>, <alias>, $parameters (also from `params`), <shape type="obj|ply|rectangle|sphere">, <bsdf> (inline, or
// top-level with id + <ref id=.../>), <texture type="bitmap"> (PFM files; nested in a <bsdf> under the parameter's
// name, or top-level with id + <ref id=... name=.../>), <emitter type="area">, <emitter type="envmap"> (PFM file), <sensor type="perspective"> with <film>,
// <sampler>, <rfilter> children, <integrator type="path|direct">; values <float> <integer> <boolean> <string> <rgb>
// <spectrum value=...>; <transform name="to_world"> of <translate> <scale> <rotate> <lookat> <matrix>.
// Anything else throws the reference's kind of error ("unexpected ..."/"Plugin ... not found").
struct LoadedScene {
    std::shared_ptr<Scene> scene;
    std::shared_ptr<PerspectiveCamera> sensor;                 // nullptr if the file has none
    std::shared_ptr<SamplingIntegrator> integrator;            // default-constructed `path` if the file has none
    std::vector<std::shared_ptr<Mesh>> shapes;
};
LoadedScene load_xml_string(const std::string &xml, const std::map<std::string, std::string> &params = {},
                            const std::string &base_dir = ".");
LoadedScene load_xml_file(const std::string &path, const std::map<std::string, std::string> &params = {});

} // namespace miwave
