// Implementation of the host classes declared in miwave_host.h, plus the C
// facade (`mih_*`) that Python's ctypes binds for tests and the benchmark.
#include "miwave_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "../csrc/miw/base.h"
#include "../csrc/miw/rng.h"
#include "../csrc/miw/bsdf.h"
#include "../csrc/miw/scene.h"
#include "../csrc/miw/special.h"
#include <cctype>

namespace miwave {

static std::string to_lower(std::string s) { for (char &c : s) c = (char) std::tolower(c); return s; }

[[noreturn]] static void Throw(const std::string &msg) { throw std::runtime_error(msg); }

// ============================================================================================
// Transform4f
// ============================================================================================
static void mat_identity(float *m) { std::memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.f; }
// enoki Matrix * Matrix: column j of the result = sum_k A.col(k) * B(k, j), fma chain
static void mat_mul(const float *a, const float *b, float *out) {
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float s = a[0 * 4 + i] * b[j * 4 + 0];
            for (int k = 1; k < 4; ++k) s = miw::fmadd(a[k * 4 + i], b[j * 4 + k], s);
            r[j * 4 + i] = s;
        }
    std::memcpy(out, r, 64);
}

Transform4f::Transform4f() { mat_identity(m); mat_identity(inv); }

Transform4f Transform4f::translate(const Vector3f &v) {
    Transform4f t;
    t.m[12] = v[0]; t.m[13] = v[1]; t.m[14] = v[2];
    t.inv[12] = -v[0]; t.inv[13] = -v[1]; t.inv[14] = -v[2];
    return t;
}
Transform4f Transform4f::scale(const Vector3f &v) {
    Transform4f t;
    t.m[0] = v[0]; t.m[5] = v[1]; t.m[10] = v[2];
    t.inv[0] = 1.f / v[0]; t.inv[5] = 1.f / v[1]; t.inv[10] = 1.f / v[2];
    return t;
}
Transform4f Transform4f::perspective(float fov, float near_, float far_) {
    float recip = 1.f / (far_ - near_);
    float tan_ = std::tan(fov * .5f * (MIW_PI / 180.f)), cot = 1.f / tan_;
    Transform4f t;
    std::memset(t.m, 0, 64); std::memset(t.inv, 0, 64);
    // trafo = diag(cot, cot, far*recip, 0); trafo(2,3) = -near*far*recip; trafo(3,2) = 1   [(row, col)]
    t.m[0] = cot; t.m[5] = cot; t.m[10] = far_ * recip;
    t.m[3 * 4 + 2] = -near_ * far_ * recip;
    t.m[2 * 4 + 3] = 1.f;
    // inv = diag(tan, tan, 0, 1/near); inv(2,3) = 1; inv(3,2) = (near - far) / (far * near)
    t.inv[0] = tan_; t.inv[5] = tan_; t.inv[15] = 1.f / near_;
    t.inv[3 * 4 + 2] = 1.f;
    t.inv[2 * 4 + 3] = (near_ - far_) / (far_ * near_);
    return t;
}
Transform4f Transform4f::look_at(const Point3f &origin, const Point3f &target, const Vector3f &up) {
    using namespace miw;
    V3 o = v3(origin[0], origin[1], origin[2]);
    V3 dir = normalize(v3(target[0], target[1], target[2]) - o);
    dir = normalize(dir);
    V3 left = normalize(cross(v3(up[0], up[1], up[2]), dir));
    V3 new_up = cross(dir, left);
    Transform4f t;
    float *m = t.m;
    m[0] = left.x;  m[1] = left.y;  m[2] = left.z;  m[3] = 0.f;
    m[4] = new_up.x; m[5] = new_up.y; m[6] = new_up.z; m[7] = 0.f;
    m[8] = dir.x;   m[9] = dir.y;   m[10] = dir.z;  m[11] = 0.f;
    m[12] = o.x;    m[13] = o.y;    m[14] = o.z;    m[15] = 1.f;
    // inverse = rows (left, new_up, dir), last column = inverse * (-origin, 1)
    float *iv = t.inv;
    std::memset(iv, 0, 64);
    iv[0] = left.x; iv[4] = left.y; iv[8] = left.z;
    iv[1] = new_up.x; iv[5] = new_up.y; iv[9] = new_up.z;
    iv[2] = dir.x; iv[6] = dir.y; iv[10] = dir.z;
    iv[15] = 1.f;
    float col[4];
    for (int i = 0; i < 4; ++i) {
        float s = iv[0 * 4 + i] * (-o.x);
        s = fmadd(iv[1 * 4 + i], -o.y, s);
        s = fmadd(iv[2 * 4 + i], -o.z, s);
        s = fmadd(iv[3 * 4 + i], 1.f, s);
        col[i] = s;
    }
    iv[12] = col[0]; iv[13] = col[1]; iv[14] = col[2]; iv[15] = col[3];
    return t;
}
Transform4f Transform4f::operator*(const Transform4f &o) const {
    Transform4f r;
    mat_mul(m, o.m, r.m);
    mat_mul(o.inv, inv, r.inv);
    return r;
}
Transform4f Transform4f::inverse() const {
    Transform4f r;
    std::memcpy(r.m, inv, 64); std::memcpy(r.inv, m, 64);
    return r;
}
bool Transform4f::has_scale() const {
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) {
            float sum = 0.f;
            for (int k = 0; k < 3; ++k) sum += m[i * 4 + k] * m[j * 4 + k];
            if (i == j && std::abs(sum - 1.f) > 1e-3f) return true;
        }
    return false;
}

// ============================================================================================
// Properties
// ============================================================================================
template <typename T> static const T *prop_get(const std::map<std::string, Properties::Value> &m, const std::string &n) {
    auto it = m.find(n);
    if (it == m.end()) return nullptr;
    return std::get_if<T>(&it->second);
}
#define MIW_PROP_GETTER(fn, T, type_name)                                                            \
    T Properties::fn(const std::string &n) const {                                                   \
        if (!has_property(n)) Throw("Property \"" + n + "\" has not been specified!");               \
        const T *v = prop_get<T>(m_values, n);                                                       \
        if (!v) Throw("The property \"" + n + "\" has the wrong type (expected <" type_name ">).");  \
        return *v;                                                                                   \
    }                                                                                                \
    T Properties::fn(const std::string &n, T def) const {                                            \
        if (!has_property(n)) return def;                                                            \
        return fn(n);                                                                                \
    }
MIW_PROP_GETTER(bool_, bool, "boolean")
MIW_PROP_GETTER(int_, int64_t, "integer")
std::string Properties::string(const std::string &n) const {
    if (!has_property(n)) Throw("Property \"" + n + "\" has not been specified!");
    const std::string *v = prop_get<std::string>(m_values, n);
    if (!v) Throw("The property \"" + n + "\" has the wrong type (expected <string>).");
    return *v;
}
std::string Properties::string(const std::string &n, const std::string &def) const {
    return has_property(n) ? string(n) : def;
}
float Properties::float_(const std::string &n) const {
    if (!has_property(n)) Throw("Property \"" + n + "\" has not been specified!");
    if (const float *v = prop_get<float>(m_values, n)) return *v;
    if (const int64_t *v = prop_get<int64_t>(m_values, n)) return (float) *v;
    Throw("The property \"" + n + "\" has the wrong type (expected <float>).");
}
float Properties::float_(const std::string &n, float def) const { return has_property(n) ? float_(n) : def; }
std::shared_ptr<BitmapTexture> Properties::bitmap(const std::string &n) const {
    if (!has_property(n)) return nullptr;
    if (const std::shared_ptr<BitmapTexture> *v = prop_get<std::shared_ptr<BitmapTexture>>(m_values, n)) return *v;
    return nullptr;
}
Color3f Properties::texture(const std::string &n) const {
    if (!has_property(n)) Throw("Property \"" + n + "\" has not been specified!");
    if (auto b = bitmap(n)) { Color3f m = b->mean(); for (float &v : m) v = std::min(std::max(v, 0.f), 1.f); return m; }
    if (const Color3f *v = prop_get<Color3f>(m_values, n)) return *v;
    if (const float *v = prop_get<float>(m_values, n)) return Color3f{ *v, *v, *v };
    Throw("The property \"" + n + "\" has the wrong type (expected <rgb> or <float>; only constant textures are supported).");
}
Color3f Properties::texture(const std::string &n, float def) const {
    return has_property(n) ? texture(n) : Color3f{ def, def, def };
}
// ---- sRGB upsampling model -------------------------------------------------------------------
int spectrum_channels() { return MIW_SPEC_N; }
namespace {
struct SRGBModel { uint32_t res = 0; std::vector<float> scale, data; };
std::mutex g_model_mutex; std::string g_model_path; SRGBModel g_model;
const SRGBModel &srgb_model() {
    std::lock_guard<std::mutex> lock(g_model_mutex);
    if (g_model.res) return g_model;
    std::string path = g_model_path;
    if (path.empty()) if (const char *e = std::getenv("MIWAVE_SRGB_COEFF")) path = e;
    if (path.empty()) Throw("Could not load sRGB-to-spectrum upsampling model ('data/srgb.coeff'): set MIWAVE_SRGB_COEFF");
    FILE *f = std::fopen(path.c_str(), "rb");                  // rgb2spec_load, rgb2spec.c:13-47
    char header[4]; uint32_t res = 0;
    if (!f || std::fread(header, 4, 1, f) != 1 || std::memcmp(header, "SPEC", 4) != 0 || std::fread(&res, 4, 1, f) != 1 || res < 2) {
        if (f) std::fclose(f);
        Throw("Could not load sRGB-to-spectrum upsampling model ('" + path + "')");
    }
    SRGBModel m; m.res = res; m.scale.resize(res); m.data.resize((size_t) res * res * res * 9);
    bool ok = std::fread(m.scale.data(), 4, res, f) == res && std::fread(m.data.data(), 4, m.data.size(), f) == m.data.size();
    std::fclose(f);
    if (!ok) Throw("Could not load sRGB-to-spectrum upsampling model ('" + path + "'): truncated file");
    g_model = std::move(m);
    return g_model;
}
}
void set_srgb_model_path(const std::string &path) { std::lock_guard<std::mutex> lock(g_model_mutex); g_model_path = path; g_model = SRGBModel(); }

// srgb.cpp:28-39 + rgb2spec_fetch (rgb2spec.c:76-124): table addressed by the largest component,
// trilinear interpolation of the three sigmoid-polynomial coefficients
std::array<float, 3> srgb_model_fetch(const Color3f &c) {
    const float inf = std::numeric_limits<float>::infinity();
    if (c[0] == 0.f && c[1] == 0.f && c[2] == 0.f) return { 0.f, 0.f, -inf };
    if (c[0] == 1.f && c[1] == 1.f && c[2] == 1.f) return { 0.f, 0.f, inf };
    const SRGBModel &m = srgb_model();
    const int res = (int) m.res;
    float rgb[3];
    for (int j = 0; j < 3; ++j) rgb[j] = std::max(std::min(c[j], 1.f), 0.f);
    int i = 0;
    for (int j = 1; j < 3; ++j) if (rgb[j] >= rgb[i]) i = j;
    const float z = rgb[i], sc = (float) (res - 1) / z, x = rgb[(i + 1) % 3] * sc, y = rgb[(i + 2) % 3] * sc;
    const uint32_t xi = std::min((uint32_t) x, (uint32_t) (res - 2)), yi = std::min((uint32_t) y, (uint32_t) (res - 2));
    int left = 0, last = res - 2, size = last;                 // rgb2spec_find_interval
    while (size > 0) {
        int half = size >> 1, middle = left + half + 1;
        if (m.scale[middle] <= z) { left = middle; size -= half + 1; } else size = half;
    }
    const uint32_t zi = (uint32_t) std::min(left, last);
    size_t offset = ((((size_t) i * res + zi) * res + yi) * res + xi) * 3;
    const size_t dx = 3, dy = 3 * (size_t) res, dz = 3 * (size_t) res * res;
    const float x1 = x - (float) xi, x0 = 1.f - x1, y1 = y - (float) yi, y0 = 1.f - y1,
                z1 = (z - m.scale[zi]) / (m.scale[zi + 1] - m.scale[zi]), z0 = 1.f - z1;
    std::array<float, 3> out;
    const float *d = m.data.data();
    for (int j = 0; j < 3; ++j, ++offset)
        out[j] = ((d[offset] * x0 + d[offset + dx] * x1) * y0 + (d[offset + dy] * x0 + d[offset + dy + dx] * x1) * y1) * z0 +
                 ((d[offset + dz] * x0 + d[offset + dz + dx] * x1) * y0 + (d[offset + dz + dy] * x0 + d[offset + dz + dy + dx] * x1) * y1) * z1;
    return out;
}

mi_texture Properties::texture_record(const std::string &n, float def, bool within_emitter, bool unbounded) const {
    mi_texture t{};
    bool is_color = false; Color3f color{ def, def, def }; float value = def;
    if (has_property(n)) {
        if (bitmap(n)) { is_color = true; color = texture(n); }   // host-side stand-in: the (clamped) mean colour
        else if (const Color3f *v = prop_get<Color3f>(m_values, n)) { is_color = true; color = *v; }
        else if (const float *v = prop_get<float>(m_values, n)) { value = *v; color = { *v, *v, *v }; }
        else Throw("The property \"" + n + "\" has the wrong type (expected <rgb> or <float>; only constant textures are supported).");
    }
#if MIW_SPECTRAL
    if (is_color) {
        if (within_emitter) {                                  // srgb_d65.cpp:33-47
            float scale = std::max(color[0], std::max(color[1], color[2])) * 2.f;
            if (scale != 0.f) { float r = 1.f / scale; color = { color[0] * r, color[1] * r, color[2] * r }; }
            auto cf = srgb_model_fetch(color);
            t.type = MI_TEX_SRGB_D65; t.v[0] = cf[0]; t.v[1] = cf[1]; t.v[2] = cf[2];
            t.v[3] = (1.f * scale) * (1.f / 10568.f);          // d65.cpp:61-62 with scale = props.scale * scale
        } else {                                               // srgb.cpp:27-35
            if (!unbounded) for (float v : color) if (v < 0.f || v > 1.f)
                Throw("Invalid RGB reflectance value, must be in the range [0, 1]!");
            auto cf = srgb_model_fetch(color);
            t.type = MI_TEX_SRGB; t.v[0] = cf[0]; t.v[1] = cf[1]; t.v[2] = cf[2];
        }
    } else if (within_emitter) { t.type = MI_TEX_D65; t.v[0] = value * (1.f / 10568.f); }   // xml.cpp:1097-1099
    else { t.type = MI_TEX_UNIFORM; t.v[0] = value; }
#else
    (void) within_emitter; (void) value;
    if (is_color && !within_emitter && !unbounded) for (float v : color) if (v < 0.f || v > 1.f)
        Throw("Invalid RGB reflectance value, must be in the range [0, 1]!");
    t.type = MI_TEX_RGB; t.v[0] = color[0]; t.v[1] = color[1]; t.v[2] = color[2];
#endif
    return t;
}

// Texture::mean() of the constant texture a property resolves to: uniform.cpp (value), srgb.cpp:52-57
// (RGB: hmean of the colour; spectral: hmean of the model over 16 wavelengths, srgb.h:26-35).
float Properties::texture_mean(const std::string &n, float def) const {
    if (has_property(n)) {
        if (const Color3f *v = prop_get<Color3f>(m_values, n)) {
#if MIW_SPECTRAL
            auto c = srgb_model_fetch(*v);
            const float step = (830.f - 360.f) / 15.f;
            float sum = 0.f;
            for (int i = 0; i < 16; ++i) {
                float lambda = std::fma((float) i, step, 360.f);
                float x = std::fma(std::fma(c[0], lambda, c[1]), lambda, c[2]);
                float r = std::isinf(c[2]) ? std::fma(c[2] < 0.f ? -1.f : 1.f, .5f, .5f)
                                           : std::max(0.f, std::fma(.5f * x, 1.f / std::sqrt(std::fma(x, x, 1.f)), .5f));
                sum += r;
            }
            return sum * (1.f / 16.f);
#else
            return (((*v)[0] + (*v)[1]) + (*v)[2]) * (1.f / 3.f);
#endif
        }
        if (const float *v = prop_get<float>(m_values, n)) return *v;
        Throw("The property \"" + n + "\" has the wrong type (expected <rgb> or <float>; only constant textures are supported).");
    }
    return def;
}

Transform4f Properties::transform(const std::string &n, const Transform4f &def) const {
    if (!has_property(n)) return def;
    const Transform4f *v = prop_get<Transform4f>(m_values, n);
    if (!v) Throw("The property \"" + n + "\" has the wrong type (expected <transform>).");
    return *v;
}

// ============================================================================================
// Reconstruction filters
// ============================================================================================
float ReconstructionFilter::eval_discretized(float x) const {
    int index = std::min((int) std::abs(x * m_scale_factor), 31);
    return m_values[index];
}
void ReconstructionFilter::init_discretization() {
    const int RES = 31;                                        // MTS_FILTER_RESOLUTION
    m_values.resize(RES + 1);
    for (int i = 0; i < RES; ++i) m_values[i] = eval((m_radius * i) / RES);
    m_values[RES] = 0;
    m_scale_factor = RES / m_radius;
    m_border_size = (uint32_t) (int) std::ceil(m_radius - .5f - 2.f * MIW_RAY_EPSILON);
}
GaussianFilter::GaussianFilter(const Properties &props) {
    m_stddev = props.float_("stddev", 0.5f);
    m_radius = 4 * m_stddev;
    m_alpha = -1.f / (2.f * m_stddev * m_stddev);
    m_bias = std::exp(m_alpha * (m_radius * m_radius));
    init_discretization();
}
float GaussianFilter::eval(float x) const { return std::max(0.f, std::exp(m_alpha * (x * x)) - m_bias); }
BoxFilter::BoxFilter(const Properties &props) {
    m_radius = props.float_("radius", .5f) + MIW_RAY_EPSILON;
    init_discretization();
}
float BoxFilter::eval(float x) const { return std::abs(x) <= m_radius ? 1.f : 0.f; }

// ============================================================================================
// Film
// ============================================================================================
Film::Film(const Properties &props) {
    m_size = { (int) props.int_("width", 768), (int) props.int_("height", 576) };
    m_crop_offset = { (int) props.int_("crop_offset_x", 0), (int) props.int_("crop_offset_y", 0) };
    m_crop_size = { (int) props.int_("crop_width", m_size[0]), (int) props.int_("crop_height", m_size[1]) };
    // set_crop_window, film.cpp:54-66
    if (m_crop_offset[0] < 0 || m_crop_offset[1] < 0 || m_crop_size[0] <= 0 || m_crop_size[1] <= 0 ||
        m_crop_offset[0] + m_crop_size[0] > m_size[0] || m_crop_offset[1] + m_crop_size[1] > m_size[1])
        Throw("Invalid crop window specification!");
    m_filter = std::make_shared<GaussianFilter>();             // film.cpp:45-49
    // hdrfilm.cpp:95-180: output format properties
    m_file_format = to_lower(props.string("file_format", "openexr"));
    m_pixel_format = to_lower(props.string("pixel_format", "rgb"));
    m_component_format = to_lower(props.string("component_format", "float16"));
    if (m_file_format != "openexr" && m_file_format != "exr" && m_file_format != "pfm")
        Throw("The \"file_format\" parameter must either be equal to \"openexr\" or \"pfm\" in this layer (\"rgbe\" is not provided). Found " + m_file_format + ".");
    if (m_pixel_format != "rgb" && m_pixel_format != "rgba")
        Throw("The \"pixel_format\" parameter must either be equal to \"rgb\" or \"rgba\" in this layer. Found " + m_pixel_format + ".");
    if (m_component_format != "float16" && m_component_format != "float32")
        Throw("The \"component_format\" parameter must either be equal to \"float16\" or \"float32\". Found " + m_component_format + " instead.");
    if (m_file_format == "pfm") { m_pixel_format = "rgb"; m_component_format = "float32"; }      // :170-180
}
void Film::prepare(const std::vector<std::string> &channels) {
    m_channels = channels;
    m_storage.assign((size_t) m_crop_size[0] * m_crop_size[1] * channels.size(), 0.f);
}
std::vector<float> Film::bitmap_rgb() const {
    size_t n = (size_t) m_crop_size[0] * m_crop_size[1];
    std::vector<float> rgb(n * 3);
    const size_t stride = std::max<size_t>(m_channels.size(), 5);      // X Y Z A W first; AOV channels (moment) behind them
    for (size_t i = 0; i < n; ++i) {
        const float *p = &m_storage[i * stride];
        float inv_w = p[4] != 0.f ? 1.f / p[4] : 0.f;         // struct.cpp:1734-1745 weight normalisation
        miw::V3 c = miw::xyz_to_srgb(miw::v3(p[0] * inv_w, p[1] * inv_w, p[2] * inv_w));
        rgb[i * 3] = c.x; rgb[i * 3 + 1] = c.y; rgb[i * 3 + 2] = c.z;
    }
    return rgb;
}

// float32 -> IEEE half, round to nearest even (what Bitmap::convert does for component_format float16)
static uint16_t float_to_half(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t) (sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));   // inf / nan
    if (x >= 0x477ff000u) return (uint16_t) (sign | 0x7c00u);                                       // overflow -> inf
    if (x < 0x33000001u) return (uint16_t) sign;                                                    // underflow -> 0
    int e = (int) (x >> 23) - 127 + 15; uint32_t m = x & 0x7fffffu;
    if (e <= 0) {                                              // subnormal half
        m |= 0x800000u; int shift = 14 - e;
        uint32_t h = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (uint16_t) (sign | h);
    }
    uint32_t h = ((uint32_t) e << 10) | (m >> 13), rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (uint16_t) (sign | h);
}

std::string Film::develop() const {
    if (m_dest_file.empty()) Throw("Destination file not specified, cannot develop.");
    const bool exr = m_file_format == "openexr" || m_file_format == "exr";
    const bool rgba = m_pixel_format == "rgba";
    std::string path = m_dest_file; const std::string ext = exr ? ".exr" : ".pfm";
    size_t dot = path.find_last_of('.'), slash = path.find_last_of("/\\");
    if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) path = path.substr(0, dot);
    path += ext;
    const int W = m_crop_size[0], H = m_crop_size[1];
    std::vector<float> rgb = bitmap_rgb();
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) Throw("Could not open \"" + path + "\" for writing");
    if (!exr) {                                                // PFM: "PF", bottom-to-top scanlines, little endian
        std::fprintf(f, "PF\n%d %d\n-1.0\n", W, H);
        for (int y = H - 1; y >= 0; --y) std::fwrite(&rgb[(size_t) y * W * 3], 4, (size_t) W * 3, f);
        std::fclose(f);
        return path;
    }
    // OpenEXR 2, single-part scanline image, no compression
    const bool half = m_component_format != "float32";
    const int nch = rgba ? 4 : 3; const char *names = rgba ? "ABGR" : "BGR";      // channels are stored alphabetically
    std::vector<unsigned char> hdr;
    auto put = [&](const void *p, size_t n) { hdr.insert(hdr.end(), (const unsigned char *) p, (const unsigned char *) p + n); };
    auto put_str = [&](const char *s) { put(s, std::strlen(s) + 1); };
    auto put_i32 = [&](int32_t v) { put(&v, 4); };
    auto put_f32 = [&](float v) { put(&v, 4); };
    auto attr = [&](const char *name, const char *type, int32_t size) { put_str(name); put_str(type); put_i32(size); };
    const uint32_t magic = 20000630u, version = 2u;
    put(&magic, 4); put(&version, 4);
    attr("channels", "chlist", nch * 18 + 1);
    for (int c = 0; c < nch; ++c) { char nm[2] = { names[c], 0 }; put_str(nm); put_i32(half ? 1 : 2); unsigned char z[4] = { 0, 0, 0, 0 }; put(z, 4); put_i32(1); put_i32(1); }
    { unsigned char z = 0; put(&z, 1); }
    attr("compression", "compression", 1); { unsigned char z = 0; put(&z, 1); }
    attr("dataWindow", "box2i", 16); put_i32(0); put_i32(0); put_i32(W - 1); put_i32(H - 1);
    attr("displayWindow", "box2i", 16); put_i32(0); put_i32(0); put_i32(W - 1); put_i32(H - 1);
    attr("lineOrder", "lineOrder", 1); { unsigned char z = 0; put(&z, 1); }
    attr("pixelAspectRatio", "float", 4); put_f32(1.f);
    attr("screenWindowCenter", "v2f", 8); put_f32(0.f); put_f32(0.f);
    attr("screenWindowWidth", "float", 4); put_f32(1.f);
    { unsigned char z = 0; put(&z, 1); }
    const size_t bpc = half ? 2 : 4, line_bytes = (size_t) W * nch * bpc;
    std::fwrite(hdr.data(), 1, hdr.size(), f);
    uint64_t offset = hdr.size() + (uint64_t) H * 8;
    for (int y = 0; y < H; ++y) { std::fwrite(&offset, 8, 1, f); offset += 8 + line_bytes; }
    std::vector<unsigned char> line(line_bytes);
    for (int y = 0; y < H; ++y) {
        int32_t yy = y, sz = (int32_t) line_bytes;
        std::fwrite(&yy, 4, 1, f); std::fwrite(&sz, 4, 1, f);
        for (int c = 0; c < nch; ++c) {
            const char ch = names[c];
            for (int x = 0; x < W; ++x) {
                const size_t i = (size_t) y * W + x;
                float v;
                if (ch == 'A') { const float *p = &m_storage[i * std::max<size_t>(m_channels.size(), 5)]; v = p[4] != 0.f ? p[3] / p[4] : 0.f; }
                else v = rgb[i * 3 + (ch == 'R' ? 0 : ch == 'G' ? 1 : 2)];
                unsigned char *dst = &line[((size_t) c * W + x) * bpc];
                if (half) { uint16_t h = float_to_half(v); std::memcpy(dst, &h, 2); } else std::memcpy(dst, &v, 4);
            }
        }
        std::fwrite(line.data(), 1, line_bytes, f);
    }
    std::fclose(f);
    return path;
}

// ============================================================================================
// Spiral
// ============================================================================================
Spiral::Spiral(std::array<int, 2> size, std::array<int, 2> offset, size_t block_size, size_t passes)
    : m_block_size(block_size), m_size(size), m_offset(offset), m_remaining_passes(passes) {
    m_blocks = { (int) std::ceil((float) m_size[0] / (float) m_block_size),
                 (int) std::ceil((float) m_size[1] / (float) m_block_size) };
    m_block_count = (size_t) m_blocks[0] * m_blocks[1];
    reset();
}
void Spiral::reset() {
    m_block_counter = 0;
    m_current_direction = Direction::Right;
    m_position = { m_blocks[0] / 2, m_blocks[1] / 2 };
    m_steps_left = 1;
    m_steps = 1;
}
Spiral::Block Spiral::next_block() {
    if (m_block_count == m_block_counter) {
        if (m_remaining_passes > 1) { --m_remaining_passes; reset(); }
        else return { { 0, 0 }, { 0, 0 }, (size_t) -1 };
    }
    size_t block_id = m_block_counter + (m_remaining_passes - 1) * m_block_count;
    std::array<int, 2> offset = { m_position[0] * (int) m_block_size, m_position[1] * (int) m_block_size };
    std::array<int, 2> size = { std::min((int) m_block_size, m_size[0] - offset[0]),
                                std::min((int) m_block_size, m_size[1] - offset[1]) };
    offset[0] += m_offset[0]; offset[1] += m_offset[1];
    ++m_block_counter;
    if (m_block_counter != m_block_count) {
        do {
            switch (m_current_direction) {
                case Direction::Right: ++m_position[0]; break;
                case Direction::Down:  ++m_position[1]; break;
                case Direction::Left:  --m_position[0]; break;
                case Direction::Up:    --m_position[1]; break;
            }
            if (--m_steps_left == 0) {
                m_current_direction = Direction(((int) m_current_direction + 1) % 4);
                if (m_current_direction == Direction::Left || m_current_direction == Direction::Right) ++m_steps;
                m_steps_left = m_steps;
            }
        } while (m_position[0] < 0 || m_position[1] < 0 || m_position[0] >= m_blocks[0] || m_position[1] >= m_blocks[1]);
    }
    return { offset, size, block_id };
}

// ============================================================================================
// Sampler
// ============================================================================================
IndependentSampler::IndependentSampler(const Properties &props) {
    m_sample_count = (size_t) props.int_("sample_count", 4);   // sampler.cpp:14-18
    m_base_seed = (uint64_t) props.int_("seed", 0);
    m_state = 0; m_inc = 0;
    seed(MIW_PCG32_DEFAULT_STATE);                             // independent.cpp:62-63
}
std::shared_ptr<IndependentSampler> IndependentSampler::clone() const {
    auto s = std::make_shared<IndependentSampler>();
    s->m_sample_count = m_sample_count; s->m_base_seed = m_base_seed;
    return s;
}
void IndependentSampler::seed(uint64_t seed_offset) {
    miw::PCG32 r; miw::pcg32_seed(r, m_base_seed + seed_offset, MIW_PCG32_DEFAULT_STREAM);
    m_state = r.state; m_inc = r.inc;
}
float IndependentSampler::next_1d() {
    miw::PCG32 r; r.state = m_state; r.inc = m_inc;
    float v = miw::pcg32_next_f32(r);
    m_state = r.state;
    return v;
}
std::array<float, 2> IndependentSampler::next_2d() { float a = next_1d(), b = next_1d(); return { a, b }; }

// ============================================================================================
// Sensor
// ============================================================================================
static float rad_to_deg(float v) { return v * (180.f / MIW_PI); }
static float deg_to_rad(float v) { return v * (MIW_PI / 180.f); }

float parse_fov(const Properties &props, float aspect) {
    if (props.has_property("fov") && props.has_property("focal_length"))
        Throw("Please specify either a focal length ('focal_length') or a field of view ('fov')!");
    float fov; std::string fov_axis;
    if (props.has_property("fov")) {
        fov = props.float_("fov");
        fov_axis = to_lower(props.string("fov_axis", "x"));
        if (fov_axis == "smaller") fov_axis = aspect > 1 ? "y" : "x";
        else if (fov_axis == "larger") fov_axis = aspect > 1 ? "x" : "y";
    } else {
        std::string f = props.string("focal_length", "50mm");
        if (f.size() >= 2 && f.substr(f.size() - 2) == "mm") f = f.substr(0, f.size() - 2);
        float value;
        try { value = std::stof(f); } catch (...) {
            Throw("Could not parse the focal length (must be of the form <x>mm, where <x> is a positive integer)!");
        }
        fov = 2.f * rad_to_deg(std::atan(std::sqrt(float(36 * 36 + 24 * 24)) / (2.f * value)));
        fov_axis = "diagonal";
    }
    float result;
    if (fov_axis == "x") result = fov;
    else if (fov_axis == "y") result = rad_to_deg(2.f * std::atan(std::tan(.5f * deg_to_rad(fov)) * aspect));
    else if (fov_axis == "diagonal") {
        float diagonal = 2.f * std::tan(.5f * deg_to_rad(fov));
        float width = diagonal / std::sqrt(1.f + 1.f / (aspect * aspect));
        result = rad_to_deg(2.f * std::atan(width * .5f));
    } else Throw("The 'fov_axis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!");
    if (result <= 0.f || result >= 180.f) Throw("The horizontal field of view must be in the range [0, 180]!");
    return result;
}

PerspectiveCamera::PerspectiveCamera(const Properties &props, std::shared_ptr<Film> film,
                                     std::shared_ptr<IndependentSampler> sampler)
    : m_film(std::move(film)), m_sampler(std::move(sampler)) {
    if (!m_film) m_film = std::make_shared<Film>();
    if (!m_sampler) m_sampler = std::make_shared<IndependentSampler>();
    m_near_clip = props.float_("near_clip", 1e-2f);            // sensor.cpp:94-96
    m_far_clip = props.float_("far_clip", 1e4f);
    if (m_near_clip <= 0.f) Throw("The 'near_clip' parameter must be greater than zero!");
    if (m_near_clip >= m_far_clip) Throw("The 'near_clip' parameter must be smaller than 'far_clip'.");
    m_to_world = props.transform("to_world", Transform4f());
    auto size = m_film->size();
    m_x_fov = parse_fov(props, size[0] / (float) size[1]);
    if (m_to_world.has_scale()) Throw("Scale factors in the camera-to-world transformation are not allowed!");
    update_camera_transforms();
    auto crop = m_film->crop_size();
    m_pp_offset = { props.float_("principal_point_offset_x", 0.f) * ((float) size[0] / (float) crop[0]),
                    props.float_("principal_point_offset_y", 0.f) * ((float) size[1] / (float) crop[1]) };
}
void PerspectiveCamera::update_camera_transforms() {
    // perspective_projection, include/mitsuba/render/sensor.h:196-231
    auto fs = m_film->size(); auto cs = m_film->crop_size(); auto co = m_film->crop_offset();
    float fx = (float) fs[0], fy = (float) fs[1];
    float rel_size_x = (float) cs[0] / fx, rel_size_y = (float) cs[1] / fy,
          rel_off_x = (float) co[0] / fx, rel_off_y = (float) co[1] / fy;
    float aspect = fx / fy;
    m_camera_to_sample =
        Transform4f::scale({ 1.f / rel_size_x, 1.f / rel_size_y, 1.f }) *
        Transform4f::translate({ -rel_off_x, -rel_off_y, 0.f }) *
        Transform4f::scale({ -0.5f, -0.5f * aspect, 1.f }) *
        Transform4f::translate({ -1.f, -1.f / aspect, 0.f }) *
        Transform4f::perspective(m_x_fov, m_near_clip, m_far_clip);
    m_sample_to_camera = m_camera_to_sample.inverse();
}
Ray3f PerspectiveCamera::sample_ray(const std::array<float, 2> &position_sample) const {
    miw::SensorRec s;
    std::memcpy(s.sample_to_camera, m_sample_to_camera.m, 64);
    std::memcpy(s.to_world, m_to_world.m, 64);
    s.near_clip = m_near_clip; s.far_clip = m_far_clip; s.pp_offset[0] = m_pp_offset[0]; s.pp_offset[1] = m_pp_offset[1];
    miw::Ray r = miw::sensor_sample_ray(s, miw::v2(position_sample[0], position_sample[1]));
    Ray3f out; out.o = { r.o.x, r.o.y, r.o.z }; out.d = { r.d.x, r.d.y, r.d.z }; out.mint = r.mint; out.maxt = r.maxt;
    return out;
}

// ============================================================================================
// BSDFs
// ============================================================================================
static const struct { const char *name; float value; } ior_data[] = {
    { "vacuum", 1.0f }, { "helium", 1.000036f }, { "hydrogen", 1.000132f }, { "air", 1.000277f },
    { "carbon dioxide", 1.00045f }, { "water", 1.3330f }, { "acetone", 1.36f }, { "ethanol", 1.361f },
    { "carbon tetrachloride", 1.461f }, { "glycerol", 1.4729f }, { "benzene", 1.501f },
    { "silicone oil", 1.52045f }, { "bromine", 1.661f }, { "water ice", 1.31f }, { "fused quartz", 1.458f },
    { "pyrex", 1.470f }, { "acrylic glass", 1.49f }, { "polypropylene", 1.49f }, { "bk7", 1.5046f },
    { "sodium chloride", 1.544f }, { "amber", 1.55f }, { "pet", 1.5750f }, { "diamond", 2.419f },
    { nullptr, 0.f }
};
float lookup_ior(const Properties &props, const std::string &name, const std::string &def) {
    auto by_name = [](const std::string &n) -> float {
        std::string l = to_lower(n);
        for (auto *e = ior_data; e->name; ++e) if (l == e->name) return e->value;
        Throw("Unable to find an IOR value for \"" + l + "\"!");
    };
    if (props.has_property(name)) {
        try { return props.float_(name); } catch (const std::runtime_error &) { return by_name(props.string(name)); }
    }
    return by_name(def);
}

static const miw::BsdfRec &as_rec(const mi_bsdf &b) { return *reinterpret_cast<const miw::BsdfRec *>(&b); }
uint32_t BSDF::flags() const {
    uint32_t f = miw::bsdf_flags(as_rec(m_rec));
    if (m_back) f |= miw::bsdf_flags(as_rec(m_back->record()));    // twosided.cpp:76-86
    return f;
}
// the plugin as the integrator sees it: a two-record table {front, back} for the twosided adapter
namespace { struct SideTable { miw::BsdfRec t[2]; std::vector<float> tables; };
SideTable side_table(const mi_bsdf &rec, const std::shared_ptr<BSDF> &back, const std::vector<float> &table) {
    SideTable s; s.t[0] = as_rec(rec); s.t[1] = back ? as_rec(back->record()) : as_rec(rec);
    s.t[0].back = 1; s.t[1].flags &= ~(uint32_t) MI_BSDF_FLAG_TWOSIDED;
    s.tables = table;                                          // front's table at offset 0, the back side's behind it
    if (s.t[0].type == miw::BSDF_TYPE_ROUGHPLASTIC) s.t[0].p[5] = 0.f;
    if (s.t[1].type == miw::BSDF_TYPE_ROUGHPLASTIC) {
        const std::vector<float> &bt = back ? back->table() : table;
        s.t[1].p[5] = (float) s.tables.size();
        s.tables.insert(s.tables.end(), bt.begin(), bt.end());
    }
    return s;
}
miw::TexCtx host_ctx(const SideTable &s) { return miw::TexCtx(miw::Wavelengths(), miw::v2(0.f, 0.f), nullptr, s.tables.empty() ? nullptr : s.tables.data()); } }
#if MIW_SPECTRAL
std::pair<BSDFSample3f, Color3f> BSDF::sample(const Vector3f &, float, const std::array<float, 2> &) const {
    Throw("BSDF::sample on the host is a scalar_rgb test helper");
}
Color3f BSDF::eval(const Vector3f &, const Vector3f &) const { Throw("BSDF::eval on the host is a scalar_rgb test helper"); }
#else
std::pair<BSDFSample3f, Color3f> BSDF::sample(const Vector3f &wi, float s1, const std::array<float, 2> &s2) const {
    miw::BSDFSample bs;
    const SideTable tab = side_table(m_rec, m_back, m_table);
    const miw::V3 wi_ = miw::v3(wi[0], wi[1], wi[2]);
    miw::V3 w = miw::bsdf_side_sample(miw::bsdf_side(tab.t, 0, wi_), wi_, s1, miw::v2(s2[0], s2[1]), bs, host_ctx(tab));
    BSDFSample3f o; o.wo = { bs.wo.x, bs.wo.y, bs.wo.z }; o.pdf = bs.pdf; o.eta = bs.eta; o.sampled_type = bs.sampled_type;
    return { o, Color3f{ w.x, w.y, w.z } };
}
Color3f BSDF::eval(const Vector3f &wi, const Vector3f &wo) const {
    const SideTable tab = side_table(m_rec, m_back, m_table);
    const miw::V3 wi_ = miw::v3(wi[0], wi[1], wi[2]);
    miw::V3 v = miw::bsdf_side_eval(miw::bsdf_side(tab.t, 0, wi_), wi_, miw::v3(wo[0], wo[1], wo[2]), host_ctx(tab));
    return { v.x, v.y, v.z };
}
#endif
float BSDF::pdf(const Vector3f &wi, const Vector3f &wo) const {
    const SideTable tab = side_table(m_rec, m_back, m_table);
    const miw::V3 wi_ = miw::v3(wi[0], wi[1], wi[2]);
    return miw::bsdf_side_pdf(miw::bsdf_side(tab.t, 0, wi_), wi_, miw::v3(wo[0], wo[1], wo[2]), host_ctx(tab));
}

void BSDF::bind_texture(int slot, const Properties &props, const std::string &name, float def, bool unbounded) {
    m_rec.tex[slot] = props.texture_record(name, def, false, unbounded);
    m_bitmaps[slot] = props.bitmap(name);
}

// ---- bitmap texture (src/textures/bitmap.cpp:85-200) ---------------------------------------------------------
BitmapTexture::BitmapTexture(const Properties &props) {
    m_to_uv = props.transform("to_uv", Transform4f());
    std::string filter_type = props.string("filter_type", "bilinear");
    if (filter_type == "nearest") m_filter = MI_BITMAP_NEAREST;
    else if (filter_type == "bilinear") m_filter = MI_BITMAP_BILINEAR;
    else Throw("Invalid filter type \"" + filter_type + "\", must be one of: \"nearest\", or \"bilinear\"!");
    std::string wrap_mode = props.string("wrap_mode", "repeat");
    if (wrap_mode == "repeat") m_wrap = MI_BITMAP_REPEAT;
    else if (wrap_mode == "mirror") m_wrap = MI_BITMAP_MIRROR;
    else if (wrap_mode == "clamp") m_wrap = MI_BITMAP_CLAMP;
    else Throw("Invalid wrap mode \"" + wrap_mode + "\", must be one of: \"repeat\", \"mirror\", or \"clamp\"!");
    m_raw = props.bool_("raw", false);
    if (props.has_property("filename")) {
        m_name = props.string("filename");
        read_pfm(m_name, m_width, m_height, m_channels, m_data);
        finish();
    }
}
void BitmapTexture::set_bitmap(uint32_t width, uint32_t height, uint32_t channels, const float *data) {
    if (channels != 1 && channels != 3) Throw("Unsupported channel count: " + std::to_string(channels) + " (expected 1 or 3)");
    if (!data || width == 0 || height == 0) Throw("BitmapTexture: empty image");
    m_width = width; m_height = height; m_channels = channels;
    m_data.assign(data, data + (size_t) width * height * channels);
    finish();
}
// bitmap.cpp:137-143 (images below 2 x 2 are up-sampled with a tent filter: here by replication, which is what that
// resampling yields for a 1-texel axis) and :150-197 (conversion to the variant's representation)
void BitmapTexture::finish() {
    if (m_width < 2 || m_height < 2) {
        const uint32_t w = std::max(m_width, 2u), h = std::max(m_height, 2u);
        std::vector<float> up((size_t) w * h * m_channels);
        for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) for (uint32_t c = 0; c < m_channels; ++c)
            up[((size_t) y * w + x) * m_channels + c] = m_data[((size_t) std::min(y, m_height - 1) * m_width + std::min(x, m_width - 1)) * m_channels + c];
        m_data.swap(up); m_width = w; m_height = h;
    }
    m_device_data.clear();
#if MIW_SPECTRAL
    if (m_channels == 3 && !m_raw) {                           // :156-165
        m_device_data.resize(m_data.size());
        for (size_t i = 0; i < m_data.size(); i += 3) {
            auto cf = srgb_model_fetch(Color3f{ m_data[i], m_data[i + 1], m_data[i + 2] });
            m_device_data[i] = cf[0]; m_device_data[i + 1] = cf[1]; m_device_data[i + 2] = cf[2];
        }
    }
#endif
}
Color3f BitmapTexture::mean() const {
    double sum[3] = { 0, 0, 0 };
    const size_t n = (size_t) m_width * m_height;
    for (size_t i = 0; i < n; ++i) for (uint32_t c = 0; c < 3; ++c) sum[c] += (double) m_data[i * m_channels + (m_channels == 3 ? c : 0)];
    return Color3f{ (float) (sum[0] / (double) n), (float) (sum[1] / (double) n), (float) (sum[2] / (double) n) };
}
mi_bitmap BitmapTexture::record() const {
    if (m_data.empty()) Throw("BitmapTexture: no image (give a \"filename\" or call set_bitmap)");
#if MIW_SPECTRAL
    if (m_channels == 3 && m_raw)                              // bitmap.cpp:269-273
        Throw("The bitmap texture " + m_name + " was queried for a spectrum, but texture conversion into spectra was explicitly disabled! (raw=true)");
#endif
    mi_bitmap b{};
    b.data = m_device_data.empty() ? m_data.data() : m_device_data.data();
    b.width = m_width; b.height = m_height; b.channels = m_channels; b.filter_type = m_filter; b.wrap_mode = m_wrap;
    // Transform4f::extract() to 3 x 3 (transform.h:324-348): upper-left 2 x 2 and the translation column
    b.to_uv[0] = m_to_uv.m[0]; b.to_uv[1] = m_to_uv.m[1]; b.to_uv[2] = m_to_uv.m[4]; b.to_uv[3] = m_to_uv.m[5];
    b.to_uv[4] = m_to_uv.m[12]; b.to_uv[5] = m_to_uv.m[13];
    return b;
}
void read_pfm(const std::string &path, uint32_t &width, uint32_t &height, uint32_t &channels, std::vector<float> &data) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) Throw("Bitmap: \"" + path + "\": file not found");
    char magic[3] = { 0, 0, 0 }; int w = 0, h = 0; float scale = 0.f;
    if (std::fscanf(f, "%2s %d %d %f", magic, &w, &h, &scale) != 4 || (std::strcmp(magic, "PF") && std::strcmp(magic, "Pf")) || w <= 0 || h <= 0 || scale == 0.f) {
        std::fclose(f); Throw("Bitmap: \"" + path + "\" is not a Portable Float Map (only PFM images are read by this layer)");
    }
    std::fgetc(f);                                             // the single whitespace byte after the header
    channels = magic[1] == 'F' ? 3u : 1u; width = (uint32_t) w; height = (uint32_t) h;
    data.resize((size_t) w * h * channels);
    const size_t row = (size_t) w * channels;
    for (int y = h - 1; y >= 0; --y)                           // bottom-to-top scanlines
        if (std::fread(data.data() + (size_t) y * row, sizeof(float), row, f) != row) { std::fclose(f); Throw("Bitmap: \"" + path + "\": truncated file"); }
    std::fclose(f);
    if (scale > 0.f)                                           // big endian
        for (float &v : data) { uint32_t u; std::memcpy(&u, &v, 4); u = __builtin_bswap32(u); std::memcpy(&v, &u, 4); }
    const float mul = std::fabs(scale);
    if (mul != 1.f) for (float &v : data) v *= mul;
}

static void check_reflectance(const Color3f &c, const char *what) {  // src/spectra/srgb.cpp:30-31
    for (float v : c) if (v < 0.f || v > 1.f) Throw(std::string(what) + ": values must be in the range [0, 1]!");
}
SmoothDiffuse::SmoothDiffuse(const Properties &props) {
    Color3f r = props.texture("reflectance", .5f);
    check_reflectance(r, "reflectance");
    m_rec.type = MI_BSDF_DIFFUSE; m_rec.flags = 0;
    m_rec.params[0] = r[0]; m_rec.params[1] = r[1]; m_rec.params[2] = r[2];
    bind_texture(0, props, "reflectance", .5f, false);
}
SmoothDielectric::SmoothDielectric(const Properties &props) {
    float int_ior = lookup_ior(props, "int_ior", "bk7"), ext_ior = lookup_ior(props, "ext_ior", "air");
    if (int_ior < 0 || ext_ior < 0) Throw("The interior and exterior indices of refraction must be positive!");
    Color3f sr = props.texture("specular_reflectance", 1.f), stt = props.texture("specular_transmittance", 1.f);
    check_reflectance(sr, "specular_reflectance"); check_reflectance(stt, "specular_transmittance");
    m_rec.type = MI_BSDF_DIELECTRIC; m_rec.flags = 0;
    m_rec.params[0] = int_ior / ext_ior;
    for (int i = 0; i < 3; ++i) { m_rec.params[1 + i] = sr[i]; m_rec.params[4 + i] = stt[i]; }
    bind_texture(0, props, "specular_reflectance", 1.f, false);
    bind_texture(1, props, "specular_transmittance", 1.f, false);
}
RoughConductor::RoughConductor(const Properties &props) {
    std::string material = props.string("material", "none");
    Color3f eta, k;
    if (props.has_property("eta") || material == "none") {
        eta = props.texture("eta", 0.f); k = props.texture("k", 1.f);
        if (material != "none") Throw("Should specify either (eta, k) or material, not both.");
    } else {
        Throw("complex_ior_from_file: the IOR data files are not available; specify 'eta' and 'k' explicitly.");
    }
    uint32_t flags = 0;
    if (props.has_property("distribution")) {
        std::string distr = to_lower(props.string("distribution"));
        if (distr == "beckmann") flags |= 0;
        else if (distr == "ggx") flags |= MI_BSDF_FLAG_GGX;
        else Throw("Specified an invalid distribution \"" + distr + "\", must be \"beckmann\" or \"ggx\"!");
    }
    if (props.bool_("sample_visible", true)) flags |= MI_BSDF_FLAG_SAMPLE_VISIBLE;
    float au, av;
    if (props.has_property("alpha_u") || props.has_property("alpha_v")) {
        if (!props.has_property("alpha_u") || !props.has_property("alpha_v"))
            Throw("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.");
        if (props.has_property("alpha")) Throw("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.");
        au = props.float_("alpha_u"); av = props.float_("alpha_v");
    } else {
        au = av = props.float_("alpha", 0.1f);
    }
    Color3f sr = props.texture("specular_reflectance", 1.f);
    check_reflectance(sr, "specular_reflectance");
    m_rec.type = MI_BSDF_ROUGHCONDUCTOR; m_rec.flags = flags;
    m_rec.params[0] = au; m_rec.params[1] = av;
    for (int i = 0; i < 3; ++i) { m_rec.params[2 + i] = eta[i]; m_rec.params[5 + i] = k[i]; m_rec.params[8 + i] = sr[i]; }
    bind_texture(0, props, "eta", 0.f, true);    // xml.cpp is_unbounded_spectrum: eta, k
    bind_texture(1, props, "k", 1.f, true);
    bind_texture(2, props, "specular_reflectance", 1.f, false);
}

SmoothConductor::SmoothConductor(const Properties &props) {
    std::string material = props.string("material", "none");
    Color3f eta, k;
    if (props.has_property("eta") || material == "none") {       // conductor.cpp:207-211
        eta = props.texture("eta", 0.f); k = props.texture("k", 1.f);
        if (material != "none") Throw("Should specify either (eta, k) or material, not both.");
    } else {
        Throw("complex_ior_from_file: the IOR data files are not available; specify 'eta' and 'k' explicitly.");
    }
    Color3f sr = props.texture("specular_reflectance", 1.f);
    check_reflectance(sr, "specular_reflectance");
    m_rec.type = MI_BSDF_CONDUCTOR; m_rec.flags = 0;
    for (int i = 0; i < 3; ++i) { m_rec.params[2 + i] = eta[i]; m_rec.params[5 + i] = k[i]; m_rec.params[8 + i] = sr[i]; }
    bind_texture(0, props, "eta", 0.f, true);
    bind_texture(1, props, "k", 1.f, true);
    bind_texture(2, props, "specular_reflectance", 1.f, false);
}
// fresnel.h:327-361
float fresnel_diffuse_reflectance(float eta) {
    if (eta < 1.f)
        return -1.4399f * (eta * eta) + 0.7099f * eta + 0.6681f + 0.0636f / eta;
    float inv_eta = 1.f / eta, inv_eta_2 = inv_eta * inv_eta, inv_eta_3 = inv_eta_2 * inv_eta,
          inv_eta_4 = inv_eta_3 * inv_eta, inv_eta_5 = inv_eta_4 * inv_eta;
    return 0.919317f - 3.4793f * inv_eta + 6.75335f * inv_eta_2 - 7.80989f * inv_eta_3 + 4.98554f * inv_eta_4 - 1.36881f * inv_eta_5;
}
SmoothPlastic::SmoothPlastic(const Properties &props) {
    float int_ior = lookup_ior(props, "int_ior", "polypropylene"), ext_ior = lookup_ior(props, "ext_ior", "air");
    if (int_ior < 0.f || ext_ior < 0.f) Throw("The interior and exterior indices of refraction must be positive!");
    const float eta = int_ior / ext_ior;
    Color3f dr = props.texture("diffuse_reflectance", .5f);
    check_reflectance(dr, "diffuse_reflectance");
    const bool has_spec = props.has_property("specular_reflectance");
    Color3f sr = props.texture("specular_reflectance", 1.f);
    if (has_spec) check_reflectance(sr, "specular_reflectance");
    // parameters_changed(), plastic.cpp:163-174
    const float d_mean = props.texture_mean("diffuse_reflectance", .5f),
                s_mean = has_spec ? props.texture_mean("specular_reflectance", 1.f) : 1.f;
    m_rec.type = MI_BSDF_PLASTIC;
    m_rec.flags = (props.bool_("nonlinear", false) ? MI_BSDF_FLAG_NONLINEAR : 0) | (has_spec ? MI_BSDF_FLAG_HAS_SPECULAR : 0);
    m_rec.params[0] = eta;
    m_rec.params[1] = 1.f / (eta * eta);
    m_rec.params[2] = fresnel_diffuse_reflectance(1.f / eta);
    m_rec.params[3] = s_mean / (d_mean + s_mean);
    for (int i = 0; i < 3; ++i) { m_rec.params[4 + i] = dr[i]; m_rec.params[7 + i] = sr[i]; }
    bind_texture(0, props, "diffuse_reflectance", .5f, false);
    bind_texture(1, props, "specular_reflectance", 1.f, false);
}
RoughDielectric::RoughDielectric(const Properties &props) {
    float int_ior = lookup_ior(props, "int_ior", "bk7"), ext_ior = lookup_ior(props, "ext_ior", "air");
    if (int_ior < 0.f || ext_ior < 0.f || int_ior == ext_ior)
        Throw("The interior and exterior indices of refraction must be positive and differ!");   // :155-157
    const float eta = int_ior / ext_ior;
    uint32_t flags = 0;
    if (props.has_property("distribution")) {                  // :162-173 (default: beckmann)
        std::string distr = to_lower(props.string("distribution"));
        if (distr == "ggx") flags |= MI_BSDF_FLAG_GGX;
        else if (distr != "beckmann") Throw("Specified an invalid distribution \"" + distr + "\", must be \"beckmann\" or \"ggx\"!");
    }
    if (props.bool_("sample_visible", true)) flags |= MI_BSDF_FLAG_SAMPLE_VISIBLE;
    float au, av;
    if (props.has_property("alpha_u") || props.has_property("alpha_v")) {
        if (!props.has_property("alpha_u") || !props.has_property("alpha_v"))
            Throw("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.");
        if (props.has_property("alpha")) Throw("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.");
        au = props.float_("alpha_u"); av = props.float_("alpha_v");
    } else {
        au = av = props.float_("alpha", 0.1f);
    }
    const bool has_r = props.has_property("specular_reflectance"), has_t = props.has_property("specular_transmittance");
    Color3f sr = props.texture("specular_reflectance", 1.f), stt = props.texture("specular_transmittance", 1.f);
    if (has_r) { check_reflectance(sr, "specular_reflectance"); flags |= MI_BSDF_FLAG_HAS_SPEC_REFLECTANCE; }
    if (has_t) { check_reflectance(stt, "specular_transmittance"); flags |= MI_BSDF_FLAG_HAS_SPEC_TRANSMITTANCE; }
    m_rec.type = MI_BSDF_ROUGHDIELECTRIC; m_rec.flags = flags;
    m_rec.params[0] = au; m_rec.params[1] = av; m_rec.params[2] = eta; m_rec.params[3] = 1.f / eta;   // parameters_changed(), :199-201
    for (int i = 0; i < 3; ++i) { m_rec.params[4 + i] = sr[i]; m_rec.params[7 + i] = stt[i]; }
    bind_texture(0, props, "specular_reflectance", 1.f, false);
    bind_texture(1, props, "specular_transmittance", 1.f, false);
}
// n-point Gauss-Legendre rule: Newton's method on P_n from Chebyshev starting points, in double (quad.cpp:7-64)
void gauss_legendre(int n, std::vector<float> &nodes, std::vector<float> &weights) {
    if (n < 1) Throw("gauss_legendre(): n must be >= 1");
    nodes.assign((size_t) n, 0.f); weights.assign((size_t) n, 0.f);
    auto legendre = [n](double x, double &p, double &dp) {     // P_n(x), P_n'(x) by the three-term recurrence
        double p0 = 1.0, p1 = x;
        if (n == 0) { p = 1.0; dp = 0.0; return; }
        for (int k = 2; k <= n; ++k) { double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k; p0 = p1; p1 = pk; }
        p = p1; dp = n * (x * p1 - p0) / (x * x - 1.0);
    };
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double x = -std::cos((2 * i + 1) / (double) (2 * n) * 3.14159265358979323846), p, dp;
        if (n % 2 == 1 && i == n / 2) x = 0.0;
        for (int it = 0; it < 30 && x != 0.0; ++it) {
            legendre(x, p, dp);
            const double step = p / dp; x -= step;
            if (std::fabs(step) <= 4 * std::fabs(x) * std::numeric_limits<double>::epsilon()) break;
        }
        if (x == 0.0) { double p0 = 1.0, p1 = 0.0; for (int k = 2; k <= n; ++k) { double pk = -((k - 1) * p0) / k; p0 = p1; p1 = pk; } dp = n * p0; }   // P_n'(0) = n P_{n-1}(0)
        else legendre(x, p, dp);
        const double w = 2.0 / ((1.0 - x * x) * dp * dp);
        nodes[i] = (float) x; nodes[n - 1 - i] = (float) -x; weights[i] = weights[n - 1 - i] = (float) w;
    }
}
// eval_transmittance / eval_reflectance (microfacet.h:454-552) for one incident direction: the visible-normal
// sampling routine of the distribution pushed through an n x n tensor Gauss-Legendre rule over the unit square
static float rough_interface_integral(const miw::Microfacet &distr, miw::V3 wi, float eta, bool transmit) {
    std::vector<float> nodes, weights;
    gauss_legendre(eta > 1.f ? 32 : 128, nodes, weights);      // :468-472 (the packet padding adds nothing at these sizes)
    double accum = 0.0;
    for (size_t a = 0; a < nodes.size(); ++a)
        for (size_t b = 0; b < nodes.size(); ++b) {
            const miw::V2 node = miw::v2(miw::fmadd(nodes[b], .5f, .5f), miw::fmadd(nodes[a], .5f, .5f));
            miw::V3 m; float pdf;
            miw::mf_sample(distr, wi, node, m, pdf);
            float f, cos_theta_t, eta_it, eta_ti;
            miw::fresnel(miw::dot(wi, m), eta, f, cos_theta_t, eta_it, eta_ti);
            float smith;
            if (transmit) {
                const miw::V3 wo = miw::refract(wi, m, cos_theta_t, eta_ti);
                smith = miw::mf_smith_g1(distr, wo, m) * (1.f - f);
                if (wo.z * wi.z >= 0.f) smith = 0.f;
            } else {
                const miw::V3 wo = miw::reflect(wi, m);
                smith = miw::mf_smith_g1(distr, wo, m) * f;
                if (wo.z <= 0.f || wi.z <= 0.f) smith = 0.f;
            }
            accum += (double) (smith * (weights[a] * weights[b]));
        }
    return (float) accum * .25f;
}
RoughPlastic::RoughPlastic(const Properties &props) {
    float int_ior = lookup_ior(props, "int_ior", "polypropylene"), ext_ior = lookup_ior(props, "ext_ior", "air");
    if (int_ior < 0.f || ext_ior < 0.f || int_ior == ext_ior)
        Throw("The interior and exterior indices of refraction must be positive and differ!");   // :155-157
    const float eta = int_ior / ext_ior;
    uint32_t flags = 0;
    if (props.has_property("distribution")) {
        std::string distr = to_lower(props.string("distribution"));
        if (distr == "ggx") flags |= MI_BSDF_FLAG_GGX;
        else if (distr != "beckmann") Throw("Specified an invalid distribution \"" + distr + "\", must be \"beckmann\" or \"ggx\"!");
    }
    if (props.bool_("sample_visible", true)) flags |= MI_BSDF_FLAG_SAMPLE_VISIBLE;
    if (props.has_property("alpha_u") || props.has_property("alpha_v"))
        Throw("The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!");   // :170-172
    const float alpha = props.float_("alpha", 0.1f);
    const bool has_spec = props.has_property("specular_reflectance");
    Color3f dr = props.texture("diffuse_reflectance", .5f), sr = props.texture("specular_reflectance", 1.f);
    check_reflectance(dr, "diffuse_reflectance");
    if (has_spec) { check_reflectance(sr, "specular_reflectance"); flags |= MI_BSDF_FLAG_HAS_SPEC_REFLECTANCE; }
    if (props.bool_("nonlinear", false)) flags |= MI_BSDF_FLAG_RP_NONLINEAR;
    // parameters_changed(), :336-371
    const float d_mean = props.texture_mean("diffuse_reflectance", .5f),
                s_mean = has_spec ? props.texture_mean("specular_reflectance", 1.f) : 1.f;
    const miw::Microfacet distr = miw::microfacet_make((flags & MI_BSDF_FLAG_GGX) ? miw::MF_GGX : miw::MF_BECKMANN, alpha, alpha, true);
    m_table.resize(MI_ROUGH_TRANSMITTANCE_RES);
    double refl = 0.0;
    for (int i = 0; i < MI_ROUGH_TRANSMITTANCE_RES; ++i) {
        const float mu = std::max(1e-6f, (float) i / (float) (MI_ROUGH_TRANSMITTANCE_RES - 1));
        const miw::V3 wi = miw::v3(std::sqrt(1.f - mu * mu), 0.f, mu);
        m_table[(size_t) i] = rough_interface_integral(distr, wi, eta, true);
        refl += (double) (rough_interface_integral(distr, wi, 1.f / eta, false) * wi.z);
    }
    m_rec.type = MI_BSDF_ROUGHPLASTIC; m_rec.flags = flags;
    m_rec.params[0] = alpha; m_rec.params[1] = eta; m_rec.params[2] = 1.f / (eta * eta);
    m_rec.params[3] = (float) (refl / MI_ROUGH_TRANSMITTANCE_RES) * 2.f;        // hmean(...) * 2, :368-369
    m_rec.params[4] = s_mean / (d_mean + s_mean);
    m_rec.params[5] = 0.f;                                     // table offset: assigned by Scene::build
    for (int i = 0; i < 3; ++i) { m_rec.params[6 + i] = dr[i]; m_rec.params[9 + i] = sr[i]; }
    bind_texture(0, props, "diffuse_reflectance", .5f, false);
    bind_texture(1, props, "specular_reflectance", 1.f, false);
}
TwoSidedBRDF::TwoSidedBRDF(std::shared_ptr<BSDF> front, std::shared_ptr<BSDF> back) {
    if (!front) Throw("A nested one-sided material is required!");
    if (front->twosided() || (back && back->twosided())) Throw("twosided: nested twosided materials are not supported");
    if (!back) back = front;
    if ((front->flags() | back->flags()) & miw::BSDF_Transmission)
        Throw("Only materials without a transmission component can be nested!");
    m_rec = front->record();
    m_rec.flags |= MI_BSDF_FLAG_TWOSIDED;
    for (int k = 0; k < 3; ++k) m_bitmaps[k] = front->bitmap(k);
    m_table = front->table();
    m_back = back;
}

AreaLight::AreaLight(const Properties &props) {
    m_radiance = props.texture("radiance", 1.f);               // area.cpp:55 (D65(1) ~ white in RGB mode)
    m_radiance_tex = props.texture_record("radiance", 1.f, true, false);
}

EnvironmentMapEmitter::EnvironmentMapEmitter(const Properties &props) {
    m_scale = props.float_("scale", 1.f);                      // envmap.cpp:124
    m_to_world = props.transform("to_world", Transform4f());
    if (props.has_property("filename")) {                      // envmap.cpp:66-75: Bitmap(file).convert(RGBA, Float32); PFM files only here
        uint32_t w, h, c; std::vector<float> px;
        read_pfm(props.string("filename"), w, h, c, px);
        std::vector<float> rgba((size_t) w * h * 4);
        for (size_t i = 0; i < (size_t) w * h; ++i) {
            for (uint32_t k = 0; k < 3; ++k) rgba[4 * i + k] = px[i * c + (c == 3 ? k : 0)];
            rgba[4 * i + 3] = 1.f;
        }
        set_bitmap(w, h, rgba.data());
    }
}
void EnvironmentMapEmitter::set_bitmap(uint32_t width, uint32_t height, const float *rgba) {
    if (width < 2 || height < 2 || !rgba) Throw("envmap: the bitmap must be at least 2x2");
    m_width = width; m_height = height;
    m_data.assign(rgba, rgba + (size_t) width * height * 4);
}

// ============================================================================================
// Mesh / Scene
// ============================================================================================
Mesh::Mesh(std::string name, std::vector<float> p, std::vector<uint32_t> f, std::vector<float> n, std::vector<float> tc)
    : m_name(std::move(name)), m_positions(std::move(p)), m_normals(std::move(n)), m_texcoords(std::move(tc)), m_faces(std::move(f)) {
    if (!m_texcoords.empty() && m_texcoords.size() / 2 != m_positions.size() / 3) Throw("Mesh: vertex texture coordinate count mismatch");
    if (m_positions.size() % 3 || m_faces.size() % 3) Throw("Mesh: buffer sizes must be multiples of 3");
    if (!m_normals.empty() && m_normals.size() != m_positions.size()) Throw("Mesh: vertex normal count mismatch");
    for (uint32_t i : m_faces) if (i >= vertex_count()) Throw("Mesh: face references a vertex out of range");
}

// enoki unit_angle(a, b) for unit vectors: 2 asin(|b -+ a| / 2), robust near 0 and pi
static float unit_angle(miw::V3 a, miw::V3 b) {
    float dot_uv = miw::dot(a, b);
    miw::V3 t = dot_uv >= 0.f ? b - a : b + a;
    float temp = 2.f * miw::asin_(.5f * miw::norm(t));
    return dot_uv >= 0.f ? temp : MIW_PI - temp;
}
void Mesh::recompute_vertex_normals() {
    const uint32_t nv = vertex_count(), nf = face_count();
    std::vector<miw::V3> acc(nv, miw::v3(0.f));
    auto P = [&](uint32_t i) { return miw::v3(m_positions[3 * i], m_positions[3 * i + 1], m_positions[3 * i + 2]); };
    for (uint32_t f = 0; f < nf; ++f) {
        const uint32_t fi[3] = { m_faces[3 * f], m_faces[3 * f + 1], m_faces[3 * f + 2] };
        miw::V3 v[3] = { P(fi[0]), P(fi[1]), P(fi[2]) };
        miw::V3 side_0 = v[1] - v[0], side_1 = v[2] - v[0];
        miw::V3 n = miw::cross(side_0, side_1);
        float length_sqr = miw::squared_norm(n);
        if (length_sqr > 0.f) {
            n = n * miw::rsqrt(length_sqr);
            const miw::V3 s1[3] = { side_0, v[2] - v[1], v[0] - v[2] }, s2[3] = { side_1, v[0] - v[1], v[1] - v[2] };
            for (int j = 0; j < 3; ++j)
                acc[fi[j]] = acc[fi[j]] + n * unit_angle(miw::normalize(s1[j]), miw::normalize(s2[j]));
        }
    }
    m_normals.assign((size_t) nv * 3, 0.f);
    for (uint32_t i = 0; i < nv; ++i) {
        miw::V3 n = acc[i];
        float length = miw::norm(n);
        if (length != 0.f) n = n / length; else n = miw::v3(1.f, 0.f, 0.f);    // "some bogus value", mesh.cpp:243
        m_normals[3 * i] = n.x; m_normals[3 * i + 1] = n.y; m_normals[3 * i + 2] = n.z;
    }
}

// ---- obj / ply ---------------------------------------------------------------------------------
namespace {
std::string read_file(const std::string &path, const char *what) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) Throw(std::string("Error while loading ") + what + " file \"" + path + "\": file not found");
    std::string data;
    char buf[1 << 16]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, n);
    std::fclose(f);
    return data;
}
std::string base_name(const std::string &path) { size_t p = path.find_last_of("/\\"); return p == std::string::npos ? path : path.substr(p + 1); }
miw::V3 xf_normal(const Transform4f &t, miw::V3 n) {            // Transform::transform_affine(Normal): inverse transpose
    const float *m = t.inv;
    return miw::v3(m[0] * n.x + m[1] * n.y + m[2] * n.z, m[4] * n.x + m[5] * n.y + m[6] * n.z, m[8] * n.x + m[9] * n.y + m[10] * n.z);
}
struct Key3 { uint32_t k[3]; bool operator<(const Key3 &o) const { return std::lexicographical_compare(k, k + 3, o.k, o.k + 3); } };
}

std::shared_ptr<Mesh> load_obj(const Properties &props) {
    const bool flip_tex_coords = props.bool_("flip_tex_coords", true), face_normals = props.bool_("face_normals", false);
    const Transform4f to_world = props.transform("to_world", Transform4f());
    const std::string path = props.string("filename"), name = base_name(path);
    const std::string data = read_file(path, "OBJ");
    auto fail = [&](const std::string &d) { Throw("Error while loading OBJ file \"" + name + "\": " + d); };
    std::vector<miw::V3> vertices, normals; std::vector<std::array<float, 2>> texcoords;
    std::vector<uint32_t> faces; std::vector<Key3> keys;        // keys[id] = (v, vt, vn) of output vertex id
    std::map<Key3, uint32_t> vertex_map;
    size_t pos = 0;
    while (pos < data.size()) {
        size_t eol = data.find('\n', pos);
        if (eol == std::string::npos) eol = data.size();
        if (eol - pos >= 1024) fail("file contains an excessively long line!");
        std::string line = data.substr(pos, eol - pos);
        pos = eol + 1;
        const char *cur = line.c_str();
        while (*cur == ' ' || *cur == '\t' || *cur == '\r') ++cur;
        bool parse_error = false;
        auto read_floats = [&](int n, float *out) { for (int i = 0; i < n; ++i) { char *end; out[i] = std::strtof(cur, &end); parse_error |= end == cur; cur = end; } };
        if (cur[0] == 'v' && (cur[1] == ' ' || cur[1] == '\t')) {
            float p[3]; cur += 2; read_floats(3, p);
            miw::V3 w = miw::xf_point_affine(to_world.m, miw::v3(p[0], p[1], p[2]));
            if (!std::isfinite(w.x) || !std::isfinite(w.y) || !std::isfinite(w.z)) fail("mesh contains invalid vertex position data");
            vertices.push_back(w);
        } else if (cur[0] == 'v' && cur[1] == 'n' && (cur[2] == ' ' || cur[2] == '\t')) {
            float p[3]; cur += 3; read_floats(3, p);
            miw::V3 n = miw::normalize(xf_normal(to_world, miw::v3(p[0], p[1], p[2])));
            if (!std::isfinite(n.x) || !std::isfinite(n.y) || !std::isfinite(n.z)) fail("mesh contains invalid vertex normal data");
            normals.push_back(n);
        } else if (cur[0] == 'v' && cur[1] == 't' && (cur[2] == ' ' || cur[2] == '\t')) {
            float p[2]; cur += 3; read_floats(2, p);
            if (flip_tex_coords) p[1] = 1.f - p[1];
            texcoords.push_back({ p[0], p[1] });
        } else if (cur[0] == 'f' && (cur[1] == ' ' || cur[1] == '\t')) {
            cur += 2;
            size_t vertex_index = 0, type_index = 0;
            Key3 key{ { 0, 0, 0 } }; uint32_t tri[3] = { 0, 0, 0 };
            while (true) {
                char *next2;
                uint32_t value = (uint32_t) std::strtoul(cur, &next2, 10);
                if (cur == next2) break;
                if (type_index < 3) key.k[type_index] = value; else { parse_error = true; break; }
                while (*next2 == '/') { type_index++; next2++; }
                if (*next2 == ' ' || *next2 == '\t' || *next2 == '\0' || *next2 == '\r') {
                    type_index = 0;
                    if ((size_t) key.k[0] - 1 >= vertices.size()) fail("reference to invalid vertex " + std::to_string(key.k[0]) + "!");
                    auto it = vertex_map.find(key);
                    uint32_t id;
                    if (it != vertex_map.end()) id = it->second;
                    else { id = (uint32_t) keys.size(); vertex_map.emplace(key, id); keys.push_back(key); }
                    if (vertex_index < 3) tri[vertex_index] = id; else { tri[1] = tri[2]; tri[2] = id; }   // polygon fan
                    vertex_index++;
                    if (vertex_index >= 3) faces.insert(faces.end(), tri, tri + 3);
                    key = Key3{ { 0, 0, 0 } };
                }
                cur = next2;
            }
        }
        if (parse_error) fail("could not parse line \"" + line + "\"");
    }
    const size_t nv = keys.size();
    std::vector<float> P(nv * 3), N, T;
    if (!texcoords.empty()) T.assign(nv * 2, 0.f);             // obj.cpp:285-286
    const bool keep_normals = !face_normals;
    if (keep_normals && !normals.empty()) N.assign(nv * 3, 0.f);
    for (size_t id = 0; id < nv; ++id) {
        const Key3 &k = keys[id];
        const miw::V3 &v = vertices[k.k[0] - 1];
        P[3 * id] = v.x; P[3 * id + 1] = v.y; P[3 * id + 2] = v.z;
        if (k.k[1] && (size_t) k.k[1] - 1 >= texcoords.size()) fail("reference to invalid texture coordinate " + std::to_string(k.k[1]) + "!");
        if (k.k[1]) { T[2 * id] = texcoords[k.k[1] - 1][0]; T[2 * id + 1] = texcoords[k.k[1] - 1][1]; }   // obj.cpp:307-312
        if (keep_normals && k.k[2]) {
            if ((size_t) k.k[2] - 1 >= normals.size()) fail("reference to invalid normal " + std::to_string(k.k[2]) + "!");
            const miw::V3 &n = normals[k.k[2] - 1];
            N[3 * id] = n.x; N[3 * id + 1] = n.y; N[3 * id + 2] = n.z;
        }
    }
    auto mesh = std::make_shared<Mesh>(name, std::move(P), std::move(faces), std::move(N), std::move(T));
    if (keep_normals && normals.empty()) mesh->recompute_vertex_normals();      // obj.cpp:339-341
    return mesh;
}

std::shared_ptr<Mesh> load_ply(const Properties &props) {
    const bool face_normals = props.bool_("face_normals", false);
    const Transform4f to_world = props.transform("to_world", Transform4f());
    const std::string path = props.string("filename"), name = base_name(path);
    const std::string data = read_file(path, "PLY");
    auto fail = [&](const std::string &d) { Throw("Error while loading PLY file \"" + name + "\": " + d); };
    struct Prop { std::string name, type, count_type; bool list = false; };
    struct Elem { std::string name; size_t count = 0; std::vector<Prop> props; };
    std::vector<Elem> elems; std::string format;
    size_t pos = 0; bool header_done = false, tag = false;
    auto next_line = [&]() { size_t e = data.find('\n', pos); if (e == std::string::npos) fail("invalid PLY header"); std::string l = data.substr(pos, e - pos); pos = e + 1; if (!l.empty() && l.back() == '\r') l.pop_back(); return l; };
    auto split = [](const std::string &l) { std::vector<std::string> t; size_t i = 0; while (i < l.size()) { while (i < l.size() && (l[i] == ' ' || l[i] == '\t')) ++i; size_t j = i; while (j < l.size() && l[j] != ' ' && l[j] != '\t') ++j; if (j > i) t.push_back(l.substr(i, j - i)); i = j; } return t; };
    while (!header_done) {
        auto t = split(next_line());
        if (t.empty()) continue;
        if (t[0] == "ply") tag = true;
        else if (t[0] == "format" && t.size() >= 3) { format = t[1]; if (t[2] != "1.0") fail("PLY file has unknown version"); }
        else if (t[0] == "comment" || t[0] == "obj_info") {}
        else if (t[0] == "element" && t.size() == 3) { Elem e; e.name = t[1]; e.count = (size_t) std::strtoull(t[2].c_str(), nullptr, 10); elems.push_back(e); }
        else if (t[0] == "property" && !elems.empty()) {
            Prop p;
            if (t.size() == 5 && t[1] == "list") { p.list = true; p.count_type = t[2]; p.type = t[3]; p.name = t[4]; }
            else if (t.size() == 3) { p.type = t[1]; p.name = t[2]; }
            else fail("invalid PLY header: could not parse a property line");
            elems.back().props.push_back(p);
        } else if (t[0] == "end_header") header_done = true;
        else fail("invalid PLY header: unknown token \"" + t[0] + "\"");
    }
    if (!tag) fail("invalid PLY header: missing \"ply\" tag");
    const bool ascii = format == "ascii", le = format == "binary_little_endian", be = format == "binary_big_endian";
    if (!ascii && !le && !be) fail("invalid PLY header: unknown format");
    auto type_size = [&](const std::string &t) -> int {
        if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
        if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
        if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
        if (t == "double" || t == "float64") return 8;
        fail("invalid PLY header: unknown format type \"" + t + "\""); return 0; };
    auto read_num = [&](const std::string &t) -> double {          // one scalar of PLY type t at `pos`
        if (ascii) {
            while (pos < data.size() && std::isspace((unsigned char) data[pos])) ++pos;
            char *end; double v = std::strtod(data.c_str() + pos, &end);
            if (end == data.c_str() + pos) fail("could not parse the body");
            pos = (size_t) (end - data.c_str());
            return v;
        }
        const int sz = type_size(t);
        if (pos + sz > data.size()) fail("file is truncated");
        unsigned char b[8];
        for (int i = 0; i < sz; ++i) b[i] = (unsigned char) data[pos + (be ? sz - 1 - i : i)];
        pos += sz;
        if (t == "char" || t == "int8") return (double) (int8_t) b[0];
        if (t == "uchar" || t == "uint8") return (double) b[0];
        if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, b, 2); return v; }
        if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, b, 2); return v; }
        if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, b, 4); return v; }
        if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, b, 4); return v; }
        if (t == "float" || t == "float32") { float v; std::memcpy(&v, b, 4); return v; }
        double v; std::memcpy(&v, b, 8); return v;
    };
    std::vector<float> P, N, T; std::vector<uint32_t> F; bool has_normals = false;
    for (const Elem &el : elems) {
        if (el.name == "vertex") {
            int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
            for (size_t i = 0; i < el.props.size(); ++i) {
                const std::string &n = el.props[i].name;
                if (n == "x") ix = (int) i; else if (n == "y") iy = (int) i; else if (n == "z") iz = (int) i;
                else if (n == "nx") inx = (int) i; else if (n == "ny") iny = (int) i; else if (n == "nz") inz = (int) i;
                else if (n == "u" || n == "texture_u" || n == "s") iu = (int) i;      // ply.cpp:159-169
                else if (n == "v" || n == "texture_v" || n == "t") iv = (int) i;
                if (el.props[i].list) fail("vertex element with a list property");
            }
            if (ix < 0 || iy < 0 || iz < 0) fail("vertex coordinates missing");
            has_normals = inx >= 0 && iny >= 0 && inz >= 0 && !face_normals;
            P.resize(el.count * 3); if (has_normals) N.resize(el.count * 3);
            if (iu >= 0 && iv >= 0) T.resize(el.count * 2);
            std::vector<double> row(el.props.size());
            for (size_t v = 0; v < el.count; ++v) {
                for (size_t i = 0; i < el.props.size(); ++i) row[i] = read_num(el.props[i].type);
                miw::V3 p = miw::xf_point_affine(to_world.m, miw::v3((float) row[ix], (float) row[iy], (float) row[iz]));
                P[3 * v] = p.x; P[3 * v + 1] = p.y; P[3 * v + 2] = p.z;
                if (has_normals) {
                    miw::V3 n = miw::normalize(xf_normal(to_world, miw::v3((float) row[inx], (float) row[iny], (float) row[inz])));
                    N[3 * v] = n.x; N[3 * v + 1] = n.y; N[3 * v + 2] = n.z;
                }
                if (!T.empty()) { T[2 * v] = (float) row[iu]; T[2 * v + 1] = (float) row[iv]; }   // ply.cpp:251-257
            }
        } else if (el.name == "face") {
            F.reserve(el.count * 3);
            for (size_t f = 0; f < el.count; ++f)
                for (const Prop &p : el.props) {
                    if (p.list) {
                        const int cnt = (int) read_num(p.count_type);
                        const bool indices = p.name == "vertex_index" || p.name == "vertex_indices";
                        if (indices && cnt != 3) fail("incompatible contents -- is this a triangle mesh?");   // ply.cpp:339
                        for (int k = 0; k < cnt; ++k) { double v = read_num(p.type); if (indices) F.push_back((uint32_t) v); }
                    } else (void) read_num(p.type);
                }
        } else {                                                  // unknown element: skipped (ply.cpp:364-366)
            for (size_t k = 0; k < el.count; ++k)
                for (const Prop &p : el.props) {
                    if (p.list) { const int cnt = (int) read_num(p.count_type); for (int q = 0; q < cnt; ++q) (void) read_num(p.type); }
                    else (void) read_num(p.type);
                }
        }
    }
    if (ascii) while (pos < data.size() && std::isspace((unsigned char) data[pos])) ++pos;
    if (pos != data.size()) fail("invalid file -- trailing content");
    auto mesh = std::make_shared<Mesh>(name, std::move(P), std::move(F), std::move(N), std::move(T));
    if (!face_normals && !has_normals) mesh->recompute_vertex_normals();        // ply.cpp:378-383
    return mesh;
}

bool PreliminaryIntersection3f::is_valid() const { return t != std::numeric_limits<float>::infinity(); }

Scene::Scene() {}
Scene::~Scene() { if (m_ctx) mi_destroy(m_ctx); }
void Scene::add_shape(std::shared_ptr<Mesh> mesh) {
    if (m_built) Throw("Scene: cannot add shapes after build()");
    m_shapes.push_back(std::move(mesh));
}
void Scene::add_emitter(std::shared_ptr<EnvironmentMapEmitter> env) {
    if (m_built) Throw("Scene: cannot add emitters after build()");
    if (m_env) Throw("Only one environment emitter can be specified per scene.");   // scene.cpp:48-49
    m_env = std::move(env); m_env_after_shapes = m_shapes.size();
}
// front and back of a twosided BSDF are the same material (twosided.cpp:72-73)
static bool miw_same_record(const mi_bsdf &back, const mi_bsdf &front_twosided) {
    mi_bsdf f = front_twosided; f.flags &= ~(uint32_t) MI_BSDF_FLAG_TWOSIDED;
    return std::memcmp(&back, &f, sizeof f) == 0;
}
std::shared_ptr<Mesh> make_rectangle(const Properties &props) {
    Transform4f tw = props.transform("to_world", Transform4f());
    if (props.bool_("flip_normals", false)) tw = tw * Transform4f::scale({ 1.f, 1.f, -1.f });   // rectangle.cpp:78-80
    const float c[4][2] = { { -1, -1 }, { 1, -1 }, { 1, 1 }, { -1, 1 } };                       // bbox(), :98-105
    std::vector<float> P;
    for (auto &q : c) { miw::V3 w = miw::xf_point_affine(tw.m, miw::v3(q[0], q[1], 0.f)); P.insert(P.end(), { w.x, w.y, w.z }); }
    auto mesh = std::make_shared<Mesh>("rectangle", std::move(P), std::vector<uint32_t>{ 0, 1, 2 });
    mesh->m_rectangle = true; mesh->m_rect_to_world = tw;
    return mesh;
}
std::shared_ptr<Mesh> make_sphere(const Properties &props) {
    Transform4f tw = props.transform("to_world", Transform4f());
    Color3f c = props.has_property("center") ? props.texture("center") : Color3f{ 0.f, 0.f, 0.f };   // a 3-vector property
    tw = tw * Transform4f::translate({ c[0], c[1], c[2] });                                          // sphere.cpp:101-102
    float rs = props.float_("radius", 1.f);
    tw = tw * Transform4f::scale({ rs, rs, rs });
    // update(), :108-131. transform_decompose belongs to enoki (not vendored); for the transforms the plugin accepts
    // (no shear, uniform scale) S = radius * I, Q = M / radius, T = the translation column.
    const float *m = tw.m;
    auto col = [&](int k) { return miw::v3(m[4 * k], m[4 * k + 1], m[4 * k + 2]); };
    const float radius = miw::norm(col(0));
    for (int k = 1; k < 3; ++k)
        if (std::fabs(miw::norm(col(k)) - radius) > 1e-4f * radius) Throw("'to_world' transform shouldn't contain non-uniform scaling!");
    if (std::fabs(miw::dot(col(0), col(1))) > 1e-4f * radius * radius || std::fabs(miw::dot(col(0), col(2))) > 1e-4f * radius * radius ||
        std::fabs(miw::dot(col(1), col(2))) > 1e-4f * radius * radius) Throw("'to_world' transform shouldn't contain any shearing!");
    if (!(radius > 0.f)) Throw("sphere: the radius must be positive");
    mi_sphere rec{};
    rec.center[0] = m[12]; rec.center[1] = m[13]; rec.center[2] = m[14];
    rec.radius = radius; rec.flip_normals = props.bool_("flip_normals", false) ? 1u : 0u;
    // transform_compose(radius, Q, T) and its inverse ((1 / radius) Q^T, -(1 / radius) Q^T T)
    const float inv_r = 1.f / radius;
    float q[3][3];
    for (int k = 0; k < 3; ++k) { miw::V3 v = col(k); q[0][k] = v.x * inv_r; q[1][k] = v.y * inv_r; q[2][k] = v.z * inv_r; }
    std::memset(rec.to_world, 0, 64); std::memset(rec.to_object, 0, 64);
    for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 3; ++r) {
        rec.to_world[cc * 4 + r] = q[r][cc] * radius;
        rec.to_object[cc * 4 + r] = q[cc][r] * inv_r;
    }
    for (int r = 0; r < 3; ++r) {
        rec.to_world[12 + r] = rec.center[r];
        rec.to_object[12 + r] = -(rec.to_object[0 + r] * rec.center[0] + rec.to_object[4 + r] * rec.center[1] + rec.to_object[8 + r] * rec.center[2]);
    }
    rec.to_world[15] = rec.to_object[15] = 1.f;
    std::vector<float> P;                                          // bbox() corners, :133-139
    for (int i = 0; i < 8; ++i)
        for (int a = 0; a < 3; ++a) P.push_back(rec.center[a] + (((i >> a) & 1) ? radius : -radius));
    auto mesh = std::make_shared<Mesh>("sphere", std::move(P), std::vector<uint32_t>{ 0, 1, 2 });
    mesh->m_sphere = true; mesh->m_sphere_rec = rec;
    return mesh;
}
static void flatten(const std::vector<std::shared_ptr<Mesh>> &shapes, std::vector<float> &pos, std::vector<float> &nrm,
                    std::vector<float> &tex, std::vector<mi_bitmap> &bitmaps, std::vector<std::shared_ptr<BitmapTexture>> &bitmap_objs,
                    std::vector<float> &tables, std::vector<uint32_t> &faces, std::vector<mi_shape> &srecs, std::vector<mi_bsdf> &brecs,
                    std::vector<mi_emitter> &erecs, std::vector<mi_rectangle> &rrecs, std::vector<mi_sphere> &sphrecs) {
    pos.clear(); nrm.clear(); tex.clear(); faces.clear(); srecs.clear(); brecs.clear(); erecs.clear(); rrecs.clear(); sphrecs.clear();
    bool any_normals = false, any_texcoords = false;
    for (auto &m : shapes) { any_normals = any_normals || m->has_vertex_normals(); any_texcoords = any_texcoords || m->has_vertex_texcoords(); }
    std::map<const BSDF *, uint32_t> bsdf_index;
    bitmaps.clear(); bitmap_objs.clear(); tables.clear();
    std::map<const BitmapTexture *, uint32_t> bitmap_index;
    // the plugin's C-ABI record, bitmap parameters resolved to entries of the scene's bitmap table
    auto push_record = [&](const BSDF *b) {
        mi_bsdf r = b->record();
        if (!b->table().empty()) {                             // roughplastic: its transmittance table joins the scene's buffer
            r.params[5] = (float) tables.size();
            tables.insert(tables.end(), b->table().begin(), b->table().end());
        }
        for (int k = 0; k < 3; ++k) {
            const std::shared_ptr<BitmapTexture> &t = b->bitmap(k);
            if (!t) continue;
            auto it = bitmap_index.find(t.get());
            if (it == bitmap_index.end()) {
                it = bitmap_index.emplace(t.get(), (uint32_t) bitmaps.size()).first;
                bitmaps.push_back(t->record()); bitmap_objs.push_back(t);
            }
            r.tex[k] = mi_texture{}; r.tex[k].type = MI_TEX_BITMAP; r.tex[k].v[0] = (float) it->second;
        }
        brecs.push_back(r);
    };
    for (auto &m : shapes) {
        uint32_t vbase = (uint32_t) (pos.size() / 3), fbase = (uint32_t) (faces.size() / 3);
        pos.insert(pos.end(), m->vertex_positions_buffer().begin(), m->vertex_positions_buffer().end());
        if (any_normals) {
            if (m->has_vertex_normals()) nrm.insert(nrm.end(), m->vertex_normals_buffer().begin(), m->vertex_normals_buffer().end());
            else nrm.insert(nrm.end(), m->vertex_positions_buffer().size(), 0.f);
        }
        if (any_texcoords) {
            if (m->has_vertex_texcoords()) tex.insert(tex.end(), m->vertex_texcoords_buffer().begin(), m->vertex_texcoords_buffer().end());
            else tex.insert(tex.end(), (size_t) m->vertex_count() * 2, 0.f);
        }
        for (uint32_t i : m->faces_buffer()) faces.push_back(i + vbase);
        mi_shape s{};
        std::shared_ptr<BSDF> b = m->bsdf();
        if (!b) {                                              // shape.cpp:75-81: diffuse, 0.5 (0 for emitters)
            Properties p("diffuse");
            if (m->emitter()) p.set_float("reflectance", 0.f);
            b = std::make_shared<SmoothDiffuse>(p);
            m->set_bsdf(b);
        }
        auto it = bsdf_index.find(b.get());
        if (it == bsdf_index.end()) {
            it = bsdf_index.emplace(b.get(), (uint32_t) brecs.size()).first;
            push_record(b.get());
            if (b->twosided()) {                                   // the back side's record follows (or is the front's own)
                const uint32_t self = it->second;
                const BSDF *back = b->back().get();
                bool same_bitmaps = true;
                for (int k = 0; k < 3; ++k) same_bitmaps = same_bitmaps && back->bitmap(k) == b->bitmap(k);
                if (same_bitmaps && miw_same_record(back->record(), b->record())) brecs[self].back = self;
                else {
                    auto jt = bsdf_index.find(back);
                    if (jt == bsdf_index.end()) { jt = bsdf_index.emplace(back, (uint32_t) brecs.size()).first; push_record(back); }
                    brecs[self].back = jt->second;
                }
            }
        }
        s.bsdf = it->second;
        s.emitter = -1;
        if (m->emitter()) {
            s.emitter = (int32_t) erecs.size();
            mi_emitter e{}; e.shape = (uint32_t) srecs.size();
            Color3f r = m->emitter()->radiance(); e.radiance[0] = r[0]; e.radiance[1] = r[1]; e.radiance[2] = r[2];
            e.radiance_tex = m->emitter()->radiance_texture();
            erecs.push_back(e);
        }
        s.flags = (m->has_vertex_normals() ? MI_SHAPE_HAS_NORMALS : 0) | (m->has_vertex_texcoords() ? MI_SHAPE_HAS_TEXCOORDS : 0);
        s.first_face = fbase; s.face_count = m->face_count();
        if (m->is_rectangle()) {
            s.flags |= MI_SHAPE_RECTANGLE;
            mi_rectangle r{}; r.shape = (uint32_t) srecs.size();
            std::memcpy(r.to_world, m->rectangle_to_world().m, 64); std::memcpy(r.to_object, m->rectangle_to_world().inv, 64);
            rrecs.push_back(r);
        }
        if (m->is_sphere()) {
            s.flags |= MI_SHAPE_SPHERE;
            mi_sphere r = m->sphere_record(); r.shape = (uint32_t) srecs.size();
            sphrecs.push_back(r);
        }
        srecs.push_back(s);
    }
}
void Scene::build(int device, int bvh_quality) {
    if (m_shapes.empty()) Throw("Scene: no shapes");
    flatten(m_shapes, m_positions, m_normals, m_texcoords, m_bitmap_recs, m_bitmap_objs, m_bsdf_tables, m_faces, m_shape_recs, m_bsdf_recs, m_emitters, m_rect_recs, m_sphere_recs);
    m_desc.spheres = m_sphere_recs.empty() ? nullptr : m_sphere_recs.data(); m_desc.sphere_count = (uint32_t) m_sphere_recs.size();
    m_desc.rectangles = m_rect_recs.empty() ? nullptr : m_rect_recs.data(); m_desc.rectangle_count = (uint32_t) m_rect_recs.size();
    m_desc.vertex_positions = m_positions.data();
    m_desc.vertex_normals = m_normals.empty() ? nullptr : m_normals.data();
    m_desc.vertex_texcoords = m_texcoords.empty() ? nullptr : m_texcoords.data();
    m_desc.bitmaps = m_bitmap_recs.empty() ? nullptr : m_bitmap_recs.data(); m_desc.bitmap_count = (uint32_t) m_bitmap_recs.size();
    m_desc.bsdf_tables = m_bsdf_tables.empty() ? nullptr : m_bsdf_tables.data(); m_desc.bsdf_table_floats = (uint32_t) m_bsdf_tables.size();
    m_desc.vertex_count = (uint32_t) (m_positions.size() / 3);
    m_desc.faces = m_faces.data(); m_desc.face_count = (uint32_t) (m_faces.size() / 3);
    m_desc.shapes = m_shape_recs.data(); m_desc.shape_count = (uint32_t) m_shape_recs.size();
    m_desc.bsdfs = m_bsdf_recs.data(); m_desc.bsdf_count = (uint32_t) m_bsdf_recs.size();
    m_desc.emitters = m_emitters.data(); m_desc.emitter_count = (uint32_t) m_emitters.size();
    m_desc.envmap = nullptr;
    if (m_env) {
        if (m_env->data().empty()) Throw("envmap: no bitmap set");
        m_env_rec.rgba = m_env->data().data(); m_env_rec.width = m_env->width(); m_env_rec.height = m_env->height();
        m_env_rec.scale = m_env->scale();
        std::memcpy(m_env_rec.to_world, m_env->world_transform().m, 64);
        // emitter order (scene.cpp:38-60): area lights of the shapes added before the envmap come first
        uint32_t before = 0;
        for (size_t i = 0; i < m_env_after_shapes && i < m_shapes.size(); ++i) if (m_shapes[i]->emitter()) ++before;
        m_env_rec.emitter_index = before;
        // scene->bbox().bounding_sphere() (bbox.h:329-332): centre of the bbox, distance to its max corner
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (size_t i = 0; i < m_positions.size(); i += 3)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], m_positions[i + a]); hi[a] = std::max(hi[a], m_positions[i + a]); }
        float c[3], d2 = 0.f;
        for (int a = 0; a < 3; ++a) { c[a] = (lo[a] + hi[a]) * .5f; float d = c[a] - hi[a]; d2 += d * d; }
        m_env_rec.bsphere_radius = std::sqrt(d2);
        m_desc.envmap = &m_env_rec;
    }
    m_built = true;
    if (device < 0) return;                                    // flatten only (host-side tests)
    if (!m_ctx) {
        mi_status st = mi_create(device, &m_ctx);
        if (st != MI_OK) Throw(std::string("mi_create failed: ") + mi_last_error(nullptr));
    }
    if (mi_scene_upload(m_ctx, &m_desc) != MI_OK) Throw(std::string("mi_scene_upload: ") + mi_last_error(m_ctx));
    if (mi_bvh_build(m_ctx, bvh_quality) != MI_OK) Throw(std::string("mi_bvh_build: ") + mi_last_error(m_ctx));
}
void Scene::ray_intersect_preliminary(const mi_rays_soa &rays, const mi_hits_soa &hits, uint64_t n) const {
    if (!m_ctx) Throw("Scene: not built on a device");
    if (mi_trace(m_ctx, &rays, &hits, n, 0) != MI_OK) Throw(std::string("mi_trace: ") + mi_last_error(m_ctx));
}
void Scene::ray_test(const mi_rays_soa &rays, float *t_out, uint64_t n) const {
    if (!m_ctx) Throw("Scene: not built on a device");
    mi_hits_soa h{}; h.t = t_out;
    if (mi_trace(m_ctx, &rays, &h, n, 1) != MI_OK) Throw(std::string("mi_trace: ") + mi_last_error(m_ctx));
}
PreliminaryIntersection3f Scene::ray_intersect_preliminary(const Ray3f &r) const {
    mi_rays_soa rays{ &r.o[0], &r.o[1], &r.o[2], &r.d[0], &r.d[1], &r.d[2], &r.mint, &r.maxt };
    PreliminaryIntersection3f pi{};
    mi_hits_soa hits{ &pi.t, &pi.u, &pi.v, &pi.prim_index, &pi.shape_index };
    ray_intersect_preliminary(rays, hits, 1);
    return pi;
}
bool Scene::ray_test(const Ray3f &r) const {
    mi_rays_soa rays{ &r.o[0], &r.o[1], &r.o[2], &r.d[0], &r.d[1], &r.d[2], &r.mint, &r.maxt };
    float t;
    ray_test(rays, &t, 1);
    return t != std::numeric_limits<float>::infinity();
}

// ============================================================================================
// PathIntegrator
// ============================================================================================
static uint32_t round_to_power_of_two(uint32_t v) { uint32_t r = 1; while (r < v) r <<= 1; return v == 0 ? 0 : r; }

SamplingIntegrator::SamplingIntegrator(const Properties &props) {
    m_block_size = (uint32_t) props.int_("block_size", 0);
    uint32_t bs = round_to_power_of_two(m_block_size);
    if (m_block_size > 0 && bs != m_block_size) m_block_size = bs;   // integrator.cpp:27-32 (warns)
    m_samples_per_pass = (uint32_t) props.int_("samples_per_pass", (int64_t) (uint32_t) -1);
    m_timeout = props.float_("timeout", -1.f);
    m_hide_emitters = props.bool_("hide_emitters", false);
}
PathIntegrator::PathIntegrator(const Properties &props) : SamplingIntegrator(props) {
    m_rr_depth = (int) props.int_("rr_depth", 5);
    if (m_rr_depth <= 0) Throw("\"rr_depth\" must be set to a value greater than zero!");
    m_max_depth = (int) props.int_("max_depth", -1);
    if (m_max_depth < 0 && m_max_depth != -1) Throw("\"max_depth\" must be set to -1 (infinite) or a value >= 0");
}
void PathIntegrator::fill_integrator(mi_render_cfg &cfg) const {
    cfg.integrator = MI_INTEGRATOR_PATH; cfg.max_depth = m_max_depth; cfg.rr_depth = m_rr_depth;
}
// direct.cpp:82-103
DirectIntegrator::DirectIntegrator(const Properties &props) : SamplingIntegrator(props) {
    if (props.has_property("shading_samples") && (props.has_property("emitter_samples") || props.has_property("bsdf_samples")))
        Throw("Cannot specify both 'shading_samples' and ('emitter_samples' and/or 'bsdf_samples').");
    size_t shading_samples = (size_t) props.int_("shading_samples", 1);
    m_emitter_samples = (size_t) props.int_("emitter_samples", (int64_t) shading_samples);
    m_bsdf_samples = (size_t) props.int_("bsdf_samples", (int64_t) shading_samples);
    if (m_emitter_samples + m_bsdf_samples == 0) Throw("Must have at least 1 BSDF or emitter sample!");
}
void DirectIntegrator::fill_integrator(mi_render_cfg &cfg) const {
    cfg.integrator = MI_INTEGRATOR_DIRECT; cfg.max_depth = -1; cfg.rr_depth = 5;
    cfg.emitter_samples = (uint32_t) m_emitter_samples; cfg.bsdf_samples = (uint32_t) m_bsdf_samples;
    cfg.hide_emitters = m_hide_emitters ? 1 : 0;
}
std::shared_ptr<SamplingIntegrator> make_integrator(const Properties &props) {
    if (props.plugin_name() == "moment") Throw("moment: needs a nested integrator (MomentIntegrator(props, nested))");
    if (props.plugin_name() == "path") return std::make_shared<PathIntegrator>(props);
    if (props.plugin_name() == "direct") return std::make_shared<DirectIntegrator>(props);
    Throw("Plugin \"" + props.plugin_name() + "\" not found!");
}
void SamplingIntegrator::cancel() { mi_ctx *c = m_active_ctx.load(); if (c) mi_cancel(c); }

// integrator.cpp:75-86
static size_t samples_per_pass_of(uint32_t samples_per_pass, size_t total_spp) {
    size_t spp_pass = (samples_per_pass == (uint32_t) -1) ? total_spp : std::min((size_t) samples_per_pass, total_spp);
    if (spp_pass == 0 || (total_spp % spp_pass) != 0)
        Throw("sample_count (" + std::to_string(total_spp) + ") must be a multiple of samples_per_pass (" + std::to_string(spp_pass) + ").");
    return spp_pass;
}
uint32_t SamplingIntegrator::pass_count(const PerspectiveCamera *sensor) const {
    size_t total_spp = sensor->sampler()->sample_count();
    return (uint32_t) (total_spp / samples_per_pass_of(m_samples_per_pass, total_spp));
}
void SamplingIntegrator::make_render_cfg(const PerspectiveCamera *sensor, mi_render_cfg &cfg,
                                     std::vector<uint32_t> &block_ids, std::vector<uint32_t> &tiles,
                                     uint32_t n_threads, uint32_t pass) const {
    std::memset(&cfg, 0, sizeof cfg);
    const Film *film = sensor->film().get();
    auto cs = film->crop_size(); auto co = film->crop_offset();
    size_t total_spp = sensor->sampler()->sample_count();
    size_t spp_pass = samples_per_pass_of(m_samples_per_pass, total_spp);
    if (pass >= total_spp / spp_pass) Throw("make_render_cfg: pass index out of range");
    // block size, integrator.cpp:88-97 (MTS_BLOCK_SIZE = 32, spiral.h:9-10)
    uint32_t bs = m_block_size;
    if (bs == 0) {
        bs = 32;
        while (true) {
            size_t blocks = (size_t) ((cs[0] + bs - 1) / bs) * ((cs[1] + bs - 1) / bs);
            if (bs == 1 || blocks >= n_threads) break;
            bs /= 2;
        }
    }
    cfg.crop_x = co[0]; cfg.crop_y = co[1]; cfg.crop_w = cs[0]; cfg.crop_h = cs[1];
    cfg.spp = (uint32_t) spp_pass;
    fill_integrator(cfg);
    cfg.accumulate = pass > 0 ? 1 : 0;
    cfg.base_seed = sensor->sampler()->base_seed();
    cfg.block_size = (int32_t) bs;
    // spiral visitation order -> block id per row-major block (spiral.cpp)
    Spiral spiral(cs, co, bs, 1);
    uint32_t nbx = (cs[0] + bs - 1) / bs;
    block_ids.assign(spiral.block_count(), 0);
    std::vector<uint32_t> by_id(spiral.block_count());
    for (size_t i = 0; i < spiral.block_count(); ++i) {
        Spiral::Block b = spiral.next_block();
        uint32_t bx = (uint32_t) (b.offset[0] - co[0]) / bs, by = (uint32_t) (b.offset[1] - co[1]) / bs;
        block_ids[by * nbx + bx] = (uint32_t) (b.block_id + (size_t) pass * spiral.block_count());   // spiral.cpp:41
        by_id[b.block_id] = by * nbx + bx;
    }
    tiles.clear();
    if (m_world > 1)                                           // interleaved shard over the spiral order
        for (size_t id = m_rank; id < by_id.size(); id += m_world) tiles.push_back(by_id[id]);
    cfg.block_ids = block_ids.data(); cfg.block_count = (uint32_t) block_ids.size();
    cfg.tile_list = m_world > 1 ? tiles.data() : nullptr; cfg.tile_count = (uint32_t) tiles.size();
    std::memcpy(cfg.sample_to_camera, sensor->sample_to_camera().m, 64);
    std::memcpy(cfg.to_world, sensor->world_transform().m, 64);
    cfg.near_clip = sensor->near_clip(); cfg.far_clip = sensor->far_clip();
    auto pp = sensor->principal_point_offset();
    cfg.principal_point_offset[0] = pp[0]; cfg.principal_point_offset[1] = pp[1];
    const ReconstructionFilter *rf = film->reconstruction_filter();
    for (int i = 0; i < 32; ++i) cfg.filter_lut[i] = rf->values()[i];
    cfg.filter_radius = rf->radius(); cfg.filter_border = (int32_t) rf->border_size();
    cfg.timeout_s = m_timeout; cfg.profile = m_profile ? 1 : 0;
    cfg.plan = m_plan;
}

bool SamplingIntegrator::render_passes(Scene *scene, PerspectiveCamera *sensor, float *film5, int moment_pass) {
    const uint32_t passes = pass_count(sensor);
    mi_counters total{};
    for (uint32_t pass = 0; pass < passes; ++pass) {
        mi_render_cfg cfg; std::vector<uint32_t> block_ids, tiles;
        make_render_cfg(sensor, cfg, block_ids, tiles, 1, pass);
        cfg.moment_pass = moment_pass;
        m_active_ctx.store(scene->ctx());
        mi_status st = mi_render(scene->ctx(), &cfg, film5);
        m_active_ctx.store(nullptr);
        mi_get_counters(scene->ctx(), &m_counters);
        if (pass > 0) {                                        // work counters add up over the passes
            m_counters.samples += total.samples; m_counters.segments += total.segments; m_counters.shadow_rays += total.shadow_rays;
            m_counters.iterations += total.iterations; m_counters.ms_render += total.ms_render;
        }
        total = m_counters;
        if (st == MI_ERR_CANCELLED) return false;
        if (st != MI_OK) Throw(std::string("mi_render: ") + mi_last_error(scene->ctx()));
    }
    return true;
}
bool SamplingIntegrator::render(Scene *scene, PerspectiveCamera *sensor) {
    if (!scene || !sensor) Throw("render(): null scene or sensor");
    if (!scene->ctx()) Throw("render(): the scene has no device context (Scene::build(device >= 0) first)");
    Film *film = sensor->film().get();
    film->prepare({ "X", "Y", "Z", "A", "W" });                // integrator.cpp:67-73
    return render_passes(scene, sensor, film->storage().data(), MI_MOMENT_OFF);
}

// moment.cpp:33-53
MomentIntegrator::MomentIntegrator(const Properties &props, std::shared_ptr<SamplingIntegrator> nested, std::string nested_name)
    : SamplingIntegrator(props), m_nested(std::move(nested)), m_name(std::move(nested_name)) {
    if (!m_nested) Throw("Child objects must be of type 'SamplingIntegrator'!");
    if (dynamic_cast<MomentIntegrator *>(m_nested.get())) Throw("moment: nested moment integrators are not supported");
#if MIW_SPECTRAL
    Throw("moment: provided by the scalar_rgb build of this layer only");
#endif
}
std::vector<std::string> MomentIntegrator::aov_names() const {
    std::vector<std::string> names = { m_name + ".X", m_name + ".Y", m_name + ".Z" };
    for (int i = 0; i < 3; ++i) names.push_back("m2_" + names[i]);
    return names;
}
bool MomentIntegrator::render(Scene *scene, PerspectiveCamera *sensor) {
    if (!scene || !sensor) Throw("render(): null scene or sensor");
    if (!scene->ctx()) Throw("render(): the scene has no device context (Scene::build(device >= 0) first)");
    Film *film = sensor->film().get();
    std::vector<std::string> channels = { "X", "Y", "Z", "A", "W" };
    for (const std::string &n : aov_names()) channels.push_back(n);
    film->prepare(channels);
    auto cs = film->crop_size();
    const size_t n = (size_t) cs[0] * cs[1];
    std::vector<float> values(n * 5), squares(n * 5);
    m_nested->set_shard(m_rank, m_world); m_nested->set_plan(m_plan); m_nested->set_profile(m_profile);
    bool ok = m_nested->render_passes(scene, sensor, values.data(), MI_MOMENT_VALUES) &&
              m_nested->render_passes(scene, sensor, squares.data(), MI_MOMENT_SQUARES);
    m_counters = m_nested->counters();
    float *out = film->storage().data();
    for (size_t i = 0; i < n; ++i) {
        const float *v = &values[i * 5], *q = &squares[i * 5];
        float *o = out + i * 11;
        for (int k = 0; k < 5; ++k) o[k] = v[k];
        for (int k = 0; k < 3; ++k) { o[5 + k] = v[k]; o[8 + k] = q[k]; }
    }
    return ok;
}

// ============================================================================================
// XML front-end (subset)
// ============================================================================================
Transform4f Transform4f::from_matrix(const float *r) {
    Transform4f t;
    for (int row = 0; row < 4; ++row) for (int col = 0; col < 4; ++col) t.m[col * 4 + row] = r[row * 4 + col];
    double a[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = r[i * 4 + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int i = 0; i < 4; ++i) {
        int piv = i;
        for (int k = i + 1; k < 4; ++k) if (std::fabs(a[k][i]) > std::fabs(a[piv][i])) piv = k;
        if (a[piv][i] == 0.0) Throw("<matrix>: singular matrix");
        for (int c = 0; c < 8; ++c) std::swap(a[i][c], a[piv][c]);
        double d = a[i][i];
        for (int c = 0; c < 8; ++c) a[i][c] /= d;
        for (int k = 0; k < 4; ++k) if (k != i) { double f = a[k][i]; for (int c = 0; c < 8; ++c) a[k][c] -= f * a[i][c]; }
    }
    for (int row = 0; row < 4; ++row) for (int col = 0; col < 4; ++col) t.inv[col * 4 + row] = (float) a[row][4 + col];
    return t;
}
Transform4f Transform4f::rotate(const Vector3f &axis_, float angle) {      // enoki::rotate(axis, rad): Rodrigues
    float len = std::sqrt(axis_[0] * axis_[0] + axis_[1] * axis_[1] + axis_[2] * axis_[2]);
    float x = axis_[0] / len, y = axis_[1] / len, z = axis_[2] / len;
    float rad = angle * (MIW_PI / 180.f), s = std::sin(rad), c = std::cos(rad), t = 1.f - c;
    const float r[16] = { t * x * x + c,     t * x * y - s * z, t * x * z + s * y, 0,
                          t * x * y + s * z, t * y * y + c,     t * y * z - s * x, 0,
                          t * x * z - s * y, t * y * z + s * x, t * z * z + c,     0,
                          0, 0, 0, 1 };
    Transform4f out;
    for (int row = 0; row < 4; ++row) for (int col = 0; col < 4; ++col) { out.m[col * 4 + row] = r[row * 4 + col]; out.inv[col * 4 + row] = r[col * 4 + row]; }
    return out;
}

namespace {
struct XmlNode { std::string tag; std::map<std::string, std::string> attr; std::vector<XmlNode> children; };

struct XmlParser {
    const std::string &s; size_t p = 0;
    explicit XmlParser(const std::string &src) : s(src) {}
    [[noreturn]] void fail(const std::string &msg) const {
        size_t line = 1 + (size_t) std::count(s.begin(), s.begin() + (long) std::min(p, s.size()), '\n');
        Throw("Error while loading XML (line " + std::to_string(line) + "): " + msg);
    }
    void skip_ws() { while (p < s.size() && std::isspace((unsigned char) s[p])) ++p; }
    bool starts(const char *t) const { return s.compare(p, std::strlen(t), t) == 0; }
    void skip_misc() {
        for (;;) {
            skip_ws();
            if (starts("<!--")) { size_t e = s.find("-->", p); if (e == std::string::npos) fail("unterminated comment"); p = e + 3; }
            else if (starts("<?")) { size_t e = s.find("?>", p); if (e == std::string::npos) fail("unterminated declaration"); p = e + 2; }
            else return;
        }
    }
    std::string name() { size_t b = p; while (p < s.size() && (std::isalnum((unsigned char) s[p]) || s[p] == '_' || s[p] == '-' || s[p] == ':')) ++p; if (p == b) fail("expected a name"); return s.substr(b, p - b); }
    XmlNode element() {
        skip_misc();
        if (p >= s.size() || s[p] != '<') fail("expected an element");
        ++p;
        XmlNode n; n.tag = name();
        for (;;) {
            skip_ws();
            if (p >= s.size()) fail("unterminated element <" + n.tag + ">");
            if (s[p] == '/') { if (!starts("/>")) fail("malformed tag"); p += 2; return n; }
            if (s[p] == '>') { ++p; break; }
            std::string key = name(); skip_ws();
            if (p >= s.size() || s[p] != '=') fail("expected '=' after attribute \"" + key + "\"");
            ++p; skip_ws();
            char q = p < s.size() ? s[p] : 0;
            if (q != '"' && q != '\'') fail("expected a quoted attribute value");
            size_t e = s.find(q, p + 1);
            if (e == std::string::npos) fail("unterminated attribute value");
            n.attr[key] = s.substr(p + 1, e - p - 1); p = e + 1;
        }
        for (;;) {
            size_t lt = s.find('<', p);
            if (lt == std::string::npos) fail("unterminated element <" + n.tag + ">");
            p = lt;
            if (starts("<!--") || starts("<?")) { skip_misc(); continue; }
            if (starts("</")) { p += 2; std::string c = name(); if (c != n.tag) fail("mismatched closing tag </" + c + ">"); skip_ws(); if (p >= s.size() || s[p] != '>') fail("malformed closing tag"); ++p; return n; }
            n.children.push_back(element());
        }
    }
};

struct XmlCtx {
    std::map<std::string, std::string> params;
    std::map<std::string, std::shared_ptr<BSDF>> bsdfs;
    std::map<std::string, std::shared_ptr<BitmapTexture>> textures;
    std::string base_dir;
    std::string subst(const std::string &v) const {           // $name parameter substitution (xml.cpp:150-180)
        std::string out; size_t i = 0;
        while (i < v.size()) {
            if (v[i] == '$') {
                size_t j = i + 1; while (j < v.size() && (std::isalnum((unsigned char) v[j]) || v[j] == '_')) ++j;
                std::string key = v.substr(i + 1, j - i - 1);
                auto it = params.find(key);
                if (it == params.end()) Throw("Error while loading XML: undefined parameter \"$" + key + "\"");
                out += it->second; i = j;
            } else out += v[i++];
        }
        return out;
    }
    std::string get(const XmlNode &n, const std::string &key) const {
        auto it = n.attr.find(key);
        if (it == n.attr.end()) Throw("Error while loading XML: <" + n.tag + "> is missing the attribute \"" + key + "\"");
        return subst(it->second);
    }
    std::string get(const XmlNode &n, const std::string &key, const std::string &def) const { return n.attr.count(key) ? subst(n.attr.at(key)) : def; }
};
std::vector<float> parse_floats(const std::string &v, const char *what) {
    std::vector<float> out; const char *c = v.c_str();
    for (;;) {
        while (*c == ' ' || *c == ',' || *c == '\t' || *c == '\n') ++c;
        if (!*c) break;
        char *e; float f = std::strtof(c, &e);
        if (e == c) Throw(std::string("Error while loading XML: could not parse ") + what + " \"" + v + "\"");
        out.push_back(f); c = e;
    }
    return out;
}
Vector3f parse_vec3(const XmlCtx &cx, const XmlNode &n, float def) {
    if (n.attr.count("value")) { auto f = parse_floats(cx.get(n, "value"), "a vector"); if (f.size() == 1) return { f[0], f[0], f[0] }; if (f.size() != 3) Throw("Error while loading XML: <" + n.tag + "> expects 1 or 3 values"); return { f[0], f[1], f[2] }; }
    auto one = [&](const char *k) { return n.attr.count(k) ? parse_floats(cx.get(n, k), "a number").at(0) : def; };
    return { one("x"), one("y"), one("z") };
}
Transform4f parse_transform(const XmlCtx &cx, const XmlNode &n) {
    Transform4f t;
    for (const XmlNode &c : n.children) {                       // each child is applied on the left (xml.cpp:880-940)
        Transform4f m;
        if (c.tag == "translate") m = Transform4f::translate(parse_vec3(cx, c, 0.f));
        else if (c.tag == "scale") m = Transform4f::scale(parse_vec3(cx, c, 1.f));
        else if (c.tag == "rotate") m = Transform4f::rotate(parse_vec3(cx, c, 0.f), parse_floats(cx.get(c, "angle"), "an angle").at(0));
        else if (c.tag == "lookat") {
            auto v = [&](const char *k, Vector3f def) { if (!c.attr.count(k)) return def; auto f = parse_floats(cx.get(c, k), "a point"); if (f.size() != 3) Throw("Error while loading XML: <lookat> expects 3 values"); return Vector3f{ f[0], f[1], f[2] }; };
            m = Transform4f::look_at(v("origin", { 0, 0, 0 }), v("target", { 0, 0, 1 }), v("up", { 0, 1, 0 }));
        } else if (c.tag == "matrix") { auto f = parse_floats(cx.get(c, "value"), "a matrix"); if (f.size() != 16) Throw("Error while loading XML: <matrix> expects 16 values"); m = Transform4f::from_matrix(f.data()); }
        else Throw("Error while loading XML: unexpected <" + c.tag + "> inside <transform>");
        t = m * t;
    }
    return t;
}
// fills `props` from the value children of `n`; returns the object children (bsdf / emitter / film / ... / ref)
std::vector<const XmlNode *> parse_properties(const XmlCtx &cx, const XmlNode &n, Properties &props) {
    std::vector<const XmlNode *> objects;
    for (const XmlNode &c : n.children) {
        if (c.tag == "float") props.set_float(cx.get(c, "name"), parse_floats(cx.get(c, "value"), "a float").at(0));
        else if (c.tag == "integer") props.set_int(cx.get(c, "name"), std::strtoll(cx.get(c, "value").c_str(), nullptr, 10));
        else if (c.tag == "boolean") { std::string v = to_lower(cx.get(c, "value")); if (v != "true" && v != "false") Throw("Error while loading XML: could not parse boolean value \"" + v + "\" -- must be \"true\" or \"false\""); props.set_bool(cx.get(c, "name"), v == "true"); }
        else if (c.tag == "string") props.set_string(cx.get(c, "name"), cx.get(c, "value"));
        else if (c.tag == "rgb") { auto f = parse_floats(cx.get(c, "value"), "an <rgb> value"); if (f.size() == 1) f = { f[0], f[0], f[0] }; if (f.size() != 3) Throw("Error while loading XML: 'rgb' tag requires one or three values"); props.set_color(cx.get(c, "name"), { f[0], f[1], f[2] }); }
        else if (c.tag == "spectrum") { auto f = parse_floats(cx.get(c, "value"), "a <spectrum> value"); if (f.size() != 1) Throw("Error while loading XML: only constant <spectrum value=\"v\"/> is supported"); props.set_float(cx.get(c, "name"), f[0]); }
        else if (c.tag == "point" || c.tag == "vector") { Vector3f v = parse_vec3(cx, c, 0.f); props.set_color(cx.get(c, "name"), Color3f{ v[0], v[1], v[2] }); }
        else if (c.tag == "transform") props.set_transform(cx.get(c, "name"), parse_transform(cx, c));
        else objects.push_back(&c);
    }
    return objects;
}
std::shared_ptr<BSDF> make_bsdf(const Properties &p) {
    const std::string &t = p.plugin_name();
    if (t == "diffuse") return std::make_shared<SmoothDiffuse>(p);
    if (t == "dielectric") return std::make_shared<SmoothDielectric>(p);
    if (t == "roughconductor") return std::make_shared<RoughConductor>(p);
    if (t == "conductor") return std::make_shared<SmoothConductor>(p);
    if (t == "plastic") return std::make_shared<SmoothPlastic>(p);
    if (t == "roughdielectric") return std::make_shared<RoughDielectric>(p);
    if (t == "roughplastic") return std::make_shared<RoughPlastic>(p);
    Throw("Plugin \"" + t + "\" not found!");
}
std::string resolve(const XmlCtx &cx, const std::string &f) { return (!f.empty() && f[0] == '/') ? f : cx.base_dir + "/" + f; }
std::shared_ptr<BitmapTexture> parse_texture(XmlCtx &cx, const XmlNode &n) {
    Properties p(cx.get(n, "type"));
    if (p.plugin_name() != "bitmap") Throw("Plugin \"" + p.plugin_name() + "\" not found!");
    auto objs = parse_properties(cx, n, p);
    if (!objs.empty()) Throw("Error while loading XML: unexpected <" + objs[0]->tag + "> inside <texture>");
    p.set_string("filename", resolve(cx, p.string("filename")));
    auto t = std::make_shared<BitmapTexture>(p);
    if (n.attr.count("id")) cx.textures[cx.get(n, "id")] = t;
    return t;
}
std::shared_ptr<BSDF> parse_bsdf(XmlCtx &cx, const XmlNode &n) {
    Properties p(cx.get(n, "type"));
    auto objs = parse_properties(cx, n, p);
    std::shared_ptr<BSDF> b;
    if (p.plugin_name() == "twosided") {                       // nested <bsdf> / <ref> children, twosided.cpp:63-73
        std::vector<std::shared_ptr<BSDF>> nested;
        for (const XmlNode *c : objs) {
            if (c->tag == "bsdf") nested.push_back(parse_bsdf(cx, *c));
            else if (c->tag == "ref") {
                auto it = cx.bsdfs.find(cx.get(*c, "id"));
                if (it == cx.bsdfs.end()) Throw("Error while loading XML: reference to unknown object \"" + cx.get(*c, "id") + "\"");
                nested.push_back(it->second);
            } else Throw("Error while loading XML: unexpected <" + c->tag + "> inside <bsdf>");
        }
        if (nested.size() > 2) Throw("At most two nested BSDFs can be specified!");
        b = std::make_shared<TwoSidedBRDF>(nested.empty() ? nullptr : nested[0], nested.size() == 2 ? nested[1] : nullptr);
    } else {
        for (const XmlNode *c : objs) {                        // <texture type="bitmap" name=...> / <ref id=... name=...>
            if (c->tag == "texture") p.set_texture(cx.get(*c, "name"), parse_texture(cx, *c));
            else if (c->tag == "ref" && c->attr.count("name")) {
                auto it = cx.textures.find(cx.get(*c, "id"));
                if (it == cx.textures.end()) Throw("Error while loading XML: reference to unknown object \"" + cx.get(*c, "id") + "\"!");
                p.set_texture(cx.get(*c, "name"), it->second);
            } else Throw("Error while loading XML: unexpected <" + c->tag + "> inside <bsdf>");
        }
        b = make_bsdf(p);
    }
    if (n.attr.count("id")) cx.bsdfs[cx.get(n, "id")] = b;
    return b;
}
}

LoadedScene load_xml_string(const std::string &xml, const std::map<std::string, std::string> &params, const std::string &base_dir) {
    XmlParser parser(xml);
    XmlNode root = parser.element();
    if (root.tag != "scene") Throw("Error while loading XML: root element must be <scene>, found <" + root.tag + ">");
    XmlCtx cx; cx.params = params; cx.base_dir = base_dir;
    LoadedScene out; out.scene = std::make_shared<Scene>();
    std::shared_ptr<EnvironmentMapEmitter> pending_env;
    for (const XmlNode &n : root.children) {
        if (n.tag == "default") { std::string k = cx.get(n, "name"); if (!cx.params.count(k)) cx.params[k] = cx.get(n, "value"); }
        else if (n.tag == "bsdf") { if (!n.attr.count("id")) Throw("Error while loading XML: a top-level <bsdf> needs an id"); parse_bsdf(cx, n); }
        else if (n.tag == "texture") { if (!n.attr.count("id")) Throw("Error while loading XML: a top-level <texture> needs an id"); parse_texture(cx, n); }
        else if (n.tag == "integrator") {
            Properties p(cx.get(n, "type"));
            if (p.plugin_name() != "path" && p.plugin_name() != "direct" && p.plugin_name() != "moment") Throw("Plugin \"" + p.plugin_name() + "\" not found!");
            auto objs = parse_properties(cx, n, p);
            if (p.plugin_name() == "moment") {                 // <integrator type="moment"><integrator type="path" name=.../></integrator>
                if (objs.size() != 1 || objs[0]->tag != "integrator") Throw("Error while loading XML: <integrator type=\"moment\"> takes one nested <integrator> in this layer");
                Properties np(cx.get(*objs[0], "type"));
                if (!parse_properties(cx, *objs[0], np).empty()) Throw("Error while loading XML: unexpected object inside the nested <integrator>");
                out.integrator = std::make_shared<MomentIntegrator>(p, make_integrator(np), cx.get(*objs[0], "name", "integrator"));
            } else {
                if (!objs.empty()) Throw("Error while loading XML: unexpected <" + objs[0]->tag + "> inside <integrator>");
                out.integrator = make_integrator(p);
            }
        } else if (n.tag == "sensor") {
            Properties p(cx.get(n, "type"));
            if (p.plugin_name() != "perspective") Throw("Plugin \"" + p.plugin_name() + "\" not found!");
            std::shared_ptr<Film> film; std::shared_ptr<IndependentSampler> sampler;
            for (const XmlNode *c : parse_properties(cx, n, p)) {
                if (c->tag == "film") {
                    Properties fp(cx.get(*c, "type"));
                    if (fp.plugin_name() != "hdrfilm") Throw("Plugin \"" + fp.plugin_name() + "\" not found!");
                    auto fobjs = parse_properties(cx, *c, fp);
                    film = std::make_shared<Film>(fp);
                    for (const XmlNode *r : fobjs) {
                        if (r->tag != "rfilter") Throw("Error while loading XML: unexpected <" + r->tag + "> inside <film>");
                        Properties rp(cx.get(*r, "type")); parse_properties(cx, *r, rp);
                        if (rp.plugin_name() == "gaussian") film->set_reconstruction_filter(std::make_shared<GaussianFilter>(rp));
                        else if (rp.plugin_name() == "box") film->set_reconstruction_filter(std::make_shared<BoxFilter>(rp));
                        else Throw("Plugin \"" + rp.plugin_name() + "\" not found!");
                    }
                } else if (c->tag == "sampler") {
                    Properties sp(cx.get(*c, "type"));
                    if (sp.plugin_name() != "independent") Throw("Plugin \"" + sp.plugin_name() + "\" not found!");
                    parse_properties(cx, *c, sp);
                    sampler = std::make_shared<IndependentSampler>(sp);
                } else Throw("Error while loading XML: unexpected <" + c->tag + "> inside <sensor>");
            }
            if (!film) film = std::make_shared<Film>(Properties("hdrfilm"));               // sensor.cpp:60-66 defaults
            if (!sampler) sampler = std::make_shared<IndependentSampler>(Properties("independent"));
            out.sensor = std::make_shared<PerspectiveCamera>(p, film, sampler);
        } else if (n.tag == "emitter") {
            Properties p(cx.get(n, "type"));
            if (p.plugin_name() != "envmap") Throw("Error while loading XML: only <emitter type=\"envmap\"> may appear at the top level (area lights belong to a shape)");
            parse_properties(cx, n, p);
            p.set_string("filename", resolve(cx, p.string("filename")));
            out.scene->add_emitter(std::make_shared<EnvironmentMapEmitter>(p));   // its place among the shapes fixes the emitter order
        } else if (n.tag == "shape") {
            Properties p(cx.get(n, "type"));
            auto objs = parse_properties(cx, n, p);
            std::shared_ptr<Mesh> mesh;
            if (p.plugin_name() == "obj" || p.plugin_name() == "ply") {
                p.set_string("filename", resolve(cx, p.string("filename")));
                mesh = p.plugin_name() == "obj" ? load_obj(p) : load_ply(p);
            } else if (p.plugin_name() == "rectangle") {
                mesh = make_rectangle(p);
            } else if (p.plugin_name() == "sphere") {
                mesh = make_sphere(p);
            } else Throw("Plugin \"" + p.plugin_name() + "\" not found!");
            for (const XmlNode *c : objs) {
                if (c->tag == "bsdf") mesh->set_bsdf(parse_bsdf(cx, *c));
                else if (c->tag == "ref") { auto it = cx.bsdfs.find(cx.get(*c, "id")); if (it == cx.bsdfs.end()) Throw("Error while loading XML: reference to unknown object \"" + cx.get(*c, "id") + "\"!"); mesh->set_bsdf(it->second); }
                else if (c->tag == "emitter") {
                    Properties ep(cx.get(*c, "type"));
                    if (ep.plugin_name() != "area") Throw("Plugin \"" + ep.plugin_name() + "\" not found!");
                    parse_properties(cx, *c, ep);
                    mesh->set_emitter(std::make_shared<AreaLight>(ep));
                } else Throw("Error while loading XML: unexpected <" + c->tag + "> inside <shape>");
            }
            out.scene->add_shape(mesh); out.shapes.push_back(mesh);
        } else Throw("Error while loading XML: unexpected <" + n.tag + "> inside <scene>");
    }
    if (!out.integrator) out.integrator = std::make_shared<PathIntegrator>(Properties("path"));
    return out;
}
LoadedScene load_xml_file(const std::string &path, const std::map<std::string, std::string> &params) {
    std::string data = read_file(path, "XML");
    size_t slash = path.find_last_of("/\\");
    return load_xml_string(data, params, slash == std::string::npos ? "." : path.substr(0, slash));
}

} // namespace miwave

// ==============================================================================================
// C facade for ctypes (tests / bench plumbing). Handles are heap boxes of shared_ptr.
// ==============================================================================================
using namespace miwave;

static thread_local std::string g_err;
#define MIH_TRY try {
#define MIH_CATCH(fail_value) } catch (const std::exception &e) { g_err = e.what(); return fail_value; }

template <typename T> struct Box { std::shared_ptr<T> p; };

extern "C" {

const char *mih_last_error() { return g_err.c_str(); }

void *mih_props_create(const char *plugin) { return new Properties(plugin ? plugin : ""); }
void mih_props_destroy(void *p) { delete (Properties *) p; }
void mih_props_set_float(void *p, const char *n, float v) { ((Properties *) p)->set_float(n, v); }
void mih_props_set_int(void *p, const char *n, int64_t v) { ((Properties *) p)->set_int(n, v); }
void mih_props_set_bool(void *p, const char *n, int v) { ((Properties *) p)->set_bool(n, v != 0); }
void mih_props_set_string(void *p, const char *n, const char *v) { ((Properties *) p)->set_string(n, v); }
void mih_props_set_texture(void *p, const char *n, void *tex) { ((Properties *) p)->set_texture(n, ((Box<BitmapTexture> *) tex)->p); }
void *mih_bitmap_create(void *props, uint32_t w, uint32_t h, uint32_t channels, const float *data) {
    MIH_TRY
        auto t = std::make_shared<BitmapTexture>(*(Properties *) props);
        if (data) t->set_bitmap(w, h, channels, data);
        return new Box<BitmapTexture>{ t }; MIH_CATCH(nullptr)
}
void mih_bitmap_destroy(void *t) { delete (Box<BitmapTexture> *) t; }
int mih_bitmap_info(void *t, uint32_t *whc, float *mean3) {
    MIH_TRY const BitmapTexture &b = *((Box<BitmapTexture> *) t)->p; whc[0] = b.width(); whc[1] = b.height(); whc[2] = b.channels();
        Color3f m = b.mean(); mean3[0] = m[0]; mean3[1] = m[1]; mean3[2] = m[2]; return 0; MIH_CATCH(-1)
}
void mih_props_set_color(void *p, const char *n, float r, float g, float b) { ((Properties *) p)->set_color(n, Color3f{ r, g, b }); }
// 4x4 row-major matrix (what <matrix value="..."/> holds, xml.cpp)
void mih_props_set_matrix(void *p, const char *n, const float *row_major16) { ((Properties *) p)->set_transform(n, Transform4f::from_matrix(row_major16)); }
void mih_props_set_lookat(void *p, const char *n, const float *origin, const float *target, const float *up) {
    ((Properties *) p)->set_transform(n, Transform4f::look_at({ origin[0], origin[1], origin[2] },
                                                              { target[0], target[1], target[2] }, { up[0], up[1], up[2] }));
}

void *mih_bsdf_create(void *props) {
    MIH_TRY
        const Properties &p = *(Properties *) props;
        std::shared_ptr<BSDF> b;
        if (p.plugin_name() == "diffuse") b = std::make_shared<SmoothDiffuse>(p);
        else if (p.plugin_name() == "dielectric") b = std::make_shared<SmoothDielectric>(p);
        else if (p.plugin_name() == "roughconductor") b = std::make_shared<RoughConductor>(p);
        else if (p.plugin_name() == "conductor") b = std::make_shared<SmoothConductor>(p);
        else if (p.plugin_name() == "plastic") b = std::make_shared<SmoothPlastic>(p);
        else if (p.plugin_name() == "roughdielectric") b = std::make_shared<RoughDielectric>(p);
        else if (p.plugin_name() == "roughplastic") b = std::make_shared<RoughPlastic>(p);
        else throw std::runtime_error("Plugin \"" + p.plugin_name() + "\" not found!");
        return new Box<BSDF>{ b }; MIH_CATCH(nullptr)
}
// <bsdf type="twosided">: front (and optionally back) are BSDF handles; the result is a new handle
void *mih_bsdf_create_twosided(void *front, void *back) {
    MIH_TRY
        std::shared_ptr<BSDF> f = front ? ((Box<BSDF> *) front)->p : nullptr, b = back ? ((Box<BSDF> *) back)->p : nullptr;
        return new Box<BSDF>{ std::make_shared<TwoSidedBRDF>(f, b) }; MIH_CATCH(nullptr)
}
float mih_fresnel_diffuse_reflectance(float eta) { return fresnel_diffuse_reflectance(eta); }
void mih_bsdf_destroy(void *b) { delete (Box<BSDF> *) b; }
int mih_bsdf_table(void *b, float *out, uint32_t cap) {          // -> entries of the plugin's float table (0: none)
    const std::vector<float> &t = ((Box<BSDF> *) b)->p->table();
    if (out && cap >= t.size()) std::memcpy(out, t.data(), t.size() * sizeof(float));
    return (int) t.size();
}
void mih_gauss_legendre(int n, float *nodes, float *weights) {
    std::vector<float> x, w; gauss_legendre(n, x, w);
    std::memcpy(nodes, x.data(), x.size() * 4); std::memcpy(weights, w.data(), w.size() * 4);
}
int mih_bsdf_record(void *b, mi_bsdf *out) { *out = ((Box<BSDF> *) b)->p->record(); return 0; }
uint32_t mih_bsdf_flags(void *b) { return ((Box<BSDF> *) b)->p->flags(); }
// out: wo.xyz, pdf, eta, sampled_type(bits), weight.rgb
int mih_bsdf_sample(void *b, const float *wi, float s1, const float *s2, float *out) {
    MIH_TRY
        auto r = ((Box<BSDF> *) b)->p->sample({ wi[0], wi[1], wi[2] }, s1, { s2[0], s2[1] });
        out[0] = r.first.wo[0]; out[1] = r.first.wo[1]; out[2] = r.first.wo[2]; out[3] = r.first.pdf; out[4] = r.first.eta;
        std::memcpy(&out[5], &r.first.sampled_type, 4);
        out[6] = r.second[0]; out[7] = r.second[1]; out[8] = r.second[2];
        return 0; MIH_CATCH(-1)
}
int mih_bsdf_eval_pdf(void *b, const float *wi, const float *wo, float *out) {
    MIH_TRY
        Color3f e = ((Box<BSDF> *) b)->p->eval({ wi[0], wi[1], wi[2] }, { wo[0], wo[1], wo[2] });
        out[0] = e[0]; out[1] = e[1]; out[2] = e[2];
        out[3] = ((Box<BSDF> *) b)->p->pdf({ wi[0], wi[1], wi[2] }, { wo[0], wo[1], wo[2] });
        return 0; MIH_CATCH(-1)
}

void *mih_emitter_create(void *props) {
    MIH_TRY
        const Properties &p = *(Properties *) props;
        if (p.plugin_name() != "area") throw std::runtime_error("Plugin \"" + p.plugin_name() + "\" not found!");
        return new Box<AreaLight>{ std::make_shared<AreaLight>(p) }; MIH_CATCH(nullptr)
}
void mih_emitter_destroy(void *e) { delete (Box<AreaLight> *) e; }

void *mih_mesh_create(const char *name, const float *pos, uint32_t nv, const uint32_t *faces, uint32_t nf, const float *normals,
                      const float *texcoords) {
    MIH_TRY
        std::vector<float> p(pos, pos + 3 * (size_t) nv);
        std::vector<uint32_t> f(faces, faces + 3 * (size_t) nf);
        std::vector<float> n; if (normals) n.assign(normals, normals + 3 * (size_t) nv);
        std::vector<float> tc; if (texcoords) tc.assign(texcoords, texcoords + 2 * (size_t) nv);
        return new Box<Mesh>{ std::make_shared<Mesh>(name ? name : "", std::move(p), std::move(f), std::move(n), std::move(tc)) }; MIH_CATCH(nullptr)
}
void *mih_sphere_create(void *props) {                       // the `sphere` shape plugin (analytic)
    MIH_TRY return new Box<Mesh>{ make_sphere(*(Properties *) props) }; MIH_CATCH(nullptr)
}
void *mih_rectangle_create(void *props) {                    // the `rectangle` shape plugin (analytic)
    MIH_TRY return new Box<Mesh>{ make_rectangle(*(Properties *) props) }; MIH_CATCH(nullptr)
}
// kind 0 = obj, 1 = ply (the `obj` / `ply` shape plugins; props: filename, face_normals, flip_tex_coords, to_world)
void *mih_mesh_load(int kind, void *props) {
    MIH_TRY return new Box<Mesh>{ kind == 0 ? load_obj(*(Properties *) props) : load_ply(*(Properties *) props) }; MIH_CATCH(nullptr)
}
int mih_mesh_recompute_normals(void *m) { MIH_TRY ((Box<Mesh> *) m)->p->recompute_vertex_normals(); return 0; MIH_CATCH(-1) }
void mih_mesh_counts(void *m, uint32_t *nv, uint32_t *nf, int *has_normals) {
    const Mesh &me = *((Box<Mesh> *) m)->p; *nv = me.vertex_count(); *nf = me.face_count();
    *has_normals = (me.has_vertex_normals() ? 1 : 0) | (me.has_vertex_texcoords() ? 2 : 0);     // bit 1: texture coordinates
}
void mih_mesh_copy_texcoords(void *m, float *texcoords) {
    const Mesh &me = *((Box<Mesh> *) m)->p;
    if (texcoords && me.has_vertex_texcoords()) std::memcpy(texcoords, me.vertex_texcoords_buffer().data(), me.vertex_texcoords_buffer().size() * 4);
}
void mih_mesh_copy(void *m, float *pos, uint32_t *faces, float *normals) {
    const Mesh &me = *((Box<Mesh> *) m)->p;
    if (pos) std::memcpy(pos, me.vertex_positions_buffer().data(), me.vertex_positions_buffer().size() * 4);
    if (faces) std::memcpy(faces, me.faces_buffer().data(), me.faces_buffer().size() * 4);
    if (normals && me.has_vertex_normals()) std::memcpy(normals, me.vertex_normals_buffer().data(), me.vertex_normals_buffer().size() * 4);
}
void mih_mesh_destroy(void *m) { delete (Box<Mesh> *) m; }
void mih_mesh_set_bsdf(void *m, void *b) { ((Box<Mesh> *) m)->p->set_bsdf(((Box<BSDF> *) b)->p); }
void mih_mesh_set_emitter(void *m, void *e) { ((Box<Mesh> *) m)->p->set_emitter(((Box<AreaLight> *) e)->p); }

int mih_spectrum_channels() { return spectrum_channels(); }
int mih_set_srgb_model(const char *path) { MIH_TRY set_srgb_model_path(path ? path : ""); return 0; MIH_CATCH(-1) }
void *mih_envmap_create(void *props, uint32_t w, uint32_t h, const float *rgba) {
    MIH_TRY
        auto e = std::make_shared<EnvironmentMapEmitter>(*(Properties *) props);
        e->set_bitmap(w, h, rgba);
        return new Box<EnvironmentMapEmitter>{ e }; MIH_CATCH(nullptr)
}
void mih_envmap_destroy(void *e) { delete (Box<EnvironmentMapEmitter> *) e; }
int mih_scene_add_envmap(void *s, void *e) { MIH_TRY ((Box<Scene> *) s)->p->add_emitter(((Box<EnvironmentMapEmitter> *) e)->p); return 0; MIH_CATCH(-1) }
// load_xml: `params` = "k1=v1\nk2=v2"; returns handles the caller owns (destroy each with its own mih_*_destroy)
int mih_load_xml(const char *xml_or_path, int is_file, const char *params, void **scene, void **sensor, void **film, void **sampler, void **integrator) {
    MIH_TRY
        std::map<std::string, std::string> pm;
        std::string ps = params ? params : ""; size_t b = 0;
        while (b < ps.size()) { size_t e = ps.find('\n', b); if (e == std::string::npos) e = ps.size(); std::string kv = ps.substr(b, e - b); size_t q = kv.find('='); if (q != std::string::npos) pm[kv.substr(0, q)] = kv.substr(q + 1); b = e + 1; }
        LoadedScene ls = is_file ? load_xml_file(xml_or_path, pm) : load_xml_string(xml_or_path, pm);
        *scene = new Box<Scene>{ ls.scene };
        *sensor = ls.sensor ? new Box<PerspectiveCamera>{ ls.sensor } : nullptr;
        *film = ls.sensor ? new Box<Film>{ ls.sensor->film() } : nullptr;
        *sampler = ls.sensor ? new Box<IndependentSampler>{ ls.sensor->sampler() } : nullptr;
        *integrator = new Box<SamplingIntegrator>{ ls.integrator };
        return 0; MIH_CATCH(-1)
}
void *mih_scene_create() { return new Box<Scene>{ std::make_shared<Scene>() }; }
void mih_scene_destroy(void *s) { delete (Box<Scene> *) s; }
int mih_scene_add_shape(void *s, void *m) { MIH_TRY ((Box<Scene> *) s)->p->add_shape(((Box<Mesh> *) m)->p); return 0; MIH_CATCH(-1) }
// device < 0: flatten only (no GPU needed); otherwise upload + build the BVH
int mih_scene_build(void *s, int device, int quality) { MIH_TRY ((Box<Scene> *) s)->p->build(device, quality); return 0; MIH_CATCH(-1) }
const mi_scene_desc *mih_scene_desc(void *s) { return &((Box<Scene> *) s)->p->desc(); }
mi_ctx *mih_scene_ctx(void *s) { return ((Box<Scene> *) s)->p->ctx(); }
int mih_scene_ray_intersect(void *s, const mi_rays_soa *rays, const mi_hits_soa *hits, uint64_t n) {
    MIH_TRY ((Box<Scene> *) s)->p->ray_intersect_preliminary(*rays, *hits, n); return 0; MIH_CATCH(-1)
}
int mih_scene_ray_test(void *s, const mi_rays_soa *rays, float *t, uint64_t n) {
    MIH_TRY ((Box<Scene> *) s)->p->ray_test(*rays, t, n); return 0; MIH_CATCH(-1)
}

void *mih_film_create(void *props) { MIH_TRY return new Box<Film>{ std::make_shared<Film>(*(Properties *) props) }; MIH_CATCH(nullptr) }
void mih_film_destroy(void *f) { delete (Box<Film> *) f; }
int mih_film_set_filter(void *f, const char *name, void *props) {
    MIH_TRY
        std::shared_ptr<ReconstructionFilter> rf;
        Properties def(name);
        const Properties &p = props ? *(Properties *) props : def;
        if (std::string(name) == "gaussian") rf = std::make_shared<GaussianFilter>(p);
        else if (std::string(name) == "box") rf = std::make_shared<BoxFilter>(p);
        else throw std::runtime_error(std::string("Plugin \"") + name + "\" not found!");
        ((Box<Film> *) f)->p->set_reconstruction_filter(rf);
        return 0; MIH_CATCH(-1)
}
const float *mih_film_data(void *f, uint64_t *count) {
    auto &s = ((Box<Film> *) f)->p->storage();
    if (count) *count = s.size();
    return s.data();
}
int mih_film_set_data(void *f, const float *xyzaw, uint64_t count) {
    MIH_TRY
        Film &film = *((Box<Film> *) f)->p;
        film.prepare({ "X", "Y", "Z", "A", "W" });
        if (count != film.storage().size()) Throw("film data size mismatch");
        std::memcpy(film.storage().data(), xyzaw, count * sizeof(float));
        return 0; MIH_CATCH(-1)
}
void mih_film_crop_size(void *f, int *w, int *h) { auto cs = ((Box<Film> *) f)->p->crop_size(); *w = cs[0]; *h = cs[1]; }
static thread_local std::string g_develop_path;
const char *mih_film_develop(void *f, const char *filename) {
    MIH_TRY
        Film &film = *((Box<Film> *) f)->p;
        if (filename) film.set_destination_file(filename);
        g_develop_path = film.develop();
        return g_develop_path.c_str(); MIH_CATCH(nullptr)
}
int mih_film_develop_rgb(void *f, float *out) {
    MIH_TRY auto rgb = ((Box<Film> *) f)->p->bitmap_rgb(); std::memcpy(out, rgb.data(), rgb.size() * 4); return 0; MIH_CATCH(-1)
}
void *mih_sampler_create(void *props) { MIH_TRY return new Box<IndependentSampler>{ std::make_shared<IndependentSampler>(*(Properties *) props) }; MIH_CATCH(nullptr) }
void mih_sampler_destroy(void *s) { delete (Box<IndependentSampler> *) s; }
void mih_sampler_seed(void *s, uint64_t off) { ((Box<IndependentSampler> *) s)->p->seed(off); }
float mih_sampler_next_1d(void *s) { return ((Box<IndependentSampler> *) s)->p->next_1d(); }

void *mih_sensor_create(void *props, void *film, void *sampler) {
    MIH_TRY
        const Properties &p = *(Properties *) props;
        if (p.plugin_name() != "perspective") throw std::runtime_error("Plugin \"" + p.plugin_name() + "\" not found!");
        return new Box<PerspectiveCamera>{ std::make_shared<PerspectiveCamera>(
            p, film ? ((Box<Film> *) film)->p : nullptr, sampler ? ((Box<IndependentSampler> *) sampler)->p : nullptr) }; MIH_CATCH(nullptr)
}
void mih_sensor_destroy(void *s) { delete (Box<PerspectiveCamera> *) s; }
int mih_sensor_sample_ray(void *s, float x, float y, float *out8) {
    MIH_TRY
        Ray3f r = ((Box<PerspectiveCamera> *) s)->p->sample_ray({ x, y });
        out8[0] = r.o[0]; out8[1] = r.o[1]; out8[2] = r.o[2]; out8[3] = r.d[0]; out8[4] = r.d[1]; out8[5] = r.d[2];
        out8[6] = r.mint; out8[7] = r.maxt; return 0; MIH_CATCH(-1)
}
float mih_sensor_x_fov(void *s) { return ((Box<PerspectiveCamera> *) s)->p->x_fov(); }

void *mih_integrator_create(void *props) {
    MIH_TRY
        const Properties &p = *(Properties *) props;
        return new Box<SamplingIntegrator>{ make_integrator(p) }; MIH_CATCH(nullptr)
}
void *mih_integrator_create_moment(void *props, void *nested, const char *name) {
    MIH_TRY return new Box<SamplingIntegrator>{ std::make_shared<MomentIntegrator>(*(Properties *) props, ((Box<SamplingIntegrator> *) nested)->p, name ? name : "integrator") }; MIH_CATCH(nullptr)
}
int mih_integrator_aov_names(void *i, char *buf, uint32_t cap) {       // comma-separated
    MIH_TRY std::string s; for (const std::string &n : ((Box<SamplingIntegrator> *) i)->p->aov_names()) s += (s.empty() ? "" : ",") + n;
        if (s.size() + 1 > cap) return -1; std::memcpy(buf, s.c_str(), s.size() + 1); return (int) s.size(); MIH_CATCH(-1)
}
void mih_integrator_destroy(void *i) { delete (Box<SamplingIntegrator> *) i; }
void mih_integrator_set_shard(void *i, uint32_t rank, uint32_t world) { ((Box<SamplingIntegrator> *) i)->p->set_shard(rank, world); }
void mih_integrator_set_profile(void *i, int on) { ((Box<SamplingIntegrator> *) i)->p->set_profile(on != 0); }
void mih_integrator_set_plan(void *i, int plan) { ((Box<SamplingIntegrator> *) i)->p->set_plan(plan); }
void mih_integrator_cancel(void *i) { ((Box<SamplingIntegrator> *) i)->p->cancel(); }
// 1 = finished, 0 = cancelled / timed out, -1 = error
int mih_integrator_render(void *i, void *scene, void *sensor) {
    MIH_TRY return ((Box<SamplingIntegrator> *) i)->p->render(((Box<Scene> *) scene)->p.get(), ((Box<PerspectiveCamera> *) sensor)->p.get()) ? 1 : 0; MIH_CATCH(-1)
}
int mih_integrator_counters(void *i, mi_counters *out) { *out = ((Box<SamplingIntegrator> *) i)->p->counters(); return 0; }
// Host-side job description (no GPU needed). block_ids / tiles must hold `capacity` entries.
int mih_make_render_cfg(void *i, void *sensor, mi_render_cfg *cfg, uint32_t *block_ids, uint32_t *tiles, uint32_t capacity, uint32_t n_threads) {
    MIH_TRY
        std::vector<uint32_t> ids, tl;
        ((Box<SamplingIntegrator> *) i)->p->make_render_cfg(((Box<PerspectiveCamera> *) sensor)->p.get(), *cfg, ids, tl, n_threads);
        if (ids.size() > capacity) throw std::runtime_error("mih_make_render_cfg: capacity too small");
        std::memcpy(block_ids, ids.data(), ids.size() * 4);
        std::memcpy(tiles, tl.data(), tl.size() * 4);
        cfg->block_ids = block_ids; cfg->tile_list = tl.empty() ? nullptr : tiles;
        return 0; MIH_CATCH(-1)
}
int mih_make_render_cfg_pass(void *i, void *sensor, mi_render_cfg *cfg, uint32_t *block_ids, uint32_t *tiles, uint32_t capacity, uint32_t n_threads, uint32_t pass) {
    MIH_TRY
        std::vector<uint32_t> ids, tl;
        ((Box<SamplingIntegrator> *) i)->p->make_render_cfg(((Box<PerspectiveCamera> *) sensor)->p.get(), *cfg, ids, tl, n_threads, pass);
        if (ids.size() > capacity) throw std::runtime_error("mih_make_render_cfg: capacity too small");
        std::memcpy(block_ids, ids.data(), ids.size() * 4);
        std::memcpy(tiles, tl.data(), tl.size() * 4);
        cfg->block_ids = block_ids; cfg->tile_list = tl.empty() ? nullptr : tiles;
        return 0; MIH_CATCH(-1)
}
int mih_integrator_pass_count(void *i, void *sensor) {
    MIH_TRY
        return (int) ((Box<SamplingIntegrator> *) i)->p->pass_count(((Box<PerspectiveCamera> *) sensor)->p.get()); MIH_CATCH(-1)
}
// Spiral walk (test hook): writes offset.xy, size.xy, block_id per block; returns block count
int mih_spiral(int w, int h, int off_x, int off_y, int block_size, int32_t *out5, int capacity) {
    Spiral sp({ w, h }, { off_x, off_y }, (size_t) block_size, 1);
    int n = (int) sp.block_count();
    for (int i = 0; i < n && i < capacity; ++i) {
        Spiral::Block b = sp.next_block();
        out5[i * 5] = b.offset[0]; out5[i * 5 + 1] = b.offset[1]; out5[i * 5 + 2] = b.size[0]; out5[i * 5 + 3] = b.size[1];
        out5[i * 5 + 4] = (int32_t) b.block_id;
    }
    return n;
}

} // extern "C"
