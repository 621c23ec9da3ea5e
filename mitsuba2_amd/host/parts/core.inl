// libmiwave_host: Transform4f, Properties, the sRGB upsampling model, reconstruction filters.
// Part of the single translation unit host/miwave_host.cpp (included there, in this order).
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
TentFilter::TentFilter(const Properties &) { m_radius = 1.f; m_inv_radius = 1.f / m_radius; init_discretization(); }
float TentFilter::eval(float x) const { return std::max(0.f, 1.f - std::abs(x * m_inv_radius)); }
MitchellNetravaliFilter::MitchellNetravaliFilter(const Properties &props) {
    m_radius = 2.f; m_b = props.float_("B", 1.f / 3.f); m_c = props.float_("C", 1.f / 3.f);
    init_discretization();
}
MitchellNetravaliFilter::MitchellNetravaliFilter(float b, float c) { m_radius = 2.f; m_b = b; m_c = c; init_discretization(); }
float MitchellNetravaliFilter::eval(float x_) const {           // mitchell.cpp:38-51, catmullrom.cpp:29-44
    const float x = std::abs(x_), x2 = x * x, x3 = x2 * x, B = m_b, C = m_c;
    const float inner = (12.f - 9.f * B - 6.f * C) * x3 + (-18.f + 12.f * B + 6.f * C) * x2 + (6.f - 2.f * B),
                outer = (-B - 6.f * C) * x3 + (6.f * B + 30.f * C) * x2 + (-12.f * B - 48.f * C) * x + (8.f * B + 24.f * C);
    const float result = (1.f / 6.f) * (x < 1.f ? inner : outer);
    return x < 2.f ? result : 0.f;
}
LanczosSincFilter::LanczosSincFilter(const Properties &props) { m_radius = (float) props.int_("lobes", 3); init_discretization(); }
float LanczosSincFilter::eval(float x_) const {                 // lanczos.cpp:40-50
    const float x = std::abs(x_), x1 = MIW_PI * x, x2 = x1 / m_radius;
    if (x < MIW_EPSILON) return 1.f;
    return x > m_radius ? 0.f : (std::sin(x1) * std::sin(x2)) / (x1 * x2);
}
std::shared_ptr<ReconstructionFilter> make_rfilter(const Properties &props) {
    const std::string &n = props.plugin_name();
    if (n == "gaussian") return std::make_shared<GaussianFilter>(props);
    if (n == "box") return std::make_shared<BoxFilter>(props);
    if (n == "tent") return std::make_shared<TentFilter>(props);
    if (n == "mitchell") return std::make_shared<MitchellNetravaliFilter>(props);
    if (n == "catmullrom") return std::make_shared<MitchellNetravaliFilter>(0.f, .5f);
    if (n == "lanczos") return std::make_shared<LanczosSincFilter>(props);
    Throw("Plugin \"" + n + "\" not found!");
}
