// libmiwave_host: BSDF plugins, the bitmap texture + PFM reader, quadrature, area light, environment map.
// Part of the single translation unit host/miwave_host.cpp (included there, in this order).
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
// the reference's (ctx, si, ...) signatures, bsdf.h:328-394
static void require_full_context(const BSDFContext &ctx) {
    if (!ctx.is_full())
        Throw("BSDFContext: only the full context (radiance transport, all components and lobes enabled) is implemented by this layer");
}
std::pair<BSDFSample3f, Color3f> BSDF::sample(const BSDFContext &ctx, const SurfaceInteraction3f &si, float sample1, const std::array<float, 2> &sample2) const {
    require_full_context(ctx);
    return sample(si.wi, sample1, sample2);
}
Color3f BSDF::eval(const BSDFContext &ctx, const SurfaceInteraction3f &si, const Vector3f &wo) const { require_full_context(ctx); return eval(si.wi, wo); }
float BSDF::pdf(const BSDFContext &ctx, const SurfaceInteraction3f &si, const Vector3f &wo) const { require_full_context(ctx); return pdf(si.wi, wo); }
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
#if MIW_SPECTRAL
    // envmap.cpp:86-116, the spectral branch: every texel becomes the coefficients of the sRGB upsampling model of its colour
    // scaled to a highest component of 50 % + that scale; the sampling density is taken from the colour before it is replaced.
    m_density.resize((size_t) width * height);
    for (uint32_t y = 0; y < height; ++y) {
        const float sin_theta = std::sin((float) y / (float) (height - 1) * 3.14159265358979323846f);   // :87-88 (float pi)
        for (uint32_t x = 0; x < width; ++x) {
            float *p = m_data.data() + 4 * ((size_t) y * width + x);
            const float lum = p[0] * 0.212671f + p[1] * 0.715160f + p[2] * 0.072169f;                    // mitsuba::luminance, spectrum.h
            const float scale = std::max(p[0], std::max(p[1], p[2])) * 2.f;                               // :106
            const float r = 1.f / std::max(1e-8f, scale);                                                 // rgb / scalar = rgb * rcp(scalar)
            const auto cf = srgb_model_fetch(Color3f{ p[0] * r, p[1] * r, p[2] * r });                    // :107-108
            m_density[(size_t) y * width + x] = lum * sin_theta;                                          // :113
            p[0] = cf[0]; p[1] = cf[1]; p[2] = cf[2]; p[3] = scale;
        }
    }
#endif
}
