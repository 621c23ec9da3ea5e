// libmiwave_host: Film (storage, develop, OpenEXR / PFM writers), Spiral, IndependentSampler, PerspectiveCamera.
// Part of the single translation unit host/miwave_host.cpp (included there, in this order).
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
// The visiting order is a TABLE here (all the device ever sees of the spiral is the block -> id table of mi_render_cfg): the
// walk of spiral.cpp:27-72 — from the centre block, legs Right, Down, Left, Up of 1, 1, 2, 2, 3, 3, ... blocks, positions outside
// the grid skipped — is laid out once per (grid size) by spiral_cells(), and next_block() reads entry after entry.
// Pinned against the reference's own vectors by tests/test_oracle_kat.py (test_spiral.py:49-80) and against the oracle's
// independently written walk (oracle/miw_oracle.cpp).
static std::vector<std::array<int, 2>> spiral_cells(int nx, int ny) {
    const size_t total = (size_t) std::max(nx, 0) * (size_t) std::max(ny, 0);
    std::vector<std::array<int, 2>> cells;
    cells.reserve(total);
    if (total == 0) return cells;
    static const int step_x[4] = { 1, 0, -1, 0 }, step_y[4] = { 0, 1, 0, -1 };     // Right, Down, Left, Up
    int x = nx / 2, y = ny / 2;
    cells.push_back({ x, y });
    for (int leg = 0; cells.size() < total; ++leg) {
        const int len = leg / 2 + 1, dir = leg & 3;
        for (int s = 0; s < len && cells.size() < total; ++s) {
            x += step_x[dir]; y += step_y[dir];
            if (x >= 0 && y >= 0 && x < nx && y < ny) cells.push_back({ x, y });
        }
    }
    return cells;
}

Spiral::Spiral(std::array<int, 2> size, std::array<int, 2> offset, size_t block_size, size_t passes)
    : m_block_size(block_size), m_size(size), m_offset(offset), m_passes_left(passes) {
    const int bs = (int) block_size;
    m_cells = spiral_cells((size[0] + bs - 1) / bs, (size[1] + bs - 1) / bs);     // ceil(size / block_size) blocks per axis
    m_cursor = 0;
}
void Spiral::reset() { m_cursor = 0; }
Spiral::Block Spiral::next_block() {
    if (m_cursor == m_cells.size()) {                          // this pass is over: the next one restarts at the centre (spiral.cpp:31-38)
        if (m_passes_left <= 1) return { { 0, 0 }, { 0, 0 }, (size_t) -1 };
        --m_passes_left; m_cursor = 0;
    }
    const int bs = (int) m_block_size;
    const std::array<int, 2> cell = m_cells[m_cursor];
    Block b;
    b.block_id = m_cursor + (m_passes_left - 1) * m_cells.size();                   // spiral.cpp:41: later passes carry smaller ids
    for (int a = 0; a < 2; ++a) {
        b.size[a] = std::min(bs, m_size[a] - cell[a] * bs);    // edge blocks are clipped to the crop window
        b.offset[a] = cell[a] * bs + m_offset[a];
    }
    ++m_cursor;
    return b;
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
