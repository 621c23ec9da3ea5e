// libmiwave_host: Mesh, obj / ply loaders, analytic rectangle / sphere, Scene (flattening into mi_scene_desc).
// Part of the single translation unit host/miwave_host.cpp (included there, in this order).
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
std::array<float, 6> Mesh::bbox() const {
    const float inf = std::numeric_limits<float>::infinity();
    std::array<float, 6> b{ inf, inf, inf, -inf, -inf, -inf };
    if (m_sphere) {                                            // sphere.cpp:133-138
        for (int k = 0; k < 3; ++k) { b[k] = m_sphere_rec.center[k] - m_sphere_rec.radius; b[3 + k] = m_sphere_rec.center[k] + m_sphere_rec.radius; }
        return b;
    }
    for (size_t i = 0; i < m_positions.size(); i += 3)
        for (int k = 0; k < 3; ++k) { b[k] = std::min(b[k], m_positions[i + k]); b[3 + k] = std::max(b[3 + k], m_positions[i + k]); }
    return b;
}
float Mesh::surface_area() const {
    if (m_sphere) return 4.f * MIW_PI * m_sphere_rec.radius * m_sphere_rec.radius;              // sphere.cpp:140-142
    if (m_rectangle) {                                         // rectangle.cpp:88-96: |dp_du x dp_dv|
        const float *m = m_rect_to_world.m;
        miw::V3 du = miw::v3(2.f * m[0], 2.f * m[1], 2.f * m[2]), dv = miw::v3(2.f * m[4], 2.f * m[5], 2.f * m[6]);
        return miw::norm(miw::cross(du, dv));
    }
    double sum = 0.0;
    for (size_t f = 0; f < m_faces.size(); f += 3) {
        auto P = [&](uint32_t i) { return miw::v3(m_positions[3 * i], m_positions[3 * i + 1], m_positions[3 * i + 2]); };
        const miw::V3 p0 = P(m_faces[f]), p1 = P(m_faces[f + 1]), p2 = P(m_faces[f + 2]);
        sum += (double) (.5f * miw::norm(miw::cross(p1 - p0, p2 - p0)));                           // face_area, mesh.h:127-134
    }
    return (float) sum;
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
Scene::~Scene() {
    // the emitters go back to "part of no scene" (they may outlive it: shared_ptr), so that another scene can take them
    for (const Emitter *e : m_emitter_objs) if (e->m_scene == this) { Emitter *w = const_cast<Emitter *>(e); w->m_scene = nullptr; w->m_index = -1; }
    for (mi_ctx *c : m_replicas) mi_destroy(c);
    if (m_ctx) mi_destroy(m_ctx);
}
std::array<float, 6> Scene::bbox() const {
    const float inf = std::numeric_limits<float>::infinity();
    std::array<float, 6> b{ inf, inf, inf, -inf, -inf, -inf };
    for (const auto &m : m_shapes) { auto s = m->bbox(); for (int k = 0; k < 3; ++k) { b[k] = std::min(b[k], s[k]); b[3 + k] = std::max(b[3 + k], s[3 + k]); } }
    return b;
}
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
    // a single-device (re)build leaves ONE context: replicas of an earlier build(devices) would keep the old scene and BVH, and
    // render() would deal spiral blocks to them. build(devices) clears them itself before it calls this, then re-creates them.
    for (mi_ctx *c : m_replicas) mi_destroy(c);
    m_replicas.clear();
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
        m_env_rec.density = m_env->density().empty() ? nullptr : m_env->density().data();
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
    // Scene::emitters(): the area lights in shape order with the environment map at its slot; every emitter learns
    // its scene and index so that Endpoint::eval / sample_direction / pdf_direction can run on the scene's device
    m_emitter_objs.clear();
    for (auto &m : m_shapes) if (m->emitter()) m_emitter_objs.push_back(m->emitter().get());
    if (m_env) m_emitter_objs.insert(m_emitter_objs.begin() + std::min<size_t>(m_env_rec.emitter_index, m_emitter_objs.size()), m_env.get());
    for (size_t i = 0; i < m_emitter_objs.size(); ++i) {
        Emitter *e = const_cast<Emitter *>(m_emitter_objs[i]);
        // an emitter answers eval / sample_direction / pdf_direction through the ONE scene it is part of (the reference's
        // emitters hold their shape, never two scenes): sharing it would leave the first scene querying the second's tables
        if (e->m_scene && e->m_scene != this)
            Throw("Scene: emitter (or the mesh carrying it) is already part of another scene; build it from its own objects");
        e->m_scene = this; e->m_index = (int32_t) i; e->m_is_env = m_env && e == m_env.get();
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
void Scene::build(const std::vector<int> &devices, int bvh_quality) {
    if (devices.empty()) Throw("Scene::build: empty device list");
    for (int d : devices) if (d < 0) Throw("Scene::build: a multi-GPU build needs device indices >= 0");
    for (mi_ctx *c : m_replicas) mi_destroy(c);                // a rebuild starts from the primary alone
    m_replicas.clear();
    build(devices[0], bvh_quality);                            // flatten + the primary context
    // the replicas: created here (mi_create is cheap), uploaded and built each on its own thread — the contexts share nothing
    std::vector<std::string> errors(devices.size());
    for (size_t i = 1; i < devices.size(); ++i) {
        mi_ctx *c = nullptr;
        if (mi_create(devices[i], &c) != MI_OK) Throw(std::string("mi_create failed: ") + mi_last_error(nullptr));
        m_replicas.push_back(c);
    }
    std::vector<std::thread> pool;
    for (size_t i = 1; i < devices.size(); ++i)
        pool.emplace_back([this, i, bvh_quality, &errors]() {
            mi_ctx *c = m_replicas[i - 1];
            if (mi_scene_upload(c, &m_desc) != MI_OK) { errors[i] = std::string("mi_scene_upload: ") + mi_last_error(c); return; }
            if (mi_bvh_build(c, bvh_quality) != MI_OK) errors[i] = std::string("mi_bvh_build: ") + mi_last_error(c);
        });
    for (auto &t : pool) t.join();
    for (const std::string &e : errors) if (!e.empty()) Throw(e);
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
// ---- Scene::ray_intersect / sample_emitter_direction / pdf_emitter_direction and the Endpoint methods ----
static SurfaceInteraction3f si_from_record(const mi_surface_interaction &r, const Scene *scene) {
    SurfaceInteraction3f si;
    si.t = r.t;
    for (int k = 0; k < 3; ++k) { si.p[k] = r.p[k]; si.n[k] = r.n[k]; si.sh_frame.s[k] = r.sh_s[k]; si.sh_frame.t[k] = r.sh_t[k]; si.sh_frame.n[k] = r.sh_n[k]; si.wi[k] = r.wi[k]; }
    si.uv = { r.uv[0], r.uv[1] };
    si.prim_index = r.prim_index; si.shape_index = r.shape_index; si.emitter_index = r.emitter_index;
    si.shape = r.shape_index < scene->shapes().size() ? scene->shapes()[r.shape_index].get() : nullptr;
    return si;
}
static mi_surface_interaction record_from_si(const SurfaceInteraction3f &si) {
    mi_surface_interaction r{};
    r.t = si.t;
    for (int k = 0; k < 3; ++k) { r.p[k] = si.p[k]; r.n[k] = si.n[k]; r.sh_s[k] = si.sh_frame.s[k]; r.sh_t[k] = si.sh_frame.t[k]; r.sh_n[k] = si.sh_frame.n[k]; r.wi[k] = si.wi[k]; }
    r.uv[0] = si.uv[0]; r.uv[1] = si.uv[1];
    r.prim_index = si.prim_index; r.shape_index = si.shape_index; r.emitter_index = si.emitter_index;
    return r;
}
static DirectionSample3f ds_from_record(const mi_direction_sample &r, const Scene *scene) {
    DirectionSample3f ds;
    for (int k = 0; k < 3; ++k) { ds.p[k] = r.p[k]; ds.n[k] = r.n[k]; ds.d[k] = r.d[k]; }
    ds.dist = r.dist; ds.pdf = r.pdf; ds.emitter_index = r.emitter_index;
    ds.object = (r.emitter_index >= 0 && (size_t) r.emitter_index < scene->emitters().size()) ? scene->emitters()[r.emitter_index] : nullptr;
    return ds;
}
static mi_direction_sample record_from_ds(const DirectionSample3f &ds) {
    mi_direction_sample r{};
    for (int k = 0; k < 3; ++k) { r.p[k] = ds.p[k]; r.n[k] = ds.n[k]; r.d[k] = ds.d[k]; }
    r.dist = ds.dist; r.pdf = ds.pdf; r.emitter_index = ds.object ? ds.object->index() : ds.emitter_index;
    return r;
}
void Scene::ray_intersect(const mi_rays_soa &rays, mi_surface_interaction *si, uint64_t n) const {
    if (!m_ctx) Throw("Scene: not built on a device");
    if (mi_ray_intersect(m_ctx, &rays, si, n) != MI_OK) Throw(std::string("mi_ray_intersect: ") + mi_last_error(m_ctx));
}
SurfaceInteraction3f Scene::ray_intersect(const Ray3f &r) const {
    mi_rays_soa rays{ &r.o[0], &r.o[1], &r.o[2], &r.d[0], &r.d[1], &r.d[2], &r.mint, &r.maxt };
    mi_surface_interaction rec{};
    ray_intersect(rays, &rec, 1);
    return si_from_record(rec, this);
}
static std::pair<DirectionSample3f, Spectrum> sample_direction_on_device(const Scene *scene, int32_t emitter, const Interaction3f &ref,
                                                                          const std::array<float, 2> &sample, bool test_visibility) {
    if (!scene || !scene->ctx()) Throw("sample_direction: the scene is not built on a device");
    mi_direction_sample rec{}; Spectrum value{};
    if (mi_sample_emitter_direction(scene->ctx(), emitter, ref.p.data(), sample.data(), ref.wavelengths.data(), test_visibility ? 1 : 0,
                                    &rec, value.data(), 1) != MI_OK)
        Throw(std::string("mi_sample_emitter_direction: ") + mi_last_error(scene->ctx()));
    return { ds_from_record(rec, scene), value };
}
std::pair<DirectionSample3f, Spectrum> Scene::sample_emitter_direction(const Interaction3f &ref, const std::array<float, 2> &sample, bool test_visibility) const {
    return sample_direction_on_device(this, -1, ref, sample, test_visibility);
}
float Scene::pdf_emitter_direction(const Interaction3f &ref, const DirectionSample3f &ds) const {
    if (!m_ctx) Throw("Scene: not built on a device");
    const mi_direction_sample rec = record_from_ds(ds);
    float pdf = 0.f;
    if (mi_pdf_emitter_direction(m_ctx, -1, ref.p.data(), &rec, &pdf, 1) != MI_OK) Throw(std::string("mi_pdf_emitter_direction: ") + mi_last_error(m_ctx));
    return pdf;
}
std::pair<DirectionSample3f, Spectrum> Emitter::sample_direction(const Interaction3f &it, const std::array<float, 2> &sample) const {
    if (m_index < 0) Throw("Emitter: not part of a built scene");
    return sample_direction_on_device(m_scene, m_index, it, sample, false);     // Endpoint::sample_direction knows no occluders
}
float Emitter::pdf_direction(const Interaction3f &it, const DirectionSample3f &ds) const {
    if (m_index < 0 || !m_scene->ctx()) Throw("Emitter: not part of a scene built on a device");
    const mi_direction_sample rec = record_from_ds(ds);
    float pdf = 0.f;
    if (mi_pdf_emitter_direction(m_scene->ctx(), m_index, it.p.data(), &rec, &pdf, 1) != MI_OK) Throw(std::string("mi_pdf_emitter_direction: ") + mi_last_error(m_scene->ctx()));
    return pdf;
}
Spectrum Emitter::eval(const SurfaceInteraction3f &si) const {
    if (m_index < 0 || !m_scene->ctx()) Throw("Emitter: not part of a scene built on a device");
    mi_surface_interaction rec = record_from_si(si);
    rec.emitter_index = m_index;                               // `this` is the emitter being evaluated
    Spectrum value{};
    if (mi_emitter_eval(m_scene->ctx(), &rec, si.wavelengths.data(), value.data(), 1) != MI_OK) Throw(std::string("mi_emitter_eval: ") + mi_last_error(m_scene->ctx()));
    return value;
}
const Emitter *SurfaceInteraction3f::emitter(const Scene *scene) const {
    return (scene && emitter_index >= 0 && (size_t) emitter_index < scene->emitters().size()) ? scene->emitters()[emitter_index] : nullptr;
}
const BSDF *SurfaceInteraction3f::bsdf() const { return shape ? shape->bsdf().get() : nullptr; }
Vector3f SurfaceInteraction3f::to_world(const Vector3f &v) const {                  // frame.h:25-37: s * v.x + t * v.y + n * v.z (fma chain)
    Vector3f r;
    for (int k = 0; k < 3; ++k) r[k] = std::fma(sh_frame.n[k], v[2], std::fma(sh_frame.t[k], v[1], sh_frame.s[k] * v[0]));
    return r;
}
Vector3f SurfaceInteraction3f::to_local(const Vector3f &v) const {
    auto dot3 = [](const Vector3f &a, const Vector3f &b) { return std::fma(a[2], b[2], std::fma(a[1], b[1], a[0] * b[0])); };
    return { dot3(v, sh_frame.s), dot3(v, sh_frame.t), dot3(v, sh_frame.n) };
}

bool Scene::ray_test(const Ray3f &r) const {
    mi_rays_soa rays{ &r.o[0], &r.o[1], &r.o[2], &r.d[0], &r.d[1], &r.d[2], &r.mint, &r.maxt };
    float t;
    ray_test(rays, &t, 1);
    return t != std::numeric_limits<float>::infinity();
}
