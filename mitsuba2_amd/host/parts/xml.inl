// libmiwave_host: the XML scene front-end subset.
// Part of the single translation unit host/miwave_host.cpp (included there, in this order).
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
    // one child of <scene>; <include> re-enters with the children of the included file's <scene> (xml.cpp:653-700)
    std::function<void(const XmlNode &, int)> handle = [&](const XmlNode &n, int depth) {
        if (n.tag == "include") {
            if (depth >= 15) Throw("Exceeded <include> recursion limit of 15");               // MTS_XML_INCLUDE_MAX_RECURSION
            const std::string file = resolve(cx, cx.get(n, "filename"));
            FILE *probe = std::fopen(file.c_str(), "rb");
            if (!probe) Throw("Error while loading XML: included file \"" + file + "\" not found");
            std::fclose(probe);
            const std::string text = read_file(file, "XML");
            XmlParser nested(text);
            XmlNode inc = nested.element();
            if (inc.tag != "scene") Throw("Error while loading XML: the included file \"" + file + "\" must have a <scene> root in this layer");
            for (const XmlNode &c : inc.children) handle(c, depth + 1);
        }
        else if (n.tag == "alias") {                           // xml.cpp:594-612
            const std::string src = cx.get(n, "id"), dst = cx.get(n, "as");
            if (cx.bsdfs.count(dst) || cx.textures.count(dst)) Throw("Error while loading XML: \"alias\" has duplicate id \"" + dst + "\"");
            if (cx.bsdfs.count(src)) cx.bsdfs[dst] = cx.bsdfs[src];
            else if (cx.textures.count(src)) cx.textures[dst] = cx.textures[src];
            else Throw("Error while loading XML: referenced id \"" + src + "\" not found");
        }
        else if (n.tag == "default") { std::string k = cx.get(n, "name"); if (!cx.params.count(k)) cx.params[k] = cx.get(n, "value"); }
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
                        film->set_reconstruction_filter(make_rfilter(rp));
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
    };
    for (const XmlNode &n : root.children) handle(n, 0);
    if (!out.integrator) out.integrator = std::make_shared<PathIntegrator>(Properties("path"));
    return out;
}
LoadedScene load_xml_file(const std::string &path, const std::map<std::string, std::string> &params) {
    std::string data = read_file(path, "XML");
    size_t slash = path.find_last_of("/\\");
    return load_xml_string(data, params, slash == std::string::npos ? "." : path.substr(0, slash));
}

} // namespace miwave
