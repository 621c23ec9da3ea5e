// libmiwave_host: the flat mih_* facade ctypes binds.
// Part of the single translation unit host/miwave_host.cpp (included there, in this order).
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
void mih_mesh_bbox_area(void *m, float *out7) {               // min xyz, max xyz, surface area
    const Mesh &me = *((Box<Mesh> *) m)->p; auto b = me.bbox(); for (int k = 0; k < 6; ++k) out7[k] = b[k]; out7[6] = me.surface_area();
}
void mih_scene_bbox(void *s, float *out6) { auto b = ((Box<Scene> *) s)->p->bbox(); for (int k = 0; k < 6; ++k) out6[k] = b[k]; }
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
// Scene::build(devices): one context per entry (an index may repeat), uploads + BVH builds in parallel; the integrators then render
// a frame over all of them with one call (mih_integrator_render / mih_render_multi)
int mih_scene_build_multi(void *s, const int *devices, int n, int quality) {
    MIH_TRY ((Box<Scene> *) s)->p->build(std::vector<int>(devices, devices + (n > 0 ? n : 0)), quality); return 0; MIH_CATCH(-1)
}
int mih_scene_device_count(void *s) { return (int) ((Box<Scene> *) s)->p->device_count(); }
const mi_scene_desc *mih_scene_desc(void *s) { return &((Box<Scene> *) s)->p->desc(); }
mi_ctx *mih_scene_ctx(void *s) { return ((Box<Scene> *) s)->p->ctx(); }
int mih_scene_ray_intersect(void *s, const mi_rays_soa *rays, const mi_hits_soa *hits, uint64_t n) {
    MIH_TRY ((Box<Scene> *) s)->p->ray_intersect_preliminary(*rays, *hits, n); return 0; MIH_CATCH(-1)
}
int mih_scene_ray_test(void *s, const mi_rays_soa *rays, float *t, uint64_t n) {
    MIH_TRY ((Box<Scene> *) s)->p->ray_test(*rays, t, n); return 0; MIH_CATCH(-1)
}
// Scene::ray_intersect -> SurfaceInteraction3f records (batch), and the same through the one-ray C++ overload
int mih_scene_ray_intersect_si(void *s, const mi_rays_soa *rays, mi_surface_interaction *si, uint64_t n) {
    MIH_TRY ((Box<Scene> *) s)->p->ray_intersect(*rays, si, n); return 0; MIH_CATCH(-1)
}
int mih_scene_ray_intersect_one(void *s, const float *ray8, mi_surface_interaction *out, int *has_bsdf, int *emitter_index) {
    MIH_TRY
        const Scene *sc = ((Box<Scene> *) s)->p.get();
        Ray3f r; r.o = { ray8[0], ray8[1], ray8[2] }; r.d = { ray8[3], ray8[4], ray8[5] }; r.mint = ray8[6]; r.maxt = ray8[7];
        SurfaceInteraction3f si = sc->ray_intersect(r);
        *out = record_from_si(si);
        *has_bsdf = si.bsdf() ? 1 : 0;
        const Emitter *e = si.emitter(sc);
        *emitter_index = e ? e->index() : -1;
        return si.is_valid() ? 1 : 0; MIH_CATCH(-1)
}
// Scene::sample_emitter_direction (emitter < 0) / Endpoint::sample_direction of Scene::emitters()[emitter], one query
int mih_scene_sample_emitter_direction(void *s, int emitter, const float *ref_p, const float *wavelengths, const float *sample2,
                                       int test_visibility, mi_direction_sample *ds, float *spec) {
    MIH_TRY
        const Scene *sc = ((Box<Scene> *) s)->p.get();
        Interaction3f it; it.p = { ref_p[0], ref_p[1], ref_p[2] };
        if (wavelengths) it.wavelengths = { wavelengths[0], wavelengths[1], wavelengths[2], wavelengths[3] };
        if (emitter >= (int) sc->emitters().size()) throw std::runtime_error("emitter index out of range");
        auto r = emitter < 0 ? sc->sample_emitter_direction(it, { sample2[0], sample2[1] }, test_visibility != 0)
                             : sc->emitters()[emitter]->sample_direction(it, { sample2[0], sample2[1] });
        *ds = record_from_ds(r.first);
        for (size_t k = 0; k < r.second.size(); ++k) spec[k] = r.second[k];
        return 0; MIH_CATCH(-1)
}
int mih_scene_pdf_emitter_direction(void *s, int emitter, const float *ref_p, const mi_direction_sample *ds, float *pdf) {
    MIH_TRY
        const Scene *sc = ((Box<Scene> *) s)->p.get();
        Interaction3f it; it.p = { ref_p[0], ref_p[1], ref_p[2] };
        if (emitter >= (int) sc->emitters().size()) throw std::runtime_error("emitter index out of range");
        DirectionSample3f d = ds_from_record(*ds, sc);
        *pdf = emitter < 0 ? sc->pdf_emitter_direction(it, d) : sc->emitters()[emitter]->pdf_direction(it, d);
        return 0; MIH_CATCH(-1)
}
// si.emitter(scene)->eval(si) (emitter < 0) or Scene::emitters()[emitter]->eval(si)
int mih_scene_emitter_eval(void *s, int emitter, const mi_surface_interaction *si, const float *wavelengths, float *spec) {
    MIH_TRY
        const Scene *sc = ((Box<Scene> *) s)->p.get();
        SurfaceInteraction3f x = si_from_record(*si, sc);
        if (wavelengths) x.wavelengths = { wavelengths[0], wavelengths[1], wavelengths[2], wavelengths[3] };
        const Emitter *e = emitter < 0 ? x.emitter(sc) : (emitter < (int) sc->emitters().size() ? sc->emitters()[emitter] : nullptr);
        Spectrum v{};
        if (e) v = e->eval(x);
        for (size_t k = 0; k < v.size(); ++k) spec[k] = v[k];
        return e ? 1 : 0; MIH_CATCH(-1)
}
int mih_scene_emitter_count(void *s) { return (int) ((Box<Scene> *) s)->p->emitters().size(); }
// BSDF::sample / eval / pdf through the reference's (ctx, si, ...) signatures; -1: the context is refused
int mih_bsdf_sample_ctx(void *b, uint32_t mode, uint32_t type_mask, uint32_t component, const float *wi, float s1, const float *s2, float *out9) {
    MIH_TRY
        BSDFContext ctx; ctx.mode = (TransportMode) mode; ctx.type_mask = type_mask; ctx.component = component;
        SurfaceInteraction3f si; si.wi = { wi[0], wi[1], wi[2] };
        auto r = ((Box<BSDF> *) b)->p->sample(ctx, si, s1, { s2[0], s2[1] });
        out9[0] = r.first.wo[0]; out9[1] = r.first.wo[1]; out9[2] = r.first.wo[2]; out9[3] = r.first.pdf; out9[4] = r.first.eta;
        std::memcpy(out9 + 5, &r.first.sampled_type, 4);
        out9[6] = r.second[0]; out9[7] = r.second[1]; out9[8] = r.second[2];
        return 0; MIH_CATCH(-1)
}

void *mih_film_create(void *props) { MIH_TRY return new Box<Film>{ std::make_shared<Film>(*(Properties *) props) }; MIH_CATCH(nullptr) }
void mih_film_destroy(void *f) { delete (Box<Film> *) f; }
int mih_film_set_filter(void *f, const char *name, void *props) {
    MIH_TRY
        std::shared_ptr<ReconstructionFilter> rf;
        Properties def(name);
        const Properties &p = props ? *(Properties *) props : def;
        Properties named(name);
        rf = make_rfilter(props ? p : named);
        ((Box<Film> *) f)->p->set_reconstruction_filter(rf);
        return 0; MIH_CATCH(-1)
}
// ReconstructionFilter::eval / eval_discretized / radius / border_size of the film's filter
int mih_film_filter_eval(void *f, float x, float *out4) {
    MIH_TRY const ReconstructionFilter *rf = ((Box<Film> *) f)->p->reconstruction_filter();
        out4[0] = rf->eval(x); out4[1] = rf->eval_discretized(x); out4[2] = rf->radius(); out4[3] = (float) rf->border_size(); return 0; MIH_CATCH(-1)
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
// SamplingIntegrator::render over a scene built on several GPUs (mih_scene_build_multi): tile shards on one host thread per context,
// one film reduce. Same return values as mih_integrator_render (which takes this path by itself for such a scene); *how = MI_REDUCE_*.
int mih_render_multi(void *i, void *scene, void *sensor, int *how) {
    MIH_TRY
        Scene *sc = ((Box<Scene> *) scene)->p.get();
        if (sc->device_count() < 2) throw std::runtime_error("mih_render_multi: the scene was not built on several contexts (mih_scene_build_multi)");
        SamplingIntegrator *it = ((Box<SamplingIntegrator> *) i)->p.get();
        const bool done = it->render(sc, ((Box<PerspectiveCamera> *) sensor)->p.get());
        if (how) *how = it->last_reduce();
        return done ? 1 : 0; MIH_CATCH(-1)
}
int mih_integrator_last_reduce(void *i) { return ((Box<SamplingIntegrator> *) i)->p->last_reduce(); }
int mih_integrator_counters(void *i, mi_counters *out) { *out = ((Box<SamplingIntegrator> *) i)->p->counters(); return 0; }
// Host-side job description (no GPU needed). block_ids / tiles must hold `capacity` entries.
int mih_make_render_cfg(void *i, void *sensor, mi_render_cfg *cfg, uint32_t *block_ids, uint32_t *tiles, uint32_t capacity, uint32_t n_threads) {
    MIH_TRY
        std::vector<uint32_t> ids, tl;
        ((Box<SamplingIntegrator> *) i)->p->make_render_cfg(((Box<PerspectiveCamera> *) sensor)->p.get(), *cfg, ids, tl, n_threads);
        if (ids.size() > capacity) throw std::runtime_error("mih_make_render_cfg: capacity too small");
        std::memcpy(block_ids, ids.data(), ids.size() * 4);
        std::memcpy(tiles, tl.data(), tl.size() * 4);
        cfg->block_ids = block_ids; cfg->tile_list = cfg->tile_list ? tiles : nullptr;   // sharded with zero tiles != unsharded
        return 0; MIH_CATCH(-1)
}
int mih_make_render_cfg_pass(void *i, void *sensor, mi_render_cfg *cfg, uint32_t *block_ids, uint32_t *tiles, uint32_t capacity, uint32_t n_threads, uint32_t pass) {
    MIH_TRY
        std::vector<uint32_t> ids, tl;
        ((Box<SamplingIntegrator> *) i)->p->make_render_cfg(((Box<PerspectiveCamera> *) sensor)->p.get(), *cfg, ids, tl, n_threads, pass);
        if (ids.size() > capacity) throw std::runtime_error("mih_make_render_cfg: capacity too small");
        std::memcpy(block_ids, ids.data(), ids.size() * 4);
        std::memcpy(tiles, tl.data(), tl.size() * 4);
        cfg->block_ids = block_ids; cfg->tile_list = cfg->tile_list ? tiles : nullptr;   // sharded with zero tiles != unsharded
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
