// libmiwave_host: SamplingIntegrator machinery, path / direct / moment integrators.
// Part of the single translation unit host/miwave_host.cpp (included there, in this order).
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
void SamplingIntegrator::cancel() {
    mi_ctx *c = m_active_ctx.load(); if (c) mi_cancel(c);
    std::lock_guard<std::mutex> lock(m_active_mutex);          // a multi-GPU frame: every context of it
    for (mi_ctx *m : m_active_multi) mi_cancel(m);
}

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
    make_render_cfg(sensor, cfg, block_ids, tiles, n_threads, pass, m_rank, m_world);
}
void SamplingIntegrator::make_render_cfg(const PerspectiveCamera *sensor, mi_render_cfg &cfg,
                                     std::vector<uint32_t> &block_ids, std::vector<uint32_t> &tiles,
                                     uint32_t n_threads, uint32_t pass, uint32_t m_rank, uint32_t m_world) const {
    std::memset(&cfg, 0, sizeof cfg);
    const Film *film = sensor->film().get();
    auto cs = film->crop_size(); auto co = film->crop_offset();
    size_t total_spp = sensor->sampler()->sample_count();
    size_t spp_pass = samples_per_pass_of(m_samples_per_pass, total_spp);
    const size_t n_passes = total_spp / spp_pass;
    if (pass >= n_passes) Throw("make_render_cfg: pass index out of range");
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
        // spiral.cpp:41: block_id = counter + (remaining_passes - 1) * block_count — the FIRST pass the reference
        // renders carries the highest offset, the last one offset 0; `pass` counts in execution order
        block_ids[by * nbx + bx] = (uint32_t) (b.block_id + (n_passes - 1 - (size_t) pass) * spiral.block_count());
        by_id[b.block_id] = by * nbx + bx;
    }
    tiles.clear();
    if (m_world > 1)                                           // interleaved shard over the spiral order
        for (size_t id = m_rank; id < by_id.size(); id += m_world) tiles.push_back(by_id[id]);
    cfg.block_ids = block_ids.data(); cfg.block_count = (uint32_t) block_ids.size();
    // a rank whose shard is empty (world > block count) renders NOTHING: mi_render reads tile_list == NULL as
    // "all blocks", so a sharded job always carries a non-null list, with tile_count == 0 if need be
    static const uint32_t no_tiles[1] = { 0 };
    cfg.tile_list = m_world > 1 ? (tiles.empty() ? no_tiles : tiles.data()) : nullptr; cfg.tile_count = (uint32_t) tiles.size();
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

// One pass of a frame over the N contexts of a multi-GPU scene (Scene::build(devices)) — SURVEY.md section 8(e) inside one
// process: context r renders tile shard (m_rank * N + r) of (m_world * N) (the spiral blocks with id % world == rank, all spp of
// their pixels, global block-id -> seed table: the film does not depend on N) on its own host thread into a film on its own
// device; mi_film_reduce sums the N films onto context 0's (RCCL over xGMI between distinct GPUs, a rank-ordered device add when
// contexts share a GPU); the root's film comes down once. Passes after the first are added to `film5` on the host (the reference
// accumulates them in the film the same way, integrator.cpp:100-131; with several GPUs the per-pass partial sums exist only after
// the reduce). Counters: work summed over the contexts, times = the slowest context's.
bool SamplingIntegrator::render_pass_multi(Scene *scene, PerspectiveCamera *sensor, float *film5, int moment_pass, uint32_t pass) {
    const size_t n = scene->device_count();
    struct Rank { mi_render_cfg cfg; std::vector<uint32_t> block_ids, tiles; void *film = nullptr; mi_status st = MI_OK; std::string err; mi_counters cnt{}; };
    std::vector<Rank> ranks(n);
    uint64_t count = 0;
    auto release = [&]() { for (size_t r = 0; r < n; ++r) if (ranks[r].film) { mi_film_free(scene->ctx(r), ranks[r].film); ranks[r].film = nullptr; } };
    for (size_t r = 0; r < n; ++r) {
        Rank &k = ranks[r];
        make_render_cfg(sensor, k.cfg, k.block_ids, k.tiles, 1, pass, m_rank * (uint32_t) n + (uint32_t) r, m_world * (uint32_t) n);
        k.cfg.moment_pass = moment_pass; k.cfg.film_on_device = 1; k.cfg.film_f64 = 0; k.cfg.accumulate = 0;
        count = (uint64_t) k.cfg.crop_w * k.cfg.crop_h * 5;
        if (mi_film_alloc(scene->ctx(r), count, &k.film) != MI_OK) { const std::string e = mi_last_error(scene->ctx(r)); release(); Throw("mi_film_alloc: " + e); }
    }
    { std::lock_guard<std::mutex> lock(m_active_mutex); for (size_t r = 0; r < n; ++r) m_active_multi.push_back(scene->ctx(r)); }
    std::vector<std::thread> pool;
    for (size_t r = 0; r < n; ++r)
        pool.emplace_back([&, r]() {
            Rank &k = ranks[r];
            k.st = mi_render(scene->ctx(r), &k.cfg, k.film);
            if (k.st != MI_OK && k.st != MI_ERR_CANCELLED) k.err = mi_last_error(scene->ctx(r));
            mi_get_counters(scene->ctx(r), &k.cnt);
        });
    for (auto &t : pool) t.join();
    { std::lock_guard<std::mutex> lock(m_active_mutex); m_active_multi.clear(); }
    bool cancelled = false;
    for (size_t r = 0; r < n; ++r) {
        if (ranks[r].st == MI_ERR_CANCELLED) cancelled = true;
        else if (ranks[r].st != MI_OK) { const std::string e = ranks[r].err; release(); Throw("mi_render (context " + std::to_string(r) + "): " + e); }
    }
    std::vector<mi_ctx *> ctxs(n); std::vector<void *> films(n);
    for (size_t r = 0; r < n; ++r) { ctxs[r] = scene->ctx(r); films[r] = ranks[r].film; }
    int32_t how = MI_REDUCE_NONE;
    if (mi_film_reduce(ctxs.data(), films.data(), (int32_t) n, count, 0, &how) != MI_OK) { const std::string e = mi_last_error(ctxs[0]); release(); Throw("mi_film_reduce: " + e); }
    m_last_reduce = how;
    if (pass == 0) {
        if (mi_film_download(ctxs[0], films[0], film5, count) != MI_OK) { const std::string e = mi_last_error(ctxs[0]); release(); Throw("mi_film_download: " + e); }
    } else {
        std::vector<float> tmp(count);
        if (mi_film_download(ctxs[0], films[0], tmp.data(), count) != MI_OK) { const std::string e = mi_last_error(ctxs[0]); release(); Throw("mi_film_download: " + e); }
        for (uint64_t i = 0; i < count; ++i) film5[i] += tmp[i];
    }
    release();
    m_counters = ranks[0].cnt;
    for (size_t r = 1; r < n; ++r) {
        const mi_counters &c = ranks[r].cnt;
        m_counters.samples += c.samples; m_counters.segments += c.segments; m_counters.shadow_rays += c.shadow_rays;
        m_counters.iterations += c.iterations; m_counters.lanes += c.lanes;
        m_counters.ms_render = std::max(m_counters.ms_render, c.ms_render); m_counters.ms_path = std::max(m_counters.ms_path, c.ms_path);
        m_counters.ms_resolve = std::max(m_counters.ms_resolve, c.ms_resolve);
    }
    return !cancelled;
}

bool SamplingIntegrator::render_passes(Scene *scene, PerspectiveCamera *sensor, float *film5, int moment_pass) {
    const uint32_t passes = pass_count(sensor);
    mi_counters total{};
    if (scene->device_count() > 1) {                            // a multi-GPU scene: every pass sharded over its contexts
        for (uint32_t pass = 0; pass < passes; ++pass) {
            const bool done = render_pass_multi(scene, sensor, film5, moment_pass, pass);
            if (pass > 0) {
                m_counters.samples += total.samples; m_counters.segments += total.segments; m_counters.shadow_rays += total.shadow_rays;
                m_counters.iterations += total.iterations; m_counters.ms_render += total.ms_render;
            }
            total = m_counters;
            if (!done) return false;
        }
        return true;
    }
    m_last_reduce = MI_REDUCE_NONE;
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
