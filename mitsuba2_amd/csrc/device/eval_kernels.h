// mi_trace and mi_eval kernels (the Scene::ray_intersect surface and the leaf-function test entry points).
// Part of the single translation unit csrc/miwave.hip (included there, in this order; not a stand-alone header).
struct SoaRays { const float *ox, *oy, *oz, *dx, *dy, *dz, *mint, *maxt; };
struct SoaHits { float *t, *u, *v; uint32_t *prim, *shape; };

template <bool AnyHit>
__global__ __launch_bounds__(MIW_BLOCK) void k_trace_soa(SceneView sc, SoaRays R, SoaHits H, uint64_t n, TraceLds cfg) {
    extern __shared__ uint4 smem[];
    stage_to_lds(sc, cfg, smem);
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Hit h;
    bool hit = trace_one<AnyHit>(sc, cfg, smem, v3(R.ox[i], R.oy[i], R.oz[i]), v3(R.dx[i], R.dy[i], R.dz[i]),
                                 R.mint[i], R.maxt[i], h);
    H.t[i] = hit ? h.t : MIW_INFINITY;
    if (H.u) H.u[i] = h.u;
    if (H.v) H.v[i] = h.v;
    if (H.prim) H.prim[i] = hit ? h.prim : 0xffffffffu;
    if (H.shape) H.shape[i] = hit ? sc.tris[h.tri].shape : 0xffffffffu;
}

// ---- the Scene query surface (include/miwave.h: mi_ray_intersect, mi_sample_emitter_direction, ...) ----
__device__ __forceinline__ void st3(float *dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

__global__ __launch_bounds__(MIW_BLOCK) void k_ray_intersect(SceneView sc, SoaRays R, mi_surface_interaction *out, uint64_t n, TraceLds cfg) {
    extern __shared__ uint4 smem[];
    stage_to_lds(sc, cfg, smem);
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o = v3(R.ox[i], R.oy[i], R.oz[i]), d = v3(R.dx[i], R.dy[i], R.dz[i]);
    Hit h;
    const bool hit = trace_one<false>(sc, cfg, smem, o, d, R.mint[i], R.maxt[i], h);
    mi_surface_interaction r;
    memset(&r, 0, sizeof r);
    if (hit) {
        SurfaceInteraction si; uint32_t bsdf_index; int32_t emitter;
        hit_surface_interaction<true, true>(sc, h.tri, h.t, h.u, h.v, [o]() { return o; }, d, si, bsdf_index, emitter);
        r.t = si.t; st3(r.p, si.p); st3(r.n, si.n); st3(r.sh_s, si.sh.s); st3(r.sh_t, si.sh.t); st3(r.sh_n, si.sh.n);
        r.uv[0] = si.uv.x; r.uv[1] = si.uv.y; st3(r.wi, si.wi);
        r.prim_index = si.prim; r.shape_index = si.shape; r.emitter_index = emitter;
    } else {                                                   // interaction.h:571-596 on an invalid pi: t = inf, wi = -ray.d
        r.t = MIW_INFINITY; st3(r.wi, -d);
        r.prim_index = r.shape_index = 0xffffffffu;
        r.emitter_index = sc.env ? (int32_t) sc.env->emitter_index : -1;   // scene.h:248-249
    }
    out[i] = r;
}

__global__ __launch_bounds__(MIW_BLOCK) void k_sample_emitter_direction(SceneView sc, int32_t emitter, const float *ref_p, const float *sample,
                                                                       const float *wavelengths, int test_visibility, mi_direction_sample *out,
                                                                       float *spec_out, uint64_t n, TraceLds cfg) {
    extern __shared__ uint4 smem[];
    stage_to_lds(sc, cfg, smem);
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Wavelengths wl;
#if MIW_SPECTRAL
    for (int k = 0; k < 4; ++k) wl.l[k] = wavelengths[4 * i + k];
#else
    (void) wavelengths;
#endif
    const V3 ref = v3(ref_p[3 * i], ref_p[3 * i + 1], ref_p[3 * i + 2]);
    const V2 u = v2(sample[2 * i], sample[2 * i + 1]);
    DirectionSample ds;
    Spec value = emitter < 0 ? sample_emitter_direction(sc, ref, u, ds, wl) : emitter_sample_direction(sc, (uint32_t) emitter, ref, u, ds, wl);
    if (test_visibility && ds.pdf != 0.f) {                    // scene.cpp:203-207
        Hit h;
        if (trace_one<true>(sc, cfg, smem, ref, ds.d, spawn_mint(ref), ds.dist * (1.f - MIW_SHADOW_EPSILON), h)) value = spec(0.f);
    }
    mi_direction_sample r;
    st3(r.p, ds.p); st3(r.n, ds.n); st3(r.d, ds.d); r.dist = ds.dist; r.pdf = ds.pdf; r.emitter_index = (int32_t) ds.emitter;
    out[i] = r;
    const float *vf = reinterpret_cast<const float *>(&value);
    for (int k = 0; k < MIW_SPEC_N; ++k) spec_out[MIW_SPEC_N * i + k] = vf[k];
}

__global__ void k_pdf_emitter_direction(SceneView sc, int32_t emitter, const float *ref_p, const mi_direction_sample *ds, float *pdf, uint64_t n) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mi_direction_sample r = ds[i];
    const V3 ref = v3(ref_p[3 * i], ref_p[3 * i + 1], ref_p[3 * i + 2]);
    const uint32_t e = emitter < 0 ? (uint32_t) r.emitter_index : (uint32_t) emitter;
    float v = 0.f;
    if (e < sc.emitter_count)
        v = emitter < 0 ? pdf_emitter_direction(sc, e, ld3(r.d), r.dist, ld3(r.n), ref) : emitter_pdf_direction(sc, e, ld3(r.d), r.dist, ld3(r.n), ref);
    pdf[i] = v;
}

__global__ void k_emitter_eval(SceneView sc, const mi_surface_interaction *si, const float *wavelengths, float *spec_out, uint64_t n) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Wavelengths wl;
#if MIW_SPECTRAL
    for (int k = 0; k < 4; ++k) wl.l[k] = wavelengths[4 * i + k];
#else
    (void) wavelengths;
#endif
    const mi_surface_interaction r = si[i];
    Spec value = spec(0.f);
    if (r.emitter_index >= 0 && (uint32_t) r.emitter_index < sc.emitter_count) {
        const EmitterRec &e = sc.emitters[r.emitter_index];
        if (e.type == EMITTER_ENVMAP) { if (sc.env) value = env_eval_spec(*sc.env, -ld3(r.wi), wl); }   // envmap.cpp:137: v = to_local(-si.wi)
        else value = emitter_eval(e, ld3(r.wi), wl);
    }
    const float *vf = reinterpret_cast<const float *>(&value);
    for (int k = 0; k < MIW_SPEC_N; ++k) spec_out[MIW_SPEC_N * i + k] = vf[k];
}

__global__ void k_eval(int op, RenderParams P, SceneView sc, const float *in, int is, float *out, int os, uint64_t n) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *a = in + i * (uint64_t) is;
    float *o = out + i * (uint64_t) os;
    switch (op) {
        case MI_EVAL_PCG32: {
            PCG32 r; pcg32_seed(r, (uint64_t) f2u(a[0]) | ((uint64_t) f2u(a[1]) << 32), MIW_PCG32_DEFAULT_STREAM);
            for (int k = 0; k < 8; ++k) o[k] = pcg32_next_f32(r);
        } break;
        case MI_EVAL_SINCOS: sincos_(a[0], o[0], o[1]); break;
        case MI_EVAL_COSINE_HEMISPHERE: {
            V3 w = square_to_cosine_hemisphere(v2(a[0], a[1]));
            o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = square_to_cosine_hemisphere_pdf(w);
        } break;
        case MI_EVAL_BSDF: {      // spectral builds: in[10..13] = wavelengths; colour outputs have MIW_SPEC_N channels
            const uint32_t b_index = f2u(a[0]);
            V3 wi = v3(a[1], a[2], a[3]), wo = v3(a[7], a[8], a[9]);
            Wavelengths wl;
#if MIW_SPECTRAL
            for (int k = 0; k < 4; ++k) wl.l[k] = a[10 + k];
#endif
            const BsdfSide b = bsdf_side(sc.bsdfs, b_index, wi);
            const TexCtx tc(wl, v2(0.f, 0.f), nullptr, sc.bsdf_tables);
            BSDFSample bs; Spec w = bsdf_side_sample(b, wi, a[4], v2(a[5], a[6]), bs, tc);
            o[0] = bs.wo.x; o[1] = bs.wo.y; o[2] = bs.wo.z; o[3] = bs.pdf; o[4] = bs.eta; o[5] = u2f(bs.sampled_type);
            Spec e = bsdf_side_eval(b, wi, wo, tc);
            const float *wf = reinterpret_cast<const float *>(&w), *ef = reinterpret_cast<const float *>(&e);
            for (int k = 0; k < MIW_SPEC_N; ++k) { o[6 + k] = wf[k]; o[6 + MIW_SPEC_N + k] = ef[k]; }
            o[6 + 2 * MIW_SPEC_N] = bsdf_side_pdf(b, wi, wo, tc);
        } break;
        case MI_EVAL_FRESNEL: fresnel(a[0], a[1], o[0], o[1], o[2], o[3]); break;
        case MI_EVAL_CAMERA_RAY: {
            V2 adj = v2((a[0] - (float) P.film.crop_x) / (float) P.film.crop_w,
                        (a[1] - (float) P.film.crop_y) / (float) P.film.crop_h);
            Ray r = sensor_sample_ray(P.sensor, adj);
            o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z; o[6] = r.mint; o[7] = r.maxt;
        } break;
        case MI_EVAL_EMITTER_SAMPLE: {   // spectral builds: in[5..8] = wavelengths
            Wavelengths wl;
#if MIW_SPECTRAL
            for (int k = 0; k < 4; ++k) wl.l[k] = a[5 + k];
#endif
            DirectionSample ds; Spec s = sample_emitter_direction(sc, v3(a[0], a[1], a[2]), v2(a[3], a[4]), ds, wl);
            o[0] = ds.d.x; o[1] = ds.d.y; o[2] = ds.d.z; o[3] = ds.dist; o[4] = ds.pdf;
            o[5] = ds.p.x; o[6] = ds.p.y; o[7] = ds.p.z; o[8] = ds.n.x; o[9] = ds.n.y; o[10] = ds.n.z;
            const float *sf = reinterpret_cast<const float *>(&s);
            for (int k = 0; k < MIW_SPEC_N; ++k) o[11 + k] = sf[k];
        } break;
        case MI_EVAL_FP_SEMANTICS: {
            float x = a[0], y = a[1], z = a[2];
            o[0] = x + y; o[1] = x * y; o[2] = x / y; o[3] = __builtin_sqrtf(abs_(x));
            o[4] = fmadd(x, y, z); o[5] = rcp(x); o[6] = min_(x, y); o[7] = max_(x, y);
        } break;
        case MI_EVAL_SPECIAL: o[0] = exp_(a[0]); o[1] = log_(a[0]); o[2] = erf_(a[0]); o[3] = erfinv_(a[0]); break;
        case MI_EVAL_ENVMAP: {           // spectral builds: in[8..11] = wavelengths; eval and spec have MIW_SPEC_N channels
            if (!sc.env) break;
            Wavelengths wl;
#if MIW_SPECTRAL
            for (int k = 0; k < 4; ++k) wl.l[k] = a[8 + k];
#endif
            V3 d = v3(a[0], a[1], a[2]);
            const Spec e = env_eval_spec(*sc.env, d, wl);
            V3 sd, sp, sn; float dist, pdf;
            const Spec ss = env_sample_direction_spec(*sc.env, v3(a[3], a[4], a[5]), v2(a[6], a[7]), sd, dist, pdf, sp, sn, wl);
            const float *ef = reinterpret_cast<const float *>(&e), *sf = reinterpret_cast<const float *>(&ss);
            for (int k = 0; k < MIW_SPEC_N; ++k) { o[k] = ef[k]; o[MIW_SPEC_N + 6 + k] = sf[k]; }
            o[MIW_SPEC_N] = env_pdf_direction(*sc.env, d);
            o[MIW_SPEC_N + 1] = sd.x; o[MIW_SPEC_N + 2] = sd.y; o[MIW_SPEC_N + 3] = sd.z; o[MIW_SPEC_N + 4] = dist; o[MIW_SPEC_N + 5] = pdf;
        } break;
#if MIW_SPECTRAL
        case MI_EVAL_SPECTRUM: {         // in: wavelength sample, c0, c1, c2 (srgb coefficients), d65 scale
            Wavelengths wl; Spec wt;
            sample_wavelengths(a[0], wl, wt);
            TexRec t; t.type = TEX_SRGB_D65; t.v[0] = a[1]; t.v[1] = a[2]; t.v[2] = a[3]; t.v[3] = a[4];
            Spec sd = tex_eval(t, wl);
            t.type = TEX_SRGB; Spec sr = tex_eval(t, wl);
            for (int k = 0; k < 4; ++k) { o[k] = wl.l[k]; o[4 + k] = wt.c[k]; o[8 + k] = sr.c[k]; o[12 + k] = sd.c[k]; }
            V3 xyz = spectrum_to_xyz(wt * sd, wl);
            o[16] = xyz.x; o[17] = xyz.y; o[18] = xyz.z;
        } break;
#endif
        case MI_EVAL_INVTRIG: o[0] = atan2_(a[0], a[1]); o[1] = acos_(a[1]); o[2] = asin_(a[1]); break;
        case MI_EVAL_TEXTURE: {
            if (!sc.bitmaps) break;
            Wavelengths wl;
#if MIW_SPECTRAL
            Spec wt; sample_wavelengths(a[3], wl, wt);
#endif
            TexRec t; t.type = TEX_BITMAP; t.v[0] = a[2]; t.v[1] = t.v[2] = t.v[3] = 0.f;
            Spec r = tex_eval(t, TexCtx(wl, v2(a[0], a[1]), sc.bitmaps));
            const float *rf = reinterpret_cast<const float *>(&r);
            for (int k = 0; k < MIW_SPEC_N; ++k) o[k] = rf[k];
        } break;
    }
}
