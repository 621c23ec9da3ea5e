// Film assembly: k_film_resolve (float64 sums), k_film_blocks (texel patches, 24-byte position log), the replays of the 16-byte
// class records — k_film_lanes (a 4 x 4 texel block per lane over the tile-interleaved log: the default from 448 tiles on), k_film_quads
// (texel groups inside DPP quads: smaller shards), k_film_columns / k_film_groups (rounds 4 / 3, kept as twins) —, k_film_merge.
// Part of the single translation unit csrc/miwave.hip (included there, in this order; not a stand-alone header).
__global__ void k_film_resolve(const double *accum, float *out32, double *out64, size_t n, int accumulate) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (out64) out64[i] = accumulate ? out64[i] + accum[i] : accum[i];
    else out32[i] = accumulate ? (float) ((double) out32[i] + accum[i]) : (float) accum[i];
}

// Ordered film assembly, step 1 (miw/film_gather.h): the bordered ImageBlock of every spiral block
// is rebuilt by TEXEL PATCHES — one wavefront owns an 8x8 patch of block texels, one texel per lane,
// the five channel sums live in registers. The wave walks the block's pixels in Morton order
// (render_block's order, integrator.cpp:196-203), skips those whose filter footprint cannot reach
// the patch, and replays each remaining pixel's sample run front to back: 64 samples are fetched with
// one coalesced load (lane i = sample i; the log is [lane][sample]), the owning lane derives what
// ImageBlock::put derives once per sample (lo, clipped extent, the discretised x/y weights,
// imageblock.cpp:114-146) and parks the weights in LDS, then the samples are broadcast one by one
// (v_readlane) and each lane adds value*wy*wx to its texel iff the footprint covers it (:148-161).
// Every texel therefore sees exactly the reference's sequence of float32 additions; there are no
// atomics and no cross-lane accumulation, and the dependent chain per texel is register-only.
#define MIW_FP_SIDE 8                  /* patch = 8 x 8 texels = one wavefront */
#define MIW_FP_MAXN 8                  /* filter footprint is at most 8 x 8 (radius <= 4) */
struct PatchArgs { uint32_t patches_x, patches_y; int32_t reach; };

template <bool wide>
__global__ __launch_bounds__(64) void k_film_blocks(FilmRec F, BlockReplayArgs A, PatchArgs PA, float *tiles) {
    __shared__ float s_lut[MIW_FILTER_RESOLUTION + 1];
    __shared__ float s_w[64 * 2 * MIW_FP_MAXN];              // per staged sample: wx[8], wy[8]
    const uint32_t l = threadIdx.x;
    const uint32_t tile = blockIdx.x / (PA.patches_x * PA.patches_y), patch = blockIdx.x % (PA.patches_x * PA.patches_y);
    const uint32_t b = A.tile_list ? A.tile_list[tile] : tile;
    const BlockGeom g = block_geom(F, A.blocks_x, b);
    const int ptx0 = (int) (patch % PA.patches_x) * MIW_FP_SIDE, pty0 = (int) (patch / PA.patches_x) * MIW_FP_SIDE;
    if (ptx0 >= g.size_x || pty0 >= g.size_y) return;        // clipped edge block: patch outside
    const int tx = ptx0 + (int) (l & 7u), ty = pty0 + (int) (l >> 3);
    if (l < 32) s_lut[l] = F.lut[l];
    __syncthreads();

    // pixels (block-local) whose samples can reach this patch
    int x0 = ptx0 - F.border - PA.reach, x1 = ptx0 + MIW_FP_SIDE - 1 - F.border + PA.reach,
        y0 = pty0 - F.border - PA.reach, y1 = pty0 + MIW_FP_SIDE - 1 - F.border + PA.reach;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > g.bw - 1) x1 = g.bw - 1;
    if (y1 > g.bh - 1) y1 = g.bh - 1;

    const float kx = (float) (g.px0 + F.crop_x - F.border) + .5f, ky = (float) (g.py0 + F.crop_y - F.border) + .5f;
    int n = ceil2int((F.radius - 2.f * MIW_RAY_EPSILON) * 2.f);
    if (n > MIW_FP_MAXN) n = MIW_FP_MAXN;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, acc4 = 0.f;
    const uint32_t bs2 = 1u << A.bs2_log2;

    for (uint32_t q = 0; q < bs2; ++q) {                     // wave-uniform scan in Morton order
        uint32_t x, y;
        morton_decode2(q, x, y);
        if ((int) x < x0 || (int) x > x1 || (int) y < y0 || (int) y > y1) continue;
        const uint32_t lane = (tile << A.bs2_log2) + q;
        const uint32_t count = (uint32_t) __builtin_amdgcn_readfirstlane((int) A.st[lane].w);
        const F2 *lp = A.log_pos + (size_t) lane * A.spp; const F4 *lv = A.log_val + (size_t) lane * A.spp;
        for (uint32_t j0 = 0; j0 < count; j0 += 64) {
            const uint32_t m = count - j0 < 64u ? count - j0 : 64u;
            // ---- stage: lane i owns sample j0 + i ----
            F2 p; p.x = __builtin_nanf(""); p.y = 0.f;
            F4 v; v.x = v.y = v.z = v.w = 0.f;
            if (l < m) { p = lp[j0 + l]; v = lv[j0 + l]; }
            int lo_x = 0, lo_y = 0, nx = 0, ny = 0;
            float wx[MIW_FP_MAXN], wy[MIW_FP_MAXN];
            for (int i = 0; i < MIW_FP_MAXN; ++i) wx[i] = wy[i] = 0.f;
            if (p.x == p.x) {                                // not a rejected sample (imageblock.cpp:98-108)
                const float posx = p.x - kx, posy = p.y - ky;                        // :114
                if (wide) {
                    lo_x = ceil2int(posx - F.radius); lo_y = ceil2int(posy - F.radius);
                    if (lo_x < 0) lo_x = 0;
                    if (lo_y < 0) lo_y = 0;
                    int hi_x = floor2int(posx + F.radius), hi_y = floor2int(posy + F.radius);
                    if (hi_x > g.size_x - 1) hi_x = g.size_x - 1;
                    if (hi_y > g.size_y - 1) hi_y = g.size_y - 1;
                    const float base_x = (float) lo_x - posx, base_y = (float) lo_y - posy;
                    for (int i = 0; i < MIW_FP_MAXN; ++i) {
                        if (i < n) {
                            int ix = (int) abs_((base_x + (float) i) * F.scale_factor),
                                iy = (int) abs_((base_y + (float) i) * F.scale_factor);
                            if (ix > MIW_FILTER_RESOLUTION) ix = MIW_FILTER_RESOLUTION;
                            if (iy > MIW_FILTER_RESOLUTION) iy = MIW_FILTER_RESOLUTION;
                            wx[i] = s_lut[ix]; wy[i] = s_lut[iy];
                        }
                    }
                    nx = hi_x - lo_x + 1; ny = hi_y - lo_y + 1;   // texels enabled by `y <= hi_y`, `x <= hi_x`
                    if (nx > n) nx = n;
                    if (ny > n) ny = n;
                    if (nx < 0) nx = 0;
                    if (ny < 0) ny = 0;
                } else {                                     // box filter, :163-170: one texel, weight 1
                    lo_x = ceil2int(posx - .5f); lo_y = ceil2int(posy - .5f);
                    const bool in = lo_x >= 0 && lo_y >= 0 && lo_x < g.size_x && lo_y < g.size_y;
                    nx = ny = in ? 1 : 0; wx[0] = wy[0] = 1.f;
                    if (!in) lo_x = lo_y = 0;
                }
            }
            __syncthreads();                                 // previous chunk's weights fully consumed
            for (int i = 0; i < MIW_FP_MAXN; ++i) { s_w[l * 16 + i] = wx[i]; s_w[l * 16 + 8 + i] = wy[i]; }
            __syncthreads();
            const int pk_lo = lo_x | (lo_y << 16), pk_n = nx | (ny << 8);
            // ---- replay: this pixel's samples back to back, four per trip ----
            // Lanes the footprint does not cover add value * 0 (= +-0: leaves a finite sum unchanged), which
            // keeps the trip branch-free; staged slots >= m carry nx = ny = 0 and value 0.
            const uint32_t m4 = (m + 3u) & ~3u;
            for (uint32_t s0 = 0; s0 < m4; s0 += 4) {
                float w[4], vx[4], vy[4], vz[4], va[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int sl = (int) (s0 + k);
                    const int slo = __builtin_amdgcn_readlane(pk_lo, sl), sn = __builtin_amdgcn_readlane(pk_n, sl);
                    const int xr = tx - (slo & 0xffff), yr = ty - (slo >> 16);
                    const bool hit = (uint32_t) xr < (uint32_t) (sn & 0xff) && (uint32_t) yr < (uint32_t) (sn >> 8);
                    const float wk = s_w[sl * 16 + 8 + (yr & 7)] * s_w[sl * 16 + (xr & 7)];            // wy * wx, :155
                    w[k] = hit ? wk : 0.f;
                    vx[k] = u2f((uint32_t) __builtin_amdgcn_readlane((int) f2u(v.x), sl));
                    vy[k] = u2f((uint32_t) __builtin_amdgcn_readlane((int) f2u(v.y), sl));
                    vz[k] = u2f((uint32_t) __builtin_amdgcn_readlane((int) f2u(v.z), sl));
                    va[k] = u2f((uint32_t) __builtin_amdgcn_readlane((int) f2u(v.w), sl));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (wide) { acc0 += vx[k] * w[k]; acc1 += vy[k] * w[k]; acc2 += vz[k] * w[k]; acc3 += va[k] * w[k]; acc4 += 1.f * w[k]; }
                    else      { acc0 += vx[k] * w[k]; acc1 += vy[k] * w[k]; acc2 += vz[k] * w[k]; acc3 += va[k] * w[k]; acc4 += w[k]; }
                }
            }
        }
    }
    if (tx < g.size_x && ty < g.size_y) {
        float *out = tiles + (size_t) tile * A.tile_stride + ((size_t) ty * g.size_x + tx) * MIW_FILM_CHANNELS;
        out[0] = acc0; out[1] = acc1; out[2] = acc2; out[3] = acc3; out[4] = acc4;
    }
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = (uint32_t) __shfl_xor((int) v, o, 64); v = t > v ? t : v; }
    return v;
}
// ---- fast form of step 1 for the filters with phase classes (miw/film.h, film_classes.h: box, tent, gaussian, mitchell,
// catmullrom) ----
//
// k_film_blocks spends one wave-iteration of all 64 texel lanes on every sample of every pixel in reach of the 8x8 patch
// (144 pixels for 64 texels), although a sample touches 16 texels: 6 % of the lane-iterations add anything. Here a wavefront
// still owns an 8x8 patch, one texel per lane, accumulators in registers, but its lanes form independent GROUPS of GW x GH
// texels, and each group walks ITS OWN Morton-ordered list of the pixels within reach of the group (48 pixels for a 4x2
// group instead of 144), so one wave-iteration serves 64 / (GW*GH) (group, sample) pairs. The render kernels logged 16 bytes
// per sample — X, Y, Z, phase class x | class y << 8 | alpha << 16 (path.h: LogSink16) — so a lane's work per sample is: read
// the record (staged through LDS with coalesced 16-byte loads, 16 samples per group per trip, the next trip's loads in
// flight), look its two weights up in the LDS copy of the class table (a lane knows its offset inside the pixel's window;
// lanes outside the window read the table's zero column, rejected samples carry the table's zero row), multiply, add.
// ~17 VALU + 3 LDS reads per (lane, sample); no position decoding, no per-sample footprint test, no k_film_pack pass.
// Every texel still sees exactly the reference's sequence of float32 additions (its pixels in Morton order, each pixel's
// samples front to back; a texel outside a sample's footprint adds value * 0 = +-0, which leaves a float32 sum that
// started at +0 unchanged), so the tiles are bit-identical to k_film_blocks' and to film_block_replay16 (film_gather.h).
#ifndef MIW_FG_CHUNK
#define MIW_FG_CHUNK 16                /* samples per group staged per trip */
#endif
#define MIW_FG_WSTRIDE 7               /* LDS stride of a class's weights: offsets 0..4 used, 5..6 zero (6 = "lane outside the window") */
template <int GW, int GH>
__global__ __launch_bounds__(64) void k_film_groups(FilmRec F, BlockReplayArgs A, PatchArgs PA, float *tiles) {
    constexpr int GL = GW * GH, NG = 64 / GL, GPX = MIW_FP_SIDE / GW;       // lanes per group, groups, groups per patch row
    constexpr int LCAP = (GW + 4) * (GH + 4), PASSES = NG * MIW_FG_CHUNK / 64;
    static_assert(PASSES >= 1, "group too large");
    extern __shared__ float s_w[];                           // (count + 1) x MIW_FG_WSTRIDE weights; row `count` = 0
    __shared__ unsigned short s_list[NG][LCAP];
    __shared__ uint32_t s_m[NG];
    __shared__ uint4 s_rec[NG][MIW_FG_CHUNK + 1];
    const uint32_t l = threadIdx.x;
    // (consecutive workgroup ids go to consecutive XCDs, one L2 each; remapping them so that the patches of one block tile —
    // whose windows share sample rows — run on ONE XCD measured no difference: 29.7 ms either way. Plain mapping.)
    const uint32_t wg = blockIdx.x;
    const uint32_t tile = wg / (PA.patches_x * PA.patches_y), patch = wg % (PA.patches_x * PA.patches_y);
    const uint32_t b = A.tile_list ? A.tile_list[tile] : tile;
    const BlockGeom g = block_geom(F, A.blocks_x, b);
    const int ptx0 = (int) (patch % PA.patches_x) * MIW_FP_SIDE, pty0 = (int) (patch / PA.patches_x) * MIW_FP_SIDE;
    if (ptx0 >= g.size_x || pty0 >= g.size_y) return;        // clipped edge block: patch outside
    const uint32_t h = l / GL, li = l % GL;
    const int tx = ptx0 + (int) (h % GPX) * GW + (int) (li % GW), ty = pty0 + (int) (h / GPX) * GH + (int) (li / GW);
    const uint32_t rej = A.cls.count;                        // the zero row (LogSink16 logs rejected samples with class `count`)
    for (uint32_t i = l; i < (rej + 1u) * MIW_FG_WSTRIDE; i += 64u) {
        const uint32_t c = i / MIW_FG_WSTRIDE, a = i % MIW_FG_WSTRIDE;
        s_w[i] = (c < rej && a < MIW_FC_STRIDE) ? A.cls.w[c * MIW_FC_STRIDE + a] : 0.f;
    }

    // ---- per group: the pixels within reach of the group, in Morton order ----
    const int reach = A.cls.reach;
    const uint32_t bs2 = 1u << A.bs2_log2, lane0 = tile << A.bs2_log2;
    uint32_t fill[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) fill[i] = 0;
    for (uint32_t q0 = 0; q0 < bs2; q0 += 64u) {
        const uint32_t q = q0 + l;
        uint32_t x, y;
        morton_decode2(q, x, y);
        const bool pixel = q < bs2 && (int) x < g.bw && (int) y < g.bh;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int gx0 = ptx0 + (i % GPX) * GW, gy0 = pty0 + (i / GPX) * GH;
            const bool in = pixel && (int) x >= gx0 - F.border - reach && (int) x <= gx0 + GW - 1 - F.border + reach &&
                                     (int) y >= gy0 - F.border - reach && (int) y <= gy0 + GH - 1 - F.border + reach;
            const unsigned long long m = __ballot(in);
            if (in) {
                const uint32_t at = fill[i] + (uint32_t) __popcll(m & ((1ull << l) - 1ull));
                if (at < (uint32_t) LCAP) s_list[i][at] = (unsigned short) q;
            }
            fill[i] += (uint32_t) __popcll(m);
        }
    }
    uint32_t max_m = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const uint32_t m = fill[i] < (uint32_t) LCAP ? fill[i] : (uint32_t) LCAP;
        if (l == 0) s_m[i] = m;
        max_m = m > max_m ? m : max_m;
    }
    __syncthreads();

    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, acc4 = 0.f;
    const uint32_t my_m = s_m[h];
    const uint32_t pad_meta = film_pack_meta(rej, rej, false);
    for (uint32_t k = 0; k < max_m; ++k) {
        // this lane's offsets inside the window of its group's k-th pixel (LDS word offsets into a class's weights)
        uint32_t ox = 6u, oy = 6u;
        if (k < my_m) {
            uint32_t x, y;
            morton_decode2((uint32_t) s_list[h][k], x, y);
            const int ax = tx - ((int) x + F.border - reach), ay = ty - ((int) y + F.border - reach);
            if ((uint32_t) ax <= (uint32_t) (2 * reach)) ox = (uint32_t) ax;
            if ((uint32_t) ay <= (uint32_t) (2 * reach)) oy = (uint32_t) ay;
        }
        // staging rows of this step: pass i loads group i * (64 / CHUNK) + l / CHUNK, sample l % CHUNK
        size_t row[PASSES]; uint32_t cnt[PASSES];
        uint32_t step_max = 0;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const uint32_t hs = (uint32_t) i * (64u / MIW_FG_CHUNK) + l / MIW_FG_CHUNK;
            cnt[i] = 0; row[i] = 0;
            if (k < s_m[hs]) {
                const uint32_t lane = lane0 + s_list[hs][k];
                cnt[i] = A.st[lane].w; row[i] = (size_t) lane * A.spp;
            }
            step_max = cnt[i] > step_max ? cnt[i] : step_max;
        }
        step_max = wave_max_u32(step_max);
        // the next chunk's loads are in flight while the current one is replayed. The loads are UNCONDITIONAL (a load inside a
        // branch is waited for at the branch's end: no overlap) — the index is clamped into the pixel's run (run 0 of the log for a
        // staging slot without a pixel) and a record past the end of the run gets the zero-weight class when it is staged
        // (logged values are finite: value * 0 adds nothing)
        const uint32_t jj = l % MIW_FG_CHUNK;
        uint4 nr[PASSES];
        auto fetch = [&](uint32_t j0) {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const uint32_t j = j0 + jj, last = cnt[i] ? cnt[i] - 1u : 0u;
                const U4 t = A.log_rec[row[i] + (j < last ? j : last)];
                nr[i] = make_uint4(t.x, t.y, t.z, t.w);
            }
        };
        fetch(0);
        for (uint32_t j0 = 0; j0 < step_max; j0 += MIW_FG_CHUNK) {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                uint4 r = nr[i];
                r.w = j0 + jj < cnt[i] ? r.w : pad_meta;
                s_rec[(uint32_t) i * (64u / MIW_FG_CHUNK) + l / MIW_FG_CHUNK][jj] = r;
            }
            fetch(j0 + MIW_FG_CHUNK);                         // (also after the last chunk: clamped, one cached record — a branch here would put the wait back)
            __syncthreads();
#pragma unroll 4
            for (int s = 0; s < MIW_FG_CHUNK; ++s) {
                const uint4 r = s_rec[h][s];
                const float w = s_w[((r.w >> 8) & 255u) * MIW_FG_WSTRIDE + oy] * s_w[(r.w & 255u) * MIW_FG_WSTRIDE + ox];   // wy * wx, :155
                acc0 += u2f(r.x) * w; acc1 += u2f(r.y) * w; acc2 += u2f(r.z) * w;
                acc3 += (r.w & 0x10000u) ? w : 0.f;         // alpha (0 or 1) * w
                acc4 += w;
            }
            __syncthreads();
        }
    }
    if (tx < g.size_x && ty < g.size_y) {
        float *out = tiles + (size_t) tile * A.tile_stride + ((size_t) ty * g.size_x + tx) * MIW_FILM_CHANNELS;
        out[0] = acc0; out[1] = acc1; out[2] = acc2; out[3] = acc3; out[4] = acc4;
    }
}

// ---- the same replay with a COLUMN of texels per lane (round 4) ----
//
// k_film_groups is bound by its LDS reads (profiles/r03: 63 % LDS-array cycles): per (texel, sample) one 16-byte record and two
// weights. A lane that owns the GH texels of one column of its group reads the record and the x weight ONCE for all of them —
// 1 + 1 + GH reads per GH texel-samples instead of 3 GH — and the unpacking of the record is shared as well. A group is still GW x GH
// texels with its own Morton-ordered pixel list, now GW lanes wide, so a wavefront serves 64 / GW groups; the groups of a block tile
// are numbered row-major and dealt to the wavefronts sixteen at a time (no 8 x 8 patch geometry: a 36 x 36 tile is 9 x 18 groups of
// 4 x 2 = 11 wavefronts with 14 idle group slots, against 25 patches of which 9 hang over the tile's edge). Every texel sees the same
// float32 additions in the same order as before (its group's pixels in Morton order, each pixel's samples front to back;
// w = wy * wx): the tiles are bit-identical to k_film_groups' and k_film_blocks'.
template <int GW, int GH>
__global__ __launch_bounds__(64) void k_film_columns(FilmRec F, BlockReplayArgs A, PatchArgs PA /* patches_x / _y = groups per tile row / column */, float *tiles) {
    constexpr int NG = 64 / GW, LCAP = (GW + 4) * (GH + 4), PASSES = NG * MIW_FG_CHUNK / 64;
    static_assert(PASSES >= 1 && 64 % GW == 0, "group shape");
    extern __shared__ float s_w[];                           // (count + 1) x MIW_FG_WSTRIDE weights; row `count` = 0
    __shared__ unsigned short s_list[NG][LCAP];
    __shared__ uint32_t s_m[NG];
    __shared__ uint4 s_rec[NG][MIW_FG_CHUNK + 1];
    const uint32_t l = threadIdx.x;
    const uint32_t n_groups = PA.patches_x * PA.patches_y, waves_per_tile = (n_groups + NG - 1) / NG;
    const uint32_t tile = blockIdx.x / waves_per_tile, gid0 = (blockIdx.x % waves_per_tile) * NG;
    const uint32_t b = A.tile_list ? A.tile_list[tile] : tile;
    const BlockGeom g = block_geom(F, A.blocks_x, b);
    const uint32_t h = l / GW, li = l % GW;
    const uint32_t my_gid = gid0 + h;
    const int tx = (int) (my_gid % PA.patches_x) * GW + (int) li, ty0 = (int) (my_gid / PA.patches_x) * GH;
    const uint32_t rej = A.cls.count;                        // the zero row (LogSink16 logs rejected samples with class `count`)
    for (uint32_t i = l; i < (rej + 1u) * MIW_FG_WSTRIDE; i += 64u) {
        const uint32_t c = i / MIW_FG_WSTRIDE, a = i % MIW_FG_WSTRIDE;
        s_w[i] = (c < rej && a < MIW_FC_STRIDE) ? A.cls.w[c * MIW_FC_STRIDE + a] : 0.f;
    }
    // ---- per group: the pixels within reach of the group, in Morton order ----
    const int reach = A.cls.reach;
    const uint32_t bs2 = 1u << A.bs2_log2, lane0 = tile << A.bs2_log2;
    uint32_t fill[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) fill[i] = 0;
    for (uint32_t q0 = 0; q0 < bs2; q0 += 64u) {
        const uint32_t q = q0 + l;
        uint32_t x, y;
        morton_decode2(q, x, y);
        const bool pixel = q < bs2 && (int) x < g.bw && (int) y < g.bh;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const uint32_t gid = gid0 + (uint32_t) i;
            const int gx0 = (int) (gid % PA.patches_x) * GW, gy0 = (int) (gid / PA.patches_x) * GH;
            const bool in = pixel && gid < n_groups && gx0 < g.size_x && gy0 < g.size_y &&
                            (int) x >= gx0 - F.border - reach && (int) x <= gx0 + GW - 1 - F.border + reach &&
                            (int) y >= gy0 - F.border - reach && (int) y <= gy0 + GH - 1 - F.border + reach;
            const unsigned long long m = __ballot(in);
            if (in) {
                const uint32_t at = fill[i] + (uint32_t) __popcll(m & ((1ull << l) - 1ull));
                if (at < (uint32_t) LCAP) s_list[i][at] = (unsigned short) q;
            }
            fill[i] += (uint32_t) __popcll(m);
        }
    }
    uint32_t max_m = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const uint32_t m = fill[i] < (uint32_t) LCAP ? fill[i] : (uint32_t) LCAP;
        if (l == 0) s_m[i] = m;
        max_m = m > max_m ? m : max_m;
    }
    __syncthreads();

    float acc[GH][MIW_FILM_CHANNELS];
#pragma unroll
    for (int r = 0; r < GH; ++r)
#pragma unroll
        for (int k = 0; k < MIW_FILM_CHANNELS; ++k) acc[r][k] = 0.f;
    const uint32_t my_m = s_m[h];
    const uint32_t pad_meta = film_pack_meta(rej, rej, false);
    for (uint32_t k = 0; k < max_m; ++k) {
        // this lane's offsets inside the window of its group's k-th pixel (LDS word offsets into a class's weights): one column, GH rows
        uint32_t ox = 6u, oy[GH];
#pragma unroll
        for (int r = 0; r < GH; ++r) oy[r] = 6u;
        if (k < my_m) {
            uint32_t x, y;
            morton_decode2((uint32_t) s_list[h][k], x, y);
            const int ax = tx - ((int) x + F.border - reach), ay = ty0 - ((int) y + F.border - reach);
            if ((uint32_t) ax <= (uint32_t) (2 * reach)) ox = (uint32_t) ax;
#pragma unroll
            for (int r = 0; r < GH; ++r) if ((uint32_t) (ay + r) <= (uint32_t) (2 * reach)) oy[r] = (uint32_t) (ay + r);
        }
        // staging rows of this step: pass i loads group i * (64 / CHUNK) + l / CHUNK, sample l % CHUNK
        size_t row[PASSES]; uint32_t cnt[PASSES];
        uint32_t step_max = 0;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const uint32_t hs = (uint32_t) i * (64u / MIW_FG_CHUNK) + l / MIW_FG_CHUNK;
            cnt[i] = 0; row[i] = 0;
            if (k < s_m[hs]) {
                const uint32_t lane = lane0 + s_list[hs][k];
                cnt[i] = A.st[lane].w; row[i] = (size_t) lane * A.spp;
            }
            step_max = cnt[i] > step_max ? cnt[i] : step_max;
        }
        step_max = wave_max_u32(step_max);
        // (the next chunk's loads are in flight while the current one is replayed; unconditional, clamped — as in k_film_groups)
        const uint32_t jj = l % MIW_FG_CHUNK;
        uint4 nr[PASSES];
        auto fetch = [&](uint32_t j0) {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                const uint32_t j = j0 + jj, last = cnt[i] ? cnt[i] - 1u : 0u;
                const U4 t = A.log_rec[row[i] + (j < last ? j : last)];
                nr[i] = make_uint4(t.x, t.y, t.z, t.w);
            }
        };
        fetch(0);
        for (uint32_t j0 = 0; j0 < step_max; j0 += MIW_FG_CHUNK) {
#pragma unroll
            for (int i = 0; i < PASSES; ++i) {
                uint4 r = nr[i];
                r.w = j0 + jj < cnt[i] ? r.w : pad_meta;
                s_rec[(uint32_t) i * (64u / MIW_FG_CHUNK) + l / MIW_FG_CHUNK][jj] = r;
            }
            fetch(j0 + MIW_FG_CHUNK);
            __syncthreads();
#pragma unroll 4
            for (int s = 0; s < MIW_FG_CHUNK; ++s) {
                const uint4 r = s_rec[h][s];
                const float wx = s_w[(r.w & 255u) * MIW_FG_WSTRIDE + ox];
                const float *wy = s_w + ((r.w >> 8) & 255u) * MIW_FG_WSTRIDE;
                const float vx = u2f(r.x), vy = u2f(r.y), vz = u2f(r.z);
                const bool alpha = (r.w & 0x10000u) != 0u;
#pragma unroll
                for (int q = 0; q < GH; ++q) {
                    const float w = wy[oy[q]] * wx;                          // wy * wx, imageblock.cpp:155
                    acc[q][0] += vx * w; acc[q][1] += vy * w; acc[q][2] += vz * w;
                    acc[q][3] += alpha ? w : 0.f;                            // alpha (0 or 1) * w
                    acc[q][4] += w;
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < GH; ++r)
        if (my_gid < n_groups && tx < g.size_x && ty0 + r < g.size_y) {
            float *out = tiles + (size_t) tile * A.tile_stride + ((size_t) (ty0 + r) * g.size_x + tx) * MIW_FILM_CHANNELS;
#pragma unroll
            for (int k = 0; k < MIW_FILM_CHANNELS; ++k) out[k] = acc[r][k];
        }
}

// ---- the group replay without the record's trip through LDS (round 5) ----
//
// k_film_columns<4,2> is bound by the LDS pipe (one 16-byte record + three weights per lane and sample, the staging writes, two
// barriers per trip) and behind that by VALU issue: ~24 vector instructions per (lane, sample) of which 9 are the sums. Here a
// group of GW x GH texels is GW lanes INSIDE A DPP QUAD (GW = 2: two groups per quad; GW = 4: one), a lane owning a column of GH
// texels:
//  * lane li of a group loads the records GW i + li of a trip straight into registers (the group's loads are GW x 16 contiguous
//    bytes) and DECODES ITS OWN records once — the byte offsets of the two class rows in the LDS weight table, alpha as a float —
//    so the decoding costs 1 / GW per sample; a record past the end of its run gets the table's zero row;
//  * every record then reaches the group's other lanes by quad_perm DPP operands: X, Y, Z and alpha as v_mov_b32_dpp, the two
//    row offsets inside the v_add_u32_dpp that forms the lane's LDS addresses (row offset + the lane's place in the window);
//  * what is left for the LDS is the class table: one x weight per lane and sample and the GH y weights of its column
//    (neighbours in a class's row, which carries zeros in front and behind: "row outside the window" is an index, not a select);
//  * the sums are packed float32 instructions over pairs of rows (v_pk_mul_f32 / v_pk_add_f32: two IEEE products / sums each).
// A column of 4 texels under a 2-wide group (k_film_quads<2, 4>) walks a 6 x 8 pixel window for 8 texels: the same
// 48 pixels per group as 4 x 2, i.e. the same log traffic, but ~31 vector instructions per (lane, sample) serve FOUR texels
// (4 x 2 in this form: ~18 for two; k_film_columns<4, 2>: ~24 + the staging for two).
// Groups are numbered through ALL tiles (a wave takes 64 / GW consecutive groups, which may straddle two tiles): 36 x 36
// texels are 162 groups, and whole waves per tile would leave one lane in six idle.
// No staging buffer, no barrier in the loop; a pixel's count comes from an LDS table filled once per wavefront, so the first
// trip of the NEXT pixel is in flight during the last trip of this one. The float32 additions of a texel are k_film_columns'
// (its group's pixels in Morton order, each pixel's samples front to back, w = wy * wx, value * w, alpha (0 or 1) * w): the tiles
// are bit-identical.
// LDS layout of a class's weights here, for columns of GH texels: GH - 1 zeros, w[0 .. 2 reach], zeros up to an odd stride of 5 + 2 GH
// (at least GH + 1 behind the window: "every row of the column outside the window" is the index 5 + GH - 1)
#define MIW_FQ_WSTRIDE(GH) (5 + 2 * (GH))
#ifndef MIW_FQ_FENCE
#define MIW_FQ_FENCE 0
#endif
typedef const __attribute__((address_space(3))) float miw_lds_cf;
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm_u32(uint32_t v) { return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, 0xf, 0xf, true); }
template <int GW, int GH, int U>
__global__ __launch_bounds__(64) void k_film_quads(FilmRec F, BlockReplayArgs A, PatchArgs PA /* patches_x / _y = groups per tile row / column */, uint32_t n_tiles, float *tiles) {
    constexpr int NG = 64 / GW, LCAP = (GW + 4) * (GH + 4), TRIP = U * GW /* U records per lane and trip */, WS = MIW_FQ_WSTRIDE(GH), LEAD = GH - 1;
    static_assert(GW == 2 || GW == 4, "a group lives inside a DPP quad");
    static_assert(GH % 2 == 0, "rows are summed in pairs");
    extern __shared__ float s_w[];                           // (count + 1) x WS weights; row `count` = 0
    __shared__ unsigned short s_list[NG][LCAP];
    __shared__ uint32_t s_cnt[NG][LCAP];
    __shared__ uint32_t s_m[NG];
    const uint32_t l = threadIdx.x;
    const uint32_t n_groups = PA.patches_x * PA.patches_y;   // per tile; >= NG (the host launches k_film_columns otherwise): a wave meets two tiles at most
    const uint32_t G0 = blockIdx.x * (uint32_t) NG, tile0 = G0 / n_groups, r0 = G0 % n_groups;
    const uint32_t h = l / GW, li = l % GW;
    const bool mine_second = r0 + h >= n_groups;
    const uint32_t tile = tile0 + (mine_second ? 1u : 0u), my_gid = r0 + h - (mine_second ? n_groups : 0u);
    const bool my_live = tile < n_tiles;
    const BlockGeom g0 = block_geom(F, A.blocks_x, A.tile_list ? A.tile_list[tile0 < n_tiles ? tile0 : 0u] : tile0);
    const BlockGeom g1 = block_geom(F, A.blocks_x, A.tile_list ? A.tile_list[tile0 + 1u < n_tiles ? tile0 + 1u : 0u] : tile0 + 1u);
    const BlockGeom g = mine_second ? g1 : g0;
    const int tx = (int) (my_gid % PA.patches_x) * GW + (int) li, ty0 = (int) (my_gid / PA.patches_x) * GH;
    const uint32_t rej = A.cls.count;                        // the zero row (LogSink16 logs rejected samples with class `count`)
    const int reach = A.cls.reach;
    for (uint32_t i = l; i < (rej + 1u) * WS; i += 64u) {
        const uint32_t c = i / WS, a = i % WS;
        s_w[i] = (c < rej && a >= (uint32_t) LEAD && a <= (uint32_t) LEAD + 2u * (uint32_t) reach) ? A.cls.w[c * MIW_FC_STRIDE + a - LEAD] : 0.f;
    }
    // ---- per group: the pixels within reach of the group, in Morton order (as k_film_columns) ----
    const uint32_t bs2 = 1u << A.bs2_log2;
    const uint32_t gx_first = r0 % PA.patches_x, gy_first = r0 / PA.patches_x;
    uint32_t fill[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) fill[i] = 0;
    for (uint32_t q0 = 0; q0 < bs2; q0 += 64u) {
        const uint32_t q = q0 + l;
        uint32_t x, y;
        morton_decode2(q, x, y);
        const bool pixel0 = q < bs2 && (int) x < g0.bw && (int) y < g0.bh, pixel1 = q < bs2 && (int) x < g1.bw && (int) y < g1.bh;
        uint32_t gx = gx_first, gy = gy_first;
        bool second = false;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int gx0 = (int) gx * GW, gy0 = (int) gy * GH;
            const bool live = (second ? tile0 + 1u : tile0) < n_tiles && gx0 < (second ? g1.size_x : g0.size_x) && gy0 < (second ? g1.size_y : g0.size_y);
            const bool in = (second ? pixel1 : pixel0) && live &&
                            (int) x >= gx0 - F.border - reach && (int) x <= gx0 + GW - 1 - F.border + reach &&
                            (int) y >= gy0 - F.border - reach && (int) y <= gy0 + GH - 1 - F.border + reach;
            const unsigned long long m = __ballot(in);
            if (in) {
                const uint32_t at = fill[i] + (uint32_t) __popcll(m & ((1ull << l) - 1ull));
                if (at < (uint32_t) LCAP) s_list[i][at] = (unsigned short) q;
            }
            fill[i] += (uint32_t) __popcll(m);
            if (++gx == PA.patches_x) { gx = 0; if (++gy == PA.patches_y) { gy = 0; second = true; } }
        }
    }
    uint32_t max_m = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const uint32_t m = fill[i] < (uint32_t) LCAP ? fill[i] : (uint32_t) LCAP;
        if (l == 0) s_m[i] = m;
        max_m = m > max_m ? m : max_m;
    }
    __syncthreads();
    const uint32_t lane0 = tile << A.bs2_log2;
    for (uint32_t i = l; i < (uint32_t) (NG * LCAP); i += 64u) {          // every listed pixel's number of logged samples, once
        const uint32_t gi = i / LCAP, k = i % LCAP;
        const uint32_t t = tile0 + (r0 + gi >= n_groups ? 1u : 0u);
        s_cnt[gi][k] = k < s_m[gi] ? A.st[(t << A.bs2_log2) + s_list[gi][k]].w : 0u;
    }
    __syncthreads();

    float acc[GH][MIW_FILM_CHANNELS];
#pragma unroll
    for (int r = 0; r < GH; ++r)
#pragma unroll
        for (int k = 0; k < MIW_FILM_CHANNELS; ++k) acc[r][k] = 0.f;
    const uint32_t my_m = s_m[h];
    const uint32_t w_base = (uint32_t) (uintptr_t) (miw_lds_cf *) s_w;
    const uint32_t off_rej = rej * (uint32_t) (WS * 4);
    // a step = the k-th pixel of every group. Per lane: the pixel's run in the log, its count, the LDS byte address of the lane's
    // x weight and of its column's first y weight inside class 0's row (the zeros in front of and behind the window: "outside")
    struct Step { const U4 *run; uint32_t cnt, bx, by; };
    auto step_of = [&](uint32_t k) {
        Step s; s.run = A.log_rec; s.cnt = 0u; s.bx = w_base + 4u * (5u + LEAD); s.by = s.bx;
        if (k < my_m) {
            const uint32_t q = s_list[h][k];
            uint32_t x, y;
            morton_decode2(q, x, y);
            const int ax = tx - ((int) x + F.border - reach), ay = ty0 - ((int) y + F.border - reach);
            if ((uint32_t) ax <= (uint32_t) (2 * reach)) s.bx = w_base + 4u * (uint32_t) (ax + LEAD);
            if ((uint32_t) (ay + LEAD) <= (uint32_t) (2 * reach + LEAD)) s.by = w_base + 4u * (uint32_t) (ay + LEAD);
            s.cnt = s_cnt[h][k]; s.run = A.log_rec + (size_t) (lane0 + q) * A.spp;
        }
        return s;
    };
    // the loads are unconditional and clamped into the run (a record past its end gets the zero row when it is decoded; logged
    // values are finite: value * 0 adds nothing), so nothing waits at a branch's end. Record i of the next trip is requested as
    // soon as record i of this trip has been decoded: most of a trip to arrive, no second buffer
    uint4 nxt[U];
    auto fetch_one = [&](int i, const U4 *run, uint32_t last, uint32_t j0) {
        const uint32_t j = j0 + (uint32_t) (GW * i) + li;
        const U4 t = run[j < last ? j : last];
        nxt[i] = make_uint4(t.x, t.y, t.z, t.w);
    };
    auto fetch = [&](const U4 *run, uint32_t cnt, uint32_t j0) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            fetch_one(i, run, cnt ? cnt - 1u : 0u, j0);
            __builtin_amdgcn_sched_barrier(0);           // record 0 first, as inside the loop: its consumer then waits for vmcnt(U - 1), not for all of them
        }
    };
    Step cur = step_of(0);
    fetch(cur.run, cur.cnt, 0u);
    for (uint32_t k = 0; k < max_m; ++k) {
        const uint32_t step_max = wave_max_u32(cur.cnt);
        const Step nx = step_of(k + 1u);
        for (uint32_t j0 = 0; j0 < step_max; j0 += (uint32_t) TRIP) {
            const bool last_trip = j0 + (uint32_t) TRIP >= step_max;                  // (wave-uniform) then: the next pixel's first trip
            const U4 *t_run = last_trip ? nx.run : cur.run;
            const uint32_t t_cnt = last_trip ? nx.cnt : cur.cnt, t_last = t_cnt ? t_cnt - 1u : 0u, t_j0 = last_trip ? 0u : j0 + (uint32_t) TRIP;
            auto one = [&](float vx, float vy, float vz, float va, uint32_t ax, uint32_t ay) {
                const float wx = *(miw_lds_cf *) (uintptr_t) ax;
                miw_lds_cf *wy = (miw_lds_cf *) (uintptr_t) ay;
#pragma unroll
                for (int q = 0; q < GH; ++q) {
                    const float w = wy[q] * wx;                              // wy * wx, imageblock.cpp:155
                    acc[q][0] += vx * w; acc[q][1] += vy * w; acc[q][2] += vz * w;
                    acc[q][3] += va * w;                                     // alpha (0 or 1) * w
                    acc[q][4] += w;
                }
            };
#pragma unroll
            for (int i = 0; i < U; ++i) {
                // this lane's record of the quartet: decoded once, for the whole group
                const uint4 r = nxt[i];
                const bool valid = j0 + (uint32_t) (GW * i) + li < cur.cnt;
                const uint32_t ox = valid ? (r.w & 255u) * (uint32_t) (WS * 4) : off_rej, oy = valid ? ((r.w >> 8) & 255u) * (uint32_t) (WS * 4) : off_rej;
                const uint32_t af = (r.w & 0x10000u) ? 0x3f800000u : 0u;
                fetch_one(i, t_run, t_last, t_j0);                                 // (after the copy-out: into the same registers)
#define MIW_FQ_ONE(C) one(u2f(quad_perm_u32<C>(r.x)), u2f(quad_perm_u32<C>(r.y)), u2f(quad_perm_u32<C>(r.z)), u2f(quad_perm_u32<C>(af)), \
                          quad_perm_u32<C>(ox) + cur.bx, quad_perm_u32<C>(oy) + cur.by)
                if (GW == 4) { MIW_FQ_ONE(0x00); MIW_FQ_ONE(0x55); MIW_FQ_ONE(0xAA); MIW_FQ_ONE(0xFF); }
                else { MIW_FQ_ONE(0xA0); MIW_FQ_ONE(0xF5); }                       // quad_perm [0,0,2,2], [1,1,3,3]: the pair's first / second lane
#undef MIW_FQ_ONE
#if MIW_FQ_FENCE
                __builtin_amdgcn_sched_barrier(0);                                 // one record's samples at a time: the scheduler otherwise spreads all of a trip's broadcasts out first (registers)
#endif
            }
        }
        if (step_max == 0u) fetch(nx.run, nx.cnt, 0u);                             // (a step without samples consumed nothing)
        cur = nx;
    }
#pragma unroll
    for (int r = 0; r < GH; ++r)
        if (my_live && tx < g.size_x && ty0 + r < g.size_y) {
            float *out = tiles + (size_t) tile * A.tile_stride + ((size_t) (ty0 + r) * g.size_x + tx) * MIW_FILM_CHANNELS;
#pragma unroll
            for (int k = 0; k < MIW_FILM_CHANNELS; ++k) out[k] = acc[r][k];
        }
}

// ---- one texel block per lane (round 5, the default for the filters with class tables) ----
//
// PMC passes over k_film_quads<2, 4> (profiles/r05_experiments.txt): the SIMDs issue an instruction in ~96 % of their cycles, ~38 per
// (lane, sample) for four texels of which 25 / 48 lie in the sample's footprint — that replay is bound by instruction issue, and
// most of what it issues multiplies by a zero weight. Here
//  * a LANE owns a block of 4 x 4 texels of one tile (80 sums in registers) and streams the samples of the 8 x 8 pixels within
//    reach itself: no records shared between lanes, so nothing to broadcast, and a pixel's run is read by 4 blocks instead of 6
//    groups (log traffic x 4 instead of x 6);
//  * a WAVEFRONT is the same block position in 64 consecutive tiles: all its lanes are at the same tile-relative pixel at every
//    step (tiles are aligned to the block size, so the Morton order of a window is the same in every tile), hence which of the
//    block's rows and columns the pixel's footprint covers is WAVE-UNIFORM: the pixel's run is replayed by a copy of the sample loop
//    specialised for that range of rows and of column pairs (30 copies, a jump per pixel), which issues the products and sums of
//    the covered texels only — 20 x 12 of 32 x 16 (row, column pair) slots over a window, 47 % of the block's 64 x 8;
//  * sums and products are packed float32 over pairs of columns (v_pk_mul_f32 / v_pk_add_f32, two IEEE operations each);
//  * the 64 lanes of a load read 64 different tiles' logs: over [lane][sample] that is 64 cache lines 8 MB apart per load (measured:
//    36.8 ms at C2, against 11.3 ms with the loads pointed at one run), so the render kernels write the log INTERLEAVED over the
//    wave's 64 tiles when this kernel will replay it (miw/film.h: log_index — [tile / 64][pixel][sample][tile % 64]): a load is 1 KB
//    of consecutive bytes, a trip's U loads 4 KB. 17.1 ms at C2 (k_film_quads<2, 4> 23.3, k_film_columns<4, 2> 26.7), the render
//    kernel's time unchanged.
// Lanes of a clipped tile (the film's last row / column of blocks) idle through the pixels their block does not have.
// The float32 additions of a texel are those of the other replay kernels (the tile's pixels in Morton order, each pixel's samples
// front to back, w = wy * wx, value * w, alpha (0 or 1) * w; a covered texel outside a sample's own footprint — a pair's second
// column — adds value * 0): the tiles are bit-identical.
// Registers: 80 sums + 2 x 16 record words + the sample's products; compiled for three wavefronts per SIMD (168 registers), what the
// compiler keeps in scratch is moved around the sample loops (once per pixel step), never inside one (tests/test_kernel_budget.py).
typedef float miw_f2 __attribute__((ext_vector_type(2)));
#define MIW_FL_BS 4
#ifndef MIW_FL_FENCE
#define MIW_FL_FENCE 0
#endif
#ifndef MIW_FL_ASM
#define MIW_FL_ASM 1
#endif
// sum += x, both halves, IN PLACE: written as the instruction so that the 40 register pairs of sums stay where they are through all
// the copies of the sample loop (as plain C++ the compiler gives every copy's sums registers of its own and moves them at each
// loop's end: 298 registers, one wavefront per SIMD)
__device__ __forceinline__ void film_pk_acc(miw_f2 &sum, const miw_f2 x) {
#if MIW_FL_ASM
    asm("v_pk_add_f32 %0, %0, %1" : "+v"(sum) : "v"(x));
#else
    sum += x;
#endif
}
// one record of the log; NT: a streaming load (the replay reads a line once per lane; its three other readers come much later)
template <int NT>
__device__ __forceinline__ uint4 film_load_rec(const U4 *p) {
    typedef uint32_t u4v_ __attribute__((ext_vector_type(4)));
    if (NT) { const u4v_ v = __builtin_nontemporal_load(reinterpret_cast<const u4v_ *>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
    const U4 t = *p;
    return make_uint4(t.x, t.y, t.z, t.w);
}
template <int R0, int R1, int P0, int P1, int U, int NT>
__device__ __forceinline__ void film_lanes_pixel(miw_f2 (&acc)[MIW_FL_BS][MIW_FL_BS / 2][MIW_FILM_CHANNELS], uint4 (&nxt)[U], uint32_t cnt,
                                                 const U4 *n_run, uint32_t n_cnt, const U4 *run, uint32_t step_max,
                                                 uint32_t bxa, uint32_t bya, uint32_t off_rej, uint32_t js) {
    constexpr uint32_t WS4 = MIW_FQ_WSTRIDE(MIW_FL_BS) * 4;
    for (uint32_t j0 = 0; j0 < step_max; j0 += (uint32_t) U) {
        const bool last_trip = j0 + (uint32_t) U >= step_max;                    // (wave-uniform) then: the next pixel's first trip
        const U4 *t_run = last_trip ? n_run : run;
        const uint32_t t_cnt = last_trip ? n_cnt : cnt, t_last = t_cnt ? t_cnt - 1u : 0u, t_j0 = last_trip ? 0u : j0 + (uint32_t) U;
        // this trip's records move out of the landing registers and the next trip's U loads are issued TOGETHER: a lane's U records
        // are 16 U contiguous bytes of its own tile's log, and the 64 lanes of a load touch 64 different cache lines — issued apart
        // (one load per consumed record) every one of them fetches its line's sector from L2 again (measured: 2 x slower than
        // k_film_quads); issued back to back the line is fetched once
        uint4 rec[U];
#pragma unroll
        for (int i = 0; i < U; ++i) rec[i] = nxt[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const uint32_t j = t_j0 + (uint32_t) i;
            nxt[i] = film_load_rec<NT>(t_run + (j < t_last ? j : t_last) * js);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const uint4 r = rec[i];
            const bool valid = j0 + (uint32_t) i < cnt;
            const uint32_t ox = valid ? (r.w & 255u) * WS4 : off_rej, oy = valid ? ((r.w >> 8) & 255u) * WS4 : off_rej;
            const float af = (r.w & 0x10000u) ? 1.f : 0.f, vx = u2f(r.x), vy = u2f(r.y), vz = u2f(r.z);
            miw_lds_cf *px = (miw_lds_cf *) (uintptr_t) (ox + bxa), *py = (miw_lds_cf *) (uintptr_t) (oy + bya);
            miw_f2 wx[MIW_FL_BS / 2];
#pragma unroll
            for (int p = P0; p <= P1; ++p) { wx[p].x = px[2 * p]; wx[p].y = px[2 * p + 1]; }
#pragma unroll
            for (int q = R0; q <= R1; ++q) {
                const float wy = py[q];
#pragma unroll
                for (int p = P0; p <= P1; ++p) {
                    const miw_f2 w = wy * wx[p];                                  // wy * wx, imageblock.cpp:155
                    film_pk_acc(acc[q][p][0], vx * w); film_pk_acc(acc[q][p][1], vy * w); film_pk_acc(acc[q][p][2], vz * w);
                    film_pk_acc(acc[q][p][3], af * w);                            // alpha (0 or 1) * w
                    film_pk_acc(acc[q][p][4], w);
                }
#if MIW_FL_FENCE
                __builtin_amdgcn_sched_barrier(0);                                // one row of one sample at a time (registers: the scheduler otherwise forms a trip's products first)
#endif
            }
        }
    }
}
#ifndef MIW_FL_WAVES
#define MIW_FL_WAVES 3
#endif
// WV = wavefronts per SIMD the kernel is compiled for: 3 (168 registers; the replay on its own) or, with U = 2, 4 (128 registers, the sample loops still free
// of scratch: tests/test_kernel_budget.py) — the form that fits BESIDE three wavefronts of the 120-register packet kernel (miwave.hip: overlap_enqueue).
template <int U, int NT, int WV = MIW_FL_WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WV, 8))) void k_film_lanes(FilmRec F, BlockReplayArgs A, PatchArgs PA /* patches_x / _y = texel blocks per tile row / column */, uint32_t n_tiles, float *tiles,
                  uint32_t group0 /* the first group of 64 tiles this launch replays (several launches when the replay runs beside the render: miwave.hip, overlap_enqueue) */,
                  uint32_t prio /* s_setprio of its wavefronts, 0 .. 3 (beside the path kernel, whose wavefronts raise theirs: resident_kernel.h tick()) */) {
    if (prio == 3u) __builtin_amdgcn_s_setprio(3); else if (prio == 2u) __builtin_amdgcn_s_setprio(2); else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
    constexpr int BS = MIW_FL_BS, WS = MIW_FQ_WSTRIDE(MIW_FL_BS), LEAD = BS - 1, LCAP = 64;
    extern __shared__ float s_w[];                           // (count + 1) x WS weights; row `count` = 0
    __shared__ unsigned short s_list[LCAP];
    const uint32_t l = threadIdx.x;
    const uint32_t n_pos = PA.patches_x * PA.patches_y;
    const uint32_t tw = blockIdx.x / n_pos + group0, pos = blockIdx.x % n_pos;
    const int tx0 = (int) (pos % PA.patches_x) * BS, ty0 = (int) (pos / PA.patches_x) * BS;      // (wave-uniform) the block inside its tile
    const uint32_t tile = tw * 64u + l;
    const bool live = tile < n_tiles;
    const BlockGeom g = block_geom(F, A.blocks_x, live ? (A.tile_list ? A.tile_list[tile] : tile) : 0u);
    const uint32_t rej = A.cls.count;                        // the zero row (LogSink16 logs rejected samples with class `count`)
    const int reach = A.cls.reach;                           // <= 2 (the host launches the other kernels otherwise)
    for (uint32_t i = l; i < (rej + 1u) * WS; i += 64u) {
        const uint32_t c = i / WS, a = i % WS;
        s_w[i] = (c < rej && a >= (uint32_t) LEAD && a <= (uint32_t) LEAD + 2u * (uint32_t) reach) ? A.cls.w[c * MIW_FC_STRIDE + a - LEAD] : 0.f;
    }
    // ---- the pixels within reach of the block, in Morton order: the same list in every tile ----
    const uint32_t bs2 = 1u << A.bs2_log2;
    uint32_t fill = 0;
    for (uint32_t q0 = 0; q0 < bs2; q0 += 64u) {
        const uint32_t q = q0 + l;
        uint32_t x, y;
        morton_decode2(q, x, y);
        const bool in = q < bs2 && (int) x >= tx0 - F.border - reach && (int) x <= tx0 + BS - 1 - F.border + reach &&
                        (int) y >= ty0 - F.border - reach && (int) y <= ty0 + BS - 1 - F.border + reach;
        const unsigned long long m = __ballot(in);
        if (in) {
            const uint32_t at = fill + (uint32_t) __popcll(m & ((1ull << l) - 1ull));
            if (at < (uint32_t) LCAP) s_list[at] = (unsigned short) q;
        }
        fill += (uint32_t) __popcll(m);
    }
    const uint32_t n_list = fill < (uint32_t) LCAP ? fill : (uint32_t) LCAP;
    __syncthreads();

    miw_f2 acc[BS][BS / 2][MIW_FILM_CHANNELS];
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int p = 0; p < BS / 2; ++p)
#pragma unroll
            for (int k = 0; k < MIW_FILM_CHANNELS; ++k) acc[r][p][k] = (miw_f2) (0.f);
    const uint32_t w_base = (uint32_t) (uintptr_t) (miw_lds_cf *) s_w;
    const uint32_t off_rej = rej * (uint32_t) (WS * 4);
    // the log: [lane][sample], or (A.log_il, path.h: log_index) the logs of the wave's 64 tiles interleaved record by record — then
    // the 64 records a load fetches are 1 KB of consecutive bytes
    const uint32_t lane0 = tile << A.bs2_log2;
    const uint32_t js = A.log_il ? 64u : 1u;
    // a step = the k-th pixel of the list, in every tile of the wave. Per lane: the pixel's run in its tile's log and its count
    struct Step { const U4 *run; uint32_t cnt; };
    auto pixel_at = [&](uint32_t k) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) (k < n_list ? (uint32_t) s_list[k] : 0u)); };
    auto step_of = [&](uint32_t k, uint32_t q) {
        Step s; s.run = A.log_rec; s.cnt = 0u;
        uint32_t x, y;
        morton_decode2(q, x, y);
        if (k < n_list && live && (int) x < g.bw && (int) y < g.bh) { s.cnt = A.st[lane0 + q].w; s.run = A.log_rec + log_index(A.log_il, lane0 + q, A.spp, 0u); }
        return s;
    };
    uint4 nxt[U];
    auto fetch = [&](const U4 *run, uint32_t cnt) {
        const uint32_t last = cnt ? cnt - 1u : 0u;
#pragma unroll
        for (int i = 0; i < U; ++i) {
            nxt[i] = film_load_rec<NT>(run + ((uint32_t) i < last ? (uint32_t) i : last) * js);
            __builtin_amdgcn_sched_barrier(0);           // record 0 first, as inside the loop
        }
    };
    uint32_t q = pixel_at(0u);
    Step cur = step_of(0u, q);
    fetch(cur.run, cur.cnt);
    for (uint32_t k = 0; k < n_list; ++k) {
        const uint32_t step_max = (uint32_t) __builtin_amdgcn_readfirstlane((int) wave_max_u32(cur.cnt));   // (a scalar for the compiler: the sample loops are uniform loops)
        const uint32_t qn = pixel_at(k + 1u);
        const Step nx = step_of(k + 1u, qn);
        uint32_t x, y;
        morton_decode2(q, x, y);
        // texel column tx0 + c takes the weight w[ax + c] of the pixel's window (0 <= ax + c <= 2 reach), rows alike
        const int ax = tx0 - ((int) x + F.border - reach), ay = ty0 - ((int) y + F.border - reach);
        const int c0 = ax < 0 ? -ax : 0, c1 = 2 * reach - ax < BS - 1 ? 2 * reach - ax : BS - 1;
        const int r0 = ay < 0 ? -ay : 0, r1 = 2 * reach - ay < BS - 1 ? 2 * reach - ay : BS - 1;
        const uint32_t bxa = w_base + 4u * (uint32_t) (LEAD + ax), bya = w_base + 4u * (uint32_t) (LEAD + ay);
        const int code = __builtin_amdgcn_readfirstlane(((r0 * 4 + r1) * 2 + (c0 >> 1)) * 2 + (c1 >> 1));
#define MIW_FL_CASE(R0, R1, P0, P1) case ((R0 * 4 + R1) * 2 + P0) * 2 + P1: \
            film_lanes_pixel<R0, R1, P0, P1, U, NT>(acc, nxt, cur.cnt, nx.run, nx.cnt, cur.run, step_max, bxa, bya, off_rej, js); break;
#define MIW_FL_ROWS(R0, R1) MIW_FL_CASE(R0, R1, 0, 0) MIW_FL_CASE(R0, R1, 0, 1) MIW_FL_CASE(R0, R1, 1, 1)
        switch (code) {
            MIW_FL_ROWS(0, 0) MIW_FL_ROWS(0, 1) MIW_FL_ROWS(0, 2) MIW_FL_ROWS(0, 3) MIW_FL_ROWS(1, 1) MIW_FL_ROWS(1, 2) MIW_FL_ROWS(1, 3)
            MIW_FL_ROWS(2, 2) MIW_FL_ROWS(2, 3) MIW_FL_ROWS(3, 3)
            default: break;
        }
#undef MIW_FL_ROWS
#undef MIW_FL_CASE
        if (step_max == 0u) fetch(nx.run, nx.cnt);                                 // (a step without samples consumed nothing)
        cur = nx; q = qn;
    }
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int c = 0; c < BS; ++c)
            if (live && tx0 + c < g.size_x && ty0 + r < g.size_y) {
                float *out = tiles + (size_t) tile * A.tile_stride + ((size_t) (ty0 + r) * g.size_x + (tx0 + c)) * MIW_FILM_CHANNELS;
#pragma unroll
                for (int k = 0; k < MIW_FILM_CHANNELS; ++k) out[k] = (c & 1) ? acc[r][c / 2][k].y : acc[r][c / 2][k].x;
            }
}

// step 2: every film texel sums the block tiles covering it, ascending block id
__global__ void k_film_merge(FilmRec F, BlockReplayArgs A, const float *tiles, float *out32, double *out64, int accumulate) {
    int fx = (int) (blockIdx.x * blockDim.x + threadIdx.x), fy = (int) blockIdx.y;
    if (fx >= F.crop_w || fy >= F.crop_h) return;
    float v[MIW_FILM_CHANNELS];
    size_t o = ((size_t) fy * F.crop_w + fx) * MIW_FILM_CHANNELS;
    if (accumulate) for (int k = 0; k < MIW_FILM_CHANNELS; ++k) v[k] = out64 ? (float) out64[o + k] : out32[o + k];
    film_merge_texel(F, A, tiles, fx, fy, v, accumulate != 0);
    for (int k = 0; k < MIW_FILM_CHANNELS; ++k) {
        if (out64) out64[o + k] = (double) v[k]; else out32[o + k] = v[k];
    }
}
