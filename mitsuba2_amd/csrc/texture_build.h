// Host-side preparation of texture data for the scene tables (shared by mi_scene_upload and the CPU checker).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../include/miwave.h"
#include "miw/bsdf.h"

namespace miw {

// Per-face texture coordinates (u0 v0 u1 v1 u2 v2, face order == global primitive id) of the shapes that carry
// MI_SHAPE_HAS_TEXCOORDS; faces of other shapes stay zero. Empty when no shape has any. Returns false on a face
// whose vertex index is out of range or when the flag is set without a vertex_texcoords array.
inline bool build_face_texcoords(const mi_scene_desc *s, std::vector<float> &out) {
    out.clear();
    bool any = false;
    for (uint32_t i = 0; i < s->shape_count; ++i) any = any || (s->shapes[i].flags & MI_SHAPE_HAS_TEXCOORDS);
    if (!any) return true;
    if (!s->vertex_texcoords) return false;
    out.assign((size_t) s->face_count * 6, 0.f);
    for (uint32_t i = 0; i < s->shape_count; ++i) {
        const mi_shape &sh = s->shapes[i];
        if (!(sh.flags & MI_SHAPE_HAS_TEXCOORDS) || (sh.flags & (MI_SHAPE_RECTANGLE | MI_SHAPE_SPHERE))) continue;
        for (uint32_t f = sh.first_face; f < sh.first_face + sh.face_count; ++f)
            for (int k = 0; k < 3; ++k) {
                const uint32_t vi = s->faces[3 * (size_t) f + k];
                if (vi >= s->vertex_count) return false;
                std::memcpy(&out[(size_t) f * 6 + 2 * k], s->vertex_texcoords + 2 * (size_t) vi, 8);
            }
    }
    return true;
}

// mi_bitmap[] -> BitmapRec[] (data pointers still the caller's). nullptr, or what is wrong with entry *bad.
inline const char *build_bitmap_table(const mi_scene_desc *s, std::vector<BitmapRec> &out, uint32_t *bad) {
    out.clear();
    if (s->bitmap_count && !s->bitmaps) { *bad = 0; return "bitmap_count without bitmaps"; }
    for (uint32_t i = 0; i < s->bitmap_count; ++i) {
        const mi_bitmap &b = s->bitmaps[i];
        *bad = i;
        if (!b.data) return "null data pointer";
        if (b.width < 2 || b.height < 2 || b.width > 32768 || b.height > 32768) return "size must be 2 .. 32768 texels a side";
        if (b.channels != 1 && b.channels != 3) return "Unsupported channel count (expected 1 or 3)";
        if (b.filter_type > MI_BITMAP_BILINEAR) return "Invalid filter type, must be one of: \"nearest\", or \"bilinear\"!";
        if (b.wrap_mode > MI_BITMAP_CLAMP) return "Invalid wrap mode, must be one of: \"repeat\", \"mirror\", or \"clamp\"!";
        BitmapRec r; std::memset(&r, 0, sizeof r);
        r.data = b.data; r.width = b.width; r.height = b.height; r.channels = b.channels;
        r.filter = b.filter_type; r.wrap = b.wrap_mode; std::memcpy(r.to_uv, b.to_uv, sizeof r.to_uv);
        out.push_back(r);
    }
    return nullptr;
}

// mi_bsdf -> BsdfRec: texture slots from the record's tex[] (scalar_spectral) or from the RGB layout of params[]
// (scalar_rgb), a MI_TEX_BITMAP slot taken as it is in either library. nullptr, or what is wrong with slot *bad.
inline const char *bsdf_record_from_abi(const mi_bsdf &b, uint32_t bitmap_count, uint32_t table_floats, BsdfRec &r, int *bad) {
    std::memset(&r, 0, sizeof r);
    r.type = b.type; r.flags = b.flags; r.back = b.back;
    std::memcpy(r.p, b.params, sizeof r.p);
    *bad = -1;
    if (b.type >= BSDF_TYPE_COUNT) return "unknown type";
    if (b.type == BSDF_TYPE_ROUGHPLASTIC) {
        const float o = b.params[5];
        if (!(o >= 0.f) || o != (float) (uint32_t) o || (uint64_t) o + MI_ROUGH_TRANSMITTANCE_RES > table_floats)
            return "roughplastic: transmittance table outside mi_scene_desc::bsdf_tables";
    }
    const int off[7][3] = { { 0, -1, -1 }, { 1, 4, -1 }, { 2, 5, 8 }, { 2, 5, 8 }, { 4, 7, -1 }, { 4, 7, -1 }, { 6, 9, -1 } };   // RGB layout of params[]
    for (int k = 0; k < 3; ++k) {
        *bad = k;
        const bool used = k < (int) bsdf_tex_slots(b.type);
        if (used && b.tex[k].type == MI_TEX_BITMAP) {
            const float idx = b.tex[k].v[0];
            if (!(idx >= 0.f) || idx >= (float) bitmap_count || idx != (float) (uint32_t) idx) return "bitmap index out of range";
            std::memcpy(&r.tex[k], &b.tex[k], sizeof(TexRec));
            continue;
        }
#if MIW_SPECTRAL
        if (!used) continue;
        if (b.tex[k].type == MI_TEX_RGB || b.tex[k].type > MI_TEX_SRGB_D65) return "the scalar_spectral library needs a spectral texture record";
        std::memcpy(&r.tex[k], &b.tex[k], sizeof(TexRec));
#else
        r.tex[k].type = TEX_RGB;
        if (off[b.type][k] >= 0) std::memcpy(r.tex[k].v, b.params + off[b.type][k], 12);
        if (b.tex[k].type != MI_TEX_RGB) return "spectral texture record passed to the scalar_rgb library";
#endif
    }
    (void) off;
    return nullptr;
}

// true when a record of the table reads a bitmap
// true when the record needs an "extended" kernel (plugins the plain kernels compile out)
inline bool bsdf_is_extended(const BsdfRec &r) { return r.type == BSDF_TYPE_ROUGHPLASTIC; }
inline bool bsdf_uses_bitmap(const BsdfRec &r) {
    for (uint32_t k = 0; k < bsdf_tex_slots(r.type); ++k) if (r.tex[k].type == TEX_BITMAP) return true;
    return false;
}

} // namespace miw
