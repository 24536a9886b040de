/* TEST INFRASTRUCTURE: a block engine for the C++ host layer (plz4_amd/csrc/host) backed by the oracle, so that
 * framing / flush / error-latching logic can be tested on a machine without a GPU.  It is plugged in through the
 * host layer's public plug-in point (plz4h_engine_vtable); it is compiled into tests/hostlib/_build/ only and is
 * never part of plz4_amd/libplz4hip.so. */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "../../oracle/plz4_oracle.h"

/* ORACLE_ENGINE_ANY_LEVEL=1: answer every level with the level-1 encoder, so that the host layer's handling of levels > 1
 * (e.g. linked HC frames have no stored-block fallback) can be exercised without a GPU.  Test switch only. */
static int any_level(void) { const char* v = getenv("ORACLE_ENGINE_ANY_LEVEL"); return v && v[0] == '1'; }

static int o_compress(void* u, int n, const void* const* src, const int32_t* sl, void* const* dst, const int32_t* dc, int level, int32_t* res)
{
    (void)u;
    if (level != 1 && !any_level()) return -4;
    for (int i = 0; i < n; i++) res[i] = orc_compress_fast((const uint8_t*)src[i], sl[i], (uint8_t*)dst[i], dc[i]);
    return 0;
}
static int o_decompress(void* u, int n, const void* const* src, const int32_t* sl, void* const* dst, const int32_t* dc, int32_t* res)
{
    (void)u;
    for (int i = 0; i < n; i++) res[i] = orc_decompress_safe((const uint8_t*)src[i], sl[i], (uint8_t*)dst[i], dc[i]);
    return 0;
}
static int o_encode_records(void* u, int n, const void* const* src, const int32_t* sl, int bsz, int level, int bc, void* const* rec, int32_t* rl)
{
    (void)u;
    if (level != 1 && !any_level()) return -4;
    for (int i = 0; i < n; i++) rl[i] = orc_block_record((const uint8_t*)src[i], sl[i], bsz, bc, (uint8_t*)rec[i]);
    return 0;
}
static int o_decode_records(void* u, int n, const void* const* rec, const int32_t* rl, int bsz, int bc, void* const* dst, int32_t* res, int32_t* st)
{
    (void)u;
    for (int i = 0; i < n; i++) {
        const uint8_t* r = (const uint8_t*)rec[i];
        uint32_t word; memcpy(&word, r, 4);
        const int sz = (int)(word & 0x7FFFFFFFu);
        res[i] = 0; st[i] = 0;
        if (sz > bsz || sz + 4 + (bc ? 4 : 0) > rl[i]) { st[i] = 2; continue; }
        if (bc) { uint32_t want; memcpy(&want, r + 4 + sz, 4); if (orc_xxh32(r + 4, (size_t)sz) != want) { st[i] = 1; continue; } }
        if (word & 0x80000000u) { memcpy(dst[i], r + 4, (size_t)sz); res[i] = sz; }
        else { res[i] = orc_decompress_safe(r + 4, sz, (uint8_t*)dst[i], bsz + 8); if (res[i] < 0) st[i] = 3; }
    }
    return 0;
}

/* ---- dictionaries / linked blocks: the oracle's stream emulation of clz4.DictCtx / StreamIndieCtx / StreamLinkedCtx */
#include <stdlib.h>
typedef struct { orc_stream ctx; uint8_t* bytes; int len; } odict;

static void* o_dict_create(void* u, const uint8_t* d, int n)
{
    (void)u;
    odict* x = (odict*)calloc(1, sizeof(odict));
    x->bytes = (uint8_t*)malloc((size_t)n + 1); memcpy(x->bytes, d, (size_t)n); x->len = n;
    orc_stream_init(&x->ctx); orc_stream_reset_fast(&x->ctx);
    orc_stream_load_dict(&x->ctx, x->bytes, n, 1);
    return x;
}
static void o_dict_destroy(void* u, void* d) { (void)u; if (d) { free(((odict*)d)->bytes); free(d); } }

static int indie_dict(const odict* d, const uint8_t* src, int n, uint8_t* dst, int cap)
{
    orc_stream s; orc_stream_init(&s); orc_stream_reset_fast(&s);
    orc_stream_attach(&s, &d->ctx);
    return orc_stream_compress(&s, src, n, dst, cap);
}
static int o_compress_dict(void* u, int n, const void* const* src, const int32_t* sl, void* const* dst, const int32_t* dc, int level, void* d, int32_t* res)
{
    (void)u; if (level != 1 && !any_level()) return -4;
    for (int i = 0; i < n; i++) res[i] = indie_dict((const odict*)d, (const uint8_t*)src[i], sl[i], (uint8_t*)dst[i], dc[i]);
    return 0;
}
static int o_decompress_dict(void* u, int n, const void* const* src, const int32_t* sl, void* const* dst, const int32_t* dc, void* d, int32_t* res)
{
    (void)u; const odict* x = (const odict*)d;
    const int k = x->len > 65536 ? 65536 : x->len;
    for (int i = 0; i < n; i++) res[i] = orc_decompress_safe_dict((const uint8_t*)src[i], sl[i], (uint8_t*)dst[i], dc[i], x->bytes + (x->len - k), k);
    return 0;
}
static int frame_record(int c, const uint8_t* src, int n, int bc, uint8_t* rec)
{
    uint32_t word = (uint32_t)c & 0x7FFFFFFFu;
    if (c == 0) { memcpy(rec + 4, src, (size_t)n); c = n; word = 0x80000000u | (uint32_t)n; }
    memcpy(rec, &word, 4);
    if (bc) { uint32_t x = orc_xxh32(rec + 4, (size_t)c); memcpy(rec + 4 + c, &x, 4); return c + 8; }
    return c + 4;
}
static int o_encode_ex(void* u, int n, const void* const* src, const int32_t* sl, int bsz, int level, int bc, int linked, void* d,
                       const void* prevTail, int prevTailLen, void* const* rec, int32_t* rl)
{
    (void)u; if (level != 1 && !any_level()) return -4;
    for (int i = 0; i < n; i++) {
        const uint8_t* s = (const uint8_t*)src[i];
        uint8_t* r = (uint8_t*)rec[i];
        int c;
        const uint8_t* tail = 0; int tl = -1;
        if (linked && i > 0) { tl = sl[i - 1] < 65536 ? sl[i - 1] : 65536; tail = (const uint8_t*)src[i - 1] + (sl[i - 1] - tl); }
        else if (linked && prevTail && prevTailLen >= 0) { tail = (const uint8_t*)prevTail; tl = prevTailLen; }
        if (tl >= 0) {
            uint8_t* copy = (uint8_t*)malloc((size_t)tl + 1); memcpy(copy, tail, (size_t)tl);      /* a separate buffer, like the pooled dict block */
            orc_stream st; orc_stream_init(&st); orc_stream_reset_fast(&st);
            orc_stream_load_dict(&st, copy, tl, 0);
            c = orc_stream_compress(&st, s, sl[i], r + 4, bsz);
            free(copy);
        } else if (d) {
            c = indie_dict((const odict*)d, s, sl[i], r + 4, bsz);
        } else {
            orc_stream st; orc_stream_init(&st); orc_stream_reset_fast(&st);
            c = orc_stream_compress(&st, s, sl[i], r + 4, bsz);
        }
        rl[i] = frame_record(c, s, sl[i], bc, r);
    }
    return 0;
}
static int o_decode_ex(void* u, int n, const void* const* rec, const int32_t* rl, int bsz, int bc, int linked, void* d,
                       void* window, int* windowLen, void* const* dst, int32_t* res, int32_t* st)
{
    (void)u;
    const odict* x = (const odict*)d;
    int dead = 0;
    for (int i = 0; i < n; i++) {
        const uint8_t* r = (const uint8_t*)rec[i];
        uint32_t word; memcpy(&word, r, 4);
        const int sz = (int)(word & 0x7FFFFFFFu);
        res[i] = 0; st[i] = 3;
        if (dead) continue;
        st[i] = 0;
        if (sz > bsz || sz + 4 + (bc ? 4 : 0) > rl[i]) { st[i] = 2; dead = linked; continue; }
        if (bc) { uint32_t want; memcpy(&want, r + 4 + sz, 4); if (orc_xxh32(r + 4, (size_t)sz) != want) { st[i] = 1; dead = linked; continue; } }
        if (word & 0x80000000u) { memcpy(dst[i], r + 4, (size_t)sz); res[i] = sz; continue; }       /* stored: no window update */
        const uint8_t* dict = 0; int dl = 0;
        if (linked) { dict = (const uint8_t*)window; dl = *windowLen; }
        else if (x) { dl = x->len > 65536 ? 65536 : x->len; dict = x->bytes + (x->len - dl); }
        res[i] = orc_decompress_safe_dict(r + 4, sz, (uint8_t*)dst[i], bsz + 8, dict, dl);
        if (res[i] < 0) { st[i] = 3; dead = linked; continue; }
        if (linked) {                                                                               /* compress/dict.go:28-41 */
            uint8_t* w = (uint8_t*)window; int wl = *windowLen; const int m = res[i]; const uint8_t* o = (const uint8_t*)dst[i];
            if (m >= 65536) { memcpy(w, o + (m - 65536), 65536); wl = 65536; }
            else { if (wl + m > 65536) { const int extra = wl + m - 65536; memmove(w, w + extra, (size_t)(wl - extra)); wl -= extra; } memcpy(w + wl, o, (size_t)m); wl += m; }
            *windowLen = wl;
        }
    }
    return 0;
}

struct vt { void* user; void* f[10]; };
void oracle_engine_vtable(struct vt* out)
{
    out->user = 0;
    out->f[0] = (void*)o_compress; out->f[1] = (void*)o_decompress;
    out->f[2] = (void*)o_encode_records; out->f[3] = (void*)o_decode_records;
    out->f[4] = (void*)o_dict_create; out->f[5] = (void*)o_dict_destroy;
    out->f[6] = (void*)o_compress_dict; out->f[7] = (void*)o_decompress_dict;
    out->f[8] = (void*)o_encode_ex; out->f[9] = (void*)o_decode_ex;
}
