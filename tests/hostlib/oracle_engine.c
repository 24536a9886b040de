/* TEST INFRASTRUCTURE: a block engine for the C++ host layer (plz4_amd/csrc/host) backed by the oracle, so that
 * framing / flush / error-latching logic can be tested on a machine without a GPU.  It is plugged in through the
 * host layer's public plug-in point (plz4h_engine_vtable); it is compiled into tests/hostlib/_build/ only and is
 * never part of plz4_amd/libplz4hip.so. */
#include <stdint.h>
#include <string.h>
#include "../../oracle/plz4_oracle.h"

static int o_compress(void* u, int n, const void* const* src, const int32_t* sl, void* const* dst, const int32_t* dc, int level, int32_t* res)
{
    (void)u;
    if (level != 1) return -4;
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
    if (level != 1) return -4;
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

struct vt { void* user; void* f[4]; };
void oracle_engine_vtable(struct vt* out)
{
    out->user = 0;
    out->f[0] = (void*)o_compress; out->f[1] = (void*)o_decompress;
    out->f[2] = (void*)o_encode_records; out->f[3] = (void*)o_decode_records;
}
