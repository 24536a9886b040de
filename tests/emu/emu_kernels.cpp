// Lane-emulation harness: compiles plz4_amd/csrc/lz4_device.inl (the SAME source hipcc builds for gfx950)
// with -DPLZ4_EMU so that the kernel logic can be checked against the oracle on a machine without a GPU.
// Test infrastructure only: built into tests/emu/_build/, never loaded by plz4_amd, not a CPU fallback.
#define PLZ4_EMU 1
#include "../../plz4_amd/csrc/lz4_device.inl"
#include "../../plz4_amd/csrc/lz4hc_device.inl"
#include <stdlib.h>

int plz4_emu_descending = 0;
static int plz4_emu_old_dict = 0;      // 1: every dictionary mode through the one-sequence-per-batch encoder (cross-check)

extern "C" {

void emu_set_descending(int d) { plz4_emu_descending = d; }
void emu_set_old_dict(int d) { plz4_emu_old_dict = d; }

int emu_encode_block(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    static thread_local uint32_t lds[plz4::kHashBytes / 4];
    return plz4::wave_encode_block(src, n, dst, cap, lds);
}

// Both builds of the decoder's vector path: the LDS-staged one the record kernels run, and the one that copies through
// memory (dictionary / raw kernels).  They must agree byte for byte; -999999 flags a disagreement.
int emu_decode_block(const uint8_t* src, int n, uint8_t* dst, int cap)
{
    static thread_local uint8_t lds[plz4::kDecLdsBytes];
    const int r1 = plz4::wave_decode_block<true>(src, n, dst, cap, nullptr, 0, lds);
    uint8_t* alt = (uint8_t*)malloc((size_t)(cap > 0 ? cap : 1) + 64);
    memset(alt, 0, (size_t)(cap > 0 ? cap : 1) + 64);
    const int r2 = plz4::wave_decode_block<false>(src, n, alt, cap);
    const bool same = (r1 == r2) && (r1 <= 0 || memcmp(dst, alt, (size_t)r1) == 0);
    free(alt);
    return same ? r1 : -999999;
}

uint32_t emu_xxh32(const uint8_t* p, int n) { return plz4::wave_xxh32(p, n); }

int emu_compress_hc(const uint8_t* src, int n, uint8_t* dst, int cap, int level)
{
    static thread_local uint8_t* ws = nullptr;
    if (!ws) ws = (uint8_t*)malloc(plz4::kHcWorkBytes);
    plz4::HcWork w;
    w.hash = (uint32_t*)ws; w.chain = (uint16_t*)(ws + plz4::kHcHashEntries * 4);
    w.opt = (plz4::HcOpt*)(ws + plz4::kHcHashEntries * 4 + plz4::kHcChainEntries * 2);
    return plz4::hc_compress(src, n, dst, cap, level, w);
}

static plz4::HcWork emu_hc_work(uint8_t* ws)
{
    plz4::HcWork w;
    w.hash = (uint32_t*)ws; w.chain = (uint16_t*)(ws + plz4::kHcHashEntries * 4);
    w.opt = (plz4::HcOpt*)(ws + plz4::kHcHashEntries * 4 + plz4::kHcChainEntries * 2);
    return w;
}

// HC with an external segment (mode 1: seg[0..segLen) is laid out right before a copy of the block, as the kernels do) or
// under a dictionary context (mode 2: tables built by hc_prime_dict, block <= 4 KiB).
int emu_compress_hc_dict(const uint8_t* src, int n, uint8_t* dst, int cap, int level, const uint8_t* seg, int segLen, int mode)
{
    static thread_local uint8_t* ws = nullptr; static thread_local uint8_t* dws = nullptr;
    if (!ws) { ws = (uint8_t*)malloc(plz4::kHcWorkBytes); dws = (uint8_t*)malloc(plz4::kHcWorkBytes); }
    plz4::HcDict d; d.mode = mode; d.len = segLen; d.bytes = nullptr; d.hash = nullptr; d.chain = nullptr;
    if (mode == plz4::kHcExt) {
        uint8_t* cat = (uint8_t*)malloc((size_t)segLen + (size_t)n + 64);
        if (segLen) memcpy(cat, seg, (size_t)segLen);
        if (n) memcpy(cat + segLen, src, (size_t)n);
        const int r = plz4::hc_compress(cat + segLen, n, dst, cap, level, emu_hc_work(ws), d);
        free(cat);
        return r;
    }
    uint8_t* dcopy = (uint8_t*)malloc((size_t)segLen + 64);
    if (segLen) memcpy(dcopy, seg, (size_t)segLen);
    const plz4::HcWork dw = emu_hc_work(dws);
    plz4::hc_prime_dict(dcopy, segLen, level, dw);
    d.bytes = dcopy; d.hash = dw.hash; d.chain = dw.chain;
    const int r = plz4::hc_compress(src, n, dst, cap, level, emu_hc_work(ws), d);
    free(dcopy);
    return r;
}

int emu_encode_block_dict(const uint8_t* src, int n, uint8_t* dst, int cap, const uint8_t* dict, int dictSize, int mode,
                          const uint32_t* dictTable)
{
    static thread_local uint32_t lds[plz4::kHashBytes / 4];
    if (mode == plz4::kDictCtxLookup || plz4_emu_old_dict) {
        plz4::DictEnc dc{dict, dictSize, mode, dictTable};
        return plz4::wave_encode_block_dict(src, n, dst, cap, dc, lds);
    }
    // as the kernels do: the segment is laid out right before (a copy of) the block
    const bool seg = (mode == plz4::kDictLoad || mode == plz4::kDictCtxCopy);
    const int segLen = seg ? dictSize : 0;
    uint8_t* cat = (uint8_t*)malloc((size_t)segLen + (size_t)n + 64);
    if (segLen) memcpy(cat, dict, (size_t)segLen);
    if (n) memcpy(cat + segLen, src, (size_t)n);
    memset(cat + segLen + n, 0, 64);
    const int r = plz4::wave_encode_block_ext(cat + segLen, n, dst, cap, mode, segLen, dictTable, lds);
    free(cat);
    return r;
}

// With a dictionary too, both builds of the vector path must agree (the record kernels run the LDS-staged one).
int emu_decode_block_dict(const uint8_t* src, int n, uint8_t* dst, int cap, const uint8_t* dict, int dictSize)
{
    static thread_local uint8_t lds[plz4::kDecLdsBytes];
    const int r1 = plz4::wave_decode_block<true>(src, n, dst, cap, dict, dictSize, lds);
    uint8_t* alt = (uint8_t*)malloc((size_t)(cap > 0 ? cap : 1) + 64);
    memset(alt, 0, (size_t)(cap > 0 ? cap : 1) + 64);
    const int r2 = plz4::wave_decode_block<false>(src, n, alt, cap, dict, dictSize);
    const bool same = (r1 == r2) && (r1 <= 0 || memcmp(dst, alt, (size_t)r1) == 0);
    free(alt);
    return same ? r1 : -999999;
}

}
