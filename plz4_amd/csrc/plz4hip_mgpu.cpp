// plz4hip_mgpu.cpp -- section D of include/plz4hip.h: several GPUs of one node behind one handle.
//
// The reference spreads the blocks of a stream over NParallel worker goroutines (internal/pkg/async/writer.go:232-282
// compressLoop, :439-467 kickoffAsync) and puts their records back in order (:284-381 writeLoop).  Here the "workers" are the
// GPUs: block i of a call goes to device i mod G (BASELINE.json north_star), every device runs its share through its own
// plz4hip_ctx, and the results come back in block order.
//   * host buffers (what the cgo shim calls): one host thread per device drives that device's ctx; records land in the caller's
//     buffers, so "assembling the frame" is the caller writing rec[0], rec[1], ... as it does today.
//   * device-resident shards: each device encodes its shard; the frame body is assembled on the owner device -- sizes to the
//     host, offsets by one prefix sum, then every other device's compacted records are copied peer to peer (xGMI) in pieces of
//     at most kPiece bytes into one of two scratch buffers on the owner and moved to their place by the record mover while
//     the next piece travels.  The owner never holds more than the frame body + two pieces.
// Only the public C ABI and the HIP runtime are used here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/plz4hip.h"

namespace {

struct DevBuf { void* p = nullptr; size_t cap = 0; };

struct Shard {
    int device = 0;
    plz4hip_ctx* ctx = nullptr;
    hipStream_t stream = nullptr, copyStream = nullptr;     // copyStream: peer copies into this device, beside the record mover
    DevBuf stage, body, recLen, recOff, srcOff, dstOff, len, pack;      // device scratch, grown on demand
    hipEvent_t ev[2] = {nullptr, nullptr};
};

struct DevGuard {
    int prev = -1;
    explicit DevGuard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; hipSetDevice(d); }
    ~DevGuard() { if (prev >= 0) hipSetDevice(prev); }
};

}  // namespace

struct plz4hip_mgpu {
    std::vector<Shard> sh;
    std::mutex mu;                      // the device-resident frame calls: they share the shards' scratch buffers and streams
    std::mutex errMu;                   // guards err alone (shard threads report concurrently)
    std::string err;
    std::atomic<unsigned> nextStart{0}; // host-buffer calls: the shard that takes a call's block 0 rotates from call to call
    DevBuf scratch[2];                  // on the owner device of the last frame call
    int scratchDev = -1;
    hipEvent_t scratchFree[2] = {nullptr, nullptr};
};

namespace {

size_t piece_bytes() { if (const char* v = getenv("PLZ4HIP_MGPU_PIECE_KB")) { const long kb = atol(v); if (kb > 0) return (size_t)kb << 10; } return (size_t)256 << 20; }

int mfail(plz4hip_mgpu* m, int code, const char* what, hipError_t e = hipSuccess)
{
    if (m) {
        char buf[512];
        if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else snprintf(buf, sizeof buf, "%s", what);
        std::lock_guard<std::mutex> g(m->errMu);
        m->err = buf;
    }
    return code;
}
#define MCHK(m, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return mfail((m), PLZ4HIP_E_DEVICE, #call, e_); } while (0)

int grow(plz4hip_mgpu* m, int device, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return PLZ4HIP_OK;
    DevGuard g(device);
    if (b.p) hipFree(b.p);
    b.p = nullptr; b.cap = 0;
    const size_t want = bytes + (bytes >> 3) + 256;
    if (hipMalloc(&b.p, want) != hipSuccess) return mfail(m, PLZ4HIP_E_NOMEM, "plz4hip_mgpu: device allocation");
    b.cap = want;
    return PLZ4HIP_OK;
}

// blocks of the call that belong to shard k: k, k + G, ...
inline int shard_blocks(int nBlocks, int G, int k) { return nBlocks > k ? (nBlocks - k + G - 1) / G : 0; }

// run fn(k) for every shard on its own host thread; the first failing shard (lowest k) wins.  Each thread keeps its own error
// text -- what its ctx reported right after the failing call, on the thread that made it (plz4hip_last_error is per thread) --
// and the handle's message is set once, after the join.
template <class F> int per_shard(plz4hip_mgpu* m, F&& fn)
{
    const int G = (int)m->sh.size();
    std::vector<int> rc((size_t)G, PLZ4HIP_OK);
    std::vector<std::string> msg((size_t)G);
    auto run = [&](int k) {
        rc[(size_t)k] = fn(k);
        if (rc[(size_t)k] != PLZ4HIP_OK) msg[(size_t)k] = plz4hip_last_error(m->sh[(size_t)k].ctx);
    };
    std::vector<std::thread> th;
    for (int k = 1; k < G; ++k) th.emplace_back(run, k);
    run(0);
    for (auto& t : th) t.join();
    for (int k = 0; k < G; ++k) if (rc[(size_t)k] != PLZ4HIP_OK) {
        std::lock_guard<std::mutex> g(m->errMu);
        if (!msg[(size_t)k].empty() || m->err.empty()) m->err = std::string("device ") + std::to_string(m->sh[(size_t)k].device) + ": " + msg[(size_t)k];
        return rc[(size_t)k];
    }
    return PLZ4HIP_OK;
}

}  // namespace

extern "C" {

int plz4hip_mgpu_create(const int* devices, int nDevices, plz4hip_mgpu** out)
{
    if (!out || nDevices < 1 || nDevices > 64 || !devices) return PLZ4HIP_E_ARG;
    *out = nullptr;
    plz4hip_mgpu* m = new (std::nothrow) plz4hip_mgpu();
    if (!m) return PLZ4HIP_E_NOMEM;
    m->sh.resize((size_t)nDevices);
    int rc = PLZ4HIP_OK;
    for (int k = 0; k < nDevices && rc == PLZ4HIP_OK; ++k) {
        Shard& s = m->sh[(size_t)k];
        s.device = devices[k];
        rc = plz4hip_ctx_create(s.device, &s.ctx);
        if (rc != PLZ4HIP_OK) break;
        DevGuard g(s.device);
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&s.copyStream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev[1], hipEventDisableTiming) != hipSuccess) rc = PLZ4HIP_E_DEVICE;
        for (int j = 0; j < k; ++j) {                   // peer access both ways (a no-op where it is on already, or the same device)
            const int o = m->sh[(size_t)j].device;
            if (o == s.device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, s.device, o) == hipSuccess && can) { hipError_t e = hipDeviceEnablePeerAccess(o, 0); if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(); }
            DevGuard g2(o);
            if (hipDeviceCanAccessPeer(&can, o, s.device) == hipSuccess && can) { hipError_t e = hipDeviceEnablePeerAccess(s.device, 0); if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(); }
        }
    }
    if (rc != PLZ4HIP_OK) { plz4hip_mgpu_destroy(m); return rc; }
    *out = m;
    return PLZ4HIP_OK;
}

void plz4hip_mgpu_destroy(plz4hip_mgpu* m)
{
    if (!m) return;
    for (Shard& s : m->sh) {
        DevGuard g(s.device);
        if (s.stream) { hipStreamSynchronize(s.stream); hipStreamDestroy(s.stream); }
        if (s.copyStream) { hipStreamSynchronize(s.copyStream); hipStreamDestroy(s.copyStream); }
        for (DevBuf* b : {&s.stage, &s.body, &s.recLen, &s.recOff, &s.srcOff, &s.dstOff, &s.len, &s.pack}) if (b->p) hipFree(b->p);
        for (hipEvent_t e : s.ev) if (e) hipEventDestroy(e);
        if (s.ctx) plz4hip_ctx_destroy(s.ctx);
    }
    if (m->scratchDev >= 0) {
        DevGuard g(m->scratchDev);
        for (DevBuf& b : m->scratch) if (b.p) hipFree(b.p);
        for (hipEvent_t e : m->scratchFree) if (e) hipEventDestroy(e);
    }
    delete m;
}

int plz4hip_mgpu_count(const plz4hip_mgpu* m) { return m ? (int)m->sh.size() : PLZ4HIP_E_ARG; }
plz4hip_ctx* plz4hip_mgpu_ctx(plz4hip_mgpu* m, int k) { return (m && k >= 0 && k < (int)m->sh.size()) ? m->sh[(size_t)k].ctx : nullptr; }
const char* plz4hip_mgpu_last_error(const plz4hip_mgpu* m)
{
    if (!m) return "null handle";
    static thread_local std::string copy;              // the caller's own copy: valid until this thread asks again
    std::lock_guard<std::mutex> g(const_cast<plz4hip_mgpu*>(m)->errMu);
    copy = m->err;
    return copy.c_str();
}

// ---------------------------------------------------------------------------------------- host buffers
// mode 0 encode_records, 1 decode_records, 2 compress_batch, 3 decompress_batch
static int host_deal(plz4hip_mgpu* m, int mode, int nBlocks, const void* const* in, const int32_t* inLen, void* const* out,
                     const int32_t* outCap, int bsz, int level, int blockChecksum, int32_t* result, int32_t* status)
{
    if (!m || nBlocks < 0 || (nBlocks && (!in || !inLen || !out || !result))) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu: bad argument");
    if (nBlocks == 0) return PLZ4HIP_OK;
    // No handle-wide lock: every ctx serialises its own calls, so concurrent callers (the reference's worker goroutines each call
    // Compress for one block, async/writer.go:232-282) only wait for each other per device.  The deal starts at a shard that
    // rotates from call to call -- block i of this call goes to shard (start + i) mod G -- so that a stream of one-block calls
    // uses every GPU instead of queueing on device 0.  Results land in the caller's arrays in block order either way.
    const int G = (int)m->sh.size();
    const int start = (int)(m->nextStart.fetch_add((unsigned)(nBlocks < G ? nBlocks : G)) % (unsigned)G);
    return per_shard(m, [&](int sk) -> int {
        const int k = (sk - start + G) % G;                     // this shard takes blocks k, k + G, ... of the call
        const int nb = shard_blocks(nBlocks, G, k);
        if (!nb) return PLZ4HIP_OK;
        std::vector<const void*> i2((size_t)nb); std::vector<void*> o2((size_t)nb);
        std::vector<int32_t> l2((size_t)nb), c2((size_t)nb), r2((size_t)nb), s2((size_t)nb);
        for (int j = 0; j < nb; ++j) { const int i = k + j * G; i2[(size_t)j] = in[i]; l2[(size_t)j] = inLen[i]; o2[(size_t)j] = out[i]; if (outCap) c2[(size_t)j] = outCap[i]; }
        plz4hip_ctx* c = m->sh[(size_t)sk].ctx;
        int rc;
        switch (mode) {
        case 0:  rc = plz4hip_encode_records(c, nb, i2.data(), l2.data(), bsz, level, blockChecksum, o2.data(), r2.data()); break;
        case 1:  rc = plz4hip_decode_records(c, nb, i2.data(), l2.data(), bsz, blockChecksum, o2.data(), r2.data(), s2.data()); break;
        case 2:  rc = plz4hip_compress_batch(c, nb, i2.data(), l2.data(), o2.data(), c2.data(), level, r2.data()); break;
        default: rc = plz4hip_decompress_batch(c, nb, i2.data(), l2.data(), o2.data(), c2.data(), r2.data()); break;
        }
        if (rc != PLZ4HIP_OK) return rc;
        for (int j = 0; j < nb; ++j) { const int i = k + j * G; result[i] = r2[(size_t)j]; if (status) status[i] = s2[(size_t)j]; }
        return PLZ4HIP_OK;
    });
}

int plz4hip_mgpu_encode_records(plz4hip_mgpu* m, int nBlocks, const void* const* src, const int32_t* srcLen, int bsz, int level,
                                int blockChecksum, void* const* rec, int32_t* recLen)
{ return host_deal(m, 0, nBlocks, src, srcLen, rec, nullptr, bsz, level, blockChecksum, recLen, nullptr); }

int plz4hip_mgpu_decode_records(plz4hip_mgpu* m, int nBlocks, const void* const* rec, const int32_t* recLen, int bsz,
                                int blockChecksum, void* const* dst, int32_t* result, int32_t* status)
{
    if (nBlocks > 0 && !status) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_decode_records: status");
    return host_deal(m, 1, nBlocks, rec, recLen, dst, nullptr, bsz, 1, blockChecksum, result, status);
}

int plz4hip_mgpu_compress_batch(plz4hip_mgpu* m, int nBlocks, const void* const* src, const int32_t* srcLen, void* const* dst,
                                const int32_t* dstCap, int level, int32_t* result)
{
    if (nBlocks > 0 && !dstCap) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_compress_batch: dstCap");
    return host_deal(m, 2, nBlocks, src, srcLen, dst, dstCap, 0, level, 0, result, nullptr);
}

int plz4hip_mgpu_decompress_batch(plz4hip_mgpu* m, int nBlocks, const void* const* src, const int32_t* srcLen, void* const* dst,
                                  const int32_t* dstCap, int32_t* result)
{
    if (nBlocks > 0 && !dstCap) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_decompress_batch: dstCap");
    return host_deal(m, 3, nBlocks, src, srcLen, dst, dstCap, 0, 1, 0, result, nullptr);
}

// ---------------------------------------------------------------------------------------- device-resident shards
int plz4hip_mgpu_dev_encode_frame(plz4hip_mgpu* m, int nBlocks, const void* const* shardSrc, const int64_t* shardBytes, int bsz, int level,
                                  int blockChecksum, int owner, void* body, int64_t bodyCap, int64_t* recOff, int64_t* bodyBytes)
{
    if (!m || nBlocks < 0 || bsz <= 0 || !shardSrc || !shardBytes || !recOff || !bodyBytes || (nBlocks && !body)) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_dev_encode_frame: bad argument");
    const int G = (int)m->sh.size();
    if (owner < 0 || owner >= G) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_dev_encode_frame: owner");
    recOff[0] = 0; *bodyBytes = 0;
    if (nBlocks == 0) return PLZ4HIP_OK;
    std::lock_guard<std::mutex> g(m->mu);
    const int64_t stride = plz4hip_dev_stage_stride(bsz);
    std::vector<std::vector<int32_t>> lens((size_t)G);
    std::vector<std::vector<int64_t>> loff((size_t)G);
    // 1. every device encodes and compacts its shard
    int rc = per_shard(m, [&](int k) -> int {
        Shard& s = m->sh[(size_t)k];
        const int nb = shard_blocks(nBlocks, G, k);
        if (!nb) return PLZ4HIP_OK;
        const int64_t need = (int64_t)(nb - 1) * bsz;
        if (shardBytes[k] <= need || shardBytes[k] > need + bsz) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_dev_encode_frame: shardBytes does not match the block count");
        if (int r = grow(m, s.device, s.stage, (size_t)nb * (size_t)stride)) return r;
        if (int r = grow(m, s.device, s.body, (size_t)nb * ((size_t)bsz + 8))) return r;
        if (int r = grow(m, s.device, s.recLen, (size_t)nb * 4)) return r;
        if (int r = grow(m, s.device, s.recOff, (size_t)(nb + 1) * 8)) return r;
        if (int r = plz4hip_dev_encode_records(s.ctx, shardSrc[k], shardBytes[k], bsz, level, blockChecksum, s.stage.p, (int32_t*)s.recLen.p, s.stream)) return r;
        if (int r = plz4hip_dev_compact_records(s.ctx, s.stage.p, stride, (const int32_t*)s.recLen.p, nb, (int64_t*)s.recOff.p, s.body.p, (int64_t)s.body.cap, s.stream)) return r;
        DevGuard dg(s.device);
        lens[(size_t)k].resize((size_t)nb); loff[(size_t)k].resize((size_t)nb + 1);
        if (hipMemcpyAsync(lens[(size_t)k].data(), s.recLen.p, (size_t)nb * 4, hipMemcpyDeviceToHost, s.stream) != hipSuccess) return PLZ4HIP_E_DEVICE;
        if (hipMemcpyAsync(loff[(size_t)k].data(), s.recOff.p, (size_t)(nb + 1) * 8, hipMemcpyDeviceToHost, s.stream) != hipSuccess) return PLZ4HIP_E_DEVICE;
        if (hipStreamSynchronize(s.stream) != hipSuccess) return PLZ4HIP_E_DEVICE;
        return PLZ4HIP_OK;
    });
    if (rc != PLZ4HIP_OK) return rc;
    // 2. the frame's offsets: one prefix sum over the records in stream order (block i is block i / G of shard i mod G)
    for (int i = 0; i < nBlocks; ++i) recOff[i + 1] = recOff[i] + lens[(size_t)(i % G)][(size_t)(i / G)];
    *bodyBytes = recOff[nBlocks];
    if (recOff[nBlocks] > bodyCap) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_dev_encode_frame: body too small");
    // 3. interleave on the owner: its own records straight from its compacted body, the others' through two scratch pieces
    Shard& o = m->sh[(size_t)owner];
    DevGuard dg(o.device);
    if (m->scratchDev != o.device) {
        if (m->scratchDev >= 0) { DevGuard g2(m->scratchDev); for (DevBuf& b : m->scratch) { if (b.p) hipFree(b.p); b.p = nullptr; b.cap = 0; } for (hipEvent_t& e : m->scratchFree) { if (e) hipEventDestroy(e); e = nullptr; } }
        m->scratchDev = o.device;
    }
    for (int q = 0; q < 2; ++q) if (!m->scratchFree[q]) MCHK(m, hipEventCreateWithFlags(&m->scratchFree[q], hipEventDisableTiming));
    int piece = 0;
    for (int k = 0; k < G; ++k) {
        Shard& s = m->sh[(size_t)k];
        const int nb = shard_blocks(nBlocks, G, k);
        if (!nb) continue;
        // (PLZ4HIP_MGPU_FORCE_PEER, tests on a one-GPU box: every entry but the owner itself goes through the scratch pieces)
        const bool local = getenv("PLZ4HIP_MGPU_FORCE_PEER") ? (k == owner) : (s.device == o.device);
        // pieces of shard k: records [cut[t], cut[t+1]) -- as many as fit kPiece (all of them when they are read in place)
        const size_t pieceMax = piece_bytes();
        std::vector<int> cut(1, 0);
        if (local) cut.push_back(nb);
        else for (int j0 = 0; j0 < nb; ) { int j1 = j0 + 1; while (j1 < nb && (size_t)(loff[(size_t)k][(size_t)j1 + 1] - loff[(size_t)k][(size_t)j0]) <= pieceMax) ++j1; cut.push_back(j1); j0 = j1; }
        std::vector<int64_t> so((size_t)nb), dd((size_t)nb); std::vector<int32_t> ll((size_t)nb);
        for (size_t t = 0; t + 1 < cut.size(); ++t) {
            const int64_t base = local ? 0 : loff[(size_t)k][(size_t)cut[t]];
            for (int j = cut[t]; j < cut[t + 1]; ++j) { so[(size_t)j] = loff[(size_t)k][(size_t)j] - base; dd[(size_t)j] = recOff[k + j * G]; ll[(size_t)j] = lens[(size_t)k][(size_t)j]; }
        }
        if (int r = grow(m, o.device, o.srcOff, (size_t)nb * 8)) return r;
        if (int r = grow(m, o.device, o.dstOff, (size_t)nb * 8)) return r;
        if (int r = grow(m, o.device, o.len, (size_t)nb * 4)) return r;
        MCHK(m, hipMemcpyAsync(o.srcOff.p, so.data(), (size_t)nb * 8, hipMemcpyHostToDevice, o.stream));      // behind the previous shard's movers
        MCHK(m, hipMemcpyAsync(o.dstOff.p, dd.data(), (size_t)nb * 8, hipMemcpyHostToDevice, o.stream));
        MCHK(m, hipMemcpyAsync(o.len.p, ll.data(), (size_t)nb * 4, hipMemcpyHostToDevice, o.stream));
        MCHK(m, hipStreamSynchronize(o.stream));
        for (size_t t = 0; t + 1 < cut.size(); ++t) {
            const int j0 = cut[t], cnt = cut[t + 1] - cut[t];
            int maxLen = 0;
            for (int j = j0; j < j0 + cnt; ++j) maxLen = std::max(maxLen, ll[(size_t)j]);
            const void* from = s.body.p;
            int q = -1;
            if (!local) {
                // the piece travels on the owner's copy stream into scratch q while the mover works on the piece before it
                q = piece++ & 1;
                const int64_t base = loff[(size_t)k][(size_t)j0], bytes = loff[(size_t)k][(size_t)(j0 + cnt)] - base;
                if (bytes > (int64_t)m->scratch[q].cap) { MCHK(m, hipStreamSynchronize(o.stream)); MCHK(m, hipStreamSynchronize(o.copyStream)); }
                if (int r = grow(m, o.device, m->scratch[q], (size_t)bytes)) return r;
                MCHK(m, hipStreamWaitEvent(o.copyStream, m->scratchFree[q], 0));       // the mover of two pieces ago is done with scratch q
                MCHK(m, hipMemcpyPeerAsync(m->scratch[q].p, o.device, (const uint8_t*)s.body.p + base, s.device, (size_t)bytes, o.copyStream));
                MCHK(m, hipEventRecord(o.ev[q], o.copyStream));
                MCHK(m, hipStreamWaitEvent(o.stream, o.ev[q], 0));
                from = m->scratch[q].p;
            }
            if (int r = plz4hip_dev_scatter_records(o.ctx, from, (const int64_t*)o.srcOff.p + j0, (const int32_t*)o.len.p + j0, (const int64_t*)o.dstOff.p + j0, cnt, maxLen, body, bodyCap, o.stream)) { m->err = plz4hip_last_error(o.ctx); return r; }
            if (q >= 0) MCHK(m, hipEventRecord(m->scratchFree[q], o.stream));
        }
    }
    MCHK(m, hipStreamSynchronize(o.stream));
    return PLZ4HIP_OK;
}

int plz4hip_mgpu_dev_decode_frame(plz4hip_mgpu* m, int nBlocks, int owner, const void* body, const int64_t* recOff, int bsz, int blockChecksum,
                                  void* const* shardDst, int64_t dstStride, int dstCap, int32_t* result, int32_t* status)
{
    if (!m || nBlocks < 0 || bsz <= 0 || !recOff || !shardDst || !result || !status || (nBlocks && !body)) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_dev_decode_frame: bad argument");
    const int G = (int)m->sh.size();
    if (owner < 0 || owner >= G) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_dev_decode_frame: owner");
    if (nBlocks == 0) return PLZ4HIP_OK;
    // the offsets come from a frame index that may be damaged: every record is its size word + at most bsz bytes (+ checksum)
    if (recOff[0] < 0) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_dev_decode_frame: recOff[0] < 0");
    for (int i = 0; i < nBlocks; ++i) {
        const int64_t l = recOff[i + 1] - recOff[i];
        if (l < 4 || l > (int64_t)bsz + 8) return mfail(m, PLZ4HIP_E_ARG, "plz4hip_mgpu_dev_decode_frame: recOff is not a list of records (length outside 4..bsz+8)");
    }
    std::lock_guard<std::mutex> g(m->mu);
    Shard& o = m->sh[(size_t)owner];
    // 1. deal the records: shard k's records are packed on the owner (record mover), then copied to device k in one go
    for (int k = 0; k < G; ++k) {
        Shard& s = m->sh[(size_t)k];
        const int nb = shard_blocks(nBlocks, G, k);
        if (!nb) continue;
        std::vector<int64_t> so((size_t)nb), dd((size_t)nb + 1); std::vector<int32_t> ll((size_t)nb);
        int maxLen = 0; dd[0] = 0;
        for (int j = 0; j < nb; ++j) { const int i = k + j * G; so[(size_t)j] = recOff[i]; ll[(size_t)j] = (int32_t)(recOff[i + 1] - recOff[i]); dd[(size_t)j + 1] = dd[(size_t)j] + ll[(size_t)j]; maxLen = std::max(maxLen, ll[(size_t)j]); }
        const size_t bytes = (size_t)dd[(size_t)nb];
        if (int r = grow(m, s.device, s.body, bytes + 16)) return r;
        if (int r = grow(m, s.device, s.recOff, (size_t)(nb + 1) * 8)) return r;
        if (int r = grow(m, s.device, s.recLen, (size_t)nb * 4)) return r;
        if (int r = grow(m, s.device, s.len, (size_t)nb * 4)) return r;
        DevGuard dg(o.device);
        if (int r = grow(m, o.device, o.pack, bytes + 16)) return r;
        if (int r = grow(m, o.device, o.srcOff, (size_t)nb * 8)) return r;
        if (int r = grow(m, o.device, o.dstOff, (size_t)nb * 8)) return r;
        if (int r = grow(m, o.device, o.len, (size_t)nb * 4)) return r;
        MCHK(m, hipMemcpyAsync(o.srcOff.p, so.data(), (size_t)nb * 8, hipMemcpyHostToDevice, o.stream));
        MCHK(m, hipMemcpyAsync(o.dstOff.p, dd.data(), (size_t)nb * 8, hipMemcpyHostToDevice, o.stream));
        MCHK(m, hipMemcpyAsync(o.len.p, ll.data(), (size_t)nb * 4, hipMemcpyHostToDevice, o.stream));
        MCHK(m, hipStreamSynchronize(o.stream));
        if (int r = plz4hip_dev_scatter_records(o.ctx, body, (const int64_t*)o.srcOff.p, (const int32_t*)o.len.p, (const int64_t*)o.dstOff.p, nb, maxLen, o.pack.p, (int64_t)o.pack.cap, o.stream)) { m->err = plz4hip_last_error(o.ctx); return r; }
        MCHK(m, hipMemcpyPeerAsync(s.body.p, s.device, o.pack.p, o.device, bytes, o.stream));
        MCHK(m, hipStreamSynchronize(o.stream));
        DevGuard dg2(s.device);
        MCHK(m, hipMemcpyAsync(s.recOff.p, dd.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, s.stream));
        MCHK(m, hipStreamSynchronize(s.stream));
    }
    // 2. every device decodes its records; results come back in block order
    return per_shard(m, [&](int k) -> int {
        Shard& s = m->sh[(size_t)k];
        const int nb = shard_blocks(nBlocks, G, k);
        if (!nb) return PLZ4HIP_OK;
        if (int r = plz4hip_dev_decode_records(s.ctx, s.body.p, (const int64_t*)s.recOff.p, nb, bsz, blockChecksum, shardDst[k], dstStride, dstCap, (int32_t*)s.recLen.p, (int32_t*)s.len.p, s.stream)) return r;
        DevGuard dg(s.device);
        std::vector<int32_t> r2((size_t)nb), s2((size_t)nb);
        if (hipMemcpyAsync(r2.data(), s.recLen.p, (size_t)nb * 4, hipMemcpyDeviceToHost, s.stream) != hipSuccess) return PLZ4HIP_E_DEVICE;
        if (hipMemcpyAsync(s2.data(), s.len.p, (size_t)nb * 4, hipMemcpyDeviceToHost, s.stream) != hipSuccess) return PLZ4HIP_E_DEVICE;
        if (hipStreamSynchronize(s.stream) != hipSuccess) return PLZ4HIP_E_DEVICE;
        for (int j = 0; j < nb; ++j) { result[k + j * G] = r2[(size_t)j]; status[k + j * G] = s2[(size_t)j]; }
        return PLZ4HIP_OK;
    });
}

}  // extern "C"
