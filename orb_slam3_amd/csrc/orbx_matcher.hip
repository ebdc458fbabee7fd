// orbx_matcher.hip -- host side of the matchers: per-thread context (own HIP stream + grow-on-demand device
// scratch), upload / launch / download for the host-pointer entry points, and the device-resident batched
// frame-to-frame matcher.  See include/orbx.h for the reference functions each entry point replaces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "matcher_kernels.hip.h"
#include "geometry_kernels.hip.h"
#include "extractor_state.h"

using namespace orbx;

namespace {

constexpr int kMaxResolveFeatures = ORBX_MAX_FRAME_FEATURES;  // claim (4 B) + angle (4 B) + occ (1 B) + octave (1 B) per feature must fit the 160 KB LDS (16000 x 10 + 200 B)
inline size_t resolve_lds_bytes(int n) { return (size_t)n * 10 + 64; }   // k_greedy_resolve: claim u32 + angle f32 + occ u8 + octave u8 per feature
// The replay of SearchByProjection's query loop.  Single calls and the batched map-point search: k_resolve_wide_t, a workgroup of 4 waves per problem
// (M1 10 000 points 241 -> 197 us per call, M2 132 -> 94, M3 111 -> 85: profiles/r06_h_resolve_ab.txt).  The frame-to-frame matcher of a BATCH keeps the
// one-wave k_greedy_resolve_t: 255 problems side by side, 75 against 83 us serialized, and a quarter of the wave slots beside the next batch's extraction.
// ORBX_RESOLVE_WAVES = 1 / 2 / 4 / 8 forces one form everywhere (A/B).
template <int WAVES, bool BRUTE>
inline int launch_resolve_wide(int np, size_t lds, hipStream_t st, const WindowProblem *dP, const ResolveProblem *dR, const GridParams &g, int cap) {
    if (lds > 64 * 1024) ORBX_HIP(hipFuncSetAttribute((const void *)k_resolve_wide_t<WAVES, BRUTE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_resolve_wide_t<WAVES, BRUTE>), dim3(np), dim3(64 * WAVES), lds, st, dP, dR, g, cap);
    return ORBX_OK;
}
template <bool BRUTE>
inline int launch_resolve(int np, hipStream_t st, const WindowProblem *dP, const ResolveProblem *dR, const GridParams &g, int cap, int nq_max, int default_waves) {
    static const int waves_env = [] { const char *v = getenv("ORBX_RESOLVE_WAVES"); return v ? atoi(v) : 0; }();
    const int waves = nq_max > 65535 ? 1 : waves_env > 0 ? waves_env : default_waves;   // the wide form carries the query index of a rotation entry in 16 bits
    const size_t lds = resolve_lds_bytes(cap);
    if (waves == 2) return launch_resolve_wide<2, BRUTE>(np, lds, st, dP, dR, g, cap);
    if (waves == 8) return launch_resolve_wide<8, BRUTE>(np, lds, st, dP, dR, g, cap);
    if (waves != 1) return launch_resolve_wide<4, BRUTE>(np, lds, st, dP, dR, g, cap);
    if (lds > 64 * 1024) ORBX_HIP(hipFuncSetAttribute((const void *)k_greedy_resolve_t<BRUTE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_greedy_resolve_t<BRUTE>, dim3(np), dim3(64), lds, st, dP, dR, g, cap);
    return ORBX_OK;
}
#define ORBX_LAUNCH_GRID_BUILD(grid, block, lds, stream, ...) hipLaunchKernelGGL(k_grid_build, grid, block, lds, stream, __VA_ARGS__)
// k_window_best2: 8 lanes per query (a window of the bench's matchers holds 1 - 10 candidates: 16 -> 8 lanes, 74 -> 59 us in round 4; 4 like 8).
// nq = queries per problem, np = problems.  From 8 problems on the launch is XCD-aware like the extractor's (extractor_kernels.hip.h, xcd_grid): x = XCD,
// problem = 8 z + x, so that ALL workgroups of a problem -- they share the frame's grid, keypoints and descriptors -- run on one XCD and fetch that frame
// into ONE L2 (round 4 spread a problem's query blocks over the eight XCDs: 134 MB of HBM reads per launch for 35 MB of data).
#define ORBX_LAUNCH_WINDOW_BEST2(nq, np, stream, ...)                                                                                          \
    do {                                                                                                                                       \
        const int np_ = (np), nb_ = ((nq) + 31) / 32;                                                                                          \
        const dim3 grid_ = np_ >= 8 ? dim3(8, (unsigned)nb_, (unsigned)((np_ + 7) / 8)) : dim3(1, (unsigned)nb_, (unsigned)np_);               \
        hipLaunchKernelGGL(k_window_best2_t<8>, grid_, dim3(256), 0, stream, __VA_ARGS__, np_);                                                \
    } while (0)
struct Arena {  // bump allocator over one device buffer, reset per call
    uint8_t *base = nullptr;
    size_t cap = 0, used = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return ORBX_OK;
        if (base) (void)hipFree(base);
        base = nullptr; cap = 0;
        bytes = (bytes * 3 / 2 + 4095) & ~(size_t)4095;
        ORBX_HIP(hipMalloc((void **)&base, bytes));
        cap = bytes;
        return ORBX_OK;
    }
    void reset() { used = 0; }
    template <typename T> T *take(size_t n) {
        used = (used + 255) & ~(size_t)255;
        T *p = reinterpret_cast<T *>(base + used);
        used += n * sizeof(T);
        return p;
    }
    static size_t pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
};

}  // namespace

// Pinned host staging of a matcher context.  No entry point ever hands a CALLER'S (pageable) pointer to the HIP runtime: uploads are
// memcpy'd into this arena first, downloads land here and are memcpy'd out after the stream synchronisation.  (The runtime's
// alternative for pageable memory is to pin the caller's pages on the fly for the duration of the DMA; keeping that machinery out of
// the hot path removes a whole class of lifetime hazards -- stack temporaries, vectors freed right after the call -- and is faster
// for the small transfers these entry points make.)
struct PinnedArena {
    uint8_t *base = nullptr;
    size_t cap = 0, used = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return ORBX_OK;
        if (base) (void)hipHostFree(base);
        base = nullptr; cap = 0;
        bytes = (bytes * 3 / 2 + 4095) & ~(size_t)4095;
        // coherent (fine-grained) whatever HIP_HOST_COHERENT says: k_xfer's lanes read the staged inputs from and write the results into this block
        // themselves, and the host reads them right after the stream synchronisation (as the extractor's h_stage, extractor_state.h)
        ORBX_HIP(hipHostMalloc((void **)&base, bytes, hipHostMallocCoherent));
        cap = bytes;
        return ORBX_OK;
    }
    void reset() { used = 0; }
    void *take(size_t bytes) {
        used = (used + 63) & ~(size_t)63;
        if (used + bytes > cap) return nullptr;
        void *p = base + used;
        used += bytes;
        return p;
    }
};

struct orbx_matcher {
    int device = 0;
    hipStream_t stream = nullptr;
    Arena arena;
    PinnedArena stage;
    // Transfers of ONE call are coalesced (round 5) and issued as KERNEL work (round 6).  `mirror` is a pinned, device-visible image of the device arena: an
    // upload to arena offset o is staged at mirror offset o and only RECORDED; exec() -- the stream as every launch, fill and download of a call obtains it --
    // first issues the recorded uploads and fills: every run of arena-adjacent buffers (adjacent = nothing but take()'s alignment padding between them, which
    // no buffer owns) is one op of a k_xfer launch whose lanes read the mirror themselves (runs above kKernelXferMax: one hipMemcpyAsync).  Downloads from the
    // arena are recorded the same way and issued as runs by deliver(): a k_xfer launch that writes the mirror.  Round 4: 14 uploads and 2 downloads of
    // 4 B .. 32 KB per projection-matcher call, each a DMA submission (~5 us); round 5: 1 + 1 DMA submissions; now none -- the call's work is one in-order
    // chain of launches in the compute queue, no hand-over to a DMA engine and back.  Transfers whose device side is not in the arena take the direct path.
    PinnedArena mirror;
    struct Span { size_t off, bytes; };
    std::vector<Span> uploads;      // recorded, not yet issued (ascending offsets: the arena is a bump allocator)
    std::vector<Span> issued;       // mirror ranges a transfer issued since the last synchronisation may still read: not staged over (such an upload goes direct)
    struct Fill { size_t off, bytes; uint32_t value; };
    std::vector<Fill> fills;        // recorded fills of arena ranges, issued with the uploads
    struct Pending { void *dst; const void *src; size_t bytes; };
    std::vector<Pending> pending;   // downloads waiting in pinned memory for the stream synchronisation
    struct Down { void *dst; size_t off, bytes; };
    std::vector<Down> downloads;    // recorded downloads from the arena (issued by deliver())
    std::vector<Down> direct;       // results a kernel wrote straight into the mirror (host_view): handed out after the synchronisation, no transfer
    hipError_t xfer_err = hipSuccess;
    int64_t xfers[6] = {0, 0, 0, 0, 0, 0};   // transfer submissions (runs) up / down and their bytes since begin(); [4] of them by a DMA engine, [5] k_xfer launches
    bool kernel_xfer = true;                 // ORBX_MATCHER_DMA=1: every run by hipMemcpyAsync, fills by hipMemsetAsync (round 5's transport, for A/B)
    bool brute_windows = true;               // ORBX_MATCHER_BRUTE=0: small single calls build the grid as large ones do (A/B)
    bool dirty = false;                      // something was enqueued since the last synchronisation
    int32_t replay_stats[3] = {0, 0, 0};     // k_replay_init_lists of the last orbx_search_for_initialization: rounds, whole-wave re-scans, queries
    static constexpr size_t kPadGap = 255;   // Arena::take aligns to 256
    static constexpr size_t kKernelXferMax = (size_t)1 << 20;
    // device scratch for one call + staging for everything that call can move in either direction
    int reserve_all(size_t device_bytes) {
        if (dirty) { (void)hipStreamSynchronize(stream); dirty = false; }   // a call that failed between exec() and deliver(): nothing of it may still read what is re-allocated here
        int r = arena.reserve(device_bytes);
        if (r != ORBX_OK) return r;
        r = mirror.reserve(arena.cap);
        if (r != ORBX_OK) return r;
        return stage.reserve(2 * device_bytes + 65536);
    }
    // begin() follows a synchronisation of the previous call: every entry point ends in deliver(); one that returned between exec() and deliver() (a failed
    // launch, an exhausted staging arena) left `dirty` set and is waited for here, before its mirror ranges are staged over
    void begin() {
        if (dirty) { (void)hipStreamSynchronize(stream); dirty = false; }
        arena.reset(); stage.reset(); pending.clear(); uploads.clear(); downloads.clear(); direct.clear(); issued.clear(); fills.clear(); xfer_err = hipSuccess;
        for (int64_t &x : xfers) x = 0;
    }
    bool in_arena(const void *p, size_t bytes) const {
        const uint8_t *q = static_cast<const uint8_t *>(p);
        return arena.base && q >= arena.base && q + bytes <= arena.base + arena.cap && arena.cap <= mirror.cap;
    }
    bool stageable(const void *p, size_t bytes) const {   // an upload that may be staged in the mirror
        if (!in_arena(p, bytes)) return false;
        const size_t o = (size_t)(static_cast<const uint8_t *>(p) - arena.base);
        for (const Span &q : issued)
            if (o < q.off + q.bytes && q.off < o + bytes) return false;
        return true;
    }
    void note(hipError_t e) { if (e != hipSuccess && xfer_err == hipSuccess) xfer_err = e; }
    // DIRECT access (round 6): a purely streaming kernel (every input read once, every output written once: k_in_frustum) is handed the MIRROR's addresses
    // of its arena buffers -- its lanes read the staged inputs over the host link and write the results into host memory themselves: one launch per call
    // instead of three (k_xfer, kernel, k_xfer).  host_view(p) = the mirror address of arena buffer p; stage_direct copies an input there without recording an
    // upload; result_direct registers an output for deliver().
    template <class T> T *host_view(T *p) const { return reinterpret_cast<T *>(mirror.base + (reinterpret_cast<const uint8_t *>(p) - arena.base)); }
    bool direct_ok(size_t total_bytes) const { return kernel_xfer && total_bytes <= kKernelXferMax * 2; }
    void stage_direct(void *arena_dst, const void *src, size_t bytes) { memcpy(host_view(static_cast<uint8_t *>(arena_dst)), src, bytes); }
    void result_direct(void *dst, const void *arena_src, size_t bytes) {
        direct.push_back(Down{dst, (size_t)(static_cast<const uint8_t *>(arena_src) - arena.base), bytes});
    }
    void record_upload(size_t o, size_t bytes) {   // staged in the mirror at offset o by the caller
        for (const Fill &f : fills)
            if (o < f.off + f.bytes + 16 && f.off < o + bytes + 16) { flush_uploads(); break; }   // never reorder an upload and a fill of one range
        uploads.push_back(Span{o, bytes});
    }
    void launch_xfer(const XferOps &X, uint32_t max_units) {
        const uint32_t grid = std::min<uint32_t>(1024u, (max_units + 255u) / 256u);
        hipLaunchKernelGGL(k_xfer, dim3(grid ? grid : 1), dim3(256), 0, stream, X);
        note(hipGetLastError());
        xfers[5]++;
    }
    // a fill of an arena range, issued with the call's uploads (the device side of the range must not be the target of a recorded upload)
    hipError_t fill(void *dst, int value, size_t bytes) {
        if (bytes == 0) return hipSuccess;
        const size_t o = (size_t)(static_cast<const uint8_t *>(dst) - arena.base);
        if (kernel_xfer && in_arena(dst, bytes) && (o & 15) == 0) {
            for (const Span &u : uploads)
                if (o < u.off + u.bytes + 16 && u.off < o + bytes + 16) { flush_uploads(); break; }   // never reorder a fill and an upload of one range
            fills.push_back(Fill{o, bytes, (uint32_t)(value & 0xff)});
            return hipSuccess;
        }
        dirty = true;
        return hipMemsetAsync(dst, value, bytes, exec());
    }
    void flush_uploads() {
        std::sort(uploads.begin(), uploads.end(), [](const Span &a, const Span &b) { return a.off < b.off; });   // (a record uploaded after its buffers were taken lies between them)
        XferOps X;
        X.n = 0;
        uint32_t max_units = 0;
        auto add = [&](uint8_t *dst, const uint8_t *src, size_t bytes, uint32_t value) {
            if (X.n == kMaxXferOps) { launch_xfer(X, max_units); X.n = 0; max_units = 0; }
            const uint32_t units = (uint32_t)((bytes + 15) / 16);
            X.op[X.n++] = XferOp{dst, src, units, value};
            max_units = std::max(max_units, units);
        };
        size_t i = 0;
        while (i < uploads.size()) {
            const size_t b = uploads[i].off;
            size_t e = b + uploads[i].bytes, j = i + 1;
            while (j < uploads.size() && uploads[j].off <= e + kPadGap) { e = std::max(e, uploads[j].off + uploads[j].bytes); j++; }
            if (kernel_xfer && e - b <= kKernelXferMax && (b & 15) == 0) {
                add(arena.base + b, mirror.base + b, e - b, 0);   // (rounded up to 16 bytes: alignment padding at most, the next buffer starts on a multiple of 256)
            } else {
                note(hipMemcpyAsync(arena.base + b, mirror.base + b, e - b, hipMemcpyHostToDevice, stream));
                xfers[4]++;
            }
            xfers[0]++; xfers[2] += (int64_t)(e - b);
            issued.push_back(Span{b, e - b});
            i = j;
        }
        uploads.clear();
        for (const Fill &f : fills) add(arena.base + f.off, nullptr, f.bytes, f.value);
        fills.clear();
        if (X.n) launch_xfer(X, max_units);
        dirty = true;
    }
    // the stream for anything that consumes the call's uploads (kernel launches, memsets, downloads, the synchronisation)
    hipStream_t exec() {
        if (!uploads.empty() || !fills.empty()) flush_uploads();
        dirty = true;
        return stream;
    }
    // issue the recorded downloads (runs of arena-adjacent buffers), wait for the stream, hand the bytes to the caller's buffers
    hipError_t deliver() {
        (void)exec();
        std::sort(downloads.begin(), downloads.end(), [](const Down &a, const Down &b) { return a.off < b.off; });
        XferOps X;
        X.n = 0;
        uint32_t max_units = 0;
        size_t i = 0;
        while (i < downloads.size()) {
            const size_t b = downloads[i].off;
            size_t e = b + downloads[i].bytes, j = i + 1;
            while (j < downloads.size() && downloads[j].off <= e + kPadGap) { e = std::max(e, downloads[j].off + downloads[j].bytes); j++; }
            if (kernel_xfer && e - b <= kKernelXferMax && (b & 15) == 0) {
                if (X.n == kMaxXferOps) { launch_xfer(X, max_units); X.n = 0; max_units = 0; }
                const uint32_t units = (uint32_t)((e - b + 15) / 16);
                X.op[X.n++] = XferOp{mirror.base + b, arena.base + b, units, 0};
                max_units = std::max(max_units, units);
            } else {
                note(hipMemcpyAsync(mirror.base + b, arena.base + b, e - b, hipMemcpyDeviceToHost, stream));
                xfers[4]++;
            }
            xfers[1]++; xfers[3] += (int64_t)(e - b);
            i = j;
        }
        if (X.n) launch_xfer(X, max_units);
        note(hipStreamSynchronize(stream));
        dirty = false;
        if (xfer_err == hipSuccess) {
            for (const Down &d : downloads) memcpy(d.dst, mirror.base + d.off, d.bytes);
            for (const Down &d : direct) memcpy(d.dst, mirror.base + d.off, d.bytes);
            for (const Pending &q : pending) memcpy(q.dst, q.src, q.bytes);
        }
        downloads.clear();
        direct.clear();
        pending.clear();
        issued.clear();
        return xfer_err;
    }
};

// ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) resident on the device
struct orbx_vocabulary {
    int device = 0, k = 0, L = 0, n_nodes = 0;
    int32_t *child_ptr = nullptr, *child_idx = nullptr, *word_id = nullptr;
    uint8_t *node_desc = nullptr;
};

extern "C" {

int orbx_matcher_create(int device, orbx_matcher **out) {
    if (!out) return ORBX_E_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        set_error("no usable HIP device (liborbx has no CPU fallback)");
        return ORBX_E_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    orbx_matcher *m = new orbx_matcher();
    m->device = device;
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); delete m; return ORBX_E_HIP; }
    const char *dma = getenv("ORBX_MATCHER_DMA");
    m->kernel_xfer = !(dma && dma[0] == '1');
    const char *br = getenv("ORBX_MATCHER_BRUTE");
    m->brute_windows = !(br && br[0] == '0');
    *out = m;
    return ORBX_OK;
}

int orbx_matcher_debug_transfers(const orbx_matcher *m, int64_t *out, int cap) {
    if (!m || !out || cap < 4) return ORBX_E_BAD_ARG;
    const int n = cap >= 6 ? 6 : 4;
    for (int i = 0; i < n; i++) out[i] = m->xfers[i];
    return n;
}

int orbx_matcher_debug_replay_stats(const orbx_matcher *m, int32_t *out3) {
    if (!m || !out3) return ORBX_E_BAD_ARG;
    for (int k = 0; k < 3; k++) out3[k] = m->replay_stats[k];
    return 3;
}

void orbx_matcher_destroy(orbx_matcher *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    if (m->arena.base) (void)hipFree(m->arena.base);
    if (m->stage.base) (void)hipHostFree(m->stage.base);
    if (m->mirror.base) (void)hipHostFree(m->mirror.base);
    delete m;
}

#define H2D(dst, src, bytes)                                                                                      \
    do {                                                                                                          \
        const size_t _b = (bytes);                                                                                \
        if (_b > 0) {                                                                                             \
            if (m->stageable((dst), _b)) {  /* recorded; issued with its arena neighbours by exec() */             \
                const size_t _o = (size_t)((const uint8_t *)(dst) - m->arena.base);                               \
                memcpy(m->mirror.base + _o, (src), _b);                                                           \
                m->record_upload(_o, _b);                                                                         \
            } else {                                                                                              \
                void *_s = m->stage.take(_b);                                                                     \
                if (!_s) { set_error("staging arena exhausted"); return ORBX_E_INTERNAL; }                        \
                memcpy(_s, (src), _b);                                                                            \
                m->dirty = true;                                                                                  \
                ORBX_HIP(hipMemcpyAsync((dst), _s, _b, hipMemcpyHostToDevice, m->stream));                        \
                m->xfers[0]++; m->xfers[2] += (int64_t)_b; m->xfers[4]++;                                         \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#define D2H(dst, src, bytes)                                                                                      \
    do {                                                                                                          \
        const size_t _b = (bytes);                                                                                \
        if (_b > 0) {                                                                                             \
            if (m->in_arena((src), _b)) {   /* recorded; issued with its arena neighbours by deliver() */          \
                m->downloads.push_back(orbx_matcher::Down{(void *)(dst), (size_t)((const uint8_t *)(src) - m->arena.base), _b}); \
            } else {                                                                                              \
                void *_s = m->stage.take(_b);                                                                     \
                if (!_s) { set_error("staging arena exhausted"); return ORBX_E_INTERNAL; }                        \
                ORBX_HIP(hipMemcpyAsync(_s, (src), _b, hipMemcpyDeviceToHost, m->exec()));                        \
                m->xfers[1]++; m->xfers[3] += (int64_t)_b; m->xfers[4]++;                                         \
                m->pending.push_back(orbx_matcher::Pending{(void *)(dst), _s, _b});                               \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#define SYNC_AND_DELIVER()                                                                                        \
    do {                                                                                                          \
        ORBX_HIP(m->deliver());                                                                                   \
    } while (0)

int orbx_hamming_csr(orbx_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt, const int32_t *row_ptr,
                     const int32_t *cand, uint16_t *dist_out) {
    if (!m || nq < 0 || nt < 0 || (nq > 0 && (!q || !row_ptr))) return ORBX_E_BAD_ARG;
    if (nq == 0) return ORBX_OK;
    const int nnz = row_ptr[nq];
    if (nnz <= 0) return ORBX_OK;
    if (!t || !cand || !dist_out) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(m->device));
    int r = m->reserve_all(Arena::pad((size_t)nq * 32) + Arena::pad((size_t)nt * 32) + Arena::pad(4 * (size_t)(nq + 1)) +
                             Arena::pad(4 * (size_t)nnz) + Arena::pad(2 * (size_t)nnz) + 4096);
    if (r != ORBX_OK) return r;
    m->begin();
    uint8_t *dq = m->arena.take<uint8_t>((size_t)nq * 32), *dt = m->arena.take<uint8_t>((size_t)nt * 32);
    int32_t *drp = m->arena.take<int32_t>(nq + 1), *dc = m->arena.take<int32_t>(nnz);
    uint16_t *dd = m->arena.take<uint16_t>(nnz);
    H2D(dq, q, (size_t)nq * 32); H2D(dt, t, (size_t)nt * 32); H2D(drp, row_ptr, 4 * (size_t)(nq + 1)); H2D(dc, cand, 4 * (size_t)nnz);
    hipLaunchKernelGGL(k_hamming_csr, dim3((nq + 3) / 4), dim3(256), 0, m->exec(), dq, nq, dt, drp, dc, dd);
    D2H(dist_out, dd, 2 * (size_t)nnz);
    SYNC_AND_DELIVER();
    return ORBX_OK;
}

int orbx_hamming_best2_csr(orbx_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt, const int32_t *row_ptr,
                           const int32_t *cand, int32_t *best_pos, int32_t *best_dist, int32_t *second_pos,
                           int32_t *second_dist) {
    if (!m || nq < 0 || nt < 0 || (nq > 0 && (!q || !row_ptr))) return ORBX_E_BAD_ARG;
    if (nq == 0) return ORBX_OK;
    const int nnz = row_ptr[nq];
    ORBX_HIP(hipSetDevice(m->device));
    int r = m->reserve_all(Arena::pad((size_t)nq * 32) + Arena::pad((size_t)nt * 32 + 32) + Arena::pad(4 * (size_t)(nq + 1)) +
                             Arena::pad(4 * (size_t)nnz + 4) + 4 * Arena::pad(4 * (size_t)nq) + 4096);
    if (r != ORBX_OK) return r;
    m->begin();
    uint8_t *dq = m->arena.take<uint8_t>((size_t)nq * 32), *dt = m->arena.take<uint8_t>((size_t)nt * 32 + 32);
    int32_t *drp = m->arena.take<int32_t>(nq + 1), *dc = m->arena.take<int32_t>(nnz + 1);
    int32_t *o[4];
    for (int k = 0; k < 4; k++) o[k] = m->arena.take<int32_t>(nq);
    H2D(dq, q, (size_t)nq * 32);
    if (nt > 0) H2D(dt, t, (size_t)nt * 32);
    H2D(drp, row_ptr, 4 * (size_t)(nq + 1));
    if (nnz > 0) H2D(dc, cand, 4 * (size_t)nnz);
    hipLaunchKernelGGL(k_hamming_best2_csr, dim3((nq + 3) / 4), dim3(256), 0, m->exec(), dq, nq, dt, drp, dc, o[0], o[1], o[2], o[3]);
    int32_t *host[4] = {best_pos, best_dist, second_pos, second_dist};
    for (int k = 0; k < 4; k++) if (host[k]) D2H(host[k], o[k], 4 * (size_t)nq);
    SYNC_AND_DELIVER();
    return ORBX_OK;
}

int orbx_knn2(orbx_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist) {
    if (!m || nq < 0 || nt < 0 || (nq > 0 && (!q || !idx || !dist))) return ORBX_E_BAD_ARG;
    if (nq == 0) return ORBX_OK;
    ORBX_HIP(hipSetDevice(m->device));
    int r = m->reserve_all(Arena::pad((size_t)nq * 32) + Arena::pad((size_t)nt * 32 + 32) + 2 * Arena::pad(8 * (size_t)nq) + 4096);
    if (r != ORBX_OK) return r;
    m->begin();
    uint8_t *dq = m->arena.take<uint8_t>((size_t)nq * 32), *dt = m->arena.take<uint8_t>((size_t)nt * 32 + 32);
    int32_t *di = m->arena.take<int32_t>(2 * (size_t)nq), *dd = m->arena.take<int32_t>(2 * (size_t)nq);
    H2D(dq, q, (size_t)nq * 32);
    if (nt > 0) H2D(dt, t, (size_t)nt * 32);
    hipLaunchKernelGGL(k_knn2, dim3((nq + 3) / 4), dim3(256), 0, m->exec(), dq, nq, dt, nt, di, dd);
    D2H(idx, di, 8 * (size_t)nq); D2H(dist, dd, 8 * (size_t)nq);
    SYNC_AND_DELIVER();
    return ORBX_OK;
}

int orbx_stereo_rowband(orbx_matcher *m, const orbx_keypoint *kl, const uint8_t *dl, int nl, const orbx_keypoint *kr,
                        const uint8_t *dr, int nr, const float *scale, int nlevels, int n_rows, float min_d, float max_d,
                        int32_t *best_idx_r, int32_t *best_dist) {
    if (!m || nl < 0 || nr < 0 || nlevels <= 0 || !scale) return ORBX_E_BAD_ARG;
    if (nl == 0) return ORBX_OK;
    if (!kl || !dl || !best_idx_r || !best_dist) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(m->device));
    int r = m->reserve_all(Arena::pad(28 * (size_t)nl) + Arena::pad(32 * (size_t)nl) + Arena::pad(28 * (size_t)nr + 32) +
                             Arena::pad(32 * (size_t)nr + 32) + Arena::pad(4 * (size_t)nlevels) + 2 * Arena::pad(4 * (size_t)nl) + 4096);
    if (r != ORBX_OK) return r;
    m->begin();
    orbx_keypoint *dkl = m->arena.take<orbx_keypoint>(nl), *dkr = m->arena.take<orbx_keypoint>(nr + 1);
    uint8_t *ddl = m->arena.take<uint8_t>(32 * (size_t)nl), *ddr = m->arena.take<uint8_t>(32 * (size_t)nr + 32);
    float *dsc = m->arena.take<float>(nlevels);
    int32_t *dbi = m->arena.take<int32_t>(nl), *dbd = m->arena.take<int32_t>(nl);
    H2D(dkl, kl, 28 * (size_t)nl); H2D(ddl, dl, 32 * (size_t)nl);
    if (nr > 0) { H2D(dkr, kr, 28 * (size_t)nr); H2D(ddr, dr, 32 * (size_t)nr); }
    H2D(dsc, scale, 4 * (size_t)nlevels);
    hipLaunchKernelGGL(k_stereo_rowband, dim3((nl + 3) / 4), dim3(256), 0, m->exec(), dkl, ddl, nl, dkr, ddr, nr, dsc, n_rows, min_d,
                       max_d, dbi, dbd);
    D2H(best_idx_r, dbi, 4 * (size_t)nl); D2H(best_dist, dbd, 4 * (size_t)nl);
    SYNC_AND_DELIVER();
    return ORBX_OK;
}

// StereoBatch's row index parameters for an image of n_rows rows and pyramid scale factors sc[0 .. nl)
static void stereo_index_params(StereoBatch &S, int n_rows, const float *sc, int nl) {
    S.band = (int)ceilf(2.0f * *std::max_element(sc, sc + nl)) + 1;
    S.row_shift = 0;
    while (((n_rows - 1) >> S.row_shift) + 1 > kStereoIndexMaxBuckets) S.row_shift++;
    S.n_buckets = ((n_rows - 1) >> S.row_shift) + 1;
}

// Frame::ComputeStereoMatches (Frame.cc:811-981): device Hamming stage, host SAD refinement on the host-resident
// pyramid (mvImagePyramid is host memory at this boundary), host median rejection.
int orbx_compute_stereo_matches(orbx_matcher *m, const orbx_keypoint *kl, const uint8_t *dl, int N, const orbx_keypoint *kr,
                                const uint8_t *dr, int Nr, const float *scale_factors, const float *inv_scale_factors,
                                int nlevels, const uint8_t *const *pyr_left, const uint8_t *const *pyr_right, const int32_t *pyr_w,
                                const int32_t *pyr_h, const size_t *pyr_stride, float bf, float b, float *u_right, float *depth) {
    if (!m || !u_right || !depth || N < 0 || Nr < 0 || !pyr_h || !pyr_w || !pyr_stride || !pyr_left || !pyr_right || nlevels <= 0 ||
        !scale_factors || !inv_scale_factors || !(b > 0.f))
        return ORBX_E_BAD_ARG;
    for (int i = 0; i < N; i++) { u_right[i] = -1.0f; depth[i] = -1.0f; }
    if (N == 0 || Nr == 0) return 0;
    ORBX_HIP(hipSetDevice(m->device));
    // the same three kernels as the device-resident batch (Hamming row band, SAD + parabola, median rejection) on a batch
    // of one: the two pyramids are uploaded as slabs addressed like the extractor's (level origin = off + kEdge rows + kRoiX)
    const size_t lead = ((size_t)kEdge * 8192 + kRoiX + 255) & ~(size_t)255;
    std::vector<LevelInfo> lv(nlevels);
    size_t slab = lead;
    for (int l = 0; l < nlevels; l++) {
        if (pyr_stride[l] > 8192 || pyr_w[l] <= 0 || pyr_h[l] <= 0) return ORBX_E_BAD_ARG;
        memset(&lv[l], 0, sizeof(LevelInfo));
        lv[l].w = pyr_w[l]; lv[l].h = pyr_h[l]; lv[l].pitch = (int32_t)pyr_stride[l];
        lv[l].off = slab - (size_t)kEdge * pyr_stride[l] - kRoiX;   // so that the kernels' (kEdge + y) * pitch + kRoiX + x lands on (x, y)
        slab += (pyr_stride[l] * (size_t)pyr_h[l] + 255) & ~(size_t)255;
    }
    const size_t need = 2 * Arena::pad(slab) + Arena::pad(28 * (size_t)N) + Arena::pad(28 * (size_t)Nr) + Arena::pad(32 * (size_t)N) +
                        Arena::pad(32 * (size_t)Nr) + 5 * Arena::pad(4 * (size_t)N) + Arena::pad(sizeof(LevelInfo) * nlevels) +
                        Arena::pad(8 * (size_t)nlevels) + Arena::pad(4 * ((size_t)kStereoIndexMaxBuckets + 1)) + Arena::pad(16 * (size_t)Nr) + 8192;
    int r = m->reserve_all(need);
    if (r != ORBX_OK) return r;
    Arena &A = m->arena;
    m->begin();
    uint8_t *dL = A.take<uint8_t>(slab), *dR = A.take<uint8_t>(slab);
    for (int l = 0; l < nlevels; l++) {
        const size_t o = lv[l].off + (size_t)kEdge * pyr_stride[l] + kRoiX, bytes = pyr_stride[l] * (size_t)(pyr_h[l] - 1) + pyr_w[l];
        H2D(dL + o, pyr_left[l], bytes);
        H2D(dR + o, pyr_right[l], bytes);
    }
    StereoBatch S;
    memset(&S, 0, sizeof(S));
    orbx_keypoint *dkl = A.take<orbx_keypoint>(N), *dkr = A.take<orbx_keypoint>(Nr);
    uint8_t *ddl = A.take<uint8_t>(32 * (size_t)N), *ddr = A.take<uint8_t>(32 * (size_t)Nr);
    H2D(dkl, kl, 28 * (size_t)N); H2D(dkr, kr, 28 * (size_t)Nr); H2D(ddl, dl, 32 * (size_t)N); H2D(ddr, dr, 32 * (size_t)Nr);
    LevelInfo *dlv = A.take<LevelInfo>(nlevels);
    H2D(dlv, lv.data(), sizeof(LevelInfo) * nlevels);
    float *dsc = A.take<float>(2 * (size_t)nlevels);
    H2D(dsc, scale_factors, 4 * (size_t)nlevels); H2D(dsc + nlevels, inv_scale_factors, 4 * (size_t)nlevels);
    int32_t *dcnt = A.take<int32_t>(4);
    const int32_t cnts[2] = {N, Nr};
    H2D(dcnt, cnts, 8);
    S.kl = dkl; S.kr = dkr; S.dl = ddl; S.dr = ddr; S.nl = dcnt; S.nr = dcnt + 1; S.capL = N; S.capR = Nr;
    S.pyrL = dL; S.pyrR = dR; S.pyr_frame_L = 0; S.pyr_frame_R = 0; S.lvL = dlv; S.lvR = dlv;
    S.scale = dsc; S.inv_scale = dsc + nlevels; S.n_rows = pyr_h[0]; S.bf = bf; S.b = b;
    S.best_idx = A.take<int32_t>(N); S.best_dist = A.take<int32_t>(N);
    S.u_right = A.take<float>(N); S.depth = A.take<float>(N); S.nmatches = A.take<int32_t>(4); S.sad = A.take<int32_t>(N);   // the three downloads side by side: one DMA
    stereo_index_params(S, pyr_h[0], scale_factors, nlevels);
    int32_t *drp = A.take<int32_t>((size_t)S.n_buckets + 1);
    uint4 *den = A.take<uint4>(Nr);
    S.row_ptr = drp; S.row_ent = den;
    hipLaunchKernelGGL(k_stereo_row_index, dim3(1), dim3(256), 4 * ((size_t)S.n_buckets + 1) + 1024, m->exec(), S, drp, den);
    hipLaunchKernelGGL(k_stereo_rowband_batch, dim3((N + 15) / 16, 1), dim3(256), 0, m->exec(), S);
    hipLaunchKernelGGL(k_stereo_sad, dim3((N + 15) / 16, 1), dim3(256), 0, m->exec(), S);
    hipLaunchKernelGGL(k_stereo_reject, dim3(1), dim3(256), 0, m->exec(), S);
    ORBX_HIP(hipGetLastError());
    int32_t nm = 0;
    D2H(u_right, S.u_right, 4 * (size_t)N); D2H(depth, S.depth, 4 * (size_t)N); D2H(&nm, S.nmatches, 4);
    SYNC_AND_DELIVER();
    return nm;
}

}  // extern "C"

namespace {

// shared driver of the two projection matchers (host-pointer form)
struct ProjArgs {
    const orbx_frame_desc *frame;
    const uint8_t *occupied;
    int nq;
    const float *qx, *qy, *qr, *qxr;
    const int32_t *qmin, *qmax;
    const uint8_t *qdesc, *qvalid, *q_has_obs;
    const float *q_angle;
    int mode;
    float nnratio;
    int check_orientation;
    int32_t *match_out;
    float max_dist;
};

int run_projection(orbx_matcher *m, const ProjArgs &a) {
    const orbx_frame_desc *F = a.frame;
    const int n = F->n, nq = a.nq;
    for (int i = 0; i < n; i++) a.match_out[i] = -1;
    if (n == 0 || nq == 0) return 0;
    if (n > kMaxResolveFeatures) return ORBX_E_TOO_LARGE;  // before anything is enqueued: the resolve pass keeps 10 B per feature in LDS
    ORBX_HIP(hipSetDevice(m->device));
    size_t need = Arena::pad(28 * (size_t)n) + Arena::pad(32 * (size_t)n) + 3 * Arena::pad((size_t)n) + Arena::pad(4 * (size_t)n) * 2 +
                  Arena::pad(4 * (size_t)nq) * 7 + Arena::pad(32 * (size_t)nq) + Arena::pad((size_t)nq) * 2 + Arena::pad(8 * (size_t)nq) * 5 +
                  Arena::pad(sizeof(WindowProblem)) + Arena::pad(sizeof(ResolveProblem)) + Arena::pad(2 * (kGridCells + 1)) + Arena::pad(2 * (size_t)n) +
                  16 * 256 + 4096;
    int r = m->reserve_all(need);
    if (r != ORBX_OK) return r;
    Arena &A = m->arena;
    m->begin();
    WindowProblem P;
    memset(&P, 0, sizeof(P));
    ResolveProblem R;
    memset(&R, 0, sizeof(R));
    orbx_keypoint *dk = A.take<orbx_keypoint>(n);
    uint8_t *dd = A.take<uint8_t>(32 * (size_t)n);
    H2D(dk, F->keypoints_un, 28 * (size_t)n); H2D(dd, F->descriptors, 32 * (size_t)n);
    P.kps = dk; P.desc = dd;
    int32_t *dcnt = A.take<int32_t>(4);
    const int32_t cnts[2] = {n, nq};
    H2D(dcnt, cnts, 8);
    P.n_ptr = dcnt; P.nq_ptr = dcnt + 1;
    if (F->u_right) { float *p = A.take<float>(n); H2D(p, F->u_right, 4 * (size_t)n); P.u_right = p; }
    if (a.occupied) { uint8_t *p = A.take<uint8_t>(n); H2D(p, a.occupied, (size_t)n); P.occupied0 = p; }
    float *f3[3]; const float *h3[3] = {a.qx, a.qy, a.qr};
    for (int k = 0; k < 3; k++) { f3[k] = A.take<float>(nq); H2D(f3[k], h3[k], 4 * (size_t)nq); }
    P.qx = f3[0]; P.qy = f3[1]; P.qr = f3[2];
    int32_t *i2[2]; const int32_t *hi2[2] = {a.qmin, a.qmax};
    for (int k = 0; k < 2; k++) { i2[k] = A.take<int32_t>(nq); H2D(i2[k], hi2[k], 4 * (size_t)nq); }
    P.qmin = i2[0]; P.qmax = i2[1];
    if (a.qxr && F->u_right) { float *p = A.take<float>(nq); H2D(p, a.qxr, 4 * (size_t)nq); P.qxr = p; }
    { uint8_t *p = A.take<uint8_t>(32 * (size_t)nq); H2D(p, a.qdesc, 32 * (size_t)nq); P.qdesc = p; }
    if (a.qvalid) { uint8_t *p = A.take<uint8_t>(nq); H2D(p, a.qvalid, (size_t)nq); P.qvalid = p; }
    R.mode = a.mode; R.nnratio = a.nnratio; R.check_orientation = a.check_orientation; R.max_dist = a.max_dist;
    R.cleared_value = -2;
    if (a.q_angle) { float *p = A.take<float>(nq); H2D(p, a.q_angle, 4 * (size_t)nq); R.q_angle = p; }
    if (a.q_has_obs) { uint8_t *p = A.take<uint8_t>(nq); H2D(p, a.q_has_obs, (size_t)nq); R.q_has_obs = p; }
    // everything the call uploads lies in ONE run of the arena (one DMA, orbx_matcher::exec): the two problem records directly behind the inputs,
    // the buffers only the device writes behind them; the downloads (match, nmatches) side by side as well
    WindowProblem *dP = A.take<WindowProblem>(1);
    ResolveProblem *dR = A.take<ResolveProblem>(1);
    P.keys = A.take<u64>((size_t)nq * kTopK); P.meta = A.take<int32_t>(nq);
    P.gstart = A.take<uint16_t>(kGridCells + 1); P.gorder = A.take<uint16_t>(n);
    R.entries = A.take<int32_t>(nq);
    R.match = A.take<int32_t>(n);
    R.nmatches = A.take<int32_t>(1);
    // a small problem skips the grid: k_window_brute walks all features per query (no k_grid_build launch, whose counting sort this one call would use once)
    const bool brute = m->brute_windows && (size_t)nq * (size_t)n <= kBruteMaxPairs;
    if (brute) { P.gstart = nullptr; P.gorder = nullptr; }
    H2D(dP, &P, sizeof(P)); H2D(dR, &R, sizeof(R));
    GridParams g;
    g.minx = F->min_x; g.miny = F->min_y;
    g.inv_w = 64.0f / (F->max_x - F->min_x);  // Frame.cc:342-343
    g.inv_h = 48.0f / (F->max_y - F->min_y);
    if (brute) {
        hipLaunchKernelGGL(k_window_brute, dim3((nq + 3) / 4), dim3(256), 0, m->exec(), dP, g);
    } else {
        ORBX_LAUNCH_GRID_BUILD( dim3(1), dim3(64), 0, m->exec(), dP, g);
        ORBX_LAUNCH_WINDOW_BEST2(nq, 1, m->exec(), dP, g);
    }
    { const int rr = brute ? launch_resolve<true>(1, m->exec(), dP, dR, g, n, nq, 4) : launch_resolve<false>(1, m->exec(), dP, dR, g, n, nq, 4); if (rr != ORBX_OK) return rr; }
    int32_t nm = 0;
    D2H(a.match_out, R.match, 4 * (size_t)n);
    D2H(&nm, R.nmatches, 4);
    SYNC_AND_DELIVER();
    return nm;
}

}  // namespace

extern "C" {

int orbx_search_by_projection_mappoints(orbx_matcher *m, const orbx_frame_desc *frame, const uint8_t *frame_occupied, int n_mp,
                                        const float *proj_x, const float *proj_y, const float *proj_xr,
                                        const int32_t *pred_level, const float *view_cos, const uint8_t *mp_desc,
                                        const uint8_t *mp_in_view, const uint8_t *mp_has_obs, float th, float nnratio,
                                        int32_t *frame_match) {
    if (!m || !frame || frame->n < 0 || (!frame_match && frame->n > 0) || n_mp < 0) return ORBX_E_BAD_ARG;   // empty frames / query sets are legal
    if (n_mp > 0 && (!proj_x || !proj_y || !pred_level || !view_cos || !mp_desc)) return ORBX_E_BAD_ARG;
    // per-query window: r = RadiusByViewingCos(viewCos) [* th] * scale[level], levels [lvl-1, lvl]  (ORBmatcher.cc:63-72)
    std::vector<float> qr(n_mp);
    std::vector<int32_t> qmin(n_mp), qmax(n_mp);
    std::vector<uint8_t> valid(n_mp);
    const bool bFactor = th != 1.0;
    for (int i = 0; i < n_mp; i++) {
        const int lvl = pred_level[i];
        valid[i] = (!mp_in_view || mp_in_view[i]) && lvl >= 0 && lvl < frame->nlevels;
        float r = (view_cos[i] > 0.998) ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        qr[i] = valid[i] ? r * frame->scale_factors[lvl] : 0.f;
        qmin[i] = lvl - 1; qmax[i] = lvl;
    }
    ProjArgs a = {frame, frame_occupied, n_mp, proj_x, proj_y, qr.data(), proj_xr, qmin.data(), qmax.data(), mp_desc, valid.data(),
                  mp_has_obs, nullptr, 1, nnratio, 0, frame_match, (float)ORBX_TH_HIGH};
    return run_projection(m, a);
}

int orbx_search_by_projection_frame(orbx_matcher *m, const orbx_frame_desc *cur, const uint8_t *cur_occupied, int n_q,
                                    const float *q_u, const float *q_v, const float *q_ur, const int32_t *q_octave,
                                    const float *q_angle, const uint8_t *q_desc, const uint8_t *q_has_obs, float th, int level_mode,
                                    int check_orientation, int32_t *cur_match) {
    if (!m || !cur || cur->n < 0 || (!cur_match && cur->n > 0) || n_q < 0) return ORBX_E_BAD_ARG;
    if (n_q > 0 && (!q_u || !q_v || !q_octave || !q_desc || (check_orientation && !q_angle))) return ORBX_E_BAD_ARG;
    std::vector<float> qr(n_q);
    std::vector<int32_t> qmin(n_q), qmax(n_q);
    std::vector<uint8_t> valid(n_q);
    for (int i = 0; i < n_q; i++) {
        const int o = q_octave[i];
        valid[i] = o >= 0 && o < cur->nlevels;
        qr[i] = valid[i] ? th * cur->scale_factors[o] : 0.f;  // :1726
        if (level_mode == 1) { qmin[i] = o; qmax[i] = -1; }          // bForward  :1731
        else if (level_mode == 2) { qmin[i] = 0; qmax[i] = o; }      // bBackward :1733
        else { qmin[i] = o - 1; qmax[i] = o + 1; }                   // :1735
    }
    ProjArgs a = {cur, cur_occupied, n_q, q_u, q_v, qr.data(), q_ur, qmin.data(), qmax.data(), q_desc, valid.data(), q_has_obs,
                  q_angle, 2, 0.f, check_orientation, cur_match, (float)ORBX_TH_HIGH};
    return run_projection(m, a);
}

}  // extern "C"

namespace {

// shared driver of the fisheye-stereo twins of the two frame projection matchers (k_replay_twin)
struct TwinArgs {
    const orbx_frame_desc *left;          // left camera: mvKeys, n_left, image bounds, scale factors; descriptors of ALL n_left + n_right features
    const orbx_keypoint *kps_right; int n_right;
    const int32_t *l2r, *r2l;
    const uint8_t *occupied;              // [n_left + n_right]
    int nq;
    const float *qx[2], *qy[2], *qr[2];   // per side: window centre and half size
    const int32_t *qmin[2], *qmax[2];
    const uint8_t *qvalid[2];
    const uint8_t *qdesc, *q_has_obs;
    const float *q_angle;
    int mode; float nnratio; int check_orientation;
    int32_t *match_out;
};

int run_projection_twin(orbx_matcher *m, const TwinArgs &a) {
    const orbx_frame_desc *F = a.left;
    const int nl = F->n, nr = a.n_right, N = nl + nr, nq = a.nq;
    for (int i = 0; i < N; i++) a.match_out[i] = -1;
    if (N == 0 || nq == 0) return 0;
    if (N > 60000) return ORBX_E_TOO_LARGE;   // 16-bit feature indices in the candidate keys; occupancy bytes in LDS
    ORBX_HIP(hipSetDevice(m->device));
    const size_t need = Arena::pad(28 * (size_t)N) + Arena::pad(32 * (size_t)N) + Arena::pad((size_t)N) + 2 * Arena::pad(4 * (size_t)N) +
                        2 * (5 * Arena::pad(4 * (size_t)nq) + Arena::pad((size_t)nq) + Arena::pad(8 * (size_t)kTopK * nq) + Arena::pad(4 * (size_t)nq) +
                             Arena::pad(2 * (kGridCells + 1)) + Arena::pad(2 * (size_t)N)) +
                        Arena::pad(32 * (size_t)nq) + Arena::pad((size_t)nq) + Arena::pad(4 * (size_t)nq) + Arena::pad(8 * (size_t)nq) +
                        Arena::pad(2 * sizeof(WindowProblem)) + 64 * 256 + 4096;
    int r = m->reserve_all(need);
    if (r != ORBX_OK) return r;
    Arena &A = m->arena;
    m->begin();
    WindowProblem P[2];
    memset(P, 0, sizeof(P));
    orbx_keypoint *dk = A.take<orbx_keypoint>(N);
    uint8_t *dd = A.take<uint8_t>(32 * (size_t)N);
    if (nl) H2D(dk, F->keypoints_un, 28 * (size_t)nl);
    if (nr) H2D(dk + nl, a.kps_right, 28 * (size_t)nr);
    H2D(dd, F->descriptors, 32 * (size_t)N);
    int32_t *dcnt = A.take<int32_t>(4);
    const int32_t cnts[3] = {nl, nr, nq};
    H2D(dcnt, cnts, 12);
    uint8_t *dqd = A.take<uint8_t>(32 * (size_t)nq);
    H2D(dqd, a.qdesc, 32 * (size_t)nq);
    for (int s = 0; s < 2; s++) {
        WindowProblem &w = P[s];
        w.kps = dk + (s ? nl : 0); w.desc = dd + (s ? (size_t)nl * 32 : 0); w.n_ptr = dcnt + s; w.nq_ptr = dcnt + 2;
        float *f3[3]; const float *h3[3] = {a.qx[s], a.qy[s], a.qr[s]};
        for (int k = 0; k < 3; k++) { f3[k] = A.take<float>(nq); H2D(f3[k], h3[k], 4 * (size_t)nq); }
        w.qx = f3[0]; w.qy = f3[1]; w.qr = f3[2];
        int32_t *i2[2]; const int32_t *hi2[2] = {a.qmin[s], a.qmax[s]};
        for (int k = 0; k < 2; k++) { i2[k] = A.take<int32_t>(nq); H2D(i2[k], hi2[k], 4 * (size_t)nq); }
        w.qmin = i2[0]; w.qmax = i2[1];
        { uint8_t *p = A.take<uint8_t>(nq); H2D(p, a.qvalid[s], (size_t)nq); w.qvalid = p; }
        w.qdesc = dqd;
        w.keys = A.take<u64>((size_t)nq * kTopK); w.meta = A.take<int32_t>(nq);
        w.gstart = A.take<uint16_t>(kGridCells + 1); w.gorder = A.take<uint16_t>(std::max(s ? nr : nl, 1));
    }
    TwinProblem T;
    memset(&T, 0, sizeof(T));
    T.mode = a.mode; T.n_left = nl; T.n_right = nr; T.nq = nq; T.nnratio = a.nnratio; T.max_dist = (float)ORBX_TH_HIGH;
    T.check_orientation = a.check_orientation; T.cleared_value = -2;
    if (a.l2r && nl) { int32_t *p = A.take<int32_t>(nl); H2D(p, a.l2r, 4 * (size_t)nl); T.l2r = p; }
    if (a.r2l && nr) { int32_t *p = A.take<int32_t>(nr); H2D(p, a.r2l, 4 * (size_t)nr); T.r2l = p; }
    if (a.occupied) { uint8_t *p = A.take<uint8_t>(N); H2D(p, a.occupied, (size_t)N); T.occupied0 = p; }
    if (a.q_has_obs) { uint8_t *p = A.take<uint8_t>(nq); H2D(p, a.q_has_obs, (size_t)nq); T.q_has_obs = p; }
    if (a.q_angle) { float *p = A.take<float>(nq); H2D(p, a.q_angle, 4 * (size_t)nq); T.q_angle = p; }
    T.match = A.take<int32_t>(N); T.nmatches = A.take<int32_t>(1); T.entries = A.take<int32_t>(2 * (size_t)nq);
    WindowProblem *dP = A.take<WindowProblem>(2);
    H2D(dP, P, sizeof(P));
    GridParams g;
    g.minx = F->min_x; g.miny = F->min_y;
    g.inv_w = 64.0f / (F->max_x - F->min_x);
    g.inv_h = 48.0f / (F->max_y - F->min_y);
    ORBX_LAUNCH_GRID_BUILD( dim3(2), dim3(64), 0, m->exec(), dP, g);
    ORBX_LAUNCH_WINDOW_BEST2(nq, 2, m->exec(), dP, g);
    const size_t lds = ((size_t)N + 63) & ~(size_t)63;
    hipLaunchKernelGGL(k_replay_twin, dim3(1), dim3(64), lds, m->exec(), dP, T, g);
    int32_t nm = 0;
    D2H(a.match_out, T.match, 4 * (size_t)N);
    D2H(&nm, T.nmatches, 4);
    SYNC_AND_DELIVER();
    return nm;
}

}  // namespace

// SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) for a fisheye-stereo frame (F.Nleft != -1), ORBmatcher.cc:43-213 whole
extern "C" int orbx_search_by_projection_mappoints_fisheye(orbx_matcher *m, const orbx_frame_desc *left, const orbx_keypoint *kps_right, int n_right,
                                                           const int32_t *left_to_right, const int32_t *right_to_left, const uint8_t *frame_occupied,
                                                           int n_mp, const uint8_t *in_view, const float *proj_x, const float *proj_y,
                                                           const int32_t *pred_level, const float *view_cos, const uint8_t *in_view_r,
                                                           const float *proj_xr, const float *proj_yr, const int32_t *pred_level_r,
                                                           const float *view_cos_r, const uint8_t *mp_desc, const uint8_t *mp_has_obs, float th,
                                                           float nnratio, int32_t *frame_match) {
    if (!m || !left || left->n < 0 || n_right < 0 || n_mp < 0 || (!frame_match && left->n + n_right > 0)) return ORBX_E_BAD_ARG;
    if ((left->n > 0 && !left_to_right) || (n_right > 0 && (!right_to_left || !kps_right))) return ORBX_E_BAD_ARG;
    if (n_mp > 0 && (!in_view || !proj_x || !proj_y || !pred_level || !view_cos || !in_view_r || !proj_xr || !proj_yr || !pred_level_r || !view_cos_r || !mp_desc))
        return ORBX_E_BAD_ARG;
    std::vector<float> qr[2];
    std::vector<int32_t> qmin[2], qmax[2];
    std::vector<uint8_t> valid[2];
    const bool bFactor = th != 1.0;
    for (int s = 0; s < 2; s++) { qr[s].resize(n_mp); qmin[s].resize(n_mp); qmax[s].resize(n_mp); valid[s].resize(n_mp); }
    for (int i = 0; i < n_mp; i++) {
        // left search :60-76
        const int lvl = pred_level[i];
        valid[0][i] = in_view[i] && lvl >= 0 && lvl < left->nlevels;
        float r = (view_cos[i] > 0.998) ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        qr[0][i] = valid[0][i] ? r * left->scale_factors[lvl] : 0.f;
        qmin[0][i] = lvl - 1; qmax[0][i] = lvl;
        // right twin :144-152: needs mbTrackInViewR and mnTrackScaleLevelR != -1; RadiusByViewingCos(mTrackViewCosR) WITHOUT the th factor
        const int lr = pred_level_r[i];
        valid[1][i] = in_view_r[i] && lr >= 0 && lr < left->nlevels;
        const float rr = (view_cos_r[i] > 0.998) ? 2.5f : 4.0f;
        qr[1][i] = valid[1][i] ? rr * left->scale_factors[lr] : 0.f;
        qmin[1][i] = lr - 1; qmax[1][i] = lr;
    }
    TwinArgs a = {left, kps_right, n_right, left_to_right, right_to_left, frame_occupied, n_mp, {proj_x, proj_xr}, {proj_y, proj_yr},
                  {qr[0].data(), qr[1].data()}, {qmin[0].data(), qmin[1].data()}, {qmax[0].data(), qmax[1].data()}, {valid[0].data(), valid[1].data()},
                  mp_desc, mp_has_obs, nullptr, 1, nnratio, 0, frame_match};
    return run_projection_twin(m, a);
}

// SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) for a fisheye-stereo current frame, ORBmatcher.cc:1676-1887 with :1794-1863
extern "C" int orbx_search_by_projection_frame_fisheye(orbx_matcher *m, const orbx_frame_desc *left, const orbx_keypoint *kps_right, int n_right,
                                                       const uint8_t *cur_occupied, int n_q, const float *q_u, const float *q_v, const float *q_ur,
                                                       const float *q_vr, const int32_t *q_octave, const float *q_angle, const uint8_t *q_desc,
                                                       const uint8_t *q_has_obs, float th, int level_mode, int check_orientation, int32_t *cur_match) {
    if (!m || !left || left->n < 0 || n_right < 0 || n_q < 0 || (!cur_match && left->n + n_right > 0) || (n_right > 0 && !kps_right)) return ORBX_E_BAD_ARG;
    if (n_q > 0 && (!q_u || !q_v || !q_ur || !q_vr || !q_octave || !q_desc || (check_orientation && !q_angle))) return ORBX_E_BAD_ARG;
    std::vector<float> qr(n_q);
    std::vector<int32_t> qmin(n_q), qmax(n_q);
    std::vector<uint8_t> valid(n_q);
    for (int i = 0; i < n_q; i++) {
        const int o = q_octave[i];
        valid[i] = o >= 0 && o < left->nlevels;
        qr[i] = valid[i] ? th * left->scale_factors[o] : 0.f;   // :1726, the twin uses the same radius (:1800)
        if (level_mode == 1) { qmin[i] = o; qmax[i] = -1; }
        else if (level_mode == 2) { qmin[i] = 0; qmax[i] = o; }
        else { qmin[i] = o - 1; qmax[i] = o + 1; }
    }
    TwinArgs a = {left, kps_right, n_right, nullptr, nullptr, cur_occupied, n_q, {q_u, q_ur}, {q_v, q_vr}, {qr.data(), qr.data()},
                  {qmin.data(), qmin.data()}, {qmax.data(), qmax.data()}, {valid.data(), valid.data()}, q_desc, q_has_obs, q_angle, 2, 0.f,
                  check_orientation, cur_match};
    return run_projection_twin(m, a);
}

extern "C" int orbx_search_by_projection_window(orbx_matcher *m, const orbx_frame_desc *frame, const uint8_t *occupied, int n_q,
                                                const float *q_x, const float *q_y, const float *q_r, const int32_t *q_min_level,
                                                const int32_t *q_max_level, const float *q_angle, const uint8_t *q_desc,
                                                const uint8_t *q_has_obs, float max_dist, int check_orientation, int32_t *match) {
    if (!m || !frame || frame->n < 0 || (!match && frame->n > 0) || n_q < 0) return ORBX_E_BAD_ARG;
    if (n_q > 0 && (!q_x || !q_y || !q_r || !q_min_level || !q_max_level || !q_desc || (check_orientation && !q_angle))) return ORBX_E_BAD_ARG;
    ProjArgs a = {frame, occupied, n_q, q_x, q_y, q_r, nullptr, q_min_level, q_max_level, q_desc, nullptr, q_has_obs,
                  q_angle, 2, 0.f, check_orientation, match, max_dist};
    return run_projection(m, a);
}

// ---------------------------------------------------------------------------------------------------------
// Matchers whose inner loop carries more state than a taken-mask: SearchForInitialization (vMatchedDistance) and the BoW
// merge-joins run entirely on the device (k_replay_init / k_replay_bow, one wave replays the reference's query order).
// SearchForTriangulation with a geometric gate keeps one host piece: the camera model's epipolarConstrain lives in the
// adapter (camera models are out of scope), so the device evaluates every candidate distance of the reference's enumeration
// and the loop calls the gate back exactly where the reference evaluates it.
// ---------------------------------------------------------------------------------------------------------
namespace {

inline int rotation_bin(float a1, float a2) {  // e.g. ORBmatcher.cc:337-343
    float rot = a1 - a2;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)std::round(rot * (1.0f / ORBX_HISTO_LENGTH));
    if (bin == ORBX_HISTO_LENGTH) bin = 0;
    return bin;
}

// ComputeThreeMaxima (ORBmatcher.cc:2012-2053) + removal of the losing bins; entries = (bin, key) in push order
struct RotHist {
    std::vector<std::pair<int, int>> entries;
    void push(int bin, int key) { entries.emplace_back(bin, key); }
    template <class Drop> void filter(Drop drop) const {
        int cnt[ORBX_HISTO_LENGTH] = {0};
        for (auto &e : entries) cnt[e.first]++;
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < ORBX_HISTO_LENGTH; i++) {
            const int s = cnt[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (auto &e : entries)
            if (e.first != ind1 && e.first != ind2 && e.first != ind3) drop(e.second);
    }
};

// merge-join of two flattened DBoW2::FeatureVector maps: equal node ids in ascending order
template <class V> void for_common_nodes(const orbx_featvec *a, const orbx_featvec *b, V visit) {
    int ia = 0, ib = 0;
    while (ia < a->n_nodes && ib < b->n_nodes) {
        const uint32_t na = a->node_id[ia], nb = b->node_id[ib];
        if (na == nb) { visit(ia, ib); ia++; ib++; }
        else if (na < nb) ia++;
        else ib++;
    }
}

// queries = features of A listed in the common nodes (those with qmask clear), candidates = the node's list in B.
// Builds the CSR in the reference's enumeration order and evaluates all distances on the GPU.
struct BowPairs {
    std::vector<int32_t> q_idx, row_ptr, cand;
    std::vector<uint16_t> dist;
};
int bow_distances(orbx_matcher *m, const uint8_t *descA, const uint8_t *skipA, const orbx_featvec *fvA, const uint8_t *descB, int nB,
                  const orbx_featvec *fvB, BowPairs &P) {
    P.row_ptr.assign(1, 0);
    for_common_nodes(fvA, fvB, [&](int ia, int ib) {
        for (int a = fvA->node_ptr[ia]; a < fvA->node_ptr[ia + 1]; a++) {
            const int i = fvA->index[a];
            if (skipA && skipA[i]) continue;
            P.q_idx.push_back(i);
            for (int b = fvB->node_ptr[ib]; b < fvB->node_ptr[ib + 1]; b++) P.cand.push_back(fvB->index[b]);
            P.row_ptr.push_back((int32_t)P.cand.size());
        }
    });
    const int nq = (int)P.q_idx.size();
    P.dist.assign(P.cand.size(), 0);
    if (nq == 0 || P.cand.empty()) return ORBX_OK;
    std::vector<uint8_t> qd((size_t)nq * 32);
    for (int k = 0; k < nq; k++) memcpy(&qd[(size_t)k * 32], descA + (size_t)P.q_idx[k] * 32, 32);
    return orbx_hamming_csr(m, qd.data(), nq, descB, nB, P.row_ptr.data(), P.cand.data(), P.dist.data());
}

}  // namespace

extern "C" {

// ORBmatcher::SearchForInitialization (ORBmatcher.cc:648-763).  Round 6: F2's grid, the candidate lists of every level-0 keypoint of F1 (k_window_best2_t<64>:
// a wave per query, the lists do not depend on the loop's state) and a one-wave replay of the loop over those lists (k_replay_init_lists);
// frames beyond kMaxResolveFeatures keep the single-wave scan k_replay_init (the replay's state no longer fits the LDS).
int orbx_search_for_initialization(orbx_matcher *m, const orbx_keypoint *kps1_un, const uint8_t *desc1, int n1, const orbx_frame_desc *F2,
                                   float *prev_matched, int window_size, float nnratio, int check_orientation, int32_t *matches12) {
    if (!m || !F2 || n1 < 0 || (n1 > 0 && (!matches12 || !kps1_un || !desc1 || !prev_matched))) return ORBX_E_BAD_ARG;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    const int n2 = F2->n;
    if (n1 == 0 || n2 == 0) return 0;
    if (n1 > 65535 || n2 > 65535) return ORBX_E_TOO_LARGE;
    ORBX_HIP(hipSetDevice(m->device));
    GridParams g;
    g.minx = F2->min_x; g.miny = F2->min_y;
    g.inv_w = 64.0f / (F2->max_x - F2->min_x);
    g.inv_h = 48.0f / (F2->max_y - F2->min_y);
    if (n1 > kMaxResolveFeatures || n2 > kMaxResolveFeatures) {
        const size_t need = Arena::pad(28 * (size_t)n1) + Arena::pad(32 * (size_t)n1) + Arena::pad(28 * (size_t)n2) + Arena::pad(32 * (size_t)n2) +
                            Arena::pad(8 * (size_t)n1) + 2 * Arena::pad(4 * (size_t)n1) + 2 * Arena::pad(4 * (size_t)n2) + 4096;
        int r = m->reserve_all(need);
        if (r != ORBX_OK) return r;
        Arena &A = m->arena;
        m->begin();
        InitProblem P;
        memset(&P, 0, sizeof(P));
        orbx_keypoint *dk1 = A.take<orbx_keypoint>(n1), *dk2 = A.take<orbx_keypoint>(n2);
        uint8_t *dd1 = A.take<uint8_t>(32 * (size_t)n1), *dd2 = A.take<uint8_t>(32 * (size_t)n2);
        float *dprev = A.take<float>(2 * (size_t)n1);
        H2D(dk1, kps1_un, 28 * (size_t)n1); H2D(dd1, desc1, 32 * (size_t)n1);
        H2D(dk2, F2->keypoints_un, 28 * (size_t)n2); H2D(dd2, F2->descriptors, 32 * (size_t)n2);
        H2D(dprev, prev_matched, 8 * (size_t)n1);
        P.kps1 = dk1; P.desc1 = dd1; P.n1 = n1; P.kps2 = dk2; P.desc2 = dd2; P.n2 = n2; P.prev_matched = dprev;
        P.window = (float)window_size; P.nnratio = nnratio; P.check_orientation = check_orientation ? 1 : 0;
        P.matches12 = A.take<int32_t>(n1); P.entries = A.take<int32_t>(n1);
        P.matches21 = A.take<int32_t>(n2); P.matched_dist = A.take<int32_t>(n2);
        P.nmatches = A.take<int32_t>(4);
        hipLaunchKernelGGL(k_replay_init, dim3(1), dim3(64), 0, m->exec(), P, g);
        int32_t nm = 0;
        D2H(matches12, P.matches12, 4 * (size_t)n1);
        D2H(prev_matched, dprev, 8 * (size_t)n1);
        D2H(&nm, P.nmatches, 4);
        SYNC_AND_DELIVER();
        return nm;
    }
    // the queries: keypoints of F1 on level 0, in index order (:661-666); window = vbPrevMatched[i1] +- windowSize on level [level1, level1] (:668)
    std::vector<int32_t> qidx;
    qidx.reserve(n1);
    for (int i = 0; i < n1; i++)
        if (kps1_un[i].octave <= 0) qidx.push_back(i);
    const int nq = (int)qidx.size();
    if (nq == 0) return 0;
    std::vector<float> qx(nq), qy(nq), qr(nq, (float)window_size);
    std::vector<int32_t> qlv(nq);
    std::vector<uint8_t> qd(32 * (size_t)nq);
    for (int k = 0; k < nq; k++) {
        const int i = qidx[k];
        qx[k] = prev_matched[2 * i]; qy[k] = prev_matched[2 * i + 1]; qlv[k] = kps1_un[i].octave;
        memcpy(&qd[32 * (size_t)k], desc1 + 32 * (size_t)i, 32);
    }
    const size_t lds = 4 * (size_t)n2 + 2 * (size_t)n2 * 2 + 2 * (size_t)n1 + 64;
    // every candidate key of every query for the replay's re-evaluations (WindowProblem::all_keys): up to 512 per query within 8 MB
    const int all_cap = (int)std::min<size_t>(512, std::max<size_t>(64, ((size_t)8 << 20) / (8 * (size_t)nq)));
    const size_t need = Arena::pad(28 * (size_t)n1) + Arena::pad(28 * (size_t)n2) + Arena::pad(32 * (size_t)n2) + Arena::pad(8 * (size_t)n1) +
                        Arena::pad(4 * (size_t)nq) * 6 + Arena::pad(32 * (size_t)nq) + Arena::pad(8 * (size_t)nq * kTopK) + Arena::pad(4 * (size_t)nq) * 2 +
                        Arena::pad(sizeof(WindowProblem)) + Arena::pad(2 * (kGridCells + 1)) + Arena::pad(2 * (size_t)n2) + Arena::pad(4 * (size_t)n1) + 16 * 256 + 4096 +
                        Arena::pad(4 * (size_t)nq) + Arena::pad(8 * (size_t)nq * all_cap);
    int r = m->reserve_all(need);
    if (r != ORBX_OK) return r;
    Arena &A = m->arena;
    m->begin();
    WindowProblem P;
    memset(&P, 0, sizeof(P));
    InitReplay R;
    memset(&R, 0, sizeof(R));
    // every upload in one run of the arena, the problem record directly behind the inputs; device-only buffers behind it; the three downloads side by side
    orbx_keypoint *dk1 = A.take<orbx_keypoint>(n1), *dk2 = A.take<orbx_keypoint>(n2);
    uint8_t *dd2 = A.take<uint8_t>(32 * (size_t)n2);
    H2D(dk1, kps1_un, 28 * (size_t)n1); H2D(dk2, F2->keypoints_un, 28 * (size_t)n2); H2D(dd2, F2->descriptors, 32 * (size_t)n2);
    P.kps = dk2; P.desc = dd2;
    int32_t *dcnt = A.take<int32_t>(4);
    const int32_t cnts[2] = {n2, nq};
    H2D(dcnt, cnts, 8);
    P.n_ptr = dcnt; P.nq_ptr = dcnt + 1;
    float *dqx = A.take<float>(nq), *dqy = A.take<float>(nq), *dqr = A.take<float>(nq);
    int32_t *dql = A.take<int32_t>(nq), *dqi = A.take<int32_t>(nq);
    uint8_t *dqd = A.take<uint8_t>(32 * (size_t)nq);
    H2D(dqx, qx.data(), 4 * (size_t)nq); H2D(dqy, qy.data(), 4 * (size_t)nq); H2D(dqr, qr.data(), 4 * (size_t)nq);
    H2D(dql, qlv.data(), 4 * (size_t)nq); H2D(dqi, qidx.data(), 4 * (size_t)nq); H2D(dqd, qd.data(), 32 * (size_t)nq);
    P.qx = dqx; P.qy = dqy; P.qr = dqr; P.qmin = dql; P.qmax = dql; P.qdesc = dqd;
    WindowProblem *dP = A.take<WindowProblem>(1);
    P.keys = A.take<u64>((size_t)nq * kTopK); P.meta = A.take<int32_t>(nq);
    P.gstart = A.take<uint16_t>(kGridCells + 1); P.gorder = A.take<uint16_t>(n2);
    P.all_cap = all_cap; P.all_cnt = A.take<int32_t>(nq); P.all_keys = A.take<u64>((size_t)nq * all_cap);
    R.q_index = dqi; R.kps1 = dk1; R.n1 = n1; R.n2 = n2; R.nq = nq; R.nnratio = nnratio; R.check_orientation = check_orientation ? 1 : 0;
    R.entries = A.take<int32_t>(nq);
    R.matches12 = A.take<int32_t>(n1);
    R.prev_matched = A.take<float>(2 * (size_t)n1);
    R.nmatches = A.take<int32_t>(4);
    H2D(R.prev_matched, prev_matched, 8 * (size_t)n1);
    H2D(dP, &P, sizeof(P));
    ORBX_LAUNCH_GRID_BUILD(dim3(1), dim3(64), 0, m->exec(), dP, g);
    hipLaunchKernelGGL(k_window_best2_t<64>, dim3(1, (unsigned)((nq + 3) / 4), 1), dim3(256), 0, m->exec(), dP, g, 1);   // a wave per query (100-px windows: hundreds of candidates)
    if (lds > 64 * 1024) ORBX_HIP(hipFuncSetAttribute((const void *)k_replay_init_lists, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_replay_init_lists, dim3(1), dim3(64), lds, m->exec(), dP, R, g);
    ORBX_HIP(hipGetLastError());
    int32_t nm[4] = {0, 0, 0, 0};
    D2H(matches12, R.matches12, 4 * (size_t)n1);
    D2H(prev_matched, R.prev_matched, 8 * (size_t)n1);
    D2H(nm, R.nmatches, 16);
    SYNC_AND_DELIVER();
    for (int k = 0; k < 3; k++) m->replay_stats[k] = nm[k + 1];
    return nm[0];
}

}  // extern "C"

namespace {
// k_replay_bow on host pointers: mode 0 SearchByBoW(KF, Frame), 1 SearchByBoW(KF, KF), 2 SearchForTriangulation without a gate
int run_bow_replay(orbx_matcher *m, int mode, const uint8_t *desc_a, const float *angle_a, const uint8_t *skip_a, int na, const orbx_featvec *fa,
                   const uint8_t *desc_b, const float *angle_b, const uint8_t *skip_b, int nb, const orbx_featvec *fb, float nnratio,
                   int check_orientation, int32_t *match_out, int n_out, const orbx_pinhole_gate *gate = nullptr, int nb_left = 0,
                   const orbx_kb8_gate *kgate = nullptr) {
    if (na > 65535 || nb > 65535) return ORBX_E_TOO_LARGE;
    ORBX_HIP(hipSetDevice(m->device));
    const size_t ia = (size_t)fa->node_ptr[fa->n_nodes], ib = (size_t)fb->node_ptr[fb->n_nodes];
    const size_t need = Arena::pad(33 * (size_t)na) + Arena::pad(33 * (size_t)nb) + 3 * Arena::pad(4 * (size_t)std::max(na, nb)) * 2 +
                        Arena::pad(8 * (size_t)fa->n_nodes + 8) + Arena::pad(8 * (size_t)fb->n_nodes + 8) + Arena::pad(4 * ia) + Arena::pad(4 * ib) +
                        Arena::pad(4 * (size_t)fa->n_nodes + 4) + 1024 +
                        Arena::pad(4 * (size_t)na) + Arena::pad(4 * (size_t)nb) + 8192 +
                        (gate ? Arena::pad(sizeof(orbx_keypoint) * (size_t)na) + Arena::pad(sizeof(orbx_keypoint) * (size_t)nb) +
                                    Arena::pad(4 * (size_t)na) + Arena::pad(4 * (size_t)nb) + 4 * Arena::pad(4 * 64) : 0) +
                        (kgate ? Arena::pad(sizeof(orbx_keypoint) * (size_t)na) + Arena::pad(sizeof(orbx_keypoint) * (size_t)nb) + 2 * Arena::pad(4 * 64) +
                                     Arena::pad(sizeof(Kb8Gate)) : 0);
    int r = m->reserve_all(need);
    if (r != ORBX_OK) return r;
    Arena &A = m->arena;
    m->begin();
    BowProblem P;
    memset(&P, 0, sizeof(P));
    auto up_fv = [&](const orbx_featvec *f, FeatVecDev &d, size_t nidx) -> int {
        uint32_t *id = A.take<uint32_t>(f->n_nodes + 1);
        int32_t *ptr = A.take<int32_t>(f->n_nodes + 1), *idx = A.take<int32_t>(nidx + 1);
        if (f->n_nodes > 0) H2D(id, f->node_id, 4 * (size_t)f->n_nodes);
        H2D(ptr, f->node_ptr, 4 * (size_t)(f->n_nodes + 1));
        if (nidx > 0) H2D(idx, f->index, 4 * nidx);
        d.node_id = id; d.node_ptr = ptr; d.index = idx; d.n_nodes = f->n_nodes;
        return ORBX_OK;
    };
    if ((r = up_fv(fa, P.fa, ia)) != ORBX_OK || (r = up_fv(fb, P.fb, ib)) != ORBX_OK) return r;
    {   // the merge-join of the two sorted node-id lists (:246-250, :800-805, :961-965) on the host: node ia of A pairs with node pair_b[ia] of B
        std::vector<int32_t> pair((size_t)fa->n_nodes + 1, -1);
        for_common_nodes(fa, fb, [&](int a_, int b_) { pair[a_] = b_; });
        int32_t *dpair = A.take<int32_t>(fa->n_nodes + 1);
        H2D(dpair, pair.data(), 4 * ((size_t)fa->n_nodes + 1));
        P.pair_b = dpair;
    }
    auto up = [&](const void *src, size_t bytes) -> const uint8_t * {
        if (!src) return nullptr;
        uint8_t *d = A.take<uint8_t>(bytes);
        if (bytes > 0) {   // recorded like H2D(): issued with its arena neighbours by exec()
            if (!m->stageable(d, bytes)) return nullptr;
            const size_t o = (size_t)(d - m->arena.base);
            memcpy(m->mirror.base + o, src, bytes);
            m->record_upload(o, bytes);
        }
        return d;
    };
    P.mode = mode; P.nb_left = nb_left;
    { const char *dbg = getenv("ORBX_BOW_DEBUG"); P.debug_stop = dbg ? atoi(dbg) : 0; }
    P.desc_a = up(desc_a, 32 * (size_t)na); P.desc_b = up(desc_b, 32 * (size_t)nb);
    P.angle_a = (const float *)up(angle_a, 4 * (size_t)na); P.angle_b = (const float *)up(angle_b, 4 * (size_t)nb);
    P.skip_a = up(skip_a, na); P.skip_b = up(skip_b, nb);
    if (!P.desc_a || !P.desc_b || (check_orientation && (!P.angle_a || !P.angle_b))) return ORBX_E_BAD_ARG;
    P.na = na; P.nb = nb; P.nnratio = nnratio; P.check_orientation = check_orientation ? 1 : 0;
    if (gate) {  // the pinhole gates of SearchForTriangulation run inside the kernel (mode 2)
        TriGate &G = P.gate;
        G.enabled = 1; G.coarse = gate->coarse ? 1 : 0; G.strict = gate->strict_fp ? 1 : 0;
        G.k1 = (const orbx_keypoint *)up(gate->kps1_un, sizeof(orbx_keypoint) * (size_t)na);
        G.k2 = (const orbx_keypoint *)up(gate->kps2_un, sizeof(orbx_keypoint) * (size_t)nb);
        G.ur1 = (const float *)up(gate->u_right1, 4 * (size_t)na);
        G.ur2 = (const float *)up(gate->u_right2, 4 * (size_t)nb);
        G.scale2 = (const float *)up(gate->scale_factors2, 4 * (size_t)gate->nlevels);
        G.sigma2_2 = (const float *)up(gate->level_sigma2_2, 4 * (size_t)gate->nlevels);
        if (!G.k1 || !G.k2 || !G.scale2 || !G.sigma2_2) return ORBX_E_BAD_ARG;
        for (int i = 0; i < 9; i++) G.F[i] = gate->F12[i];
        G.ex = gate->ep_x; G.ey = gate->ep_y;
    }
    const bool kb8_kernel = kgate && !kgate->coarse;   // bCoarse: no gate at all for such key frames (no epipole test either, :1026) = k_replay_bow without a gate
    if (kb8_kernel) {  // fisheye key frames: KannalaBrandt8::epipolarConstrain on the device (k_tri_kb8)
        TriGate &G = P.gate;
        G.enabled = 1; G.coarse = 0; G.strict = 1;
        G.k1 = (const orbx_keypoint *)up(kgate->kps1, sizeof(orbx_keypoint) * (size_t)na);
        G.k2 = (const orbx_keypoint *)up(kgate->kps2, sizeof(orbx_keypoint) * (size_t)nb);
        G.sigma2_2 = (const float *)up(kgate->level_sigma2_2, 4 * (size_t)kgate->nlevels);
        Kb8Gate K;
        memset(&K, 0, sizeof(K));
        K.n_left1 = kgate->n_left1; K.n_left2 = kgate->n_left2;
        K.sigma2_1 = (const float *)up(kgate->level_sigma2_1, 4 * (size_t)kgate->nlevels);
        memcpy(K.cam[0], kgate->cam1, sizeof(float) * 16);
        memcpy(K.cam[2], kgate->cam2, sizeof(float) * 16);
        memcpy(K.R12, kgate->R12, sizeof(K.R12));
        memcpy(K.t12, kgate->t12, sizeof(K.t12));
        if (!G.k1 || !G.k2 || !G.sigma2_2 || !K.sigma2_1) return ORBX_E_BAD_ARG;
        G.kb8 = (const Kb8Gate *)up(&K, sizeof(K));
        if (!G.kb8) return ORBX_E_BAD_ARG;
    }
    // the two downloads side by side (one DMA), the two zero-filled buffers side by side (one fill)
    P.match = A.take<int32_t>(n_out); P.nmatches = A.take<int32_t>(4);
    P.taken_b = A.take<uint8_t>(nb);
    P.hist = A.take<int32_t>(ORBX_HISTO_LENGTH + 2); P.counters = P.hist + ORBX_HISTO_LENGTH;
    P.entries = A.take<int32_t>(2 * (size_t)std::max(na, nb));
    ORBX_HIP(m->fill(P.match, 0xff, 4 * (size_t)n_out));     // -1: no match.  Both fills ride in the launch that brings the inputs (orbx_matcher::fill)
    ORBX_HIP(m->fill(P.taken_b, 0, (size_t)((const uint8_t *)(P.hist + ORBX_HISTO_LENGTH + 2) - P.taken_b)));   // taken_b, (padding,) hist + counters
    if (fa->n_nodes > 0) {   // a wave per vocabulary node
        if (kb8_kernel) hipLaunchKernelGGL(k_tri_kb8, dim3((fa->n_nodes + 3) / 4), dim3(256), 0, m->exec(), P);
        else hipLaunchKernelGGL(k_replay_bow, dim3((fa->n_nodes + 3) / 4), dim3(256), 0, m->exec(), P);
    }
    hipLaunchKernelGGL(k_replay_bow_finish, dim3(1), dim3(64), 0, m->exec(), P);
    int32_t nm = 0;
    D2H(match_out, P.match, 4 * (size_t)n_out);
    D2H(&nm, P.nmatches, 4);
    SYNC_AND_DELIVER();
    return nm;
}
}  // namespace

extern "C" {

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (ORBmatcher.cc:223-425), monocular form: k_replay_bow mode 0
int orbx_search_by_bow_frame(orbx_matcher *m, const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid, int n_kf,
                             const orbx_featvec *kf_fv, const uint8_t *f_desc, const float *f_angle, int n_f, const orbx_featvec *f_fv,
                             float nnratio, int check_orientation, int32_t *f_match) {
    if (!m || !kf_fv || !f_fv || (!f_match && n_f > 0) || n_kf < 0 || n_f < 0) return ORBX_E_BAD_ARG;
    for (int i = 0; i < n_f; i++) f_match[i] = -1;
    if (n_kf == 0 || n_f == 0) return 0;
    std::vector<uint8_t> skip(n_kf);
    for (int i = 0; i < n_kf; i++) skip[i] = kf_valid ? !kf_valid[i] : 0;
    return run_bow_replay(m, 0, kf_desc, kf_angle, skip.data(), n_kf, kf_fv, f_desc, f_angle, nullptr, n_f, f_fv, nnratio, check_orientation,
                          f_match, n_f);
}

// the same for a fisheye-stereo frame (F.Nleft != -1, ORBmatcher.cc:283-392): features >= n_f_left belong to the right camera
int orbx_search_by_bow_frame_fisheye(orbx_matcher *m, const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid, int n_kf,
                                     const orbx_featvec *kf_fv, const uint8_t *f_desc, const float *f_angle, int n_f, int n_f_left,
                                     const orbx_featvec *f_fv, float nnratio, int check_orientation, int32_t *f_match) {
    if (!m || !kf_fv || !f_fv || (!f_match && n_f > 0) || n_kf < 0 || n_f < 0 || n_f_left < 0 || n_f_left > n_f) return ORBX_E_BAD_ARG;
    for (int i = 0; i < n_f; i++) f_match[i] = -1;
    if (n_kf == 0 || n_f == 0) return 0;
    std::vector<uint8_t> skip(n_kf);
    for (int i = 0; i < n_kf; i++) skip[i] = kf_valid ? !kf_valid[i] : 0;
    return run_bow_replay(m, 3, kf_desc, kf_angle, skip.data(), n_kf, kf_fv, f_desc, f_angle, nullptr, n_f, f_fv, nnratio, check_orientation,
                          f_match, n_f, nullptr, n_f_left);
}

// ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (ORBmatcher.cc:765-905): k_replay_bow mode 1
int orbx_search_by_bow_keyframes(orbx_matcher *m, const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1,
                                 const orbx_featvec *fv1, const uint8_t *desc2, const float *angle2, const uint8_t *valid2, int n2,
                                 const orbx_featvec *fv2, float nnratio, int check_orientation, int32_t *match12) {
    if (!m || !fv1 || !fv2 || (!match12 && n1 > 0) || n1 < 0 || n2 < 0) return ORBX_E_BAD_ARG;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0) return 0;
    std::vector<uint8_t> skip1(n1), skip2(n2);
    for (int i = 0; i < n1; i++) skip1[i] = valid1 ? !valid1[i] : 0;
    for (int i = 0; i < n2; i++) skip2[i] = valid2 ? !valid2[i] : 0;
    return run_bow_replay(m, 1, desc1, angle1, skip1.data(), n1, fv1, desc2, angle2, skip2.data(), n2, fv2, nnratio, check_orientation, match12, n1);
}

// ORBmatcher::SearchForTriangulation (ORBmatcher.cc:907-1146)
int orbx_search_for_triangulation(orbx_matcher *m, const uint8_t *desc1, const float *angle1, const uint8_t *skip1, int n1,
                                  const orbx_featvec *fv1, const uint8_t *desc2, const float *angle2, const uint8_t *skip2, int n2,
                                  const orbx_featvec *fv2, int check_orientation, orbx_pair_predicate pair_ok, void *user,
                                  int32_t *matches12) {
    if (!m || !fv1 || !fv2 || (!matches12 && n1 > 0) || n1 < 0 || n2 < 0) return ORBX_E_BAD_ARG;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return 0;
    // without a geometric gate (bCoarse) nothing but distances decides: the whole loop runs on the device (k_replay_bow mode 2).
    // With a gate, the camera model's epipolarConstrain is host code of the adapter: the device evaluates every candidate
    // distance and the loop below calls the gate exactly where the reference evaluates it.
    if (!pair_ok)
        return run_bow_replay(m, 2, desc1, angle1, skip1, n1, fv1, desc2, angle2, skip2, n2, fv2, 0.f, check_orientation, matches12, n1);
    BowPairs P;
    const int r = bow_distances(m, desc1, skip1, fv1, desc2, n2, fv2, P);
    if (r != ORBX_OK) return r;
    int nmatches = 0;
    RotHist hist;
    for (size_t k = 0; k < P.q_idx.size(); k++) {
        const int i1 = P.q_idx[k];
        int best = ORBX_TH_LOW, best2 = -1;
        for (int c = P.row_ptr[k]; c < P.row_ptr[k + 1]; c++) {
            const int i2 = P.cand[c], d = P.dist[c];
            if (skip2 && skip2[i2]) continue;            // pMP2 (vbMatched2 is never set in v1.0)
            if (d > ORBX_TH_LOW || d > best) continue;     // :1017 -- '>' : a later equal candidate wins
            if (!pair_ok || pair_ok(user, i1, i2)) { best2 = i2; best = d; }  // epipole gate + epipolarConstrain / bCoarse
        }
        if (best2 >= 0) {
            matches12[i1] = best2;
            nmatches++;
            if (check_orientation) hist.push(rotation_bin(angle1[i1], angle2[best2]), i1);
        }
    }
    if (check_orientation) hist.filter([&](int i1) { matches12[i1] = -1; nmatches--; });
    return nmatches;
}

// SearchForTriangulation for pinhole key frames with both geometric gates on the device (k_replay_bow mode 2 + tri_gate): the
// epipole-distance test (ORBmatcher.cc:1026-1034) and Pinhole::epipolarConstrain (CameraModels/Pinhole.cpp:107-129) on the caller's
// F12.  No callback, no host loop.
int orbx_search_for_triangulation_pinhole(orbx_matcher *m, const uint8_t *desc1, const uint8_t *skip1, int n1, const orbx_featvec *fv1,
                                          const uint8_t *desc2, const uint8_t *skip2, int n2, const orbx_featvec *fv2,
                                          int check_orientation, const orbx_pinhole_gate *gate, int32_t *matches12) {
    if (!m || !fv1 || !fv2 || (!matches12 && n1 > 0) || !gate || n1 < 0 || n2 < 0) return ORBX_E_BAD_ARG;
    if (!gate->kps1_un || !gate->kps2_un || !gate->scale_factors2 || !gate->level_sigma2_2 || gate->nlevels <= 0 || gate->nlevels > 64)
        return ORBX_E_BAD_ARG;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return 0;
    std::vector<float> a1(n1), a2(n2);  // kp.angle of mvKeysUn (:1086-1094)
    for (int i = 0; i < n1; i++) a1[i] = gate->kps1_un[i].angle;
    for (int i = 0; i < n2; i++) a2[i] = gate->kps2_un[i].angle;
    return run_bow_replay(m, 2, desc1, a1.data(), skip1, n1, fv1, desc2, a2.data(), skip2, n2, fv2, 0.f, check_orientation, matches12, n1, gate);
}

// SearchForTriangulation between two key frames of a FISHEYE rig (pKF->mpCamera2 != NULL): the gate of :1036-1072 -- KannalaBrandt8::epipolarConstrain with the
// camera pair and relative pose the two feature indices select -- evaluated by k_tri_kb8 over the node's list of pairs within TH_LOW (until round 6 a host
// callback around a download of every candidate distance); bCoarse: no gate at all, k_replay_bow mode 2
int orbx_search_for_triangulation_kb8(orbx_matcher *m, const uint8_t *desc1, const uint8_t *skip1, int n1, const orbx_featvec *fv1, const uint8_t *desc2,
                                      const uint8_t *skip2, int n2, const orbx_featvec *fv2, int check_orientation, const orbx_kb8_gate *gate,
                                      int32_t *matches12) {
    if (!m || !fv1 || !fv2 || (!matches12 && n1 > 0) || !gate || n1 < 0 || n2 < 0) return ORBX_E_BAD_ARG;
    if (!gate->kps1 || !gate->kps2 || !gate->level_sigma2_1 || !gate->level_sigma2_2 || gate->nlevels <= 0 || gate->nlevels > 64) return ORBX_E_BAD_ARG;
    if (gate->n_left1 < 0 || gate->n_left1 > n1 || gate->n_left2 < 0 || gate->n_left2 > n2) return ORBX_E_BAD_ARG;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return 0;
    for (int i = 0; i < n1; i++) if (gate->kps1[i].octave < 0 || gate->kps1[i].octave >= gate->nlevels) return ORBX_E_BAD_ARG;
    for (int i = 0; i < n2; i++) if (gate->kps2[i].octave < 0 || gate->kps2[i].octave >= gate->nlevels) return ORBX_E_BAD_ARG;
    std::vector<float> a1(n1), a2(n2);  // kp.angle (:1086-1094)
    for (int i = 0; i < n1; i++) a1[i] = gate->kps1[i].angle;
    for (int i = 0; i < n2; i++) a2[i] = gate->kps2[i].angle;
    return run_bow_replay(m, 2, desc1, a1.data(), skip1, n1, fv1, desc2, a2.data(), skip2, n2, fv2, 0.f, check_orientation, matches12, n1, nullptr, 0, gate);
}

// test hook: KannalaBrandt8::epipolarConstrain of n independent pairs on the device (k_debug_kb8_gate); sel[i] = 2 * right1 + right2 picks cam1[right1],
// cam2[right2] and R12 / t12 [sel] as the search does per candidate
int orbx_debug_kb8_epipolar(orbx_matcher *m, const float *cam1_2x8, const float *cam2_2x8, const float *R12_4x9, const float *t12_4x3, int n, const float *xy1,
                            const float *xy2, const float *sigma1, const float *sigma2, const uint8_t *sel, uint8_t *ok) {
    if (!m || !cam1_2x8 || !cam2_2x8 || !R12_4x9 || !t12_4x3 || n < 0 || (n > 0 && (!xy1 || !xy2 || !sigma1 || !sigma2 || !sel || !ok))) return ORBX_E_BAD_ARG;
    if (n == 0) return ORBX_OK;
    for (int i = 0; i < n; i++) if (sel[i] > 3) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(m->device));
    const size_t N = (size_t)n;
    int r = m->reserve_all(2 * Arena::pad(8 * N) + 2 * Arena::pad(4 * N) + 2 * Arena::pad(N) + Arena::pad(sizeof(Kb8Gate)) + 4096);
    if (r != ORBX_OK) return r;
    Arena &A = m->arena;
    m->begin();
    float *d1 = A.take<float>(2 * N), *d2 = A.take<float>(2 * N), *s1 = A.take<float>(N), *s2 = A.take<float>(N);
    uint8_t *dsel = A.take<uint8_t>(N), *dok = A.take<uint8_t>(N);
    Kb8Gate *dg = A.take<Kb8Gate>(1);
    Kb8Gate K;
    memset(&K, 0, sizeof(K));
    memcpy(K.cam[0], cam1_2x8, sizeof(float) * 16);
    memcpy(K.cam[2], cam2_2x8, sizeof(float) * 16);
    memcpy(K.R12, R12_4x9, sizeof(K.R12));
    memcpy(K.t12, t12_4x3, sizeof(K.t12));
    H2D(dg, &K, sizeof(K));
    H2D(d1, xy1, 8 * N); H2D(d2, xy2, 8 * N); H2D(s1, sigma1, 4 * N); H2D(s2, sigma2, 4 * N); H2D(dsel, sel, N);
    hipLaunchKernelGGL(k_debug_kb8_gate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->exec(), (const Kb8Gate *)dg, n, (const float *)d1, (const float *)d2,
                       (const float *)s1, (const float *)s2, (const uint8_t *)dsel, dok);
    ORBX_HIP(hipGetLastError());
    D2H(ok, dok, N);
    SYNC_AND_DELIVER();
    return ORBX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Device-resident batched frame-to-frame matcher (throughput path): problem f-1 matches frame f-1 (queries)
// into frame f of the extractor's last batch; everything stays in HBM on the extractor's stream.
// ---------------------------------------------------------------------------------------------------------
extern "C" int orbx_match_consecutive_device(orbx_extractor *ex, float th, float du, float dv, int check_orientation,
                                             int32_t *d_match, int32_t *d_nmatches) {
    RoctxRange rr("orbx:match_consecutive");
    if (!ex) return ORBX_E_BAD_ARG;
    const int n = ex->last_batch;
    if (n < 2) return ORBX_OK;
    if (ex->cap > kMaxResolveFeatures) return ORBX_E_TOO_LARGE;
    ORBX_HIP(hipSetDevice(ex->device));
    const int np = n - 1, cap = ex->cap;
    int r;
    if (!d_match || !d_nmatches) {  // internal result buffers (downloaded by orbx_batch_download_async)
        if (ex->internal_match_owner == 2) { set_error("the internal match buffers hold the map-point search of this batch: download it first, or pass result buffers"); return ORBX_E_BAD_ARG; }
        ex->internal_match_owner = 1;
        if (ex->d_match.bytes < 4 * (size_t)cap * ex->batch_cap || ex->d_nmatch.bytes < 4 * (size_t)ex->batch_cap) {
            // growing frees the old buffers: an earlier matcher / download may still be using them
            ORBX_HIP(hipStreamSynchronize(ex->match_stream)); ORBX_HIP(hipStreamSynchronize(ex->copy_stream));
        }
        if ((r = ex->d_match.ensure(4 * (size_t)cap * ex->batch_cap)) != ORBX_OK) return r;
        if ((r = ex->d_nmatch.ensure(4 * (size_t)ex->batch_cap)) != ORBX_OK) return r;
        d_match = (int32_t *)ex->d_match.p;
        d_nmatches = (int32_t *)ex->d_nmatch.p;
    }
    const orbx_keypoint *kps = (const orbx_keypoint *)ex->match_kps();   // mvKeysUn (== mvKeys without a distortion model)
    orbx_extractor::MatchKey key;
    key.n = n; key.cap = cap; key.match = d_match; key.nm = d_nmatches; key.th = th; key.du = du; key.dv = dv;
    key.ori = check_orientation; key.kps = kps;
    const orbx_extractor::MatchKey &old = ex->mkey;
    const bool cached = old.n == key.n && old.cap == key.cap && old.match == key.match && old.nm == key.nm && old.th == key.th &&
                        old.du == key.du && old.dv == key.dv && old.ori == key.ori && old.kps == key.kps;
    if (!cached) {  // (re)build the per-pair problem descriptors; steady-state batches reuse them without any host sync
        ORBX_HIP(hipStreamSynchronize(ex->match_stream));   // the scratch below may be in use by the previous batch's matcher
#define ENS(buf, bytes) if ((r = (buf).ensure(bytes)) != ORBX_OK) return r
        ENS(ex->d_mkey1, 8 * (size_t)kTopK * cap * np);
        ENS(ex->d_mkey2, 4 * (size_t)cap * np);
        ENS(ex->d_mentries, 4 * (size_t)cap * np);
        ENS(ex->d_mgrid, 2 * ((size_t)kGridCells + 1 + 63 + cap) * np);
        ENS(ex->d_mprobs, sizeof(WindowProblem) * (size_t)np);
        ENS(ex->d_mres, sizeof(ResolveProblem) * (size_t)np);
        ENS(ex->d_mscale, sizeof(float) * ex->prm.nlevels);
#undef ENS
        std::vector<WindowProblem> P(np);
        std::vector<ResolveProblem> R(np);
        const uint8_t *desc = (const uint8_t *)ex->d_desc.p;
        const int32_t *count = (const int32_t *)ex->d_count.p;
        for (int p = 0; p < np; p++) {
            const int f = p + 1;
            WindowProblem &w = P[p];
            memset(&w, 0, sizeof(w));
            w.kps = kps + (size_t)f * cap; w.desc = desc + (size_t)f * cap * 32; w.n_ptr = count + f;
            w.q_from_kps = kps + (size_t)(f - 1) * cap; w.qdesc = desc + (size_t)(f - 1) * cap * 32; w.nq_ptr = count + (f - 1);
            w.th = th; w.du = du; w.dv = dv;
            w.scale = (const float *)ex->d_mscale.p;  // mvScaleFactors
            w.gstart = (uint16_t *)ex->d_mgrid.p + (size_t)p * (kGridCells + 64 + cap);
            w.gorder = w.gstart + kGridCells + 64;
            w.keys = (u64 *)ex->d_mkey1.p + (size_t)p * cap * kTopK; w.meta = (int32_t *)ex->d_mkey2.p + (size_t)p * cap;
            ResolveProblem &q = R[p];
            memset(&q, 0, sizeof(q));
            q.mode = 2; q.check_orientation = check_orientation; q.max_dist = (float)ORBX_TH_HIGH; q.cleared_value = -1;
            q.match = d_match + (size_t)f * cap; q.nmatches = d_nmatches + f;
            q.entries = (int32_t *)ex->d_mentries.p + (size_t)p * cap;
        }
        ORBX_HIP(hipMemcpyAsync(ex->d_mscale.p, ex->scale.data(), sizeof(float) * ex->prm.nlevels, hipMemcpyHostToDevice, ex->stream));
        ORBX_HIP(hipMemcpyAsync(ex->d_mprobs.p, P.data(), sizeof(WindowProblem) * np, hipMemcpyHostToDevice, ex->stream));
        ORBX_HIP(hipMemcpyAsync(ex->d_mres.p, R.data(), sizeof(ResolveProblem) * np, hipMemcpyHostToDevice, ex->stream));
        ORBX_HIP(hipStreamSynchronize(ex->stream));  // P/R are host temporaries
        ex->mkey = key;
    }
    GridParams g;
    g.minx = ex->bounds[0]; g.miny = ex->bounds[2];  // mnMinX.. of the extractor's camera (image rectangle without one, Frame.cc:804-807)
    g.inv_w = 64.0f / (ex->bounds[1] - ex->bounds[0]);   // Frame.cc:342-343
    g.inv_h = 48.0f / (ex->bounds[3] - ex->bounds[2]);
    // the matcher of batch i runs on its own stream beside the pyramid / FAST / quad-tree of batch i+1
    const bool side = !ex->profile && ex->side_streams;
    hipStream_t ms = side ? ex->match_stream : ex->stream;
    if (side) ORBX_HIP(hipStreamWaitEvent(ms, ex->ev_describe, 0));
    hipEvent_t e0 = ex->ev0, e1 = ex->ev1;
    if (ex->profile) (void)hipEventRecord(e0, ms);
    ORBX_LAUNCH_GRID_BUILD( dim3(np), dim3(64), 0, ms, (const WindowProblem *)ex->d_mprobs.p, g);
    ORBX_LAUNCH_WINDOW_BEST2(cap, np, ms, (const WindowProblem *)ex->d_mprobs.p, g);
    if (ex->profile) {
        (void)hipEventRecord(e1, ms); (void)hipEventSynchronize(e1);
        float t = 0; (void)hipEventElapsedTime(&t, e0, e1);
        ex->prof_ms[K_MATCH_SCAN] += t; ex->prof_n[K_MATCH_SCAN]++;
        (void)hipEventRecord(e0, ms);
    }
    { const int rr = launch_resolve<false>(np, ms, (const WindowProblem *)ex->d_mprobs.p, (const ResolveProblem *)ex->d_mres.p, g, cap, cap, 1); if (rr != ORBX_OK) return rr; }
    if (ex->profile) {
        (void)hipEventRecord(e1, ms); (void)hipEventSynchronize(e1);
        float t = 0; (void)hipEventElapsedTime(&t, e0, e1);
        ex->prof_ms[K_MATCH_RESOLVE] += t; ex->prof_n[K_MATCH_RESOLVE]++;
    }
    ORBX_HIP(hipEventRecord(ex->ev_match, ms));
    ex->match_pending = true; ex->copy_covers_match = false;
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// SearchByProjection(Frame&, vector<MapPoint*>&, th, bFarPoints, thFarPoints) (ORBmatcher.cc:39-141; Tracking::SearchLocalPoints,
// Tracking.cc:3390-3413) for every frame of the resident batch against device-resident map-point data: per (frame, map point)
// the projection (mTrackProjX/Y), predicted level and viewing cosine that Frame::isInFrustum left in the MapPoint, plus the
// map points' descriptors.  Monocular / Nleft == -1 form with all features free on entry.
// ---------------------------------------------------------------------------------------------------------
extern "C" int orbx_search_mappoints_batch_device(orbx_extractor *ex, int n_mp, const float *d_proj_x, const float *d_proj_y,
                                                  const int32_t *d_level, const float *d_view_cos, const uint8_t *d_in_view,
                                                  const uint8_t *d_mp_desc, size_t desc_frame_stride, float th, float nnratio,
                                                  int32_t *d_match, int32_t *d_nmatches) {
    RoctxRange rr("orbx:search_mappoints");
    if (!ex || n_mp < 0 || (n_mp > 0 && (!d_proj_x || !d_proj_y || !d_level || !d_view_cos || !d_mp_desc))) return ORBX_E_BAD_ARG;
    const int n = ex->last_batch;
    if (n < 1) return ORBX_E_BAD_ARG;
    if (ex->cap > kMaxResolveFeatures) return ORBX_E_TOO_LARGE;
    ORBX_HIP(hipSetDevice(ex->device));
    const int cap = ex->cap;
    int r;
    if (!d_match || !d_nmatches) {  // internal result buffers (downloaded by orbx_batch_download_async)
        if (ex->internal_match_owner == 1) { set_error("the internal match buffers hold the frame-to-frame matches of this batch: download them first, or pass result buffers"); return ORBX_E_BAD_ARG; }
        ex->internal_match_owner = 2;
        if (ex->d_match.bytes < 4 * (size_t)cap * ex->batch_cap || ex->d_nmatch.bytes < 4 * (size_t)ex->batch_cap) {
            ORBX_HIP(hipStreamSynchronize(ex->match_stream)); ORBX_HIP(hipStreamSynchronize(ex->copy_stream));
        }
        if ((r = ex->d_match.ensure(4 * (size_t)cap * ex->batch_cap)) != ORBX_OK) return r;
        if ((r = ex->d_nmatch.ensure(4 * (size_t)ex->batch_cap)) != ORBX_OK) return r;
        d_match = (int32_t *)ex->d_match.p;
        d_nmatches = (int32_t *)ex->d_nmatch.p;
    }
    const orbx_keypoint *kps = (const orbx_keypoint *)ex->match_kps();
    orbx_extractor::MpKey key;
    key.n = n; key.cap = cap; key.n_mp = n_mp; key.px = d_proj_x; key.py = d_proj_y; key.lvl = d_level; key.vc = d_view_cos; key.iv = d_in_view;
    key.desc = d_mp_desc; key.match = d_match; key.nm = d_nmatches; key.kps = kps; key.dstride = desc_frame_stride; key.th = th; key.ratio = nnratio;
    const orbx_extractor::MpKey &o = ex->mpkey;
    const bool cached = o.n == key.n && o.cap == key.cap && o.n_mp == key.n_mp && o.px == key.px && o.py == key.py && o.lvl == key.lvl &&
                        o.vc == key.vc && o.iv == key.iv && o.desc == key.desc && o.match == key.match && o.nm == key.nm && o.kps == key.kps &&
                        o.dstride == key.dstride && o.th == key.th && o.ratio == key.ratio;
    const size_t nq_all = (size_t)std::max(n_mp, 1) * n;
    if (!cached) {  // (re)build the per-frame problem descriptors; steady-state batches reuse them without a host sync
        ORBX_HIP(hipStreamSynchronize(ex->match_stream));   // the scratch below may be in use by the previous batch's matcher
#define ENS(buf, bytes) if ((r = (buf).ensure(bytes)) != ORBX_OK) return r
        ENS(ex->d_mp_qr, 4 * nq_all); ENS(ex->d_mp_qmin, 4 * nq_all); ENS(ex->d_mp_qmax, 4 * nq_all); ENS(ex->d_mp_valid, nq_all);
        ENS(ex->d_mp_keys, 8 * (size_t)kTopK * nq_all); ENS(ex->d_mp_meta, 4 * nq_all);
        ENS(ex->d_mp_grid, 2 * ((size_t)kGridCells + 1 + 63 + cap) * n);
        ENS(ex->d_mp_entries, 4 * nq_all);
        ENS(ex->d_mp_probs, sizeof(WindowProblem) * (size_t)n);
        ENS(ex->d_mp_res, sizeof(ResolveProblem) * (size_t)n);
        ENS(ex->d_mp_misc, 256 + sizeof(float) * ex->prm.nlevels);
#undef ENS
        std::vector<WindowProblem> P(n);
        std::vector<ResolveProblem> R(n);
        const uint8_t *desc = (const uint8_t *)ex->d_desc.p;
        const int32_t *count = (const int32_t *)ex->d_count.p;
        int32_t *d_nq = (int32_t *)ex->d_mp_misc.p;
        float *d_scale = (float *)((uint8_t *)ex->d_mp_misc.p + 256);
        for (int f = 0; f < n; f++) {
            WindowProblem &w = P[f];
            memset(&w, 0, sizeof(w));
            w.kps = kps + (size_t)f * cap; w.desc = desc + (size_t)f * cap * 32; w.n_ptr = count + f;
            w.qx = d_proj_x + (size_t)f * n_mp; w.qy = d_proj_y + (size_t)f * n_mp;
            w.qr = (const float *)ex->d_mp_qr.p + (size_t)f * n_mp;
            w.qmin = (const int32_t *)ex->d_mp_qmin.p + (size_t)f * n_mp; w.qmax = (const int32_t *)ex->d_mp_qmax.p + (size_t)f * n_mp;
            w.qdesc = d_mp_desc + (size_t)f * desc_frame_stride;
            w.qvalid = (const uint8_t *)ex->d_mp_valid.p + (size_t)f * n_mp;
            w.nq_ptr = d_nq;
            w.gstart = (uint16_t *)ex->d_mp_grid.p + (size_t)f * (kGridCells + 64 + cap);
            w.gorder = w.gstart + kGridCells + 64;
            w.keys = (u64 *)ex->d_mp_keys.p + (size_t)f * n_mp * kTopK; w.meta = (int32_t *)ex->d_mp_meta.p + (size_t)f * n_mp;
            ResolveProblem &q = R[f];
            memset(&q, 0, sizeof(q));
            q.mode = 1; q.nnratio = nnratio; q.max_dist = (float)ORBX_TH_HIGH; q.cleared_value = -1;
            q.match = d_match + (size_t)f * cap; q.nmatches = d_nmatches + f;
            q.entries = (int32_t *)ex->d_mp_entries.p + (size_t)f * n_mp;
        }
        const int32_t nq_host = n_mp;
        ORBX_HIP(hipMemcpyAsync(d_nq, &nq_host, 4, hipMemcpyHostToDevice, ex->stream));
        ORBX_HIP(hipMemcpyAsync(d_scale, ex->scale.data(), sizeof(float) * ex->prm.nlevels, hipMemcpyHostToDevice, ex->stream));
        ORBX_HIP(hipMemcpyAsync(ex->d_mp_probs.p, P.data(), sizeof(WindowProblem) * n, hipMemcpyHostToDevice, ex->stream));
        ORBX_HIP(hipMemcpyAsync(ex->d_mp_res.p, R.data(), sizeof(ResolveProblem) * n, hipMemcpyHostToDevice, ex->stream));
        ORBX_HIP(hipStreamSynchronize(ex->stream));  // P/R are host temporaries
        ex->mpkey = key;
    }
    GridParams g;
    g.minx = ex->bounds[0]; g.miny = ex->bounds[2];  // mnMinX.. of the extractor's camera (image rectangle without one, Frame.cc:804-807)
    g.inv_w = 64.0f / (ex->bounds[1] - ex->bounds[0]);   // Frame.cc:342-343
    g.inv_h = 48.0f / (ex->bounds[3] - ex->bounds[2]);
    const bool side = !ex->profile && ex->side_streams;
    hipStream_t ms = side ? ex->match_stream : ex->stream;
    if (side) ORBX_HIP(hipStreamWaitEvent(ms, ex->ev_describe, 0));
    if (n_mp > 0)
        hipLaunchKernelGGL(k_mappoint_windows, dim3((n_mp + 255) / 256, n), dim3(256), 0, ms, n_mp, d_level, d_view_cos, d_in_view,
                           (const float *)((const uint8_t *)ex->d_mp_misc.p + 256), ex->prm.nlevels, th, (float *)ex->d_mp_qr.p,
                           (int32_t *)ex->d_mp_qmin.p, (int32_t *)ex->d_mp_qmax.p, (uint8_t *)ex->d_mp_valid.p);
    ORBX_LAUNCH_GRID_BUILD( dim3(n), dim3(64), 0, ms, (const WindowProblem *)ex->d_mp_probs.p, g);
    if (n_mp > 0)
        ORBX_LAUNCH_WINDOW_BEST2(n_mp, n, ms, (const WindowProblem *)ex->d_mp_probs.p, g);
    { const int rr = launch_resolve<false>(n, ms, (const WindowProblem *)ex->d_mp_probs.p, (const ResolveProblem *)ex->d_mp_res.p, g, cap, 0, 4); if (rr != ORBX_OK) return rr; }
    ORBX_HIP(hipEventRecord(ex->ev_match, ms));
    ex->match_pending = true; ex->copy_covers_match = false;
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Candidate generation: Frame::UndistortKeyPoints, ComputeImageBounds, isInFrustum
// ---------------------------------------------------------------------------------------------------------
namespace {
inline CameraModel model_of(const orbx_camera *c) { return CameraModel{c->fx, c->fy, c->cx, c->cy, c->k1, c->k2, c->p1, c->p2, c->k3}; }
inline FrustumFrame frustum_frame(const orbx_camera *cam, const orbx_frame_pose *pose, const float *b, float log_sf, int nlevels, float cos_limit) {
    FrustumFrame F;
    memcpy(F.Rcw, pose->Rcw, sizeof(F.Rcw)); memcpy(F.tcw, pose->tcw, sizeof(F.tcw)); memcpy(F.Ow, pose->Ow, sizeof(F.Ow));
    F.fx = cam->fx; F.fy = cam->fy; F.cx = cam->cx; F.cy = cam->cy; F.mbf = cam->bf;
    F.minx = b[0]; F.maxx = b[1]; F.miny = b[2]; F.maxy = b[3];
    F.log_scale_factor = log_sf; F.nlevels = nlevels; F.cos_limit = cos_limit;
    return F;
}
}  // namespace

extern "C" int orbx_image_bounds(const orbx_camera *cam, int width, int height, float *bounds4) {
    if (!cam || !bounds4 || width <= 0 || height <= 0) return ORBX_E_BAD_ARG;
    image_bounds(model_of(cam), width, height, bounds4);
    return ORBX_OK;
}

extern "C" int orbx_undistort_keypoints(orbx_matcher *m, const orbx_camera *cam, const orbx_keypoint *kps, int n, orbx_keypoint *kps_un) {
    if (!m || !cam || n < 0 || (n > 0 && (!kps || !kps_un))) return ORBX_E_BAD_ARG;
    if (n == 0) return ORBX_OK;
    ORBX_HIP(hipSetDevice(m->device));
    int r = m->reserve_all(2 * Arena::pad(sizeof(orbx_keypoint) * (size_t)n) + 4096);
    if (r != ORBX_OK) return r;
    Arena &A = m->arena;
    m->begin();
    orbx_keypoint *di = A.take<orbx_keypoint>(n), *dou = A.take<orbx_keypoint>(n);
    H2D(di, kps, sizeof(orbx_keypoint) * (size_t)n);
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256, 1), dim3(256), 0, m->exec(), model_of(cam), (const orbx_keypoint *)di, (const int32_t *)nullptr, n, dou);
    D2H(kps_un, dou, sizeof(orbx_keypoint) * (size_t)n);
    SYNC_AND_DELIVER();
    return ORBX_OK;
}

extern "C" int orbx_is_in_frustum(orbx_matcher *m, const orbx_camera *cam, const orbx_frame_pose *pose, const float *bounds4, float log_scale_factor,
                                  int nlevels, float viewing_cos_limit, int n_mp, const float *pos, const float *normal, const float *min_dist,
                                  const float *max_dist, uint8_t *in_view, float *proj_x, float *proj_y, float *proj_xr, float *depth, int32_t *level,
                                  float *view_cos) {
    if (!m || !cam || !pose || !bounds4 || nlevels < 1 || n_mp < 0) return ORBX_E_BAD_ARG;
    if (n_mp > 0 && (!pos || !normal || !min_dist || !max_dist || !in_view || !proj_x || !proj_y || !proj_xr || !depth || !level || !view_cos)) return ORBX_E_BAD_ARG;
    if (n_mp == 0) return ORBX_OK;
    ORBX_HIP(hipSetDevice(m->device));
    const size_t n = (size_t)n_mp;
    int r = m->reserve_all(2 * Arena::pad(12 * n) + 8 * Arena::pad(4 * n) + Arena::pad(n) + Arena::pad(sizeof(FrustumFrame)) + 8192);
    if (r != ORBX_OK) return r;
    Arena &A = m->arena;
    m->begin();
    float *dp = A.take<float>(3 * n), *dn = A.take<float>(3 * n), *dmn = A.take<float>(n), *dmx = A.take<float>(n);
    uint8_t *div = A.take<uint8_t>(n);
    float *dx = A.take<float>(n), *dy = A.take<float>(n), *dxr = A.take<float>(n), *dd = A.take<float>(n), *dvc = A.take<float>(n);
    int32_t *dl = A.take<int32_t>(n);
    FrustumFrame *dF = A.take<FrustumFrame>(1);
    const FrustumFrame F = frustum_frame(cam, pose, bounds4, log_scale_factor, nlevels, viewing_cos_limit);
    if (m->direct_ok(57 * n)) {   // streaming kernel, small call: its lanes read the staged inputs from and write the results into the pinned mirror (one launch)
        m->stage_direct(dF, &F, sizeof(F));
        m->stage_direct(dp, pos, 12 * n); m->stage_direct(dn, normal, 12 * n); m->stage_direct(dmn, min_dist, 4 * n); m->stage_direct(dmx, max_dist, 4 * n);
        hipLaunchKernelGGL(k_in_frustum, dim3((n_mp + 255) / 256, 1), dim3(256), 0, m->exec(), (const FrustumFrame *)m->host_view(dF), n_mp,
                           (const float *)m->host_view(dp), (const float *)m->host_view(dn), (const float *)m->host_view(dmn), (const float *)m->host_view(dmx),
                           m->host_view(div), m->host_view(dx), m->host_view(dy), m->host_view(dxr), m->host_view(dd), m->host_view(dl), m->host_view(dvc));
        m->result_direct(in_view, div, n); m->result_direct(proj_x, dx, 4 * n); m->result_direct(proj_y, dy, 4 * n); m->result_direct(proj_xr, dxr, 4 * n);
        m->result_direct(depth, dd, 4 * n); m->result_direct(level, dl, 4 * n); m->result_direct(view_cos, dvc, 4 * n);
        SYNC_AND_DELIVER();
        return ORBX_OK;
    }
    H2D(dF, &F, sizeof(F));
    H2D(dp, pos, 12 * n); H2D(dn, normal, 12 * n); H2D(dmn, min_dist, 4 * n); H2D(dmx, max_dist, 4 * n);
    hipLaunchKernelGGL(k_in_frustum, dim3((n_mp + 255) / 256, 1), dim3(256), 0, m->exec(), (const FrustumFrame *)dF, n_mp, (const float *)dp,
                       (const float *)dn, (const float *)dmn, (const float *)dmx, div, dx, dy, dxr, dd, dl, dvc);
    D2H(in_view, div, n); D2H(proj_x, dx, 4 * n); D2H(proj_y, dy, 4 * n); D2H(proj_xr, dxr, 4 * n); D2H(depth, dd, 4 * n);
    D2H(level, dl, 4 * n); D2H(view_cos, dvc, 4 * n);
    SYNC_AND_DELIVER();
    return ORBX_OK;
}

extern "C" int orbx_is_in_frustum_checks(orbx_matcher *m, const orbx_fisheye_view *views, int n_views, const float *bounds4, float log_scale_factor,
                                         int nlevels, float viewing_cos_limit, int n_mp, const float *pos, const float *normal, const float *min_dist,
                                         const float *max_dist, uint8_t *in_view, float *proj_x, float *proj_y, float *depth, int32_t *level,
                                         float *view_cos) {
    if (!m || !views || n_views < 1 || n_views > 2 || !bounds4 || nlevels < 1 || n_mp < 0) return ORBX_E_BAD_ARG;
    if (n_mp > 0 && (!pos || !normal || !min_dist || !max_dist || !in_view || !proj_x || !proj_y || !depth || !level || !view_cos)) return ORBX_E_BAD_ARG;
    if (n_mp == 0) return ORBX_OK;
    ORBX_HIP(hipSetDevice(m->device));
    const size_t n = (size_t)n_mp, no = n * (size_t)n_views;
    int r = m->reserve_all(2 * Arena::pad(12 * n) + 2 * Arena::pad(4 * n) + 5 * Arena::pad(4 * no) + Arena::pad(no) + Arena::pad(sizeof(FrustumChecks)) + 8192);
    if (r != ORBX_OK) return r;
    Arena &A = m->arena;
    m->begin();
    float *dp = A.take<float>(3 * n), *dn = A.take<float>(3 * n), *dmn = A.take<float>(n), *dmx = A.take<float>(n);
    uint8_t *div = A.take<uint8_t>(no);
    float *dx = A.take<float>(no), *dy = A.take<float>(no), *dd = A.take<float>(no), *dvc = A.take<float>(no);
    int32_t *dl = A.take<int32_t>(no);
    FrustumChecks *dF = A.take<FrustumChecks>(1);
    FrustumChecks F;
    memset(&F, 0, sizeof(F));
    static_assert(sizeof(FisheyeView) == sizeof(orbx_fisheye_view), "orbx_fisheye_view layout");
    memcpy(F.view, views, sizeof(FisheyeView) * (size_t)n_views);
    F.minx = bounds4[0]; F.maxx = bounds4[1]; F.miny = bounds4[2]; F.maxy = bounds4[3];
    F.log_scale_factor = log_scale_factor; F.nlevels = nlevels; F.cos_limit = viewing_cos_limit;
    H2D(dF, &F, sizeof(F));
    H2D(dp, pos, 12 * n); H2D(dn, normal, 12 * n); H2D(dmn, min_dist, 4 * n); H2D(dmx, max_dist, 4 * n);
    hipLaunchKernelGGL(k_in_frustum_checks, dim3((n_mp + 255) / 256, n_views), dim3(256), 0, m->exec(), (const FrustumChecks *)dF, n_mp, (const float *)dp,
                       (const float *)dn, (const float *)dmn, (const float *)dmx, div, dx, dy, dd, dl, dvc);
    D2H(in_view, div, no); D2H(proj_x, dx, 4 * no); D2H(proj_y, dy, 4 * no); D2H(depth, dd, 4 * no); D2H(level, dl, 4 * no); D2H(view_cos, dvc, 4 * no);
    SYNC_AND_DELIVER();
    return ORBX_OK;
}

extern "C" int orbx_frustum_batch_device(orbx_extractor *ex, const orbx_camera *cam, const orbx_frame_pose *poses, int n_frames, const float *bounds4,
                                         float viewing_cos_limit, int n_mp, const float *d_pos, const float *d_normal, const float *d_min_dist,
                                         const float *d_max_dist, uint8_t *d_in_view, float *d_proj_x, float *d_proj_y, float *d_proj_xr,
                                         float *d_depth, int32_t *d_level, float *d_view_cos) {
    if (!ex || !cam || !poses || n_frames < 1 || n_mp < 0) return ORBX_E_BAD_ARG;
    if (n_mp > 0 && (!d_pos || !d_normal || !d_min_dist || !d_max_dist || !d_in_view || !d_proj_x || !d_proj_y || !d_proj_xr || !d_depth || !d_level || !d_view_cos))
        return ORBX_E_BAD_ARG;
    if (n_mp == 0) return ORBX_OK;
    ORBX_HIP(hipSetDevice(ex->device));
    const bool side = !ex->profile && ex->side_streams;
    hipStream_t ms = side ? ex->match_stream : ex->stream;
    if (ex->d_frustum_frames.bytes < sizeof(FrustumFrame) * (size_t)n_frames) ORBX_HIP(hipStreamSynchronize(ms));
    int r = ex->d_frustum_frames.ensure(sizeof(FrustumFrame) * (size_t)n_frames);
    if (r != ORBX_OK) return r;
    const float *b = bounds4 ? bounds4 : ex->bounds;
    if (!bounds4 && ex->width <= 0) return ORBX_E_BAD_ARG;
    const float log_sf = logf((float)(double)ex->prm.scale_factor);   // Frame::mfLogScaleFactor = log(mfScaleFactor) (Frame.cc:120)
    // the per-batch pose upload goes through a pinned ring of three slots owned by the library (no caller or pageable memory reaches the
    // HIP runtime, and the host never waits for the matcher stream): a slot is reused once the copy out of it has run
    const unsigned slot = ex->frustum_issued % 3u;
    const size_t fbytes = sizeof(FrustumFrame) * (size_t)n_frames;
    if (ex->frustum_used[slot]) ORBX_HIP(hipEventSynchronize(ex->ev_frustum[slot]));
    if (ex->h_frustum_bytes[slot] < fbytes) {
        if (ex->h_frustum[slot]) ORBX_HIP(hipHostFree(ex->h_frustum[slot]));
        ex->h_frustum[slot] = nullptr; ex->h_frustum_bytes[slot] = 0;
        ORBX_HIP(hipHostMalloc(&ex->h_frustum[slot], fbytes, hipHostMallocDefault));
        ex->h_frustum_bytes[slot] = fbytes;
    }
    if (!ex->ev_frustum[slot]) ORBX_HIP(hipEventCreateWithFlags(&ex->ev_frustum[slot], hipEventDisableTiming));
    FrustumFrame *F = (FrustumFrame *)ex->h_frustum[slot];
    for (int f = 0; f < n_frames; f++) F[f] = frustum_frame(cam, poses + f, b, log_sf, ex->prm.nlevels, viewing_cos_limit);
    ORBX_HIP(hipMemcpyAsync(ex->d_frustum_frames.p, F, fbytes, hipMemcpyHostToDevice, ms));
    ORBX_HIP(hipEventRecord(ex->ev_frustum[slot], ms));
    ex->frustum_used[slot] = true;
    ex->frustum_issued++;
    hipLaunchKernelGGL(k_in_frustum, dim3((n_mp + 255) / 256, n_frames), dim3(256), 0, ms, (const FrustumFrame *)ex->d_frustum_frames.p, n_mp, d_pos,
                       d_normal, d_min_dist, d_max_dist, d_in_view, d_proj_x, d_proj_y, d_proj_xr, d_depth, d_level, d_view_cos);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (Frame.cc:811-981) for every frame of two resident batches (left / right extractor):
// row-band Hamming, SAD refinement on the device-resident pyramids, median rejection -- nothing leaves HBM until
// the results are downloaded.  Runs on the left extractor's stream behind both extractions.
// ---------------------------------------------------------------------------------------------------------
extern "C" int orbx_stereo_batch_device(orbx_extractor *L, orbx_extractor *R, float bf, float b) {
    RoctxRange rr("orbx:stereo");
    if (!L || !R || !(b > 0.f)) return ORBX_E_BAD_ARG;
    if (L->last_batch <= 0 || L->last_batch != R->last_batch || L->width != R->width || L->height != R->height ||
        L->prm.nlevels != R->prm.nlevels || L->device != R->device)
        return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(L->device));
    const int n = L->last_batch, capL = L->cap, nl = L->prm.nlevels;
    int r;
#define ENS(buf, bytes) if ((r = (buf).ensure(bytes)) != ORBX_OK) return r
    ENS(L->d_st_bidx, 4 * (size_t)capL * L->batch_cap);
    ENS(L->d_st_bdist, 4 * (size_t)capL * L->batch_cap);
    ENS(L->d_st_ur, 4 * (size_t)capL * L->batch_cap);
    ENS(L->d_st_depth, 4 * (size_t)capL * L->batch_cap);
    ENS(L->d_st_sad, 4 * (size_t)capL * L->batch_cap);
    ENS(L->d_st_nm, 4 * (size_t)L->batch_cap);
    ENS(L->d_st_scales, sizeof(float) * 2 * nl);
    StereoBatch S;
    stereo_index_params(S, L->height, L->scale.data(), nl);
    ENS(L->d_st_rowptr, 4 * ((size_t)S.n_buckets + 1) * L->batch_cap);
    ENS(L->d_st_rowidx, 16 * (size_t)R->cap * L->batch_cap);
#undef ENS
    // From now on both extractors alternate between two pyramid slabs: these kernels run on the left extractor's MATCH stream beside the next
    // pair of extractions, and k_stereo_sad reads the pyramids of THIS pair.  (The slab written next is the one the stereo stage before this
    // one read: every extraction waits for the rig's previous stereo stage before its k_finalize -- match_pending -- and the one after it
    // follows on the same stream.)
    for (orbx_extractor *e : {L, R}) {
        // a pair extracted before the rig existed may have read its level 0 in place: the SAD stage needs the padded level in the slab
        if ((r = orbx_materialize_level0(e)) != ORBX_OK) return r;
        if (!e->pyr_double) {
            if ((r = e->d_pyr2.ensure(e->d_pyr.bytes)) != ORBX_OK) return r;
            e->pyr_double = true;   // pyr_slot stays: the current batch lies in the slab it was extracted into
        }
    }
    const bool side = !L->profile && L->side_streams;
    hipStream_t st = side ? L->match_stream : L->stream;
    ORBX_HIP(hipMemcpyAsync(L->d_st_scales.p, L->scale.data(), sizeof(float) * nl, hipMemcpyHostToDevice, st));
    ORBX_HIP(hipMemcpyAsync((float *)L->d_st_scales.p + nl, L->inv_scale.data(), sizeof(float) * nl, hipMemcpyHostToDevice, st));
    ORBX_HIP(hipStreamWaitEvent(st, L->ev_describe, 0));  // the two extractions of this batch
    ORBX_HIP(hipStreamWaitEvent(st, R->ev_describe, 0));
    if (L->stereo_copy_issued) ORBX_HIP(hipStreamWaitEvent(st, L->ev_stereo_copy[(L->stereo_copy_issued - 1) & 1], 0));  // the previous results may still be on their way to the host
    S.kl = (const orbx_keypoint *)L->d_kps.p; S.kr = (const orbx_keypoint *)R->d_kps.p;
    S.dl = (const uint8_t *)L->d_desc.p; S.dr = (const uint8_t *)R->d_desc.p;
    S.nl = (const int32_t *)L->d_count.p; S.nr = (const int32_t *)R->d_count.p;
    S.capL = capL; S.capR = R->cap;
    S.pyrL = L->pyr_cur(); S.pyrR = R->pyr_cur();
    S.pyr_frame_L = L->pyr_frame; S.pyr_frame_R = R->pyr_frame;
    S.lvL = (const LevelInfo *)L->d_lv.p; S.lvR = (const LevelInfo *)R->d_lv.p;
    S.scale = (const float *)L->d_st_scales.p; S.inv_scale = S.scale + nl;
    S.n_rows = L->height;
    S.bf = bf; S.b = b;
    S.best_idx = (int32_t *)L->d_st_bidx.p; S.best_dist = (int32_t *)L->d_st_bdist.p;
    S.u_right = (float *)L->d_st_ur.p; S.depth = (float *)L->d_st_depth.p;
    S.sad = (int32_t *)L->d_st_sad.p; S.nmatches = (int32_t *)L->d_st_nm.p;
    S.row_ptr = (const int32_t *)L->d_st_rowptr.p; S.row_ent = (const uint4 *)L->d_st_rowidx.p;
    hipLaunchKernelGGL(k_stereo_row_index, dim3(n), dim3(256), 4 * ((size_t)S.n_buckets + 1) + 1024, st, S, (int32_t *)L->d_st_rowptr.p, (uint4 *)L->d_st_rowidx.p);
    hipLaunchKernelGGL(k_stereo_rowband_batch, dim3((capL + 15) / 16, n), dim3(256), 0, st, S);
    hipLaunchKernelGGL(k_stereo_sad, dim3((capL + 15) / 16, n), dim3(256), 0, st, S);
    hipLaunchKernelGGL(k_stereo_reject, dim3(n), dim3(256), 0, st, S);
    ORBX_HIP(hipGetLastError());
    // the extractors must not overwrite their outputs / pyramids before these kernels are done
    ORBX_HIP(hipEventRecord(L->ev_match, st));
    L->match_pending = true; L->copy_covers_match = false;
    ORBX_HIP(hipEventRecord(R->ev_match, st));   // the right extractor's next k_finalize waits for it as for a matcher of its own
    R->match_pending = true; R->copy_covers_match = false;
    return ORBX_OK;
}

extern "C" int orbx_stereo_batch_download(orbx_extractor *L, int frame, float *u_right, float *depth, int *n_left, int *n_matches) {
    if (!L || frame < 0 || frame >= L->last_batch || !L->d_st_ur.p) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(L->device));
    ORBX_HIP(hipStreamWaitEvent(L->stream, L->ev_match, 0));   // the stereo stage runs on the match stream
    const size_t fb = 4 * (size_t)L->cap, o_u = 64, o_d = o_u + ((fb + 63) & ~(size_t)63);
    int r = L->d2h_staged_begin(o_d + fb);
    if (r != ORBX_OK) return r;
    if ((r = L->d2h_staged(0, (int32_t *)L->d_count.p + frame, 4)) != ORBX_OK) return r;
    if ((r = L->d2h_staged(4, (int32_t *)L->d_st_nm.p + frame, 4)) != ORBX_OK) return r;
    if ((r = L->d2h_staged(o_u, (float *)L->d_st_ur.p + (size_t)frame * L->cap, fb)) != ORBX_OK) return r;
    if ((r = L->d2h_staged(o_d, (float *)L->d_st_depth.p + (size_t)frame * L->cap, fb)) != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(L->stream));
    int32_t h[2];
    memcpy(h, L->staged(0), 8);
    if (n_left) *n_left = h[0];
    if (n_matches) *n_matches = h[1];
    const int nlv = std::min(std::max(h[0], 0), L->cap);
    if (u_right) memcpy(u_right, L->staged(o_u), 4 * (size_t)nlv);
    if (depth) memcpy(depth, L->staged(o_d), 4 * (size_t)nlv);
    return ORBX_OK;
}

// all frames of the last stereo batch at once: u_right / depth [n_frames][cap] (-1 where unmatched; entries beyond a frame's
// keypoint count are unspecified), n_matches [n_frames]; synchronous on the left extractor's stream.  The destinations may be
// pinned (copied directly would be possible) or pageable: they are always filled from the pinned staging buffer.
extern "C" int orbx_stereo_batch_download_all(orbx_extractor *L, float *u_right, float *depth, int32_t *n_matches) {
    if (!L || L->last_batch <= 0 || !L->d_st_ur.p) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(L->device));
    ORBX_HIP(hipStreamWaitEvent(L->stream, L->ev_match, 0));   // the stereo stage runs on the match stream
    const size_t n = (size_t)L->last_batch, fb = 4 * n * L->cap, o_d = (fb + 63) & ~(size_t)63, o_n = 2 * o_d;
    int r = L->d2h_staged_begin(o_n + 4 * n);
    if (r != ORBX_OK) return r;
    if (u_right && (r = L->d2h_staged(0, L->d_st_ur.p, fb)) != ORBX_OK) return r;
    if (depth && (r = L->d2h_staged(o_d, L->d_st_depth.p, fb)) != ORBX_OK) return r;
    if (n_matches && (r = L->d2h_staged(o_n, L->d_st_nm.p, 4 * n)) != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(L->stream));
    if (u_right) memcpy(u_right, L->staged(0), fb);
    if (depth) memcpy(depth, L->staged(o_d), fb);
    if (n_matches) memcpy(n_matches, L->staged(o_n), 4 * n);
    return ORBX_OK;
}

// the same into PINNED host buffers, asynchronously on the left extractor's copy stream behind the stereo kernels: the pipelined form
// (the next pair of batches is extracted while these results travel).  At most two such downloads in flight; orbx_stereo_download_wait ends the older, and
// the next orbx_stereo_batch_device waits for it on the device before it overwrites the result buffers.
extern "C" int orbx_stereo_batch_download_async(orbx_extractor *L, float *u_right, float *depth, int32_t *n_matches) {
    if (!L || L->last_batch <= 0 || !L->d_st_ur.p) return ORBX_E_BAD_ARG;
    if (L->stereo_copy_issued - L->stereo_copy_waited >= 2) { set_error("two stereo downloads already in flight: call orbx_stereo_download_wait first"); return ORBX_E_BAD_ARG; }
    for (const void *p : {(const void *)u_right, (const void *)depth, (const void *)n_matches}) {
        hipPointerAttribute_t a;
        if (p && (hipPointerGetAttributes(&a, p) != hipSuccess || a.type != hipMemoryTypeHost)) {
            (void)hipGetLastError();
            set_error("orbx_stereo_batch_download_async needs pinned host buffers (hipHostMalloc / hipHostRegister)");
            return ORBX_E_BAD_ARG;
        }
    }
    ORBX_HIP(hipSetDevice(L->device));
    for (hipEvent_t &ev : L->ev_stereo_copy) if (!ev) ORBX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const size_t n = (size_t)L->last_batch, fb = 4 * n * L->cap;
    hipStream_t cs = L->copy_stream;
    ORBX_HIP(hipStreamWaitEvent(cs, L->ev_match, 0));   // recorded behind k_stereo_reject
    if (u_right) ORBX_HIP(hipMemcpyAsync(u_right, L->d_st_ur.p, fb, hipMemcpyDeviceToHost, cs));
    if (depth) ORBX_HIP(hipMemcpyAsync(depth, L->d_st_depth.p, fb, hipMemcpyDeviceToHost, cs));
    if (n_matches) ORBX_HIP(hipMemcpyAsync(n_matches, L->d_st_nm.p, 4 * n, hipMemcpyDeviceToHost, cs));
    ORBX_HIP(hipEventRecord(L->ev_stereo_copy[L->stereo_copy_issued & 1], cs));
    L->stereo_copy_issued++;
    return ORBX_OK;
}

extern "C" int orbx_stereo_download_wait(orbx_extractor *L) {
    if (!L) return ORBX_E_BAD_ARG;
    if (L->stereo_copy_issued == L->stereo_copy_waited) return ORBX_OK;
    ORBX_HIP(hipSetDevice(L->device));
    ORBX_HIP(hipEventSynchronize(L->ev_stereo_copy[L->stereo_copy_waited & 1]));   // the OLDEST one in flight
    L->stereo_copy_waited++;
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// DBoW2 vocabulary on the device + TemplatedVocabulary::transform for all features of a frame (Frame::ComputeBoW,
// Frame.cc:738-745).  The tf-idf weighting / L1 normalisation of the BowVector (double arithmetic in std::map order)
// stays in the adapter: it needs only the word ids returned here.
// ---------------------------------------------------------------------------------------------------------
extern "C" int orbx_vocabulary_create(int device, int L, int n_nodes, const int32_t *child_ptr, const int32_t *child_idx,
                                      const uint8_t *node_desc, const int32_t *word_id, orbx_vocabulary **out) {
    if (!out || n_nodes <= 0 || !child_ptr || !child_idx || !node_desc || !word_id || L < 1) return ORBX_E_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        set_error("no usable HIP device (liborbx has no CPU fallback)");
        return ORBX_E_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    orbx_vocabulary *v = new orbx_vocabulary();
    v->device = device; v->L = L; v->n_nodes = n_nodes;
    const int nchild = child_ptr[n_nodes];
    ORBX_HIP(hipMalloc((void **)&v->child_ptr, 4 * (size_t)(n_nodes + 1)));
    ORBX_HIP(hipMalloc((void **)&v->child_idx, 4 * (size_t)std::max(nchild, 1)));
    ORBX_HIP(hipMalloc((void **)&v->word_id, 4 * (size_t)n_nodes));
    ORBX_HIP(hipMalloc((void **)&v->node_desc, 32 * (size_t)n_nodes));
    ORBX_HIP(hipMemcpy(v->child_ptr, child_ptr, 4 * (size_t)(n_nodes + 1), hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(v->child_idx, child_idx, 4 * (size_t)nchild, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(v->word_id, word_id, 4 * (size_t)n_nodes, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(v->node_desc, node_desc, 32 * (size_t)n_nodes, hipMemcpyHostToDevice));
    *out = v;
    return ORBX_OK;
}

extern "C" void orbx_vocabulary_destroy(orbx_vocabulary *v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    (void)hipFree(v->child_ptr); (void)hipFree(v->child_idx); (void)hipFree(v->word_id); (void)hipFree(v->node_desc);
    delete v;
}

extern "C" int orbx_bow_transform(orbx_matcher *m, const orbx_vocabulary *v, const uint8_t *desc, int n, int levelsup, int32_t *word_id,
                                  int32_t *node_id) {
    if (!m || !v || n < 0 || (n > 0 && (!desc || !word_id || !node_id)) || m->device != v->device) return ORBX_E_BAD_ARG;
    if (n == 0) return ORBX_OK;
    ORBX_HIP(hipSetDevice(m->device));
    int r = m->reserve_all(Arena::pad(32 * (size_t)n) + 2 * Arena::pad(4 * (size_t)n) + 4096);
    if (r != ORBX_OK) return r;
    m->begin();
    uint8_t *dd = m->arena.take<uint8_t>(32 * (size_t)n);
    int32_t *dw = m->arena.take<int32_t>(n), *dn = m->arena.take<int32_t>(n);
    H2D(dd, desc, 32 * (size_t)n);
    hipLaunchKernelGGL(k_bow_transform, dim3((n + 15) / 16), dim3(256), 0, m->exec(), v->child_ptr, v->child_idx, v->node_desc, v->word_id,
                       v->L, levelsup, dd, n, dw, dn);
    D2H(word_id, dw, 4 * (size_t)n); D2H(node_id, dn, 4 * (size_t)n);
    SYNC_AND_DELIVER();
    return ORBX_OK;
}

// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:329-403) for a batch of map points
extern "C" int orbx_distinctive_descriptors(orbx_matcher *m, const uint8_t *desc, const int32_t *set_ptr, int n_sets, int32_t *best_idx) {
    if (!m || n_sets < 0 || (n_sets > 0 && (!set_ptr || !best_idx))) return ORBX_E_BAD_ARG;
    if (n_sets == 0) return ORBX_OK;
    const int total = set_ptr[n_sets];
    if (total > 0 && !desc) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(m->device));
    int r = m->reserve_all(Arena::pad(32 * (size_t)total + 32) + Arena::pad(4 * (size_t)(n_sets + 1)) + Arena::pad(4 * (size_t)n_sets) + 4096);
    if (r != ORBX_OK) return r;
    m->begin();
    uint8_t *dd = m->arena.take<uint8_t>(32 * (size_t)total + 32);
    int32_t *dp = m->arena.take<int32_t>(n_sets + 1), *db = m->arena.take<int32_t>(n_sets);
    if (total > 0) H2D(dd, desc, 32 * (size_t)total);
    H2D(dp, set_ptr, 4 * (size_t)(n_sets + 1));
    hipLaunchKernelGGL(k_distinctive, dim3((n_sets + 3) / 4), dim3(256), 0, m->exec(), dd, dp, n_sets, db);
    D2H(best_idx, db, 4 * (size_t)n_sets);
    SYNC_AND_DELIVER();
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Candidate loop of ORBmatcher::Fuse x2 (ORBmatcher.cc:1246-1306, 1405-1433): queries do not interact, so the device
// returns the best feature per projected map point; the Replace / AddObservation logic stays in the adapter.
// ---------------------------------------------------------------------------------------------------------
extern "C" int orbx_fuse_search(orbx_matcher *m, const orbx_frame_desc *kf, const float *inv_level_sigma2, int n_q, const float *q_u,
                                const float *q_v, const float *q_ur, const float *q_r, const int32_t *q_level, const uint8_t *q_desc,
                                int strict_fp, int32_t *best_idx, int32_t *best_dist) {
    if (!m || !kf || n_q < 0 || (n_q > 0 && (!q_u || !q_v || !q_r || !q_level || !q_desc || !best_idx || !best_dist))) return ORBX_E_BAD_ARG;
    for (int i = 0; i < n_q; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    const int n = kf->n;
    if (n == 0 || n_q == 0) return ORBX_OK;
    if (n > 65535) return ORBX_E_TOO_LARGE;
    ORBX_HIP(hipSetDevice(m->device));
    const int nl = kf->nlevels;
    size_t need = Arena::pad(28 * (size_t)n) + Arena::pad(32 * (size_t)n) + Arena::pad(4 * (size_t)n) + Arena::pad(4 * (size_t)nl) +
                  Arena::pad(4 * (size_t)n_q) * 6 + Arena::pad(32 * (size_t)n_q) + Arena::pad(8 * (size_t)n_q * kTopK) + Arena::pad(4 * (size_t)n_q) +
                  Arena::pad(sizeof(WindowProblem)) + Arena::pad(2 * (kGridCells + 1)) + Arena::pad(2 * (size_t)n) + 16 * 256 + 4096;
    int r = m->reserve_all(need);
    if (r != ORBX_OK) return r;
    Arena &A = m->arena;
    m->begin();
    WindowProblem P;
    memset(&P, 0, sizeof(P));
    orbx_keypoint *dk = A.take<orbx_keypoint>(n);
    uint8_t *dd = A.take<uint8_t>(32 * (size_t)n);
    H2D(dk, kf->keypoints_un, 28 * (size_t)n); H2D(dd, kf->descriptors, 32 * (size_t)n);
    P.kps = dk; P.desc = dd;
    int32_t *dcnt = A.take<int32_t>(4);
    const int32_t cnts[2] = {n, n_q};
    H2D(dcnt, cnts, 8);
    P.n_ptr = dcnt; P.nq_ptr = dcnt + 1;
    if (kf->u_right) { float *p = A.take<float>(n); H2D(p, kf->u_right, 4 * (size_t)n); P.u_right = p; }
    if (inv_level_sigma2) { float *p = A.take<float>(nl); H2D(p, inv_level_sigma2, 4 * (size_t)nl); P.inv_sigma2 = p; }
    P.chi2_fma = strict_fp ? 0 : 1;
    float *f3[3]; const float *h3[3] = {q_u, q_v, q_r};
    for (int k = 0; k < 3; k++) { f3[k] = A.take<float>(n_q); H2D(f3[k], h3[k], 4 * (size_t)n_q); }
    P.qx = f3[0]; P.qy = f3[1]; P.qr = f3[2];
    std::vector<int32_t> qmin(n_q), qmax(n_q);
    for (int i = 0; i < n_q; i++) { qmin[i] = q_level[i] - 1; qmax[i] = q_level[i]; }  // kpLevel<nPredictedLevel-1 || kpLevel>nPredictedLevel
    int32_t *dmin = A.take<int32_t>(n_q), *dmax = A.take<int32_t>(n_q);
    H2D(dmin, qmin.data(), 4 * (size_t)n_q); H2D(dmax, qmax.data(), 4 * (size_t)n_q);
    P.qmin = dmin; P.qmax = dmax;
    if (q_ur && kf->u_right) { float *p = A.take<float>(n_q); H2D(p, q_ur, 4 * (size_t)n_q); P.qxr = p; }
    { uint8_t *p = A.take<uint8_t>(32 * (size_t)n_q); H2D(p, q_desc, 32 * (size_t)n_q); P.qdesc = p; }
    WindowProblem *dP = A.take<WindowProblem>(1);   // directly behind the inputs: the call's uploads are one run of the arena (one DMA)
    P.keys = A.take<u64>((size_t)n_q * kTopK); P.meta = A.take<int32_t>(n_q);
    P.gstart = A.take<uint16_t>(kGridCells + 1); P.gorder = A.take<uint16_t>(n);
    const bool brute = m->brute_windows && (size_t)n_q * (size_t)n <= kBruteMaxPairs;
    if (brute) { P.gstart = nullptr; P.gorder = nullptr; }
    H2D(dP, &P, sizeof(P));
    GridParams g;
    g.minx = kf->min_x; g.miny = kf->min_y;
    g.inv_w = 64.0f / (kf->max_x - kf->min_x);
    g.inv_h = 48.0f / (kf->max_y - kf->min_y);
    if (brute) {
        hipLaunchKernelGGL(k_window_brute, dim3((n_q + 3) / 4), dim3(256), 0, m->exec(), dP, g);
    } else {
        ORBX_LAUNCH_GRID_BUILD( dim3(1), dim3(64), 0, m->exec(), dP, g);
        ORBX_LAUNCH_WINDOW_BEST2(n_q, 1, m->exec(), dP, g);
    }
    std::vector<u64> keys((size_t)n_q * kTopK);
    D2H(keys.data(), P.keys, 8 * (size_t)n_q * kTopK);
    SYNC_AND_DELIVER();
    for (int i = 0; i < n_q; i++) {
        const u64 k = keys[(size_t)i * kTopK];
        if (k != kNoKey) { best_idx[i] = (int32_t)(k & 0xffff); best_dist[i] = (int32_t)(k >> 32); }
    }
    return ORBX_OK;
}
