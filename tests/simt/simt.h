// simt.h -- a minimal SIMT emulator for running liborbx's wave64 HIP device code on the CPU (test infrastructure only).
//
// Every lane of a workgroup is a cooperative fiber (ucontext) of ONE OS thread; a fiber runs until it reaches a wave- or workgroup-
// level collective (__ballot, __shfl*, DPP, readlane, __syncthreads, wave barriers ...), deposits its operand and yields; the
// last participating lane completes the collective and everybody proceeds.  That reproduces the semantics the kernels rely on:
//   * lanes of a wave see each other's LDS / global writes at every collective or wave barrier (the hardware's lock step is
//     stronger; code that needs MORE than this -- communication through memory without any ordering point -- would be a bug
//     the emulator exposes as a wrong result),
//   * lanes that have returned do not take part (their ballot bit is 0),
//   * a collective reached by only a part of the lanes that are still alive (divergent control flow around a collective) cannot
//     be emulated: the scheduler detects the stall and aborts with a message.
// Only what octree.hip.h / octree_par.hip.h use is provided.
#pragma once
#include <ucontext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
// static LDS arrays of the kernels land in one section of the emulated library, so that the launcher can fill them with garbage before
// every workgroup (launch.cc): on the hardware they hold whatever the previous workgroup of that CU left there
#define __shared__ static __attribute__((section("simt_lds")))
#define __ATOMIC_SEQ_CST_SIMT 5

namespace simt {

struct Dim3 { unsigned x = 1, y = 1, z = 1; };

struct Fiber {
    ucontext_t ctx;
    bool done = false;
    int tid = 0;
};
constexpr size_t kFiberStack = 192 * 1024;
inline char *fiber_stack(int t) {   // stacks are reused by every block (not zeroed)
    static char *pool = (char *)aligned_alloc(4096, 1024 * kFiberStack);
    return pool + (size_t)t * kFiberStack;
}

struct Rendezvous {   // one per wave (64 lanes) and one per workgroup
    int n = 64;           // slots in use (64 for a wave, the block size for the workgroup)
    int waiting = 0;
    unsigned long long generation = 0;
    uint64_t slot[1024];
    uint8_t present[1024];
    uint64_t result[2][1024];
    uint8_t rpresent[2][1024];
};

struct Block {
    int nthreads = 0;
    std::vector<Fiber> fibers;
    ucontext_t sched;
    int cur = -1;
    int alive_block = 0;
    int alive_wave[16] = {0};
    Rendezvous wave_rv[16], block_rv;
    std::vector<std::pair<uint32_t *, uint32_t>> deferred[16];   // per wave: additions that take effect at its next collective
    unsigned long long progress = 0;
    Dim3 block_idx, grid_dim;
    std::function<void()> body;
};

inline Block *&blk() { static Block *b = nullptr; return b; }
inline int cur_tid() { return blk()->cur; }

inline void yield_to_scheduler() {
    Block *b = blk();
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}

inline void apply_deferred(Rendezvous &rv);
// complete a rendezvous if every alive participant has arrived
inline void try_complete(Rendezvous &rv, int alive) {
    if (rv.waiting > 0 && rv.waiting == alive) {
        apply_deferred(rv);
        const int g = (int)(rv.generation & 1);
        memcpy(rv.result[g], rv.slot, sizeof(uint64_t) * (size_t)rv.n);
        memcpy(rv.rpresent[g], rv.present, (size_t)rv.n);
        memset(rv.present, 0, (size_t)rv.n);
        rv.waiting = 0;
        rv.generation++;
        blk()->progress++;
    }
}

inline void apply_deferred(Rendezvous &rv) {
    Block *b = blk();
    for (int w = 0; w < 16; w++)
        if (&rv == &b->wave_rv[w] || &rv == &b->block_rv) {
            for (auto &d : b->deferred[w]) *d.first += d.second;
            b->deferred[w].clear();
        }
}
// An LDS update that the hardware's lock step orders AFTER every lane's earlier read of the same instruction sequence (all lanes
// read a running offset, then all lanes bump it): the emulator's lanes run one after the other, so the update takes effect when the
// wave reaches its next collective.  Used through a rewrite in build.py for the one place that needs it.
inline void defer_add(uint32_t *p, uint32_t v) {
    Block *b = blk();
    b->deferred[b->cur >> 6].push_back({p, v});
}
// deposit `v` for lane index `idx` of the group; returns the generation slot holding everybody's values
inline int rendezvous(Rendezvous &rv, int idx, uint64_t v, int *alive_counter) {
    rv.slot[idx] = v;
    rv.present[idx] = 1;
    rv.waiting++;
    const unsigned long long g = rv.generation;
    try_complete(rv, *alive_counter);
    while (rv.generation == g) yield_to_scheduler();
    return (int)(g & 1);
}

struct WaveView {   // all lanes' operands of one wave collective
    const uint64_t *v;
    const uint8_t *p;
};
inline WaveView wave_collect(uint64_t v) {
    Block *b = blk();
    const int tid = b->cur, w = tid >> 6, lane = tid & 63;
    Rendezvous &rv = b->wave_rv[w];
    const int g = rendezvous(rv, lane, v, &b->alive_wave[w]);
    return WaveView{rv.result[g], rv.rpresent[g]};
}
inline void block_barrier() {
    Block *b = blk();
    rendezvous(b->block_rv, b->cur, 0, &b->alive_block);
}

inline void fiber_entry() {
    Block *b = blk();
    b->body();
    const int tid = b->cur;
    b->fibers[tid].done = true;
    b->alive_block--;
    b->alive_wave[tid >> 6]--;
    b->progress++;
    try_complete(b->wave_rv[tid >> 6], b->alive_wave[tid >> 6]);
    try_complete(b->block_rv, b->alive_block);
    swapcontext(&b->fibers[tid].ctx, &b->sched);
}

// run one workgroup of `nthreads` lanes
inline void run_block(Dim3 grid, Dim3 bidx, int nthreads, const std::function<void()> &body) {
    Block *b = new Block();
    blk() = b;
    b->nthreads = nthreads; b->grid_dim = grid; b->block_idx = bidx; b->body = body;
    b->fibers.resize(nthreads);
    b->alive_block = nthreads;
    for (int t = 0; t < nthreads; t++) b->alive_wave[t >> 6]++;
    for (int w = 0; w < 16; w++) memset(b->wave_rv[w].present, 0, sizeof(b->wave_rv[w].present));
    memset(b->block_rv.present, 0, sizeof(b->block_rv.present));
    b->block_rv.n = nthreads;
    for (int t = 0; t < nthreads; t++) {
        Fiber &f = b->fibers[t];
        f.tid = t;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = fiber_stack(t);
        f.ctx.uc_stack.ss_size = kFiberStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    // SIMT_SHUFFLE=<seed>: the waves of the workgroup are resumed in a random order that changes every round (lanes of a wave stay in
    // lane order).  A kernel whose result depends on that order has a data race between its waves (a missing barrier).
    static const char *shuffle_env = getenv("SIMT_SHUFFLE");
    static unsigned long long rng = shuffle_env ? strtoull(shuffle_env, nullptr, 10) * 2654435761ull + 88172645463325252ull : 0;
    const int nwaves = (nthreads + 63) / 64;
    while (b->alive_block > 0) {
        const unsigned long long before = b->progress;
        int order[16];
        for (int i = 0; i < 16; i++) order[i] = i;
        if (shuffle_env)
            for (int i = nwaves - 1; i > 0; i--) {
                rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
                const int j = (int)(rng % (unsigned)(i + 1));
                std::swap(order[i], order[j]);
            }
        // ... and each wave takes part in a round with probability 1/2 (SIMT_SHUFFLE only), so that waves drift apart by any number of
        // collectives, as they do on the hardware -- bounded only by the workgroup barriers
        unsigned skip = 0;
        if (shuffle_env) {
            rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
            skip = (unsigned)(rng >> 20) & ((1u << nwaves) - 1u);
            if (skip == (1u << nwaves) - 1u) skip = 0;
        }
        // SIMT_LANE_ORDER=reverse|<seed>: the lanes of a wave run their stretch between two collectives last lane first / in a random order
        // that changes every round.  The hardware executes them in lock step; what a wave's lanes do to the SAME location within one
        // instruction (atomics handing out slots) has no defined order among the lanes, and a result that changes with this setting
        // depends on one (or on a lower lane's plain store being visible to a higher lane within the stretch -- an emulator artefact).
        static const char *lane_env = getenv("SIMT_LANE_ORDER");
        static unsigned long long lrng = lane_env ? strtoull(lane_env, nullptr, 10) * 0x9E3779B97F4A7C15ull + 12345 : 0;
        for (int wi = 0; wi < nwaves; wi++) {
            if ((skip >> order[wi]) & 1u) continue;
            const int t0 = order[wi] * 64, cnt = std::min(nthreads, t0 + 64) - t0;
            int lanes[64];
            for (int i = 0; i < cnt; i++) lanes[i] = lane_env && !strcmp(lane_env, "reverse") ? cnt - 1 - i : i;
            if (lane_env && strcmp(lane_env, "reverse"))
                for (int i = cnt - 1; i > 0; i--) { lrng ^= lrng << 13; lrng ^= lrng >> 7; lrng ^= lrng << 17; std::swap(lanes[i], lanes[lrng % (unsigned)(i + 1)]); }
            for (int i = 0; i < cnt; i++) {
                const int t = t0 + lanes[i];
                if (b->fibers[t].done) continue;
                b->cur = t;
                swapcontext(&b->sched, &b->fibers[t].ctx);
            }
        }
        if (b->progress == before && b->alive_block > 0 && skip == 0) {
            fprintf(stderr, "simt: stall -- a collective was reached by only part of the live lanes (divergent control flow around it)\n");
            abort();
        }
    }
    delete b;
    blk() = nullptr;
}

}  // namespace simt

// ---- the HIP surface the kernels use -----------------------------------------------------------------------------------------------
struct simt_tid { operator unsigned() const { return 0; } };
struct simt_threadidx { unsigned get() const { return (unsigned)simt::cur_tid(); } };
struct simt_idx3 {
    struct X { operator unsigned() const { return (unsigned)simt::cur_tid(); } } x;
};
static simt_idx3 threadIdx;
struct simt_bidx3 {
    struct X { operator unsigned() const { return simt::blk()->block_idx.x; } } x;
    struct Y { operator unsigned() const { return simt::blk()->block_idx.y; } } y;
    struct Z { operator unsigned() const { return simt::blk()->block_idx.z; } } z;
};
static simt_bidx3 blockIdx;
struct simt_gdim3 {
    struct X { operator unsigned() const { return simt::blk()->grid_dim.x; } } x;
    struct Y { operator unsigned() const { return simt::blk()->grid_dim.y; } } y;
    struct Z { operator unsigned() const { return simt::blk()->grid_dim.z; } } z;
};
static simt_gdim3 gridDim;

inline void __syncthreads() { simt::block_barrier(); }
inline void __builtin_amdgcn_wave_barrier() { simt::wave_collect(0); }
#define __builtin_amdgcn_fence(...) ((void)0)
inline void __builtin_amdgcn_s_barrier() { simt::block_barrier(); }

inline unsigned long long __ballot(bool pred) {
    const simt::WaveView w = simt::wave_collect(pred ? 1 : 0);
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++) if (w.p[l] && w.v[l]) m |= 1ull << l;
    return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long)v); }
inline int __mul24(int a, int b) { return (int)((int64_t)((a << 8) >> 8) * (int64_t)((b << 8) >> 8)); }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline long long wall_clock64() { return 0; }

template <typename T> inline uint64_t simt_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&b, &v, sizeof(T)); return b; }
template <typename T> inline T simt_from(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

// A shuffle / readlane whose source lane has left the kernel: the hardware returns 0 (ds_bpermute) or whatever the dead lane's
// register holds (v_readlane, DPP-lowered shuffles) -- nothing a result may depend on.  The emulator returns 0 and counts the reads;
// SIMT_STRICT_LANES=1 aborts at the first one.
namespace simt {
inline uint64_t lane_value(const WaveView &w, int src, const char *what) {
    if (w.p[src & 63]) return w.v[src & 63];
    static const bool strict = getenv("SIMT_STRICT_LANES") != nullptr;
    if (strict) { fprintf(stderr, "simt: %s reads lane %d, which has left the kernel\n", what, src & 63); abort(); }
    return 0;
}
}
template <typename T> inline T __shfl(T v, int src) {
    const simt::WaveView w = simt::wave_collect(simt_bits(v));
    return simt_from<T>(simt::lane_value(w, src, "__shfl"));
}
template <typename T> inline T __shfl_up(T v, unsigned d) {
    const int lane = simt::cur_tid() & 63;
    const simt::WaveView w = simt::wave_collect(simt_bits(v));
    return lane >= (int)d ? simt_from<T>(simt::lane_value(w, lane - (int)d, "__shfl_up")) : v;
}
template <typename T> inline T __shfl_xor(T v, int m) {
    const int lane = simt::cur_tid() & 63;
    const simt::WaveView w = simt::wave_collect(simt_bits(v));
    return simt_from<T>(simt::lane_value(w, lane ^ m, "__shfl_xor"));
}
// v_mbcnt_lo / v_mbcnt_hi: bits of the mask below this lane
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add) {
    const int lane = simt::cur_tid() & 63;
    return add + (unsigned)__builtin_popcount(mask & (lane >= 32 ? 0xffffffffu : ((1u << lane) - 1u)));
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add) {
    const int lane = simt::cur_tid() & 63;
    return add + (lane > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (lane - 32)) - 1u)) : 0u);
}
inline int __builtin_amdgcn_readlane(int v, int l) {
    const simt::WaveView w = simt::wave_collect(simt_bits(v));
    return simt_from<int>(simt::lane_value(w, l, "v_readlane"));
}
// v_readfirstlane_b32: the kernels use it on values that ARE wave-uniform (to tell the compiler so), also under partial EXEC masks
// (inside helpers some lanes have left) -- a collective would not be reached by every live lane, and the value is the lane's own anyway
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
// v_mov_b32_dpp as used here: row_shr:n (0x111..0x11f), row_ror:n (0x121..0x12f), wave_shr:1 (0x138), wave_shl:1 (0x130), row_bcast15 (0x142), row_bcast31 (0x143);
// bank_mask 0xf, bound_ctrl off: a lane without a valid source, or in a row the row mask disables, keeps `old`
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)bank_mask; (void)bound_ctrl;
    const int lane = simt::cur_tid() & 63, row = lane >> 4, li = lane & 15;
    const simt::WaveView w = simt::wave_collect(simt_bits(src));
    if (!((row_mask >> row) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; if (li >= n) from = lane - n; }
    else if (ctrl >= 0x121 && ctrl <= 0x12f) { const int n = ctrl - 0x120; from = 16 * row + ((li - n) & 15); }   // row_ror:n
    else if (ctrl == 0x138) { if (lane >= 1) from = lane - 1; }
    else if (ctrl == 0x130) { if (lane <= 62) from = lane + 1; }   // wave_shl:1
    else if (ctrl == 0x142) { if (row >= 1) from = 16 * row - 1; }
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }
    else { fprintf(stderr, "simt: DPP control 0x%x not emulated\n", ctrl); abort(); }
    if (from < 0 || !w.p[from]) return old;
    return simt_from<int>(w.v[from]);
}

template <typename T> inline T atomicAdd(T *p, T v) { const T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicOr(T *p, T v) { const T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicExch(T *p, T v) { const T o = *p; *p = v; return o; }
template <typename T> inline T atomicMax(T *p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T *p, T v) { const T o = *p; if (v < o) *p = v; return o; }

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

using std::min;
using std::max;
inline int min(int a, unsigned b) { return (int)std::min<long long>(a, b); }
inline int min(unsigned a, int b) { return (int)std::min<long long>(a, b); }
inline int max(int a, unsigned b) { return (int)std::max<long long>(a, b); }
inline int max(unsigned a, int b) { return (int)std::max<long long>(a, b); }

// ---- more of the device surface (extractor / matcher kernels) ------------------------------------------------------------------
struct uint3 { uint32_t x, y, z; };
struct char4 { signed char x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };
inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

inline int __builtin_amdgcn_s_waitcnt_dummy() { return 0; }
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

// V_PERM_B32: byte i of the result = byte sel[i] of {a (bytes 4..7), b (bytes 0..3)}; 0x0c -> 0x00, >= 0x0d -> 0xff
inline uint32_t __builtin_amdgcn_perm(uint32_t a, uint32_t b, uint32_t sel) {
    const uint64_t src = ((uint64_t)a << 32) | b;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t s = (sel >> (8 * i)) & 0xff;
        uint32_t v;
        if (s <= 7) v = (uint32_t)(src >> (8 * s)) & 0xff;
        else if (s == 0x0c) v = 0;
        else if (s >= 0x0d) v = 0xff;
        else { fprintf(stderr, "simt: v_perm_b32 selector 0x%x not emulated\n", s); abort(); }
        r |= v << (8 * i);
    }
    return r;
}
inline uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t shift) {
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (shift & 3)));
}
inline uint32_t __builtin_amdgcn_sad_u8(uint32_t a, uint32_t b, uint32_t c) {
    for (int i = 0; i < 4; i++) { const int x = (a >> (8 * i)) & 0xff, y = (b >> (8 * i)) & 0xff; c += (uint32_t)(x > y ? x - y : y - x); }
    return c;
}
inline uint32_t __builtin_amdgcn_udot4(uint32_t a, uint32_t b, uint32_t c, bool) {
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xff) * ((b >> (8 * i)) & 0xff);
    return c;
}
typedef unsigned short simt_u16x2 __attribute__((ext_vector_type(2)));
inline uint32_t __builtin_amdgcn_udot2(simt_u16x2 a, simt_u16x2 b, uint32_t c, bool) { return (uint32_t)a.x * b.x + (uint32_t)a.y * b.y + c; }
inline unsigned short __builtin_amdgcn_ashr_pk_u8_i32(int a, int b, int s) {
    auto sat = [](int v) { return (unsigned)(v < 0 ? 0 : v > 255 ? 255 : v); };
    return (unsigned short)(sat(a >> (s & 31)) | (sat(b >> (s & 31)) << 8));
}
inline uint32_t simt_pk_min_u16(uint32_t a, uint32_t b) {
    const uint32_t lo = std::min(a & 0xffffu, b & 0xffffu), hi = std::min(a >> 16, b >> 16);
    return lo | (hi << 16);
}
// v_pk_minimum3_f16 / v_pk_maximum3_f16 on the operands the kernels give them: integers 0 .. 255 per half (non-negative f16 bit patterns order like
// the integers, and minimum / maximum return an input unchanged)
inline uint32_t simt_pk_min3_u16(uint32_t a, uint32_t b, uint32_t c) { return simt_pk_min_u16(a, simt_pk_min_u16(b, c)); }
inline uint32_t simt_pk_max3_u16(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t lo = std::max(a & 0xffffu, std::max(b & 0xffffu, c & 0xffffu)), hi = std::max(a >> 16, std::max(b >> 16, c >> 16));
    return lo | (hi << 16);
}
// raw buffer descriptor: base + byte range; a dword load that is not fully inside the range returns 0 without touching memory
struct simt_rsrc { const uint8_t *base; uint32_t num_records; };
inline simt_rsrc __builtin_amdgcn_make_buffer_rsrc(void *p, short, int num_records, int) { return simt_rsrc{(const uint8_t *)p, (uint32_t)num_records}; }
inline uint32_t __builtin_amdgcn_raw_buffer_load_b32(simt_rsrc r, int voffset, int soffset, int) {
    const uint32_t off = (uint32_t)voffset + (uint32_t)soffset;
    if ((uint64_t)off + 4 > r.num_records) return 0;
    uint32_t v; memcpy(&v, r.base + off, 4); return v;
}
typedef uint32_t simt_u32x3 __attribute__((ext_vector_type(3)));
inline simt_u32x3 __builtin_amdgcn_raw_buffer_load_b96(simt_rsrc r, int voffset, int soffset, int) {   // range check per dword
    simt_u32x3 v;
    v.x = __builtin_amdgcn_raw_buffer_load_b32(r, voffset, soffset, 0);
    v.y = __builtin_amdgcn_raw_buffer_load_b32(r, voffset + 4, soffset, 0);
    v.z = __builtin_amdgcn_raw_buffer_load_b32(r, voffset + 8, soffset, 0);
    return v;
}
#undef __builtin_amdgcn_readfirstlane_collective
inline int __syncthreads_or(int pred) {
    simt::Block *b = simt::blk();
    const int g = simt::rendezvous(b->block_rv, b->cur, pred ? 1 : 0, &b->alive_block);
    int r = 0;
    for (int t = 0; t < b->nthreads; t++) if (b->block_rv.rpresent[g][t] && b->block_rv.result[g][t]) r = 1;
    return r;
}
template <typename T> inline T __shfl_down(T v, unsigned d) {
    const int lane = simt::cur_tid() & 63;
    const simt::WaveView w = simt::wave_collect(simt_bits(v));
    return lane + (int)d < 64 ? simt_from<T>(simt::lane_value(w, lane + (int)d, "__shfl_down")) : v;
}

inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline double __fma_rn(double a, double b, double c) { return fma(a, b, c); }
inline double __dmul_rn(double a, double b) { return a * b; }
inline double __dadd_rn(double a, double b) { return a + b; }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float2int_rn(float f) { return (int)lrintf(f); }
inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

// the 32-bit LDS address a kernel computes by hand (k_describe) back to a host pointer: statics of this library share the upper half
namespace simt { uint8_t *dyn_lds(); uintptr_t bss_anchor(); inline const uint8_t *lds_ptr(uint32_t a) {
    return (const uint8_t *)((bss_anchor() & ~(uintptr_t)0xffffffffull) | a); } }
inline void __threadfence_block() {}
inline void __threadfence() {}
inline void __builtin_amdgcn_s_sleep(int) {}
inline uint32_t __builtin_amdgcn_s_getreg(int) { return blockIdx.x & 7u; }   // XCC_ID of the probe: workgroups go to the XCDs round-robin
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
