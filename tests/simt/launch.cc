// launch.cc -- grid loop + dynamic LDS of the SIMT emulator (one definition per emulated library)
#include "hip/hip_runtime.h"
#include <algorithm>
#include <vector>
#include <cstring>
extern "C" char __start_simt_lds[], __stop_simt_lds[];   // the section of the kernels' static LDS arrays (simt.h)
namespace simt {
static __attribute__((section("simt_lds"), used)) char g_lds_section_anchor[16];   // the section exists even without a static array
static uint8_t *g_lds = nullptr;
static uint8_t g_anchor[16];
uintptr_t bss_anchor() { return (uintptr_t)g_anchor; }   // static LDS arrays of the kernels live in this library's .bss too
uint8_t *dyn_lds() {
    if (!g_lds) g_lds = (uint8_t *)aligned_alloc(4096, 256 * 1024);
    return g_lds;
}
// Workgroups of a launch run one after another.  SIMT_BLOCK_ORDER=reverse runs them last to first, SIMT_BLOCK_ORDER=<seed> in a
// different random order for every launch: a result that changes with it depends on the order in which workgroups reach a
// global atomic (list appends, counters) -- which the hardware does not define.
// (No kernel of the product may depend on it: HIP promises nothing about dispatch order.  Round 3's k_pyr_resize_chain_ordered did and was removed.)
std::vector<uint32_t> block_order(dim3 grid, const char *name) {
    static const char *order_env = getenv("SIMT_BLOCK_ORDER");
    const char *order = order_env;
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    std::vector<uint32_t> perm(nblocks);
    for (size_t i = 0; i < nblocks; i++) perm[i] = (uint32_t)i;
    if (order && !strcmp(order, "reverse")) std::reverse(perm.begin(), perm.end());
    else if (order) {
        static unsigned long long st = strtoull(order, nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
        for (size_t i = nblocks; i > 1; i--) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; std::swap(perm[i - 1], perm[st % i]); }
    }
    return perm;
}
// SIMT_KERNEL_SPLIT=<n> (with SIMT_STREAM_FUZZ): a launch is queued as up to n pieces (runs of workgroups), so that the scheduler
// interleaves the workgroups of kernels that sit on different streams -- kernels overlapping in time, at workgroup granularity.
int kernel_split() {
    static const int n = [] { const char *e = getenv("SIMT_KERNEL_SPLIT"); return e ? atoi(e) : 0; }();
    return n;
}
void launch_blocks(const char *name, dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body, const uint32_t *ids, size_t n,
                   bool announce) {
    static const bool trace = getenv("SIMT_TRACE") != nullptr;
    static const bool trace_pieces = trace && atoi(getenv("SIMT_TRACE")) >= 2;   // SIMT_TRACE=2: one line per piece of a split launch
    if (trace && announce) fprintf(stderr, "simt: %s grid (%u, %u, %u) block %u lds %zu\n", name, grid.x, grid.y, grid.z, block.x, lds_bytes);
    else if (trace_pieces) fprintf(stderr, "simt:   piece of %s from workgroup #%u (%zu)\n", name, n ? ids[0] : 0u, n);
    if (lds_bytes > 200 * 1024) { fprintf(stderr, "simt: %zu bytes of dynamic LDS\n", lds_bytes); abort(); }
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads > 1024) { fprintf(stderr, "simt: %d threads per block\n", nthreads); abort(); }
    Dim3 g; g.x = grid.x; g.y = grid.y; g.z = grid.z;
    for (size_t i = 0; i < n; i++) {
        const size_t id = ids[i];
        Dim3 b; b.x = (unsigned)(id % grid.x); b.y = (unsigned)(id / grid.x % grid.y); b.z = (unsigned)(id / ((size_t)grid.x * grid.y));
        // LDS is not zero on entry: a fixed pattern, or (SIMT_LDS_RANDOM=<seed>) different garbage for every block -- a result
        // that changes with it depends on LDS the kernel never wrote
        static const char *rnd = getenv("SIMT_LDS_RANDOM");
        if (rnd) {
            static unsigned long long st = strtoull(rnd, nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
            uint64_t *p = (uint64_t *)dyn_lds();
            for (size_t j = 0; j < (lds_bytes + 7) / 8; j++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; p[j] = st; }
            for (char *q = __start_simt_lds; q < __stop_simt_lds; q++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; *q = (char)st; }
        } else {
            memset(dyn_lds(), 0xcd, lds_bytes);
            memset(__start_simt_lds, 0xcd, (size_t)(__stop_simt_lds - __start_simt_lds));
        }
        run_block(g, b, nthreads, body);
    }
}
void launch(const char *name, dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body) {
    const std::vector<uint32_t> ids = block_order(grid, name);
    launch_blocks(name, grid, block, lds_bytes, body, ids.data(), ids.size(), true);
}
}  // namespace simt

// ---- stream / event runtime ------------------------------------------------------------------------------------------------------
#include <deque>
#include <map>
struct simt_event { unsigned long long recorded = 0, done = 0; };   // tickets: number of records issued / executed
struct simt_stream {
    struct Op { std::function<void()> fn; simt_event *wait = nullptr; unsigned long long wait_ticket = 0; simt_event *rec = nullptr; unsigned long long rec_ticket = 0; };
    std::deque<Op> q;
};
namespace simt {
struct Rt {
    std::vector<simt_stream *> streams;
    bool fuzz = false;
    int policy = 0;   // 0 random, 1 = the stream created first that can run (the main stream runs ahead, side streams starve), 2 = the last
    unsigned long long rng = 0;
    Rt() {
        if (const char *e = getenv("SIMT_STREAM_FUZZ")) {
            fuzz = true;
            if (!strcmp(e, "first")) policy = 1;
            else if (!strcmp(e, "last")) policy = 2;
            else rng = strtoull(e, nullptr, 10) * 0x9E3779B97F4A7C15ull + 12345;
        }
    }
    unsigned next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 11); }
};
Rt &rt() { static Rt r; return r; }
bool fuzz() { return rt().fuzz; }
hipStream_t new_stream() { auto *s = new simt_stream(); rt().streams.push_back(s); return s; }
hipEvent_t new_event() { return new simt_event(); }
// SIMT_MEMSET_ASYNC=1 (with SIMT_STREAM_FUZZ): hipMemset(), like cudaMemset(), may return before the fill has run -- it is queued on the
// null stream, which the (non-blocking) streams of the library do not wait for.  The null stream counts as the stream created first.
hipStream_t null_stream() {
    static hipStream_t ns = [] () -> hipStream_t {
        if (!rt().fuzz || !getenv("SIMT_MEMSET_ASYNC")) return nullptr;
        auto *s = new simt_stream(); rt().streams.insert(rt().streams.begin(), s); return s;
    }();
    return ns;
}
static bool runnable(const simt_stream::Op &op) { return !op.wait || op.wait->done >= op.wait_ticket; }
static void run_op(simt_stream *s) {
    simt_stream::Op op = std::move(s->q.front());
    s->q.pop_front();
    if (op.fn) op.fn();
    if (op.rec) op.rec->done = std::max(op.rec->done, op.rec_ticket);
}
// one scheduling step: a random stream whose head operation may run; false if nothing can
static bool step() {
    Rt &r = rt();
    std::vector<simt_stream *> ready;
    for (auto *s : r.streams) if (!s->q.empty() && runnable(s->q.front())) ready.push_back(s);
    if (ready.empty()) return false;
    run_op(r.policy == 1 ? ready.front() : r.policy == 2 ? ready.back() : ready[r.next() % ready.size()]);
    return true;
}
static void stall() { fprintf(stderr, "simt: stream deadlock (an operation waits for an event record that is not queued)\n"); abort(); }
void enqueue(hipStream_t s, std::function<void()> op) {
    if (!rt().fuzz || !s) { op(); return; }   // default, and the legacy null stream: immediate
    simt_stream::Op o; o.fn = std::move(op); s->q.push_back(std::move(o));
}
void record(hipEvent_t e, hipStream_t s) {
    if (!e) return;
    e->recorded++;
    if (!rt().fuzz || !s) { e->done = e->recorded; return; }
    simt_stream::Op o; o.rec = e; o.rec_ticket = e->recorded; s->q.push_back(std::move(o));
}
void wait_event(hipStream_t s, hipEvent_t e) {   // waits for the records issued so far (none recorded: no wait)
    if (!rt().fuzz || !s || !e || e->recorded == 0) return;
    simt_stream::Op o; o.wait = e; o.wait_ticket = e->recorded; s->q.push_back(std::move(o));
}
void sync_stream(hipStream_t s) {
    if (!rt().fuzz || !s) return;
    while (!s->q.empty()) if (!step()) stall();
}
void sync_event(hipEvent_t e) {
    if (!rt().fuzz || !e) return;
    while (e->done < e->recorded) if (!step()) stall();
}
void sync_all() {
    if (!rt().fuzz) return;
    for (;;) {
        bool any = false;
        for (auto *s : rt().streams) any = any || !s->q.empty();
        if (!any) return;
        if (!step()) stall();
    }
}
}  // namespace simt
