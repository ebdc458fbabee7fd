// orbx_extractor.hip -- host side of the extractor: tables (ORBextractor ctor), workspace layout in HBM,
// kernel launch sequence on the extractor's HIP stream, and the C ABI of include/orbx.h.
//
// HBM layout per extractor for a (width, height, batch) configuration, all slabs frame-major:
//   pyr    [B][pyr_frame]   8 padded levels; row = 45 B slack | 19 B ring | w ROI | 19 B ring | pad to 64 B
//   blur   [B][blur_frame]  8 blurred levels (no ring), 64-B pitched rows
//   cellcnt[B][cells]       FAST survivors per cell          cellent[B][cand] per-cell slots (packed keys)
//   keys0/1[B][cand]        quad-tree ping-pong key buffers   lvlkp [B][lvl]  selected keypoints per level
//   work   [B][cap]         keypoints in level order + output slot
//   kps    [B][cap] (28 B)  desc [B][cap][32]  count[B] mono[B]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

#include "extractor_kernels.hip.h"
#include "geometry_kernels.hip.h"
#include "extractor_state.h"

namespace orbx {

thread_local std::string g_last_error;

void set_error(const std::string &s) { g_last_error = s; }

namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *v = getenv("ORBX_ROCTX");
        if (!v || v[0] != '1') return;
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
Roctx &roctx() { static Roctx r; return r; }
}  // namespace
RoctxRange::RoctxRange(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
RoctxRange::~RoctxRange() { if (on) roctx().pop(); }

int guard_fill() {
    static const int m = [] { const char *v = getenv("ORBX_GUARD_FILL"); return v ? (atoi(v) & 0xff) : 0xCB; }();
    return m;
}
int guard_mode() {
    static const int m = [] { const char *v = getenv("ORBX_GUARD"); return v ? atoi(v) : 0; }();
    return m;
}

static const int8_t kPatternData[1024] = {
#include "orb_pattern.inc"
};

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

static const char *kKernelNames[K_COUNT] = {"k_pyr_base", "k_pyr_resize", "k_fast_strip", "k_octree", "k_finalize",
                                            "k_blur", "k_describe", "k_window_best2", "k_greedy_resolve"};

}  // namespace orbx

using namespace orbx;

namespace orbx {

// ---------------------------------------------------------------------------------------------------------
// k_pyr_stream's per-geometry tables (pyr_stream.hip.h): the bands of a frame, the row schedule of every band (which rows of which level are
// produced in which step), the LDS rings and the task list the waves of a workgroup work through.
//   level 0 = the frame's rows, staged `rows0` per step;  level l rows become computable once both their source rows are complete;
//   ring of level l = the largest distance, over the steps, between the newest row written and the oldest row its reader still needs.
// Returns false when the geometry does not fit (a scale factor above 2, LDS budget, 16-bit table fields): the per-level launches stay.
// ---------------------------------------------------------------------------------------------------------
struct PyrStreamPlan {
    PyrStreamGeom geom;
    std::vector<PyrStreamLevel> levels;   // [nlevels], entry 0 unused
    std::vector<PyrColumn> cols;
    std::vector<PyrStep> steps;      // [bands][steps_per_band]
    std::vector<PyrTask> tasks;      // all bands
    std::vector<uint32_t> band_task0;
    int bands = 0;
    size_t lds_bytes = 0;
};

static bool build_pyr_stream(const std::vector<LevelInfo> &lv, const std::vector<ResizeTap> &ytab, const std::vector<ResizeGroup> &xg, const bool *march_ok,
                             int bands, int rows0, int workers, size_t lds_budget, PyrStreamPlan &out) {
    const int nl = (int)lv.size();
    if (nl < 2 || nl > kMaxLevels || lv[0].w < 16) return false;
    for (int l = 1; l < nl; l++) if (!march_ok[l]) return false;
    PyrStreamPlan P;
    memset(&P.geom, 0, sizeof(P.geom));
    PyrStreamGeom &G = P.geom;
    P.levels.assign(nl, PyrStreamLevel{0u, 0u, 0u, 0u, 0u, 0u, 0, 0u});
    // ---- column tables ----
    std::vector<int> nchunk(nl, 0);
    uint32_t xg_bytes = 0;
    for (int l = 1; l < nl; l++) {
        const int nd = lv[l].pitch / 4;
        int first = -1, last = -1;
        for (int c = 0; c < nd; c++) if (xg[lv[l].xg_off + c].valid != 2) { if (first < 0) first = c; last = c; }
        if (first < 0) return false;
        PyrStreamLevel &S = P.levels[l];
        S.xg_lds = xg_bytes; S.col0 = (uint32_t)first; S.ncol = (uint32_t)(last - first + 1);
        S.pitch = (uint32_t)lv[l].pitch; S.off_lo = (uint32_t)lv[l].off; S.off_hi = (uint32_t)(lv[l].off >> 32);
        S.h = lv[l].h; S.roi_dw = (uint32_t)((lv[l].w + 3) / 4);
        nchunk[l] = (int)(S.ncol + 63) / 64;
        if (nchunk[l] > 15) return false;
        for (int c = first; c <= last; c++) {
            const ResizeGroup &g = xg[lv[l].xg_off + c];
            if (g.valid == 1 && (g.base < 0 || g.base > 0xffff)) return false;
            PyrColumn pc;
            pc.base_valid = (g.valid == 1 ? (uint32_t)g.base : 0u) | ((uint32_t)g.valid << 30);
            pc.sel = g.sel;
            for (int k = 0; k < 4; k++) pc.cc[k] = g.cc[k];
            P.cols.push_back(pc);
        }
        xg_bytes += S.ncol * (uint32_t)sizeof(PyrColumn);
    }
    if (P.cols.size() & 1) P.cols.push_back(PyrColumn{0u, 0u, {0u, 0u, 0u, 0u}});   // the tables are staged into LDS sixteen bytes per thread
    xg_bytes = (uint32_t)(P.cols.size() * sizeof(PyrColumn));
    G.xg_bytes = xg_bytes;
    // source rows of output row r of level l (clamped as k_pyr_resize_march clamps them)
    auto src0 = [&](int l, int r) { return std::min(std::max(ytab[lv[l].ytab_off + r].ofs, 0), lv[l - 1].h - 1); };
    auto src1 = [&](int l, int r) { return std::min(std::max(ytab[lv[l].ytab_off + r].ofs + 1, 0), lv[l - 1].h - 1); };
    G.workers = (uint32_t)std::min(std::max(workers, 1), 15);
    G.cpr0 = (uint32_t)((lv[0].w + 15) / 16);
    G.cpr0_rcp = (uint32_t)((0x100000000ull + G.cpr0 - 1) / G.cpr0);
    G.w0 = lv[0].w;
    if ((uint32_t)rows0 * G.cpr0 > (uint32_t)(kPyrStreamStage * 64)) rows0 = (int)((kPyrStreamStage * 64) / G.cpr0);
    if (rows0 < 2) return false;
    // ---- per band: compute ranges, schedule, ring sizes ----
    struct Band { std::vector<std::pair<int, int>> own, comp; std::vector<std::vector<std::pair<int, int>>> steps; std::vector<int> ring; };
    std::vector<Band> B(bands);
    std::vector<int> ring(nl, 0);
    size_t max_steps = 0;
    for (int k = 0; k < bands; k++) {
        Band &b = B[k];
        b.own.assign(nl, {0, 0}); b.comp.assign(nl, {0, 0});
        for (int l = 1; l < nl; l++) b.own[l] = {(int)((long)k * lv[l].h / bands), (int)((long)(k + 1) * lv[l].h / bands)};
        b.comp[nl - 1] = b.own[nl - 1];
        for (int l = nl - 2; l >= 0; l--) {
            int lo = src0(l + 1, b.comp[l + 1].first), hi = src1(l + 1, b.comp[l + 1].second - 1);
            if (b.comp[l + 1].second <= b.comp[l + 1].first) { lo = b.own[l].first; hi = b.own[l].first - 1; }
            if (l >= 1) { lo = std::min(lo, b.own[l].first); hi = std::max(hi, b.own[l].second - 1); }
            b.comp[l] = {lo, hi + 1};
        }
        std::vector<int> rb(nl);
        for (int l = 0; l < nl; l++) rb[l] = b.comp[l].first;
        auto done = [&]() { for (int l = 0; l < nl; l++) if (rb[l] < b.comp[l].second) return false; return true; };
        while (!done()) {
            const std::vector<int> prev = rb;
            rb[0] = std::min(b.comp[0].second, prev[0] + rows0);
            for (int l = 1; l < nl; l++) {
                int r = prev[l];
                while (r < b.comp[l].second && src1(l, r) < prev[l - 1]) r++;
                rb[l] = r;
            }
            std::vector<std::pair<int, int>> st(nl);
            for (int l = 0; l < nl; l++) st[l] = {prev[l], rb[l]};
            b.steps.push_back(st);
            if (b.steps.size() > 4096) return false;
        }
        for (auto &st : b.steps)
            for (int l = 0; l + 1 < nl; l++) {
                const int a = st[l + 1].first;
                const int lo = a < b.comp[l + 1].second ? src0(l + 1, a) : st[l].second;
                ring[l] = std::max(ring[l], st[l].second - lo);
            }
        max_steps = std::max(max_steps, b.steps.size());
    }
    // ---- LDS layout ----
    std::vector<uint32_t> ring_off(nl, 0), ring_pitch(nl, 0);
    size_t lds = xg_bytes;
    for (int l = 0; l + 1 < nl; l++) {
        ring[l] = std::max(ring[l], 2);
        ring_pitch[l] = (uint32_t)(((lv[l].w + 15) & ~15) + 16);
        ring_off[l] = (uint32_t)lds;
        lds += (size_t)ring[l] * ring_pitch[l];
    }
    lds += 16;
    if (lds > lds_budget || lds / 16 >= 0xffff) return false;
    G.ring0_off = ring_off[0]; G.ring0_pitch = ring_pitch[0]; G.ring0_rows = (uint32_t)ring[0];
    G.steps_per_band = (uint32_t)max_steps;
    // ---- steps and tasks ----
    P.steps.assign((size_t)bands * max_steps, PyrStep{0u, 0u, 0u, 0u});
    for (int k = 0; k < bands; k++) {
        const Band &b = B[k];
        P.band_task0.push_back((uint32_t)P.tasks.size());
        const size_t t0 = P.tasks.size();
        auto slot_off = [&](int l, int row) { return (uint16_t)((ring_off[l] + (uint32_t)((row - b.comp[l].first) % ring[l]) * ring_pitch[l]) / 16u); };
        for (size_t s = 0; s < max_steps; s++) {
            PyrStep &d = P.steps[(size_t)k * max_steps + s];
            d.task_begin = d.task_end = (uint32_t)(P.tasks.size() - t0);
            if (s >= b.steps.size()) continue;
            const auto &st = b.steps[s];
            const int y0 = st[0].first, ny = st[0].second - st[0].first;
            if (y0 > 0xffff || ny > 0xffff) return false;
            d.y0_rows = (uint32_t)y0 | ((uint32_t)ny << 16);
            d.slot0 = (uint32_t)((y0 - b.comp[0].first) % ring[0]);
            for (int l = 1; l < nl; l++) {
                int r = st[l].first;
                const int re = st[l].second;
                while (r < re) {
                    PyrTask T;
                    memset(&T, 0, sizeof(T));
                    const int a0 = src0(l, r), a1 = src1(l, r);
                    int rows = 1, nsrc = 2;
                    if (r + 1 < re && a1 == a0 + 1) {   // a pair: rows r, r + 1 from three (shared middle) or four consecutive source rows
                        const int c0 = src0(l, r + 1), c1 = src1(l, r + 1);
                        if (c0 == a1 && c1 == a1 + 1) { rows = 2; nsrc = 3; }
                        else if (c0 == a1 + 1 && c1 == a1 + 2) { rows = 2; nsrc = 4; }
                    }
                    const uint32_t s0 = slot_off(l - 1, a0), s1 = slot_off(l - 1, a1);
                    const uint32_t s2 = nsrc >= 3 ? slot_off(l - 1, a0 + 2) : s1, s3 = nsrc >= 4 ? slot_off(l - 1, a0 + 3) : s2;
                    T.src01 = s0 | (s1 << 16);
                    T.src23 = s2 | (s3 << 16);
                    const PyrStreamLevel &S = P.levels[l];
                    const int h = lv[l].h;
                    if (h < 2 * kEdge + 2) return false;   // a row with a copy in both rings: not handled (levels are at least 67 rows high)
                    for (int c = 0; c < nchunk[l]; c++) {
                        const uint32_t dcol0 = S.col0 + 64u * (uint32_t)c;                               // dword column of lane 0 inside the padded row
                        const int nlive = std::min(64, (int)S.ncol - 64 * c);
                        const int roi_first = std::max(0, kRoiX / 4 - (int)dcol0);                        // first lane whose dword lies in the ROI
                        const int roi_n = std::max(0, std::min(nlive, kRoiX / 4 + (int)S.roi_dw - (int)dcol0) - roi_first);
                        T.hdr = (uint32_t)(rows - 1) | ((uint32_t)nsrc << 1) | ((uint32_t)nlive << 4) | ((uint32_t)roi_first << 11) | ((uint32_t)roi_n << 18);
                        T.xg = S.xg_lds + 64u * (uint32_t)c * (uint32_t)sizeof(PyrColumn);
                        for (int q = 0; q < 2; q++) {
                            const int rr = r + (q < rows ? q : 0);
                            const ResizeTap &ty = ytab[lv[l].ytab_off + rr];
                            T.b[q] = (uint32_t)(uint16_t)ty.c0 | ((uint32_t)(uint16_t)ty.c1 << 16);
                            T.goff[q] = T.moff[q] = 0xffffffffu;
                            if (q == 0) T.dlds = 0xffffffffu;
                            if (q >= rows) continue;
                            if (l + 1 < nl && roi_n > 0)
                                T.dlds = (T.dlds & ~(0xffffu << (16 * q))) |
                                         ((((uint32_t)slot_off(l, rr) * 16u + 4u * (dcol0 + (uint32_t)roi_first - (uint32_t)(kRoiX / 4))) / 4u) << (16 * q));
                            if (rr >= b.own[l].first && rr < b.own[l].second) {
                                const uint64_t row0 = lv[l].off + 4ull * dcol0;
                                T.goff[q] = (uint32_t)(row0 + (uint64_t)(kEdge + rr) * lv[l].pitch);
                                if (rr >= 1 && rr <= kEdge) T.moff[q] = (uint32_t)(row0 + (uint64_t)(kEdge - rr) * lv[l].pitch);
                                if (rr <= h - 2 && rr >= h - 1 - kEdge) T.moff[q] = (uint32_t)(row0 + (uint64_t)(kEdge + 2 * (h - 1) - rr) * lv[l].pitch);
                            }
                        }
                        P.tasks.push_back(T);
                    }
                    r += rows;
                }
            }
            d.task_end = (uint32_t)(P.tasks.size() - t0);
            // the costliest tasks first: with the waves taking tasks w, w + W, ... the step's last round holds the cheap ones
            std::stable_sort(P.tasks.begin() + (ptrdiff_t)(t0 + d.task_begin), P.tasks.end(), [](const PyrTask &a, const PyrTask &b) {
                auto cost = [](const PyrTask &t) { return (int)((t.hdr & 1u) * 2u + ((t.hdr >> 1) & 7u)); };
                return cost(a) > cost(b);
            });
        }
    }
    if (P.tasks.empty()) return false;
    P.band_task0.push_back((uint32_t)P.tasks.size());   // sentinel: band b's tasks are [band_task0[b], band_task0[b + 1])
    for (int k = 0; k < bands; k++) if (P.band_task0[k + 1] == P.band_task0[k]) return false;   // a band without a task: not a geometry this form takes
    P.bands = bands;
    P.lds_bytes = lds;
    out = std::move(P);
    return true;
}

static int configure(orbx_extractor *ex, int width, int height, int batch) {
    if (width > kMaxDim || height > kMaxDim) return ORBX_E_TOO_LARGE;
    const bool same_geom = (width == ex->width && height == ex->height);
    if (same_geom && batch <= ex->batch_cap) return ORBX_OK;
    const int nl = ex->prm.nlevels;
    std::vector<LevelInfo> lv(nl);
    std::vector<ResizeTap> xtab, ytab;
    std::vector<ResizeGroup> xgtab;
    std::vector<TileRef> fast_tiles;
    std::vector<BlurItem> blur_items;
    std::vector<StripTile> strips;
    std::vector<int> strip_level_rows;
    bool fast_strip = true;
    bool fused_blur = false;   // committed to ex together with the geometry, once every allocation and upload has succeeded
    size_t pyr_off = 0, blur_off = 0;
    uint32_t cand_off = 0, lvl_off = 0;
    int cell_base = 0, cap = 0, max_pool = 0;
    size_t fast_lds = 0;
    int fast_wave_maxw = 0, fast_wave_rows = 0, fast_wave_qfull = 16;
    bool fast_wave = true;
    for (int l = 0; l < nl; l++) {
        LevelInfo &L = lv[l];
        memset(&L, 0, sizeof(L));
        // ComputePyramid :1174-1175
        L.w = cv_round_f((float)width * ex->inv_scale[l]);
        L.h = cv_round_f((float)height * ex->inv_scale[l]);
        const float fw = (float)(L.w - 2 * kBorder), fh = (float)(L.h - 2 * kBorder);
        if (fw < 35.f || fh < 35.f) return ORBX_E_TOO_SMALL;
        L.pitch = (kRoiX + L.w + kEdge + 63) & ~63;
        L.bpitch = (L.w + 63) & ~63;
        L.off = pyr_off;
        pyr_off += (size_t)L.pitch * (L.h + 2 * kEdge);
        pyr_off = (pyr_off + 255) & ~(size_t)255;
        L.boff = blur_off;
        blur_off += (size_t)L.bpitch * L.h;
        blur_off = (blur_off + 255) & ~(size_t)255;
        // ComputeKeyPointsOctTree :794-803
        const float W = 35;
        L.nCols = (int)(fw / W);
        L.nRows = (int)(fh / W);
        L.wCell = (int)std::ceil(fw / L.nCols);
        L.hCell = (int)std::ceil(fh / L.nRows);
        L.cell_base = cell_base;
        cell_base += L.nCols * L.nRows;
        // NMS survivors are pairwise non-adjacent => at most ceil(w/2)*ceil(h/2) per cell
        L.cell_cap = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);
        L.cand_off = cand_off;
        L.cand_cap = (uint32_t)(L.nCols * L.nRows * L.cell_cap);
        cand_off += (L.cand_cap + 63u) & ~63u;
        L.quota = ex->quota[l];
        // DistributeOctTree :559-560
        L.nIni = (int)std::round((float)(L.w - 2 * kBorder) / (float)(L.h - 2 * kBorder));
        if (L.nIni < 1) return ORBX_E_TOO_SMALL;  // the reference divides by zero for such aspect ratios
        L.hX = (float)(L.w - 2 * kBorder) / L.nIni;
        L.lvl_cap = std::max(L.quota + 4, 4 * L.nIni) + 4;
        L.pool = std::max(L.quota, 4 * L.nIni) + 16;
        if (L.pool >= 0xfff0) return ORBX_E_TOO_LARGE;
        max_pool = std::max(max_pool, L.pool);
        L.lvl_off = lvl_off;
        lvl_off += (uint32_t)((L.lvl_cap + 63) & ~63);
        cap += L.lvl_cap;
        L.scale = ex->scale[l];
        L.size = (float)(int)(kPatch * ex->scale[l]);  // :880
        // resize tables ([OCV] resize INTER_LINEAR 8U), from level l-1
        ex->resize_march_ok[l] = true;
        L.xtab_off = (uint32_t)xtab.size();
        L.ytab_off = (uint32_t)ytab.size();
        if (l > 0) {
            const LevelInfo &P = lv[l - 1];
            auto build = [](int ssize, int dsize, bool horizontal, std::vector<ResizeTap> &out) {
                const double inv = (double)dsize / ssize;
                const double sc = 1. / inv;
                for (int d = 0; d < dsize; d++) {
                    float fx = (float)((d + 0.5) * sc - 0.5);
                    int s = (int)std::floor(fx);
                    fx -= s;
                    if (horizontal) {
                        if (s < 0) { fx = 0; s = 0; }
                        if (s >= ssize - 1) { fx = 0; s = ssize - 1; }
                    }
                    ResizeTap t;
                    t.ofs = s;
                    t.c0 = (int16_t)std::min(std::max(cv_round_f((1.f - fx) * 2048.f), -32768), 32767);
                    t.c1 = (int16_t)std::min(std::max(cv_round_f(fx * 2048.f), -32768), 32767);
                    out.push_back(t);
                }
            };
            build(P.w, L.w, true, xtab);
            build(P.h, L.h, false, ytab);
            L.xg_off = (uint32_t)xgtab.size();
            for (int wi = 0; wi < L.pitch / 4; wi++) {
                ResizeGroup g;
                memset(&g, 0, sizeof(g));
                int lo = INT32_MAX, hi = INT32_MIN, ofs[4] = {0, 0, 0, 0};
                bool in[4];
                for (int k = 0; k < 4; k++) {
                    const int x = wi * 4 + k - kRoiX;
                    in[k] = x >= -kEdge && x < L.w + kEdge;
                    if (!in[k]) continue;
                    const int rx = x < 0 ? -x : (x >= L.w ? 2 * L.w - 2 - x : x);  // REFLECT_101
                    const ResizeTap &t = xtab[L.xtab_off + rx];
                    ofs[k] = t.ofs;
                    g.cc[k] = (uint32_t)(uint16_t)t.c0 | ((uint32_t)(uint16_t)t.c1 << 16);
                    lo = std::min(lo, t.ofs); hi = std::max(hi, t.ofs);
                }
                if (lo == INT32_MAX) { g.base = 0; g.valid = 2; }  // pitch padding outside the ring: the kernel stores zeros
                else {
                    g.base = lo;
                    g.valid = (hi + 1 - lo <= 7) ? 1 : 0;
                    if (!g.valid) ex->resize_march_ok[l] = false;
                    for (int k = 0; k < 4; k++) g.sel |= (uint32_t)(in[k] ? (ofs[k] - lo) & 7 : 0) << (8 * k);
                }
                xgtab.push_back(g);
            }
        }
        for (int i = 0; i < L.nRows; i++)
            for (int j = 0; j < L.nCols; j++) {
                fast_tiles.push_back(TileRef{(int16_t)l, (int16_t)i, (int16_t)j, 0});
            }
        strip_level_rows.push_back(std::min(L.hCell, L.h - 2 * kBorder - 6) + 6);   // tile rows of this level's strips (fast_strip.hip.h)
        {
            auto strip = [&](int y0, int rows_out, int x0) {
                BlurItem bi;
                bi.src_off = (uint32_t)(L.off + (size_t)(kEdge + y0 - 3) * L.pitch + kRoiX + x0);
                bi.dst_off = (uint32_t)(L.boff + (size_t)y0 * L.bpitch + x0);
                bi.pitches = (uint32_t)L.pitch | ((uint32_t)L.bpitch << 16);
                bi.misc = (uint32_t)(L.h + kEdge - 1 + 3 - y0)               // rmax: the level's last ring row
                          | ((uint32_t)rows_out << 16) | ((uint32_t)((std::min(kBlurTW, L.w - x0) + 3) / 4) << 24);
                return bi;
            };
            // k_blur_stream: one item per 256-pixel x 42-row strip
            for (int y0 = 0; y0 < L.h; y0 += kBlurRows)
                for (int x0 = 0; x0 < L.w; x0 += kBlurTW) blur_items.push_back(strip(y0, std::min(kBlurRows, L.h - y0), x0));
        }
        const int cols = L.wCell + 6, rows = L.hCell + 6;
        const size_t pp = (size_t)((cols + 3 + 3) & ~3);
        const size_t lds = 32 + ((rows * pp + 15) & ~(size_t)15) + ((((size_t)(L.hCell + 2) * (L.wCell + 2)) + 15) & ~(size_t)15) +
                           2 * (size_t)L.wCell * L.hCell + 64;
        fast_lds = std::max(fast_lds, lds);
        if (L.wCell + 6 > 63) fast_wave = false;  // k_fast_wave: the sub-image (+1 byte) must fit its 64-byte LDS pitch
        fast_wave_maxw = std::max(fast_wave_maxw, L.wCell);
        fast_wave_rows = std::max(fast_wave_rows, rows);
        fast_wave_qfull = std::max(fast_wave_qfull, (L.wCell * L.hCell + 15) & ~15);
    }
    if (oct_lds_bytes(max_pool) > 150 * 1024) return ORBX_E_TOO_LARGE;
    // k_pyr_stream: two bands per frame and as many frame rows per step as keep two workgroups on a CU (LDS <= 80 KB); wider frames take more bands,
    // then fewer rows per step
    // LDS budget of a workgroup: two of them AND a workgroup of the matcher's k_grid_build (24.5 KB static) fit a CU's 160 KB.  With 72 KB (ten frame
    // rows per step) the kernel alone is 8 % faster (157 against 170 us) but the previous batch's matcher cannot start beside it -- k_grid_build waits for
    // a CU with LDS to spare, the whole match stream slides behind the pyramid -- and the pipelined step is 2 % slower (0.880 against 0.861 ms,
    // profiles/r05_ab_rows.log)
    constexpr size_t kPyrStreamLdsBudget = 67 * 1024;
    PyrStreamPlan plans[4];
    bool plan_ok[4] = {false, false, false, false};
    bool ps_ok = false;
    for (int k = 0; k < 3; k++) {   // 1, 2, 4 bands per frame (8 bands: slower than 4 at every batch size measured, profiles/r05_q2_*)
        // (measured, profiles/r05_p3 / p4: eight workers, ten rows per step; more workers and 5 / 12 / 16 rows are slower)
        const int kb = 1 << k;
        if (kb * 16 > height) continue;
        int r_first = 10;
        if (const char *v = getenv("ORBX_PYR_STREAM_MIN")) { const char *c = strchr(v, ','); c = c ? strchr(c + 1, ',') : nullptr; if (c) r_first = std::max(2, atoi(c + 1)); }   // test hook, third field: rows per step
        for (int r0 : {r_first, 8, 6, 5, 4, 3})
            if ((plan_ok[k] = build_pyr_stream(lv, ytab, xgtab, ex->resize_march_ok, kb, r0, 8, kPyrStreamLdsBudget, plans[k]))) break;
        ps_ok = ps_ok || plan_ok[k];
    }

    ORBX_HIP(hipSetDevice(ex->device));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    for (hipStream_t s : {ex->copy_stream, ex->aux_stream, ex->match_stream, ex->in_stream}) if (s) ORBX_HIP(hipStreamSynchronize(s));
    ex->copy_pending = false; ex->match_pending = false; ex->copy_covers_match = false; ex->copy_issued = ex->copy_waited = 0;
    ex->mkey = orbx_extractor::MatchKey();
    ex->mpkey = orbx_extractor::MpKey();
    const int B = std::max(batch, ex->batch_cap);
    // from here on device tables and buffers change: a failure below must not leave the extractor claiming its old geometry (the early return at
    // the top would then launch on half-updated tables) -- the geometry is committed again at the end
    ex->width = 0; ex->height = 0;
    int r;
#define ENS(buf, bytes) if ((r = (buf).ensure(bytes)) != ORBX_OK) return r
    ENS(ex->d_lv, sizeof(LevelInfo) * nl);
    ENS(ex->d_xtab, sizeof(ResizeTap) * std::max<size_t>(xtab.size(), 1));
    ENS(ex->d_ytab, sizeof(ResizeTap) * std::max<size_t>(ytab.size(), 1));
    ENS(ex->d_xgtab, sizeof(ResizeGroup) * std::max<size_t>(xgtab.size(), 1));
    ENS(ex->d_fast_tiles, sizeof(TileRef) * fast_tiles.size());
    ENS(ex->d_blur_items, sizeof(BlurItem) * blur_items.size());
    for (int k = 0; k < 4; k++)
        if (plan_ok[k]) {
            ENS(ex->ps_plan[k].cols, plans[k].cols.size() * sizeof(PyrColumn));
            ENS(ex->ps_plan[k].steps, plans[k].steps.size() * sizeof(PyrStep));
            ENS(ex->ps_plan[k].tasks, plans[k].tasks.size() * sizeof(PyrTask));
            ENS(ex->ps_plan[k].band0, plans[k].band_task0.size() * sizeof(uint32_t));
        }
    ENS(ex->d_pyr, pyr_off * B);
    if (ex->pyr_double) ENS(ex->d_pyr2, pyr_off * B);
    // The blur on demand (k_describe_fused) filters 43 x 43 pixels per keypoint (VALU bound: +39 us per 256 k keypoints over k_describe), the blur pass
    // (k_blur_stream) every pixel of the pyramid once (HBM bound, 200 us per 285 M pixels, about half of it hidden beside the FAST strips).  Measured
    // (profiles/r04_s_ab_*, r04_t_ab_*): EuRoC 752x480 / 1000 (the keypoints' windows = 1.2 x the pyramid) step 1.053 -> 0.972 ms, TUM-VI 1024x1024 / 1500
    // (0.6 x) 1.54 -> 1.33, KITTI 1241x376 / 2000 (1.9 x) 1.43 -> 1.34.  By those rates the blur pass wins beyond 3 - 6 x: on demand up to 4 x
    // (e.g. not for the 5 x nFeatures extractor of the monocular initialisation).  ORBX_FUSED_BLUR=0 / 1 forces either form.
    {
        size_t pyr_px = 0;
        for (int l = 0; l < nl; l++) pyr_px += (size_t)lv[l].w * lv[l].h;
        const char *v = getenv("ORBX_FUSED_BLUR");
        fused_blur = v && (v[0] == '0' || v[0] == '1') ? v[0] == '1' : (size_t)std::max(ex->prm.nfeatures, 0) * 1369 <= 4 * pyr_px;
    }
    if (!fused_blur) ENS(ex->d_blur, blur_off * B);   // (orbx_debug_level_blurred allocates it when it is the only user)
    ENS(ex->d_cellcnt, sizeof(int32_t) * (size_t)cell_base * B);
    ENS(ex->d_cellent, sizeof(uint32_t) * (size_t)cand_off * B);
    ENS(ex->d_keys0, sizeof(uint32_t) * (size_t)cand_off * B);
    ENS(ex->d_keys1, sizeof(uint32_t) * (size_t)cand_off * B);
    ENS(ex->d_fast_ovf, 4 * (16 + fast_tiles.size() * (size_t)B) + 64);
    ENS(ex->d_nof0, sizeof(uint16_t) * (size_t)cand_off * B);
    ENS(ex->d_nof1, sizeof(uint16_t) * (size_t)cand_off * B);
    ENS(ex->d_lvlkp, sizeof(uint32_t) * (size_t)lvl_off * B);
    ENS(ex->d_lvlcnt, sizeof(int32_t) * (size_t)nl * B);
    ENS(ex->d_candtot, sizeof(int32_t) * (size_t)nl * B);
    ENS(ex->d_work, sizeof(WorkItem) * (size_t)cap * B);
    ENS(ex->d_kps, sizeof(orbx_keypoint) * (size_t)cap * B);
    ENS(ex->d_desc, (size_t)32 * cap * B);
    if (ex->has_camera) { ENS(ex->d_kps_un, sizeof(orbx_keypoint) * (size_t)cap * B); }
    ENS(ex->d_count, sizeof(int32_t) * (size_t)B);
    ENS(ex->d_mono, sizeof(int32_t) * (size_t)B);
    ENS(ex->d_err, sizeof(int32_t));
#undef ENS
    ORBX_HIP(hipMemcpy(ex->d_lv.p, lv.data(), sizeof(LevelInfo) * nl, hipMemcpyHostToDevice));
    if (!xtab.empty()) ORBX_HIP(hipMemcpy(ex->d_xtab.p, xtab.data(), sizeof(ResizeTap) * xtab.size(), hipMemcpyHostToDevice));
    if (!ytab.empty()) ORBX_HIP(hipMemcpy(ex->d_ytab.p, ytab.data(), sizeof(ResizeTap) * ytab.size(), hipMemcpyHostToDevice));
    if (!xgtab.empty()) ORBX_HIP(hipMemcpy(ex->d_xgtab.p, xgtab.data(), sizeof(ResizeGroup) * xgtab.size(), hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(ex->d_fast_tiles.p, fast_tiles.data(), sizeof(TileRef) * fast_tiles.size(), hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(ex->d_blur_items.p, blur_items.data(), sizeof(BlurItem) * blur_items.size(), hipMemcpyHostToDevice));
    ex->ps_ok = false;
    for (int k = 0; k < 4; k++) {
        ex->ps_plan[k].ok = false;
        if (!plan_ok[k]) continue;
        const PyrStreamPlan &pl = plans[k];
        ORBX_HIP(hipMemcpy(ex->ps_plan[k].cols.p, pl.cols.data(), pl.cols.size() * sizeof(PyrColumn), hipMemcpyHostToDevice));
        ORBX_HIP(hipMemcpy(ex->ps_plan[k].steps.p, pl.steps.data(), pl.steps.size() * sizeof(PyrStep), hipMemcpyHostToDevice));
        ORBX_HIP(hipMemcpy(ex->ps_plan[k].tasks.p, pl.tasks.data(), pl.tasks.size() * sizeof(PyrTask), hipMemcpyHostToDevice));
        ORBX_HIP(hipMemcpy(ex->ps_plan[k].band0.p, pl.band_task0.data(), pl.band_task0.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (pl.lds_bytes > 64 * 1024) ORBX_HIP(hipFuncSetAttribute((const void *)k_pyr_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    }
    ORBX_HIP(hipMemset(ex->d_err.p, 0, sizeof(int32_t)));
    // Cells the reference skips (empty interior: iniX >= maxBorderX - 6 / iniY >= maxBorderY - 3, ORBextractor.cc:810,819 -- e.g. cell column 33 of
    // level 0 of a 1226 x 370 image) belong to no strip and are never written by the FAST stage; compact_level reads every cell's count.  A
    // buffer kept from another geometry holds that geometry's counts there, and a buffer re-allocated for a larger batch of the same geometry starts
    // with the allocator's fill (poison bytes in guard mode): cleared at every (re)configuration -- this is a cold path.
    ORBX_HIP(hipMemset(ex->d_cellcnt.p, 0, ex->d_cellcnt.bytes));
    ORBX_HIP(hipDeviceSynchronize());   // the fill has run (DevBuf::ensure, extractor_state.h)
    ex->lv = lv;
    ex->ps_ok = ps_ok;
    for (int k = 0; k < 4; k++)
        if (plan_ok[k]) {
            ex->ps_plan[k].geom = plans[k].geom; ex->ps_plan[k].bands = plans[k].bands; ex->ps_plan[k].lds = plans[k].lds_bytes; ex->ps_plan[k].ok = true;
            if (getenv("ORBX_DEBUG_ALLOC"))
                fprintf(stderr, "[orbx pyr] k_pyr_stream plan: %d bands, %u steps, %zu tasks, LDS %zu B (tables %u B)\n", plans[k].bands, plans[k].geom.steps_per_band,
                        plans[k].tasks.size(), plans[k].lds_bytes, plans[k].geom.xg_bytes);
        }
    if (const char *v = getenv("ORBX_PYR_STREAM_MIN")) {   // test hook "<frames>[,<workgroups>]": the smallest batch that takes k_pyr_stream, the launch size the band plan aims at
        ex->ps_min_frames = std::max(1, atoi(v));
        if (const char *c = strchr(v, ',')) ex->ps_wg_target = std::max(1, atoi(c + 1));
    }
    ex->fused_blur = fused_blur;
    ex->width = width; ex->height = height; ex->batch_cap = B;
    ex->pyr_frame = pyr_off; ex->blur_frame = blur_off; ex->cand_frame = cand_off; ex->lvl_frame = lvl_off;
    ex->total_cells = cell_base; ex->cap = cap; ex->max_pool = max_pool; ex->fast_lds = fast_lds; ex->fast_wave = fast_wave && fast_tiles.size() < 65536 && batch < 65536;
    // k_fast_strip's tiles: the cells of a cell row in strips of floor(interior / wCell) cells.  The LDS budget of a workgroup's pixel tile is
    // set by the ordinary levels (full 272-byte pitch x the row count that covers 85 % of the cells); a level with taller cells gets a narrower
    // pitch (fewer cells per strip), so that ONE launch with one LDS size serves every level.
    {
        std::vector<std::pair<int, int>> rows_cells;   // (tile rows, cells) per level
        int all_cells = 0;
        for (int l = 0; l < nl; l++) { rows_cells.push_back({strip_level_rows[l], lv[l].nCols * lv[l].nRows}); all_cells += lv[l].nCols * lv[l].nRows; }
        std::sort(rows_cells.begin(), rows_cells.end());
        int rows_main = rows_cells.back().first, acc = 0;
        for (auto &rc : rows_cells) { acc += rc.second; if (acc * 100 >= all_cells * 85) { rows_main = rc.first; break; } }
        const int budget = rows_main * kStripPitch;
        ex->strip_pix_bytes = budget;
        for (int l = 0; l < nl && fast_strip; l++) {
            const LevelInfo &L = lv[l];
            const int maxBX = L.w - kBorder, maxBY = L.h - kBorder;
            const uint32_t rcpw = (65536u + (uint32_t)L.wCell - 1u) / (uint32_t)L.wCell;
            for (int x = 0; x < 256; x++) if ((int)(((uint32_t)x * rcpw) >> 16) != x / L.wCell) fast_strip = false;
            if (L.hCell > 63) fast_strip = false;
            const int R = strip_level_rows[l];
            int pitch = 0;
            for (int p : {kStripPitch, kStripPitchMid, kStripPitchLow})
                if (!pitch && R * p <= budget && strip_max_interior(p) >= L.wCell) pitch = p;
            if (!pitch) { fast_strip = false; break; }
            const int kmax = std::max(1, std::min(strip_max_interior(pitch) / L.wCell, kStripMaxCells - 1));
            for (int i = 0; i < L.nRows; i++) {
                const int Y0 = kBorder + 3 + i * L.hCell, IH = std::min(Y0 + L.hCell, maxBY - 3) - Y0;
                if (IH <= 0) continue;   // :810 and sub-images with fewer than 7 rows
                for (int j0 = 0; j0 < L.nCols; j0 += kmax) {
                    const int j1 = std::min(j0 + kmax, L.nCols);
                    const int X0 = kBorder + 3 + j0 * L.wCell, IW = std::min(kBorder + 3 + j1 * L.wCell, maxBX - 3) - X0;
                    if (IW <= 0) continue;   // :819 and sub-images with fewer than 7 columns
                    StripTile T;
                    memset(&T, 0, sizeof(T));
                    T.src_off = (uint32_t)(L.off + (size_t)(kEdge + Y0 - 3) * L.pitch + kRoiX + X0 - 4);
                    T.pitch = L.pitch;
                    T.iw = (int16_t)IW; T.ih = (int16_t)IH;
                    T.ox = (int16_t)(3 + j0 * L.wCell); T.oy = (int16_t)(3 + i * L.hCell);
                    T.cell0 = (uint32_t)(L.cell_base + i * L.nCols + j0);
                    T.slot0 = L.cand_off + (uint32_t)(i * L.nCols + j0) * (uint32_t)L.cell_cap;
                    T.cell_cap = (uint32_t)L.cell_cap;
                    T.rcp_wcell = rcpw;
                    const uint32_t G = (uint32_t)((IW + 3) >> 2);
                    T.rcp_groups = ((1u << 20) + G - 1u) / G;
                    T.rows_per_iter = (uint16_t)(64u / G);
                    T.ncell = (uint16_t)((IW + L.wCell - 1) / L.wCell);
                    T.lds_pitch = (uint32_t)pitch;
                    T.wcell = (uint32_t)L.wCell;
                    if ((IH + 6) * pitch > budget) fast_strip = false;
                    strips.push_back(T);
                }
            }
        }
        if (fast_strip && !strips.empty()) {
            // a failure here must not leave the new geometry committed beside the old strip tables: width / height = 0 sends the next call through prepare again
            int rr = ex->d_strips.ensure(sizeof(StripTile) * strips.size());
            if (rr != ORBX_OK) { ex->width = 0; ex->height = 0; return rr; }
            const hipError_t he = hipMemcpy(ex->d_strips.p, strips.data(), sizeof(StripTile) * strips.size(), hipMemcpyHostToDevice);
            if (he != hipSuccess) { ex->width = 0; ex->height = 0; ORBX_HIP(he); }
        }
    }
    ex->n_strips = (int)strips.size();
    ex->n_strips0 = 0;
    {   // what a wave's queues would have to hold if every group / pixel of its band passed (the ceiling of orbx_tune_fast_queues)
        int gmax = 64, qmax = 256;
        for (const StripTile &T : strips) {
            const int bh = (T.ih + 3) / 4, G = (T.iw + 3) >> 2;
            gmax = std::max(gmax, bh * G); qmax = std::max(qmax, bh * 4 * G);
        }
        ex->strip_gmax = (gmax + 7) & ~7; ex->strip_qmax = (qmax + 15) & ~15;
        ex->strip_gcap = std::min(ex->strip_gcap, std::max(ex->strip_gmax, orbx_extractor::kStripGcap0));   // a geometry with shorter bands than the one the queues were enlarged for
        ex->strip_qcap = std::min(ex->strip_qcap, std::max(ex->strip_qmax, orbx_extractor::kStripQcap0));
    }
    for (const StripTile &T : strips) if (T.src_off < lv[0].off + (size_t)lv[0].pitch * (lv[0].h + 2 * kEdge)) ex->n_strips0++;   // level 0 lies first in the slab
    ex->fast_strip = fast_strip && ex->fast_wave && !strips.empty();
    if (ex->strip_qcap > 2 * ex->strip_gcap) ex->strip_gcap = (ex->strip_qcap / 2 + 7) & ~7;   // the scores reuse the group queue's bytes
    if (getenv("ORBX_DEBUG_ALLOC"))
        fprintf(stderr, "[orbx fast] %s: %d strips per frame, %zu cells, pixel tile %d B, LDS %zu B per workgroup\n", ex->fast_strip ? "k_fast_strip" : "per-cell kernels",
                ex->n_strips, fast_tiles.size(), ex->strip_pix_bytes, fast_strip_lds_bytes(4, ex->strip_pix_bytes, ex->strip_gcap, ex->strip_qcap));
    ex->fast_wave_pitch = (fast_wave_maxw + 7 <= 48) ? 48 : 64;
    ex->fast_wave_rows = fast_wave_rows;
    ex->fast_wave_qfull = fast_wave_qfull;
    { const char *v = getenv("ORBX_OCTREE"); ex->oct_par = !(v && v[0] == 's') && oct_par_lds_bytes(max_pool) <= 150 * 1024;
      ex->oct_single_wave = v && v[0] == 'w'; }   // ORBX_OCTREE=w1: the global-memory form on ONE wave (round 5's; the whole workgroup since round 6)
    ex->n_fast_tiles = (int)fast_tiles.size(); ex->n_blur_items = (int)blur_items.size();
    ex->last_batch = 0;
    ex->lvl0_inplace = false;
    if (ex->has_camera) {
        CameraModel c = {ex->cam_params[0], ex->cam_params[1], ex->cam_params[2], ex->cam_params[3], ex->cam_params[4], ex->cam_params[5],
                         ex->cam_params[6], ex->cam_params[7], ex->cam_params[8]};
        image_bounds(c, width, height, ex->bounds);
    } else {
        ex->bounds[0] = 0.f; ex->bounds[1] = (float)width; ex->bounds[2] = 0.f; ex->bounds[3] = (float)height;   // Frame.cc:804-807
    }
    return ORBX_OK;
}

struct ProfScope {
    orbx_extractor *ex;
    int k;
    ProfScope(orbx_extractor *e, int kernel) : ex(e), k(kernel) {
        if (ex->profile) (void)hipEventRecord(ex->ev0, ex->stream);
    }
    ~ProfScope() {
        if (ex->profile) {
            (void)hipEventRecord(ex->ev1, ex->stream);
            (void)hipEventSynchronize(ex->ev1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ex->ev0, ex->ev1);
            ex->prof_ms[k] += ms;
            ex->prof_n[k] += 1;
        }
    }
};

// the float forms the descriptor kernels evaluate (kFpDescStrict | kFpAtanFma) from orbx_params.flags
static int fp_mode_of(uint32_t flags) { return ((flags & ORBX_FLAG_DESC_STRICT) ? kFpDescStrict : 0) | ((flags & ORBX_FLAG_ATAN_FMA) ? kFpAtanFma : 0); }

// taps of GaussianBlur(7x7, sigma 2) in 8-bit fixed point: [OCV] >= 4.5.1 / <= 4.5.0 (ORBX_FLAG_BLUR_OCV440); the older ones sum to more than 1.0
// and can exceed 255 (the saturating forms of the kernels)
struct BlurTaps { int g[4]; bool sat; };
static BlurTaps blur_taps(const orbx_extractor *ex) {
    static const int kBlurNew[4] = {18, 34, 48, 56}, kBlurOld[4] = {18, 34, 49, 55};
    const int *bg = (ex->prm.flags & ORBX_FLAG_BLUR_OCV440) ? kBlurOld : kBlurNew;
    return BlurTaps{{bg[0], bg[1], bg[2], bg[3]}, 2 * (bg[0] + bg[1] + bg[2]) + bg[3] > 256};
}
// k_blur_stream over every strip of every level of the first n frames of the pyramid slab `pyr` into the blur slab
static void launch_blur_stream(orbx_extractor *ex, int n, const uint8_t *pyr, hipStream_t bs) {
    const BlurTaps bt = blur_taps(ex);
    const int *bg = bt.g;
    uint8_t *blur_slab = (uint8_t *)ex->d_blur.p;
    const BlurItem *items = (const BlurItem *)ex->d_blur_items.p;
    const int nitems = ex->n_blur_items;
    const int nx = n >= 8 ? 8 : 1;
    const long items_per_group = (long)((n + nx - 1) / nx) * nitems;
    const int K = (int)std::max<long>(1, std::min<long>(ex->blur_waves / nx, items_per_group));
    if (bt.sat) hipLaunchKernelGGL(k_blur_stream<true>, dim3(K * nx), dim3(64), 0, bs, items, nitems, pyr, ex->pyr_frame, blur_slab, ex->blur_frame,
                                   bg[0], bg[1], bg[2], bg[3], n, nx);
    else hipLaunchKernelGGL(k_blur_stream<false>, dim3(K * nx), dim3(64), 0, bs, items, nitems, pyr, ex->pyr_frame, blur_slab, ex->blur_frame,
                            bg[0], bg[1], bg[2], bg[3], n, nx);
}

// k_pyr_base: the frames into the padded level-0 slab of `pyr` (ring = REFLECT_101 of the image)
static int launch_pyr_base(orbx_extractor *ex, const uint8_t *d_images, int n, size_t row_stride, size_t frame_stride, uint8_t *pyr, hipStream_t st, bool zero_word) {
    ProfScope ps(ex, K_PYR_BASE);
    const LevelInfo &L = ex->lv[0];
    const dim3 grid = xcd_grid(((L.pitch / 16) * (L.h + 2 * kEdge) + 255) / 256, n, true);   // a frame's workgroups stay on one XCD
    hipLaunchKernelGGL(k_pyr_base, grid, dim3(256), 0, st, L, d_images, row_stride, frame_stride, pyr, ex->pyr_frame,
                       zero_word ? (int32_t *)ex->d_fast_ovf.p : (int32_t *)nullptr,
                       (uint32_t)((0x100000000ull + (uint64_t)(L.pitch / 16) - 1) / (uint64_t)(L.pitch / 16)), n);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

}  // namespace orbx

// The padded level 0 of the last batch, if that batch was extracted in place (extractor_state.h): written now, on the extractor's stream, from the
// frames the caller still holds; ev_describe is recorded again so that whoever waits for the extraction waits for this too.
int orbx_materialize_level0(orbx_extractor *ex) {
    if (!ex) return ORBX_E_BAD_ARG;
    if (!ex->lvl0_inplace) return ORBX_OK;
    if (ex->in0_released) {
        set_error("level 0 of the last batch was read in place from the caller's frames and never copied; orbx_sync / orbx_download_wait have since released "
                  "those frames to the caller -- ask for level 0 (orbx_get_level / orbx_get_level_device) before that call");
        return ORBX_E_STALE;
    }
    ORBX_HIP(hipSetDevice(ex->device));
    int r = orbx::launch_pyr_base(ex, ex->in0_images, ex->last_batch, ex->in0_row_stride, ex->in0_frame_stride, ex->pyr_cur(), ex->stream, false);
    if (r != ORBX_OK) return r;
    ORBX_HIP(hipEventRecord(ex->ev_describe, ex->stream));
    if (ex->in0_event) ORBX_HIP(hipEventRecord(ex->in0_event, ex->stream));   // the upload slab the frames lie in stays busy until k_pyr_base has read it
    ex->lvl0_inplace = false;
    return ORBX_OK;
}

namespace orbx {

// enqueue the whole extraction of `n` device-resident frames on ex->stream
static int enqueue_extract(orbx_extractor *ex, const uint8_t *d_images, int n, size_t row_stride, size_t frame_stride,
                           int lap0, int lap1, hipEvent_t ev_input_consumed = nullptr, const HostMirror *mirror = nullptr) {
    RoctxRange rr("orbx:extract");
    ex->internal_match_owner = 0;   // a new batch: its first batched matcher owns the internal match buffers
    const int nl = ex->prm.nlevels;
    const LevelInfo *d_lv = (const LevelInfo *)ex->d_lv.p;
    if (ex->pyr_double) ex->pyr_slot ^= 1;   // the previous batch's pyramid may still be read by the rig's SAD stage (orbx_stereo_batch_device)
    uint8_t *pyr = ex->pyr_cur();
    uint8_t *blur_slab = (uint8_t *)ex->d_blur.p;
    hipStream_t st = ex->stream;
    hipStream_t pst = st;   // stream of the pyramid stage
    const bool pyr_local = true;   // a frame's pyramid workgroups stay on one XCD
    // the plan whose band count brings the launch closest to ps_wg_target workgroups (two per CU); fewer bands win a tie (less overlap work)
    const orbx_extractor::PyrPlanDev *pp = nullptr;
    // (an extractor of a stereo rig keeps its padded level 0 for the SAD stage, so k_pyr_base runs anyway -- and beside the rig's second extractor the
    // per-level chain is the faster form: KITTI 1.34 against 1.37 - 1.41 ms per 128 pairs, profiles/r05_ab_bands.log)
    if (ex->ps_ok && n >= ex->ps_min_frames && !ex->pyr_double) {
        long best = -1;
        for (int k = 0; k < 4; k++) {
            if (!ex->ps_plan[k].ok) continue;
            const long wgs = (long)n * ex->ps_plan[k].bands, miss = wgs >= ex->ps_wg_target ? wgs - ex->ps_wg_target : 2 * (ex->ps_wg_target - wgs);
            if (best < 0 || miss < best) { best = miss; pp = &ex->ps_plan[k]; }
        }
    }
    const bool stream = pp != nullptr;
    // level 0 in place: every reader of level 0 in this call can take the caller's frames (the FAST strips never touch the ring, k_describe_fused
    // reflects the few windows that cross the border itself, k_pyr_stream reads the frames anyway)
    const bool inplace0 = stream && ex->fused_blur && ex->fast_strip && !ex->pyr_double && mirror == nullptr;
    const Level0Src src0 = inplace0 ? Level0Src{d_images, row_stride, frame_stride} : Level0Src{nullptr, 0, 0};
    ex->lvl0_inplace = inplace0;
    ex->in0_images = d_images; ex->in0_row_stride = row_stride; ex->in0_frame_stride = frame_stride;
    ex->in0_released = false; ex->in0_copy_seq = ex->copy_issued; ex->in0_event = inplace0 ? ev_input_consumed : nullptr;
    if (!inplace0) {
        int r = launch_pyr_base(ex, d_images, n, row_stride, frame_stride, pyr, pst, true);
        if (r != ORBX_OK) return r;
    }
    if (ev_input_consumed && !stream) ORBX_HIP(hipEventRecord(ev_input_consumed, pst));  // k_pyr_base is the only reader of the input frames
    const int ini_th = std::min(std::max(ex->prm.ini_th_fast, 0), 255), min_th = std::min(std::min(std::max(ex->prm.min_th_fast, 0), 255), ini_th);
    const BlurTaps bt = blur_taps(ex);
    if (stream) {   // levels 1 .. nl-1 in one launch, straight from the caller's frames
        ProfScope ps(ex, K_PYR_RESIZE);
        hipLaunchKernelGGL(k_pyr_stream, xcd_grid(pp->bands, n, pyr_local), dim3(64 * (pp->geom.workers + 1)), pp->lds, pst, pp->geom, (const uint4 *)pp->cols.p,
                           (const PyrStep *)pp->steps.p, (const PyrTask *)pp->tasks.p, (const uint32_t *)pp->band0.p, d_images, row_stride,
                           frame_stride, pyr, ex->pyr_frame, inplace0 ? (int32_t *)ex->d_fast_ovf.p : (int32_t *)nullptr, n);
        if (ev_input_consumed && !inplace0) ORBX_HIP(hipEventRecord(ev_input_consumed, pst));   // k_pyr_base and k_pyr_stream have read the frames
    }
    for (int l = 1; l < nl && !stream; l++) {
        ProfScope ps(ex, K_PYR_RESIZE);
        const LevelInfo &L = ex->lv[l];
        if (ex->resize_march_ok[l]) {
            // register-marching form: one wave = 64 dword columns x rb output rows (rb 16 / 32 and 4 / 8 source rows in flight measured
            // alike, 64 rows per block 5 % slower: profiles/r03_c_ab_resize_march_strip_waves_prime.log)
            // A few frames cannot fill the machine and the launch lasts as long as ONE wave's march: short blocks then (single-frame call: 32 / 16 /
            // 8 / 4 rows per block = 0.215 / 0.202 / 0.190 / 0.181 ms, profiles/r03_s_single_frame_latency.log)
            const int nstrips = (L.pitch / 4 + 63) / 64, rb = n <= 8 ? 4 : L.h >= 256 ? 32 : 16, n_items = nstrips * ((L.h + rb - 1) / rb);
            const uint32_t rcp = (uint32_t)((0x100000000ull + (uint64_t)nstrips - 1) / (uint64_t)nstrips);
            hipLaunchKernelGGL(k_pyr_resize_march<8>, xcd_grid((n_items + 3) / 4, n, pyr_local), dim3(256), 0, pst, L, ex->lv[l - 1],
                               (const ResizeTap *)ex->d_ytab.p, (const ResizeGroup *)ex->d_xgtab.p, pyr, ex->pyr_frame, rb, nstrips, rcp, n_items, n);
            continue;
        }
        // scale factors above 2 (a dword column's taps further apart than 8 source bytes): the table form
        const uint32_t wpc = (uint32_t)(L.pitch / 8);
        const dim3 grid2 = xcd_grid((int)((wpc * (uint32_t)((L.h + 2 * kEdge + kResizeRows - 1) / kResizeRows) + 255u) / 256u), n, pyr_local);
        hipLaunchKernelGGL(k_pyr_resize2, grid2, dim3(256), 0, pst, L, ex->lv[l - 1], (const ResizeTap *)ex->d_xtab.p,
                           (const ResizeTap *)ex->d_ytab.p, (const ResizeGroup *)ex->d_xgtab.p, pyr, ex->pyr_frame,
                           (uint32_t)((0x100000000ull + (uint64_t)wpc - 1) / (uint64_t)wpc), n);
    }
    auto launch_blur = [&]() -> int {   // one launch over all levels, beside FAST on the aux stream (ORBX_SIDE_STREAMS=0 / profile mode: main stream)
        const bool side = !ex->profile && ex->side_streams;
        hipStream_t bs = side ? ex->aux_stream : st;
        if (side) ORBX_HIP(hipStreamWaitEvent(bs, ex->ev_pyr, 0));
        ProfScope ps(ex, K_BLUR);
        launch_blur_stream(ex, n, pyr, bs);
        if (side) ORBX_HIP(hipEventRecord(ex->ev_blur, bs));
        return ORBX_OK;
    };
    ORBX_HIP(hipEventRecord(ex->ev_pyr, pst));                         // the pyramid of this batch is complete
    if (!ex->fused_blur) { int r = launch_blur(); if (r != ORBX_OK) return r; }
    {
        ProfScope ps(ex, K_FAST);
        // one score map at min(ini, min) serves both passes of :830-846; for ini < min the reference's second pass FAST(min) is a
        // subset of its first, so its result is FAST(ini) alone -- the same as running this stage with min := ini
        const int ini = ini_th, mn = min_th;
        if (ex->fast_strip) {
            int32_t *ovf_count = (int32_t *)ex->d_fast_ovf.p;
            uint32_t *ovf_list = (uint32_t *)ex->d_fast_ovf.p + 16;
            // first pass of :826 for every cell, one workgroup per strip of cells; cells it leaves empty go to the list pass below
            // (ini <= min: the second pass is a subset of the first, an empty cell stays empty)
            // (Level 0's strips on the aux stream beside the latency-bound resize chain were measured: EuRoC 1.10 vs 1.11 ms, but TUM-VI 1.02 vs
            // 0.84 ms -- with the default four hardware queues the aux stream shares one with the matcher, with eight everything else slows:
            // profiles/r03_j_*, r03_k_*.  One launch on the main stream.)
#define ORBX_FAST_STRIP(W)                                                                                                                                   \
    hipLaunchKernelGGL(k_fast_strip<W>, xcd_grid(ex->n_strips, n), dim3(64 * W), fast_strip_lds_bytes(W, ex->strip_pix_bytes, ex->strip_gcap, ex->strip_qcap), st, \
                       (const StripTile *)ex->d_strips.p, (const uint8_t *)pyr, ex->pyr_frame, (int32_t *)ex->d_cellcnt.p, ex->total_cells,                        \
                       (uint32_t *)ex->d_cellent.p, ex->cand_frame, ini, mn, ex->strip_pix_bytes, ex->strip_gcap, ex->strip_qcap, ovf_list, ovf_count,             \
                       ini > mn ? 1 : 0, n, src0, ex->n_strips0)
            ORBX_FAST_STRIP(4);   // (two waves per strip, bands of twice the rows: fewer part-filled iterations, half the waves per CU -- 339 against 297 us, profiles/r05_q6)
#undef ORBX_FAST_STRIP
            ORBX_HIP(hipGetLastError());   // e.g. an LDS budget the device refuses: fail here, not as silently missing candidates
            // second pass (:843-846) and strips whose queues overflowed: one wave per listed cell, queue sized for a whole cell
            const size_t lds_full = fast_wave_lds_bytes(ex->fast_wave_pitch, ex->fast_wave_rows, ex->fast_wave_qfull);
#define ORBX_FAST_WAVE_LIST(PITCH)                                                                                                  \
    hipLaunchKernelGGL(k_fast_wave_list<PITCH>, dim3(n < 8 ? 256 : 4096), dim3(64), lds_full, st, d_lv, (const TileRef *)ex->d_fast_tiles.p,         \
                       (const uint8_t *)pyr, ex->pyr_frame, (int32_t *)ex->d_cellcnt.p, ex->total_cells, (uint32_t *)ex->d_cellent.p, \
                       ex->cand_frame, ini, mn, ex->fast_wave_rows, ex->fast_wave_qfull, (const uint32_t *)ovf_list,                  \
                       (const int32_t *)ovf_count, src0)
            if (ex->fast_wave_pitch == 48) ORBX_FAST_WAVE_LIST(48); else ORBX_FAST_WAVE_LIST(64);
#undef ORBX_FAST_WAVE_LIST
        } else {
            // a level with cells wider than 57 or higher than 63 pixels (levels under ~100 px): the generic workgroup-per-cell kernel for all
            hipLaunchKernelGGL(k_fast_cells<64>, dim3(ex->n_fast_tiles, n), dim3(64), ex->fast_lds, st, d_lv, (const TileRef *)ex->d_fast_tiles.p,
                               (const uint8_t *)pyr, ex->pyr_frame, (int32_t *)ex->d_cellcnt.p, ex->total_cells, (uint32_t *)ex->d_cellent.p,
                               ex->cand_frame, ini, mn);
        }
    }
    {
        ProfScope ps(ex, K_OCTREE);
        if (ex->oct_par) {
            // vToDistributeKeys of every (frame, level) is gathered by the first tier itself (compact_level): as a launch of its own (k_compact, rounds 1-3)
            // it took 24 us + a launch boundary on the main stream's latency-bound stretch, inside the tier 13 us (profiles/r03_am)
            const size_t lds = oct_par_lds_bytes(ex->max_pool), lds1 = oct_par_pool_bytes(ex->max_pool);
            const size_t lds_s = oct_par_pool_bytes(ex->max_pool) + (size_t)kOctTier1Keys * 6;
            const size_t lds_r = std::max(lds, lds1);
            if (lds_r > 64 * 1024) ORBX_HIP(hipFuncSetAttribute((const void *)k_octree_rest, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
            if (lds_s > 64 * 1024) ORBX_HIP(hipFuncSetAttribute((const void *)k_octree_par_t<kOctTier1Keys, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
            // the 256-thread form for levels with at most 2048 candidates (every level of the EuRoC-shaped bench), then ONE launch for everything else:
            // 2049 .. 4096 candidates in the 256-thread form with the larger key buffers, the rest in the single-wave chunked form (k_octree_rest)
            hipLaunchKernelGGL((k_octree_par_t<kOctTier1Keys, -1>), dim3(n, nl), dim3(256), lds_s, st, d_lv, ex->cand_frame, (uint32_t *)ex->d_keys1.p,
                               (uint32_t *)ex->d_lvlkp.p, ex->lvl_frame, (int32_t *)ex->d_lvlcnt.p, nl, (int32_t *)ex->d_candtot.p,
                               (int32_t *)ex->d_err.p, ex->max_pool, (const int32_t *)ex->d_cellcnt.p,
                               ex->total_cells, (const uint32_t *)ex->d_cellent.p);
            hipLaunchKernelGGL(k_octree_rest, dim3(n, nl), dim3(256), lds_r, st, d_lv, ex->cand_frame, (uint32_t *)ex->d_keys0.p,
                               (uint32_t *)ex->d_keys1.p, (uint16_t *)ex->d_nof0.p, (uint16_t *)ex->d_nof1.p, (uint32_t *)ex->d_lvlkp.p,
                               ex->lvl_frame, (int32_t *)ex->d_lvlcnt.p, nl, (const int32_t *)ex->d_candtot.p, (int32_t *)ex->d_err.p,
                               ex->max_pool, ex->oct_single_wave ? 1 : 0);
        } else {
            if (oct_lds_bytes(ex->max_pool) > 64 * 1024)
                ORBX_HIP(hipFuncSetAttribute((const void *)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oct_lds_bytes(ex->max_pool)));
            hipLaunchKernelGGL(k_octree, dim3(nl, n), dim3(64), oct_lds_bytes(ex->max_pool), st, d_lv,
                               (const int32_t *)ex->d_cellcnt.p, ex->total_cells, (const uint32_t *)ex->d_cellent.p, ex->cand_frame,
                               (uint32_t *)ex->d_keys0.p, (uint32_t *)ex->d_keys1.p, (uint32_t *)ex->d_lvlkp.p, ex->lvl_frame,
                               (int32_t *)ex->d_lvlcnt.p, nl, (int32_t *)ex->d_candtot.p, (int32_t *)ex->d_err.p, ex->max_pool);
        }
    }
    if (ex->copy_pending) {  // outputs of the previous batch may still be in flight to the host
        ORBX_HIP(hipStreamWaitEvent(st, ex->ev_copy_done[(ex->copy_issued - 1) & 1], 0));  // the most recent download
    }
    if (ex->match_pending && !(ex->copy_pending && ex->copy_covers_match)) {  // the previous batch's matcher still reads its counts / keypoints / descriptors
                                                                              // (a download issued behind it on the copy stream has waited for it already: one barrier less)
        ORBX_HIP(hipStreamWaitEvent(st, ex->ev_match, 0));
    }
    {
        ProfScope ps(ex, K_FINALIZE);
        hipLaunchKernelGGL(k_finalize, dim3(n), dim3(256), 0, st, d_lv, nl, (const uint32_t *)ex->d_lvlkp.p, ex->lvl_frame,
                           (const int32_t *)ex->d_lvlcnt.p, (WorkItem *)ex->d_work.p, ex->cap, (int32_t *)ex->d_count.p,
                           (int32_t *)ex->d_mono.p, lap0, lap1, (int32_t *)ex->d_err.p);
    }
    if (!ex->fused_blur && !ex->profile && ex->side_streams) ORBX_HIP(hipStreamWaitEvent(st, ex->ev_blur, 0));
    {
        ProfScope ps(ex, K_DESCRIBE);
        const HostMirror hm = mirror && n == 1 ? *mirror : HostMirror{nullptr, nullptr, nullptr, nullptr, nullptr};
        const int strict = fp_mode_of(ex->prm.flags);
        const dim3 grid = xcd_grid((ex->cap + 7) / 8, n);
        if (!ex->fused_blur)
            hipLaunchKernelGGL(k_describe, grid, dim3(256), 0, st, (const DescConst *)ex->d_dc.p, (const WorkItem *)ex->d_work.p,
                               (const int32_t *)ex->d_count.p, ex->cap, (const uint8_t *)pyr, ex->pyr_frame, (const uint8_t *)blur_slab,
                               ex->blur_frame, (orbx_keypoint *)ex->d_kps.p, (uint8_t *)ex->d_desc.p, strict, n, hm);
        else if (bt.sat)   // the blur on demand, around the keypoints only (the blur slab is filled by orbx_debug_level_blurred alone)
            hipLaunchKernelGGL(k_describe_fused<true>, grid, dim3(256), 0, st, (const DescConst *)ex->d_dc.p, (const WorkItem *)ex->d_work.p,
                               (const int32_t *)ex->d_count.p, ex->cap, (const uint8_t *)pyr, ex->pyr_frame, bt.g[0], bt.g[1], bt.g[2], bt.g[3],
                               (orbx_keypoint *)ex->d_kps.p, (uint8_t *)ex->d_desc.p, strict, n, hm, src0, ex->width, ex->height, (uint8_t *)nullptr, -1);
        else
            hipLaunchKernelGGL(k_describe_fused<false>, grid, dim3(256), 0, st, (const DescConst *)ex->d_dc.p, (const WorkItem *)ex->d_work.p,
                               (const int32_t *)ex->d_count.p, ex->cap, (const uint8_t *)pyr, ex->pyr_frame, bt.g[0], bt.g[1], bt.g[2], bt.g[3],
                               (orbx_keypoint *)ex->d_kps.p, (uint8_t *)ex->d_desc.p, strict, n, hm, src0, ex->width, ex->height, (uint8_t *)nullptr, -1);
    }
    if (ev_input_consumed && inplace0) ORBX_HIP(hipEventRecord(ev_input_consumed, st));   // in place: the descriptor stage is the frames' last reader
    if (ex->has_camera) {   // Frame::UndistortKeyPoints for the whole batch (mvKeysUn for the batched matchers)
        CameraModel c = {ex->cam_params[0], ex->cam_params[1], ex->cam_params[2], ex->cam_params[3], ex->cam_params[4], ex->cam_params[5],
                         ex->cam_params[6], ex->cam_params[7], ex->cam_params[8]};
        hipLaunchKernelGGL(k_undistort, dim3((ex->cap + 255) / 256, n), dim3(256), 0, st, c, (const orbx_keypoint *)ex->d_kps.p,
                           (const int32_t *)ex->d_count.p, ex->cap, (orbx_keypoint *)ex->d_kps_un.p);
    }
    ORBX_HIP(hipEventRecord(ex->ev_describe, st));
    ORBX_HIP(hipGetLastError());
    ex->last_batch = n;
    return ORBX_OK;
}

// The asynchronous entry points hand the caller's host pointers straight to the copy engine, so they insist on pinned memory
// (hipHostMalloc / hipHostRegister, e.g. torch's pin_memory()): pageable memory would be pinned page by page by the runtime behind
// the caller's back and the copy would not be asynchronous at all.
static bool is_pinned_host(const void *p) {
    if (!p) return true;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

static int check_device_error(orbx_extractor *ex) {
    int r0 = ex->ensure_stage(64);
    if (r0 != ORBX_OK) return r0;
    ORBX_HIP(hipMemcpyAsync(ex->h_stage, ex->d_err.p, sizeof(int32_t), hipMemcpyDeviceToHost, ex->stream));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    int32_t e = 0;
    memcpy(&e, ex->h_stage, sizeof(e));
    if (e != 0) {
        set_error("device-side consistency check failed, code " + std::to_string(e));
        (void)hipMemsetAsync(ex->d_err.p, 0, sizeof(int32_t), ex->stream);
        return ORBX_E_INTERNAL;
    }
    return ORBX_OK;
}

}  // namespace orbx

extern "C" {

const char *orbx_last_error(void) { return g_last_error.c_str(); }

const char *orbx_status_string(int s) {
    switch (s) {
        case ORBX_OK: return "ok";
        case ORBX_E_EMPTY: return "empty image";
        case ORBX_E_BAD_ARG: return "bad argument";
        case ORBX_E_TOO_SMALL: return "image too small for the pyramid";
        case ORBX_E_CAPACITY: return "output capacity too small";
        case ORBX_E_NO_DEVICE: return "no usable HIP device";
        case ORBX_E_HIP: return "HIP runtime error";
        case ORBX_E_TOO_LARGE: return "image or batch too large";
        case ORBX_E_INTERNAL: return "internal device-side check failed";
        case ORBX_E_STALE: return "level 0 requested after the caller's frames were released";
        default: return s > 0 ? "ok" : "unknown error";
    }
}

int orbx_create(const orbx_params *p, int device, int max_width, int max_height, int max_batch, orbx_extractor **out) {
    if (!p || !out) return ORBX_E_BAD_ARG;
    *out = nullptr;
    if (p->nlevels < 1 || p->nlevels > kMaxLevels || p->nfeatures < 1 || !(p->scale_factor > 1.0f)) return ORBX_E_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        set_error("no usable HIP device (liborbx has no CPU fallback)");
        return ORBX_E_NO_DEVICE;
    }
    ORBX_HIP(hipSetDevice(device));
    orbx_extractor *ex = new orbx_extractor();
    ex->prm = *p;
    ex->device = device;
    const int nl = p->nlevels;
    // ---- ORBextractor ctor :414-445 ----
    const double scaleFactor = p->scale_factor;  // the member is a double (ORBextractor.h:94)
    ex->scale.resize(nl); ex->sigma2.resize(nl); ex->inv_scale.resize(nl); ex->inv_sigma2.resize(nl); ex->quota.resize(nl);
    ex->scale[0] = 1.0f; ex->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        ex->scale[i] = (float)(ex->scale[i - 1] * scaleFactor);
        ex->sigma2[i] = ex->scale[i] * ex->scale[i];
    }
    for (int i = 0; i < nl; i++) {
        ex->inv_scale[i] = 1.0f / ex->scale[i];
        ex->inv_sigma2[i] = 1.0f / ex->sigma2[i];
    }
    const float factor = (float)(1.0f / scaleFactor);
    float desired = p->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        ex->quota[l] = cv_round_f(desired);
        sum += ex->quota[l];
        desired *= factor;
    }
    ex->quota[nl - 1] = std::max(p->nfeatures - sum, 0);
    // ---- umax :453-468 ----
    {
        int v, v0;
        const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
        const int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
        const double hp2 = kHalfPatch * kHalfPatch;
        for (v = 0; v < 16; v++) ex->umax[v] = 0;
        for (v = 0; v <= vmax; ++v) ex->umax[v] = cv_round_d(std::sqrt(hp2 - v * v));
        for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
            while (ex->umax[v0] == ex->umax[v0 + 1]) ++v0;
            ex->umax[v] = v0;
            ++v0;
        }
    }
    // all streams at the default priority (priorities measured: no gain for one extractor, a 20 % loss when two share the device)
    const int prio_lo = 0, prio_hi = 0;
    hipError_t e = hipStreamCreateWithPriority(&ex->stream, hipStreamNonBlocking, prio_hi);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); delete ex; return ORBX_E_HIP; }
    (void)hipEventCreate(&ex->ev0);
    (void)hipEventCreate(&ex->ev1);
    (void)hipStreamCreateWithPriority(&ex->copy_stream, hipStreamNonBlocking, prio_lo);
    { const char *v = getenv("ORBX_FAST_QCAP"); if (v && atoi(v) >= 16) ex->strip_qcap = atoi(v) & ~15; }  // test hook: k_fast_strip's pixel queues overflow, every cell takes the list pass
    (void)hipStreamCreateWithPriority(&ex->aux_stream, hipStreamNonBlocking, prio_hi);
    (void)hipStreamCreateWithPriority(&ex->match_stream, hipStreamNonBlocking, prio_lo);
    (void)hipStreamCreateWithPriority(&ex->in_stream, hipStreamNonBlocking, prio_hi);
    // A sixth, unused stream.  The HIP runtime deals streams onto its GPU_MAX_HW_QUEUES (4) hardware queues round-robin in creation order, and
    // streams that share a queue serialize.  With five streams per extractor the second extractor of a stereo rig starts one queue further on:
    // its main stream shares a queue with the first one's copy stream, its aux stream with the first one's matcher -- KITTI stereo 1.85 ms per
    // step; with six (the count the library had through round 2) the second extractor starts two queues on and the step is 1.67 ms (seven:
    // 1.71, eight: 1.70; one extractor per process: no difference).  profiles/r03_l_kitti_stream_mapping.log, r03_n_ab_spare_streams.log
    (void)hipStreamCreateWithFlags(&ex->spare_stream, hipStreamNonBlocking);
    for (hipEvent_t *ev : {&ex->ev_in_free[0], &ex->ev_in_free[1], &ex->ev_in_ready[0], &ex->ev_in_ready[1]}) (void)hipEventCreateWithFlags(ev, hipEventDisableTiming);
    for (hipEvent_t *ev : {&ex->ev_pyr, &ex->ev_blur, &ex->ev_describe, &ex->ev_match}) (void)hipEventCreateWithFlags(ev, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&ex->ev_copy_done[0], hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&ex->ev_copy_done[1], hipEventDisableTiming);
    (void)hipHostMalloc((void **)&ex->h_err, 2 * sizeof(int32_t), hipHostMallocDefault);
    if (ex->h_err) { ex->h_err[0] = 0; ex->h_err[1] = 0; }
    // descriptor constants: orientation disc offsets + BRIEF pattern
    DescConst dc;
    memset(&dc, 0, sizeof(dc));
    int n = 0;
    for (int u = 0; u <= kHalfPatch; u++) {   // column form of the disc {(u, v) : |u| <= umax[|v|]} (:82-100)
        int vm = -1;
        for (int v = 0; v <= kHalfPatch; v++) if (ex->umax[v] >= u) vm = v;
        dc.vmax_of_u[u] = (int8_t)vm;
        if (vm >= 0) n += (u == 0 ? 1 : 2) * (2 * vm + 1);
    }
    for (int r = 0; r < 31; r++)   // row form for k_describe_fused (row 31 stays empty)
        for (int ax = 0; ax < 4; ax++)
            for (int t = 0; t < 40; t++) {
                const int v = r - kHalfPatch, u = t - 18 - ax;
                if (std::abs(u) <= ex->umax[std::abs(v)]) dc.ic_mask[r][ax][t >> 2] |= 0xffu << (8 * (t & 3));
            }
    // the column form equals the row form of :82-100 only for a monotone umax; 749 = pixel count of the reference's disc
    bool mono = true;
    for (int v = 0; v < kHalfPatch; v++) mono = mono && ex->umax[v] >= ex->umax[v + 1];
    if (n != 749 || !mono) { set_error("orientation disc has " + std::to_string(n) + " pixels"); orbx_destroy(ex); return ORBX_E_INTERNAL; }
    memcpy(dc.pat, kPatternData, 1024);
    for (int i = 0; i < 1024; i++) dc.patf[i >> 2][i & 3] = (float)kPatternData[i];
    int r = ex->d_dc.ensure(sizeof(DescConst));
    if (r != ORBX_OK) { orbx_destroy(ex); return r; }
    if (hipMemcpy(ex->d_dc.p, &dc, sizeof(dc), hipMemcpyHostToDevice) != hipSuccess) { orbx_destroy(ex); return ORBX_E_HIP; }
    if (max_width > 0 && max_height > 0) {
        r = configure(ex, max_width, max_height, std::max(max_batch, 1));
        if (r != ORBX_OK) { orbx_destroy(ex); return r; }
    }
    *out = ex;
    return ORBX_OK;
}

void orbx_destroy(orbx_extractor *ex) {
    if (!ex) return;
    (void)hipSetDevice(ex->device);
    if (ex->stream) (void)hipStreamSynchronize(ex->stream);
    for (hipStream_t s : {ex->copy_stream, ex->aux_stream, ex->match_stream, ex->in_stream, ex->spare_stream})
        if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipEvent_t ev : {ex->ev_in_free[0], ex->ev_in_free[1], ex->ev_in_ready[0], ex->ev_in_ready[1]}) if (ev) (void)hipEventDestroy(ev);
    ex->d_in[0].release(); ex->d_in[1].release();
    for (hipEvent_t ev : {ex->ev_pyr, ex->ev_blur, ex->ev_describe, ex->ev_match}) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ex->ev_stereo_copy) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ex->ev_copy_done) if (ev) (void)hipEventDestroy(ev);
    if (ex->h_err) (void)hipHostFree(ex->h_err);
    for (int i = 0; i < 3; i++) { if (ex->h_frustum[i]) (void)hipHostFree(ex->h_frustum[i]); if (ex->ev_frustum[i]) (void)hipEventDestroy(ex->ev_frustum[i]); }
    ex->d_match.release(); ex->d_nmatch.release();
    for (DevBuf *b : {&ex->d_st_bidx, &ex->d_st_bdist, &ex->d_st_ur, &ex->d_st_depth, &ex->d_st_sad, &ex->d_st_nm, &ex->d_st_scales, &ex->d_st_rowptr, &ex->d_st_rowidx}) b->release();
    DevBuf *bufs[] = {&ex->d_lv, &ex->d_xtab, &ex->d_ytab, &ex->d_fast_tiles, &ex->d_blur_items, &ex->d_dc, &ex->d_pyr, &ex->d_pyr2,
                      &ex->d_blur, &ex->d_cellcnt, &ex->d_cellent, &ex->d_keys0, &ex->d_keys1, &ex->d_nof0, &ex->d_nof1, &ex->d_fast_ovf, &ex->d_lvlkp, &ex->d_lvlcnt,
                      &ex->d_candtot, &ex->d_work, &ex->d_kps, &ex->d_desc, &ex->d_count, &ex->d_mono, &ex->d_err,
                      &ex->d_mkey1, &ex->d_mkey2, &ex->d_mocc, &ex->d_mentries, &ex->d_mprobs, &ex->d_mres, &ex->d_mscale, &ex->d_mgrid, &ex->d_xgtab,
                      &ex->d_mp_qr, &ex->d_mp_qmin, &ex->d_mp_qmax, &ex->d_mp_valid, &ex->d_mp_keys, &ex->d_mp_meta, &ex->d_mp_grid, &ex->d_mp_probs,
                      &ex->d_mp_res, &ex->d_mp_misc, &ex->d_mp_entries, &ex->d_kps_un, &ex->d_frustum_frames, &ex->d_strips,
                      &ex->ps_plan[0].cols, &ex->ps_plan[0].steps, &ex->ps_plan[0].tasks, &ex->ps_plan[0].band0, &ex->ps_plan[1].cols, &ex->ps_plan[1].steps,
                      &ex->ps_plan[1].tasks, &ex->ps_plan[1].band0, &ex->ps_plan[2].cols, &ex->ps_plan[2].steps, &ex->ps_plan[2].tasks, &ex->ps_plan[2].band0,
                      &ex->ps_plan[3].cols, &ex->ps_plan[3].steps, &ex->ps_plan[3].tasks, &ex->ps_plan[3].band0};
    for (DevBuf *b : bufs) b->release();
    if (ex->h_stage) (void)hipHostFree(ex->h_stage);
    if (ex->ev0) (void)hipEventDestroy(ex->ev0);
    if (ex->ev1) (void)hipEventDestroy(ex->ev1);
    if (ex->stream) (void)hipStreamDestroy(ex->stream);
    delete ex;
}

int orbx_extract_batch_device(orbx_extractor *ex, const uint8_t *d_images, int n_frames, int width, int height,
                              size_t row_stride, size_t frame_stride, int lap0, int lap1) {
    if (!ex) return ORBX_E_BAD_ARG;
    if (!d_images || width <= 0 || height <= 0 || n_frames <= 0) return ORBX_E_EMPTY;
    if (row_stride < (size_t)width) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    int r = configure(ex, width, height, n_frames);
    if (r != ORBX_OK) return r;
    return enqueue_extract(ex, d_images, n_frames, row_stride, frame_stride, lap0, lap1);
}

int orbx_extract_batch_host(orbx_extractor *ex, const uint8_t *h_images, int n_frames, int width, int height,
                            size_t row_stride, size_t frame_stride, int lap0, int lap1) {
    if (!ex) return ORBX_E_BAD_ARG;
    if (!h_images || width <= 0 || height <= 0 || n_frames <= 0) return ORBX_E_EMPTY;
    if (row_stride < (size_t)width || frame_stride < row_stride * (size_t)height) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    if (!is_pinned_host(h_images)) { set_error("orbx_extract_batch_host needs pinned frames (hipHostMalloc / hipHostRegister)"); return ORBX_E_BAD_ARG; }
    int r = configure(ex, width, height, n_frames);
    if (r != ORBX_OK) return r;
    const unsigned slot = ex->in_issued & 1u;
    // the contract of orbx.h: the PREVIOUS call's frames may be overwritten once this call has returned -- so its upload must have
    // landed (it was queued behind ev_in_free of its slab and may still be pending)
    if (ex->in_used[slot ^ 1u]) ORBX_HIP(hipEventSynchronize(ex->ev_in_ready[slot ^ 1u]));
    orbx::DevBuf &din = ex->d_in[slot];
    const size_t fbytes = (size_t)width * height, need = fbytes * n_frames;
    if (need > din.bytes) {  // growing: nothing may still read the old slab
        ORBX_HIP(hipStreamSynchronize(ex->stream));
        ORBX_HIP(hipStreamSynchronize(ex->in_stream));
        if ((r = din.ensure(need)) != ORBX_OK) return r;
        ex->in_used[slot] = false;
    }
    hipStream_t is = ex->in_stream;
    if (ex->in_used[slot]) ORBX_HIP(hipStreamWaitEvent(is, ex->ev_in_free[slot], 0));  // the batch that last used this slab
    if (frame_stride == row_stride * (size_t)height) {  // rows of all frames equally spaced: one (2-D) copy into the packed slab
        if (row_stride == (size_t)width) ORBX_HIP(hipMemcpyAsync(din.p, h_images, need, hipMemcpyHostToDevice, is));
        else ORBX_HIP(hipMemcpy2DAsync(din.p, width, h_images, row_stride, width, (size_t)height * n_frames, hipMemcpyHostToDevice, is));
    } else {
        for (int f = 0; f < n_frames; f++)
            ORBX_HIP(hipMemcpy2DAsync((uint8_t *)din.p + f * fbytes, width, h_images + f * frame_stride, row_stride, width, height,
                                      hipMemcpyHostToDevice, is));
    }
    ORBX_HIP(hipEventRecord(ex->ev_in_ready[slot], is));
    ORBX_HIP(hipStreamWaitEvent(ex->stream, ex->ev_in_ready[slot], 0));   // the stream of k_pyr_base
    ex->in_used[slot] = true;
    ex->in_issued++;
    return enqueue_extract(ex, (const uint8_t *)din.p, n_frames, width, fbytes, lap0, lap1, ex->ev_in_free[slot]);
}

int orbx_set_camera(orbx_extractor *ex, const orbx_camera *cam) {
    if (!ex) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    for (hipStream_t s : {ex->copy_stream, ex->aux_stream, ex->match_stream, ex->in_stream}) if (s) ORBX_HIP(hipStreamSynchronize(s));
    ex->has_camera = cam != nullptr;
    if (cam) {
        const float p[9] = {cam->fx, cam->fy, cam->cx, cam->cy, cam->k1, cam->k2, cam->p1, cam->p2, cam->k3};
        memcpy(ex->cam_params, p, sizeof(p));
        ex->cam_bf = cam->bf;
    }
    // re-derive the workspace (d_kps_un) and the image bounds for the current geometry
    const int w = ex->width, h = ex->height, b = ex->batch_cap;
    ex->width = 0; ex->height = 0;
    ex->mkey = orbx_extractor::MatchKey(); ex->mpkey = orbx_extractor::MpKey();
    if (w > 0 && h > 0) return configure(ex, w, h, std::max(b, 1));
    return ORBX_OK;
}

int orbx_batch_download_keypoints_un(orbx_extractor *ex, int frame, orbx_keypoint *kps_un, int cap, int *n_out) {
    if (!ex || frame < 0 || frame >= ex->last_batch) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    int r = ex->d2h_staged_begin(64 + sizeof(orbx_keypoint) * (size_t)ex->cap);
    if (r != ORBX_OK) return r;
    if ((r = ex->d2h_staged(0, (int32_t *)ex->d_count.p + frame, 4)) != ORBX_OK) return r;
    if ((r = ex->d2h_staged(64, (const orbx_keypoint *)ex->match_kps() + (size_t)frame * ex->cap, sizeof(orbx_keypoint) * (size_t)ex->cap)) != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    int32_t n = 0;
    memcpy(&n, ex->staged(0), 4);
    if (n_out) *n_out = n;
    if (n > cap) return ORBX_E_CAPACITY;
    if (n > 0 && kps_un) memcpy(kps_un, ex->staged(64), sizeof(orbx_keypoint) * (size_t)n);
    return ORBX_OK;
}

int orbx_output_capacity(orbx_extractor *ex, int width, int height) {
    if (!ex) return ORBX_E_BAD_ARG;
    if (hipSetDevice(ex->device) != hipSuccess) return ORBX_E_HIP;
    int r = configure(ex, width, height, std::max(ex->batch_cap, 1));
    if (r != ORBX_OK) return r;
    return ex->cap;
}

int orbx_sync(orbx_extractor *ex) {
    if (!ex) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    ORBX_HIP(hipStreamSynchronize(ex->aux_stream));
    ORBX_HIP(hipStreamSynchronize(ex->match_stream));
    ORBX_HIP(hipStreamSynchronize(ex->in_stream));
    if (ex->lvl0_inplace) ex->in0_released = true;   // the caller may overwrite the frames from here on (include/orbx.h)
    return ORBX_OK;
}

int orbx_batch_view_get(orbx_extractor *ex, orbx_batch_view *v) {
    if (!ex || !v) return ORBX_E_BAD_ARG;
    v->n_frames = ex->last_batch;
    v->cap = ex->cap;
    v->d_keypoints = (const orbx_keypoint *)ex->d_kps.p;
    v->d_descriptors = (const uint8_t *)ex->d_desc.p;
    v->d_count = (const int32_t *)ex->d_count.p;
    v->d_mono_index = (const int32_t *)ex->d_mono.p;
    v->d_keypoints_un = (const orbx_keypoint *)ex->match_kps();
    return ORBX_OK;
}

int orbx_batch_download(orbx_extractor *ex, int frame, orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out, int *mono) {
    if (!ex || frame < 0 || frame >= ex->last_batch) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    int r = check_device_error(ex);
    if (r != ORBX_OK) return r;
    if ((r = ex->d2h_staged_begin(64 + (size_t)ex->cap * (sizeof(orbx_keypoint) + 32))) != ORBX_OK) return r;
    if ((r = ex->d2h_staged(0, (int32_t *)ex->d_count.p + frame, 4)) != ORBX_OK) return r;
    if ((r = ex->d2h_staged(4, (int32_t *)ex->d_mono.p + frame, 4)) != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    int32_t cm[2];
    memcpy(cm, ex->staged(0), 8);
    if (n_out) *n_out = cm[0];
    if (mono) *mono = cm[1];
    if (cm[0] > cap) return ORBX_E_CAPACITY;
    if (cm[0] > 0) {
        const size_t kb = sizeof(orbx_keypoint) * (size_t)cm[0], db = (size_t)32 * cm[0], off_d = 64 + (size_t)ex->cap * sizeof(orbx_keypoint);
        if (kps && (r = ex->d2h_staged(64, (orbx_keypoint *)ex->d_kps.p + (size_t)frame * ex->cap, kb)) != ORBX_OK) return r;
        if (desc && (r = ex->d2h_staged(off_d, (uint8_t *)ex->d_desc.p + (size_t)frame * ex->cap * 32, db)) != ORBX_OK) return r;
        ORBX_HIP(hipStreamSynchronize(ex->stream));
        if (kps) memcpy(kps, ex->staged(64), kb);
        if (desc) memcpy(desc, ex->staged(off_d), db);
    }
    return ORBX_OK;
}

int orbx_batch_download_all(orbx_extractor *ex, orbx_keypoint *kps, uint8_t *desc, int32_t *counts, int32_t *mono) {
    if (!ex || ex->last_batch <= 0) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    const size_t n = (size_t)ex->last_batch;
    const size_t cb = 4 * n, kb = sizeof(orbx_keypoint) * (size_t)ex->cap * n, db = (size_t)32 * ex->cap * n;
    const size_t o_m = (cb + 63) & ~(size_t)63, o_k = 2 * o_m, o_d = o_k + ((kb + 63) & ~(size_t)63);
    int r = check_device_error(ex);
    if (r != ORBX_OK) return r;
    if ((r = ex->d2h_staged_begin(o_d + db)) != ORBX_OK) return r;
    if (counts && (r = ex->d2h_staged(0, ex->d_count.p, cb)) != ORBX_OK) return r;
    if (mono && (r = ex->d2h_staged(o_m, ex->d_mono.p, cb)) != ORBX_OK) return r;
    if (kps && (r = ex->d2h_staged(o_k, ex->d_kps.p, kb)) != ORBX_OK) return r;
    if (desc && (r = ex->d2h_staged(o_d, ex->d_desc.p, db)) != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    if (counts) memcpy(counts, ex->staged(0), cb);
    if (mono) memcpy(mono, ex->staged(o_m), cb);
    if (kps) memcpy(kps, ex->staged(o_k), kb);
    if (desc) memcpy(desc, ex->staged(o_d), db);
    return ORBX_OK;
}

int orbx_batch_download_async(orbx_extractor *ex, orbx_keypoint *kps, uint8_t *desc, int32_t *counts, int32_t *mono,
                              int32_t *match, int32_t *nmatches) {
    RoctxRange rr("orbx:download");
    if (!ex || ex->last_batch <= 0) return ORBX_E_BAD_ARG;
    if (ex->copy_issued - ex->copy_waited >= 2) { set_error("two downloads already in flight: call orbx_download_wait first"); return ORBX_E_BAD_ARG; }
    for (const void *p : {(const void *)kps, (const void *)desc, (const void *)counts, (const void *)mono, (const void *)match, (const void *)nmatches})
        if (!is_pinned_host(p)) { set_error("orbx_batch_download_async needs pinned host buffers (hipHostMalloc / hipHostRegister)"); return ORBX_E_BAD_ARG; }
    ORBX_HIP(hipSetDevice(ex->device));
    const int n = ex->last_batch;
    hipStream_t cs = ex->copy_stream;
    // behind the extraction itself (ev_describe), not behind whatever else the main stream has been given since: the stereo kernels of
    // orbx_stereo_batch_device run there, and the keypoints / descriptors (94 % of the bytes) need not wait for them
    ORBX_HIP(hipStreamWaitEvent(cs, ex->ev_describe, 0));
    // keypoints and descriptors leave as soon as the extraction is done, BESIDE the matcher (they are 94 % of the bytes and the matcher
    // does not write them); only the match vectors wait for it.  With the wait in front of everything the next batch's k_finalize sat
    // behind matcher + all copies in series.
    // (Round 3: the copies run as blit kernels of the runtime; in the pipelined loop the kernels beside them stretch -- k_window_best2 76 -> 380 us,
    // the next k_pyr_base 54 -> 350 us -- but a copy kernel of our own on 4 .. 64 workgroups made the step SLOWER (1.28 - 1.35 vs 1.10 ms), forcing
    // SDMA changed nothing and an HBM-bound kernel beside such copies alone loses 1 - 5 %: profiles/r03_g_*, r03_h_*.  The stretch is the sharing
    // of a machine that the main stream's kernels already fill, not a property of the copy.)
    if (counts) ORBX_HIP(hipMemcpyAsync(counts, ex->d_count.p, 4 * (size_t)n, hipMemcpyDeviceToHost, cs));
    if (mono) ORBX_HIP(hipMemcpyAsync(mono, ex->d_mono.p, 4 * (size_t)n, hipMemcpyDeviceToHost, cs));
    if (kps) ORBX_HIP(hipMemcpyAsync(kps, ex->d_kps.p, sizeof(orbx_keypoint) * (size_t)ex->cap * n, hipMemcpyDeviceToHost, cs));
    if (desc) ORBX_HIP(hipMemcpyAsync(desc, ex->d_desc.p, (size_t)32 * ex->cap * n, hipMemcpyDeviceToHost, cs));
    if (ex->match_pending) ORBX_HIP(hipStreamWaitEvent(cs, ex->ev_match, 0));
    ex->copy_covers_match = ex->match_pending;
    if (match && ex->d_match.p) ORBX_HIP(hipMemcpyAsync(match, ex->d_match.p, 4 * (size_t)ex->cap * n, hipMemcpyDeviceToHost, cs));
    if (nmatches && ex->d_nmatch.p) ORBX_HIP(hipMemcpyAsync(nmatches, ex->d_nmatch.p, 4 * (size_t)n, hipMemcpyDeviceToHost, cs));
    const unsigned slot = ex->copy_issued & 1;
    ORBX_HIP(hipMemcpyAsync(ex->h_err + slot, ex->d_err.p, sizeof(int32_t), hipMemcpyDeviceToHost, cs));
    ORBX_HIP(hipEventRecord(ex->ev_copy_done[slot], cs));
    ex->copy_issued++;
    ex->copy_pending = true;
    return ORBX_OK;
}

// Waits for the OLDEST download still in flight (at most two are).
int orbx_download_wait(orbx_extractor *ex) {
    if (!ex) return ORBX_E_BAD_ARG;
    if (ex->copy_issued == ex->copy_waited) return ORBX_OK;
    ORBX_HIP(hipSetDevice(ex->device));
    const unsigned slot = ex->copy_waited & 1;
    ORBX_HIP(hipEventSynchronize(ex->ev_copy_done[slot]));
    ex->copy_waited++;
    if (ex->lvl0_inplace && ex->copy_waited > ex->in0_copy_seq) ex->in0_released = true;   // a download issued after the in-place batch has completed: its frames are the caller's again
    if (ex->h_err[slot] != 0) {
        set_error("device-side consistency check failed, code " + std::to_string(ex->h_err[slot]));
        ex->h_err[slot] = 0;
        (void)hipMemsetAsync(ex->d_err.p, 0, sizeof(int32_t), ex->stream);
        return ORBX_E_INTERNAL;
    }
    return ORBX_OK;
}

int orbx_extract(orbx_extractor *ex, const uint8_t *image, int width, int height, size_t stride, int lap0, int lap1,
                 orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out, int *mono_index) {
    if (n_out) *n_out = 0;
    if (mono_index) *mono_index = 0;
    if (!ex) return ORBX_E_BAD_ARG;
    if (!image || width <= 0 || height <= 0) return ORBX_E_EMPTY;  // ORBextractor.cc:1090
    if (stride < (size_t)width) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    int r = configure(ex, width, height, 1);
    if (r != ORBX_OK) return r;
    const size_t dpitch = ((size_t)width + 63) & ~(size_t)63;
    // the caller's image goes into the pinned staging block (rows packed to the device pitch), never to the runtime directly, and k_pyr_base
    // reads that block over the link itself (6.7 -> 13 us for the kernel, but no copy command and no wait for it: 0.190 -> 0.180 ms per call);
    // the results come back into the same block, written by k_describe (HostMirror): ONE synchronisation ends the call
    const size_t img_bytes = (dpitch * height + 255) & ~(size_t)255, kp_bytes = (sizeof(orbx_keypoint) * (size_t)ex->cap + 63) & ~(size_t)63;
    if ((r = ex->ensure_stage(img_bytes + 64 + kp_bytes + (size_t)32 * ex->cap)) != ORBX_OK) return r;
    for (int y = 0; y < height; y++) memcpy((uint8_t *)ex->h_stage + (size_t)y * dpitch, image + (size_t)y * stride, (size_t)width);
    uint8_t *hb = (uint8_t *)ex->h_stage + img_bytes;
    const HostMirror hm = {(int32_t *)hb, (orbx_keypoint *)(hb + 64), hb + 64 + kp_bytes, (const int32_t *)ex->d_err.p, (const int32_t *)ex->d_mono.p};
    r = enqueue_extract(ex, (const uint8_t *)ex->h_stage, 1, dpitch, dpitch * height, lap0, lap1, nullptr, &hm);
    if (r != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    int32_t hdr[3];
    memcpy(hdr, hm.hdr, sizeof(hdr));
    if (hdr[0] != 0) {
        set_error("device-side consistency check failed, code " + std::to_string(hdr[0]));
        (void)hipMemsetAsync(ex->d_err.p, 0, sizeof(int32_t), ex->stream);
        return ORBX_E_INTERNAL;
    }
    if (n_out) *n_out = hdr[1];
    if (mono_index) *mono_index = hdr[2];
    if (hdr[1] > cap) return ORBX_E_CAPACITY;
    if (hdr[1] > 0) {
        if (kps) memcpy(kps, hm.kps, sizeof(orbx_keypoint) * (size_t)hdr[1]);
        if (desc) memcpy(desc, hm.desc, (size_t)32 * hdr[1]);
    }
    return ORBX_OK;
}

int orbx_level_size(orbx_extractor *ex, int width, int height, int level, int *w, int *h) {
    if (!ex || level < 0 || level >= ex->prm.nlevels) return ORBX_E_BAD_ARG;
    if (w) *w = cv_round_f((float)width * ex->inv_scale[level]);
    if (h) *h = cv_round_f((float)height * ex->inv_scale[level]);
    return ORBX_OK;
}

int orbx_get_level(orbx_extractor *ex, int frame, int level, uint8_t *dst, size_t dst_stride) {
    if (!ex || !dst || frame < 0 || frame >= ex->last_batch || level < 0 || level >= ex->prm.nlevels) return ORBX_E_BAD_ARG;
    const LevelInfo &L = ex->lv[level];
    if (dst_stride < (size_t)(L.w + 2 * kEdge)) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    if (level == 0) { int r0 = orbx_materialize_level0(ex); if (r0 != ORBX_OK) return r0; }
    const uint8_t *src = (const uint8_t *)ex->pyr_cur() + (size_t)frame * ex->pyr_frame + L.off;
    const size_t bytes = (size_t)L.pitch * (L.h + 2 * kEdge);
    int r = ex->d2h_staged_begin(bytes);
    if (r != ORBX_OK) return r;
    if ((r = ex->d2h_staged(0, src, bytes)) != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    for (int y = 0; y < L.h + 2 * kEdge; y++) memcpy(dst + (size_t)y * dst_stride, ex->staged((size_t)y * L.pitch + kRingX), (size_t)(L.w + 2 * kEdge));
    return ORBX_OK;
}

int orbx_get_level_device(orbx_extractor *ex, int frame, int level, const uint8_t **d_padded, size_t *pitch) {
    if (!ex || frame < 0 || frame >= ex->last_batch || level < 0 || level >= ex->prm.nlevels) return ORBX_E_BAD_ARG;
    const LevelInfo &L = ex->lv[level];
    if (level == 0 && ex->lvl0_inplace) {   // not a pure getter then: the level is written now -- complete before the pointer is handed out
        int r0 = orbx_materialize_level0(ex);
        if (r0 != ORBX_OK) return r0;
        ORBX_HIP(hipStreamSynchronize(ex->stream));
    }
    if (d_padded) *d_padded = (const uint8_t *)ex->pyr_cur() + (size_t)frame * ex->pyr_frame + L.off + kRingX;
    if (pitch) *pitch = L.pitch;
    return ORBX_OK;
}

int orbx_get_levels(const orbx_extractor *ex) { return ex ? ex->prm.nlevels : ORBX_E_BAD_ARG; }
float orbx_get_scale_factor(const orbx_extractor *ex) { return ex ? (float)(double)ex->prm.scale_factor : 0.f; }
int orbx_get_scale_tables(const orbx_extractor *ex, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2) {
    if (!ex) return ORBX_E_BAD_ARG;
    const size_t b = sizeof(float) * ex->prm.nlevels;
    if (scale) memcpy(scale, ex->scale.data(), b);
    if (inv_scale) memcpy(inv_scale, ex->inv_scale.data(), b);
    if (sigma2) memcpy(sigma2, ex->sigma2.data(), b);
    if (inv_sigma2) memcpy(inv_sigma2, ex->inv_sigma2.data(), b);
    return ex->prm.nlevels;
}
int orbx_get_feature_tables(const orbx_extractor *ex, int32_t *quota, int32_t *umax16) {
    if (!ex) return ORBX_E_BAD_ARG;
    if (quota) memcpy(quota, ex->quota.data(), sizeof(int32_t) * ex->prm.nlevels);
    if (umax16) memcpy(umax16, ex->umax, sizeof(ex->umax));
    return ex->prm.nlevels;
}

int orbx_debug_level_candidates(orbx_extractor *ex, int frame, int level, orbx_keypoint *out, int cap) {
    if (!ex || frame < 0 || frame >= ex->last_batch || level < 0 || level >= ex->prm.nlevels) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    const LevelInfo &L = ex->lv[level];
    const int ncell = L.nCols * L.nRows;
    const size_t cb = 4 * (size_t)ncell, eb = 4 * (size_t)L.cand_cap, eoff = (cb + 63) & ~(size_t)63;
    int r = ex->d2h_staged_begin(eoff + eb);
    if (r != ORBX_OK) return r;
    if ((r = ex->d2h_staged(0, (int32_t *)ex->d_cellcnt.p + (size_t)frame * ex->total_cells + L.cell_base, cb)) != ORBX_OK) return r;
    if ((r = ex->d2h_staged(eoff, (uint32_t *)ex->d_cellent.p + (size_t)frame * ex->cand_frame + L.cand_off, eb)) != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    const int32_t *cnt = (const int32_t *)ex->staged(0);
    const uint32_t *ent = (const uint32_t *)ex->staged(eoff);
    int n = 0;
    for (int c = 0; c < ncell; c++)
        for (int k = 0; k < cnt[c]; k++) {
            const uint32_t key = ent[(size_t)c * L.cell_cap + k];
            if (out && n < cap) out[n] = orbx_keypoint{(float)key_x(key), (float)key_y(key), 7.f, -1.f, (float)key_s(key), 0, -1};
            n++;
        }
    return n;
}

int orbx_debug_level_keypoints(orbx_extractor *ex, int frame, int level, orbx_keypoint *out, int cap) {
    if (!ex || frame < 0 || frame >= ex->last_batch || level < 0 || level >= ex->prm.nlevels) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    const LevelInfo &L = ex->lv[level];
    int r = ex->d2h_staged_begin(64 + 4 * (size_t)L.lvl_cap);
    if (r != ORBX_OK) return r;
    if ((r = ex->d2h_staged(0, (int32_t *)ex->d_lvlcnt.p + (size_t)frame * ex->prm.nlevels + level, 4)) != ORBX_OK) return r;
    if ((r = ex->d2h_staged(64, (uint32_t *)ex->d_lvlkp.p + (size_t)frame * ex->lvl_frame + L.lvl_off, 4 * (size_t)L.lvl_cap)) != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    int32_t n = 0;
    memcpy(&n, ex->staged(0), 4);
    n = std::min(std::max(n, 0), L.lvl_cap);
    const uint32_t *keys = (const uint32_t *)ex->staged(64);
    for (int i = 0; i < n && i < cap && out; i++)
        out[i] = orbx_keypoint{(float)key_x(keys[i]), (float)key_y(keys[i]), L.size, -1.f, (float)key_s(keys[i]), level, -1};
    return n;
}

int orbx_debug_level_blurred(orbx_extractor *ex, int frame, int level, uint8_t *dst, size_t dst_stride) {
    if (!ex || !dst || frame < 0 || frame >= ex->last_batch || level < 0 || level >= ex->prm.nlevels) return ORBX_E_BAD_ARG;
    const LevelInfo &L = ex->lv[level];
    if (dst_stride < (size_t)L.w) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    const size_t bytes = (size_t)L.bpitch * L.h;
    int r = ex->d2h_staged_begin(bytes);
    if (r != ORBX_OK) return r;
    if (ex->fused_blur) {   // k_describe_fused blurs around the keypoints only: fill the blur slab of the last batch now
        if ((r = ex->d_blur.ensure(ex->blur_frame * (size_t)ex->batch_cap)) != ORBX_OK) return r;
        if ((r = orbx_materialize_level0(ex)) != ORBX_OK) return r;
        launch_blur_stream(ex, ex->last_batch, ex->pyr_cur(), ex->stream);
        ORBX_HIP(hipGetLastError());
    }
    if ((r = ex->d2h_staged(0, (const uint8_t *)ex->d_blur.p + (size_t)frame * ex->blur_frame + L.boff, bytes)) != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    for (int y = 0; y < L.h; y++) memcpy(dst + (size_t)y * dst_stride, ex->staged((size_t)y * L.bpitch), (size_t)L.w);
    return ORBX_OK;
}

int orbx_debug_fused_patches(orbx_extractor *ex, int frame, uint8_t *dst, int cap_keypoints) {
    if (!ex || !dst || frame < 0 || frame >= ex->last_batch || cap_keypoints <= 0) return ORBX_E_BAD_ARG;
    if (!ex->fused_blur) return ORBX_E_BAD_ARG;   // k_describe reads the blurred slab: orbx_debug_level_blurred
    ORBX_HIP(hipSetDevice(ex->device));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    DevBuf scratch;
    int r = scratch.ensure((size_t)ex->cap * 37 * 37);
    if (r != ORBX_OK) return r;
    // the descriptor kernel of the last batch once more, writing what it wrote before plus the blurred patches of `frame` (the frames the batch was
    // extracted from must still be valid when level 0 was read in place)
    const BlurTaps bt = blur_taps(ex);
    const int n = ex->last_batch, strict = fp_mode_of(ex->prm.flags);
    const Level0Src src0 = ex->lvl0_inplace ? Level0Src{ex->in0_images, ex->in0_row_stride, ex->in0_frame_stride} : Level0Src{nullptr, 0, 0};
    const HostMirror hm = HostMirror{nullptr, nullptr, nullptr, nullptr, nullptr};
    const dim3 grid = xcd_grid((ex->cap + 7) / 8, n);
    if (bt.sat)
        hipLaunchKernelGGL(k_describe_fused<true>, grid, dim3(256), 0, ex->stream, (const DescConst *)ex->d_dc.p, (const WorkItem *)ex->d_work.p, (const int32_t *)ex->d_count.p,
                           ex->cap, (const uint8_t *)ex->pyr_cur(), ex->pyr_frame, bt.g[0], bt.g[1], bt.g[2], bt.g[3], (orbx_keypoint *)ex->d_kps.p, (uint8_t *)ex->d_desc.p,
                           strict, n, hm, src0, ex->width, ex->height, (uint8_t *)scratch.p, frame);
    else
        hipLaunchKernelGGL(k_describe_fused<false>, grid, dim3(256), 0, ex->stream, (const DescConst *)ex->d_dc.p, (const WorkItem *)ex->d_work.p, (const int32_t *)ex->d_count.p,
                           ex->cap, (const uint8_t *)ex->pyr_cur(), ex->pyr_frame, bt.g[0], bt.g[1], bt.g[2], bt.g[3], (orbx_keypoint *)ex->d_kps.p, (uint8_t *)ex->d_desc.p,
                           strict, n, hm, src0, ex->width, ex->height, (uint8_t *)scratch.p, frame);
    ORBX_HIP(hipGetLastError());
    int32_t cnt = 0;
    ORBX_HIP(hipMemcpyAsync(&cnt, (const int32_t *)ex->d_count.p + frame, 4, hipMemcpyDeviceToHost, ex->stream));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    const int m = std::min(cnt, cap_keypoints);
    // patches are indexed by the keypoint's position in LEVEL order (the work list); the caller gets them by output slot
    std::vector<WorkItem> work((size_t)std::max(cnt, 1));
    std::vector<uint8_t> all((size_t)std::max(cnt, 1) * 37 * 37);
    if (cnt > 0) {
        ORBX_HIP(hipMemcpy(work.data(), (const WorkItem *)ex->d_work.p + (size_t)frame * ex->cap, sizeof(WorkItem) * (size_t)cnt, hipMemcpyDeviceToHost));
        ORBX_HIP(hipMemcpy(all.data(), scratch.p, (size_t)cnt * 37 * 37, hipMemcpyDeviceToHost));
        for (int i = 0; i < cnt; i++)
            if (work[i].pos >= 0 && work[i].pos < m) memcpy(dst + (size_t)work[i].pos * 37 * 37, &all[(size_t)i * 37 * 37], 37 * 37);
    }
    scratch.release();
    return m;
}

int orbx_debug_stage_stats(orbx_extractor *ex, int64_t *out, int cap) {
    if (!ex || !out || cap < 7 || ex->last_batch <= 0) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    const int nl = ex->prm.nlevels, B = ex->last_batch;
    const size_t cb = sizeof(int32_t) * (size_t)nl * B;
    int r = ex->d2h_staged_begin(64 + cb);
    if (r != ORBX_OK) return r;
    if ((r = ex->d2h_staged(0, ex->d_fast_ovf.p, 4)) != ORBX_OK) return r;
    if ((r = ex->d2h_staged(64, ex->d_candtot.p, cb)) != ORBX_OK) return r;
    ORBX_HIP(hipStreamSynchronize(ex->stream));
    int32_t listed = 0;
    memcpy(&listed, ex->staged(0), 4);
    const int32_t *ct = (const int32_t *)ex->staged(64);
    int64_t t1 = 0, t2 = 0, t3 = 0, total = 0, most = 0;
    for (int i = 0; i < nl * B; i++) {
        const int64_t c = ct[i];
        total += c; most = std::max(most, c);
        if (c <= kOctTier1Keys) t1++; else if (c <= kOctParLdsKeys) t2++; else t3++;
    }
    out[0] = ex->fast_strip ? listed : -1; out[1] = (int64_t)ex->total_cells * B; out[2] = t1; out[3] = t2; out[4] = t3; out[5] = total; out[6] = most;
    return 7;
}

int orbx_tune_fast_queues(orbx_extractor *ex, int mode, int32_t info[4]) {
    if (!ex || mode < 0 || mode > 2) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(ex->device));
    int32_t listed = 0;
    const int64_t cells = (int64_t)ex->total_cells * std::max(ex->last_batch, 0);
    if (ex->last_batch > 0 && ex->fast_strip) {
        ORBX_HIP(hipStreamSynchronize(ex->stream));
        int r = ex->d2h_staged_begin(64);
        if (r != ORBX_OK) return r;
        if ((r = ex->d2h_staged(0, ex->d_fast_ovf.p, 4)) != ORBX_OK) return r;
        ORBX_HIP(hipStreamSynchronize(ex->stream));
        memcpy(&listed, ex->staged(0), 4);
    }
    const int g0 = ex->strip_gcap, q0 = ex->strip_qcap;
    const int gdef = orbx_extractor::kStripGcap0, qdef = orbx_extractor::kStripQcap0;
    if (mode == 2) { ex->strip_gcap = gdef; ex->strip_qcap = qdef; }
    else if (mode == 1 && ex->fast_strip && cells > 0) {
        const int gmax = std::max(ex->strip_gmax, gdef), qmax = std::max(ex->strip_qmax, qdef);
        if ((int64_t)listed * 10 > cells) {          // the queues overflow all over the batch
            ex->strip_qcap = std::min(qmax, (2 * q0 + 15) & ~15);
            ex->strip_gcap = gmax;                    // the group queue is the small one (2 bytes per 4 pixels): its ceiling at once
        } else if ((int64_t)listed * 200 < cells && q0 > qdef) {
            ex->strip_qcap = std::max(qdef, (q0 / 2 + 15) & ~15);
            if (ex->strip_qcap == qdef) ex->strip_gcap = gdef;
        }
    }
    if (ex->strip_qcap > 2 * ex->strip_gcap) ex->strip_gcap = (ex->strip_qcap / 2 + 7) & ~7;   // the scores reuse the group queue's bytes
    if (fast_strip_lds_bytes(4, ex->strip_pix_bytes, ex->strip_gcap, ex->strip_qcap) > 64 * 1024) { ex->strip_gcap = g0; ex->strip_qcap = q0; }   // (cannot happen: 13 + 40 KB)
    if (info) { info[0] = listed; info[1] = (int32_t)std::min<int64_t>(cells, INT32_MAX); info[2] = ex->strip_gcap; info[3] = ex->strip_qcap; }
    return (ex->strip_gcap != g0 || ex->strip_qcap != q0) ? 1 : 0;
}

int orbx_debug_sort_nodes(int device, const int32_t *count, const int32_t *ulx, int n, int32_t *perm) {
    if (n <= 0 || n > 65535) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(device));
    int32_t *d_c, *d_u, *d_p;
    uint64_t *d_s;
    ORBX_HIP(hipMalloc((void **)&d_c, 4 * (size_t)n));
    ORBX_HIP(hipMalloc((void **)&d_u, 4 * (size_t)n));
    ORBX_HIP(hipMalloc((void **)&d_p, 4 * (size_t)n));
    ORBX_HIP(hipMalloc((void **)&d_s, 8 * (size_t)n));
    ORBX_HIP(hipMemcpy(d_c, count, 4 * (size_t)n, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(d_u, ulx, 4 * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_sort, dim3(1), dim3(64), 0, 0, d_c, d_u, n, d_p, d_s);
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpy(perm, d_p, 4 * (size_t)n, hipMemcpyDeviceToHost));
    (void)hipFree(d_c); (void)hipFree(d_u); (void)hipFree(d_p); (void)hipFree(d_s);
    return ORBX_OK;
}


int orbx_debug_sort_nodes_par(int device, const int32_t *count, const int32_t *ulx, int n, int32_t *perm, int32_t *fell_back) {
    if (n <= 0 || n > 65535) return ORBX_E_BAD_ARG;
    ORBX_HIP(hipSetDevice(device));
    int32_t *d_c, *d_u, *d_p, *d_f;
    uint64_t *d_s, *d_t;
    uint16_t *d_l, *d_r;
    ORBX_HIP(hipMalloc((void **)&d_c, 4 * (size_t)n));
    ORBX_HIP(hipMalloc((void **)&d_u, 4 * (size_t)n));
    ORBX_HIP(hipMalloc((void **)&d_p, 4 * (size_t)n));
    ORBX_HIP(hipMalloc((void **)&d_f, 4));
    ORBX_HIP(hipMalloc((void **)&d_s, 8 * (size_t)n));
    ORBX_HIP(hipMalloc((void **)&d_t, 8 * (size_t)n));
    ORBX_HIP(hipMalloc((void **)&d_l, 2 * (size_t)n));
    ORBX_HIP(hipMalloc((void **)&d_r, 2 * (size_t)n));
    ORBX_HIP(hipMemset(d_f, 0, 4));
    ORBX_HIP(hipMemcpy(d_c, count, 4 * (size_t)n, hipMemcpyHostToDevice));
    ORBX_HIP(hipMemcpy(d_u, ulx, 4 * (size_t)n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_sort_par, dim3(1), dim3(64), 0, 0, d_c, d_u, n, d_p, d_s, d_t, d_l, d_r, d_f);
    ORBX_HIP(hipDeviceSynchronize());
    ORBX_HIP(hipMemcpy(perm, d_p, 4 * (size_t)n, hipMemcpyDeviceToHost));
    if (fell_back) ORBX_HIP(hipMemcpy(fell_back, d_f, 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_c); (void)hipFree(d_u); (void)hipFree(d_p); (void)hipFree(d_s); (void)hipFree(d_t); (void)hipFree(d_l); (void)hipFree(d_r);
    (void)hipFree(d_f);
    return ORBX_OK;
}

int orbx_profile_enable(orbx_extractor *ex, int enable) {
    if (!ex) return ORBX_E_BAD_ARG;
    ex->profile = enable != 0;
    for (int k = 0; k < K_COUNT; k++) { ex->prof_ms[k] = 0; ex->prof_n[k] = 0; }
    return ORBX_OK;
}
int orbx_profile_read(orbx_extractor *ex, const char **names, double *avg_ms, int64_t *launches, int cap) {
    if (!ex) return ORBX_E_BAD_ARG;
    for (int k = 0; k < K_COUNT && k < cap; k++) {
        if (names) names[k] = kKernelNames[k];
        if (avg_ms) avg_ms[k] = ex->prof_n[k] ? ex->prof_ms[k] / (double)ex->prof_n[k] : 0.0;
        if (launches) launches[k] = ex->prof_n[k];
    }
    return K_COUNT;
}

}  // extern "C"
