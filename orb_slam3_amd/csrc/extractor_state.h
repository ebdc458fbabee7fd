// extractor_state.h -- host-side state of an extractor (shared by orbx_extractor.hip and orbx_matcher.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "orbx_internal.h"

namespace orbx {

extern thread_local std::string g_last_error;
void set_error(const std::string &s);

#define ORBX_HIP(expr)                                                                                 \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            orbx::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                        \
            return ORBX_E_HIP;                                                                         \
        }                                                                                              \
    } while (0)

// Device buffer.  Normal mode: hipMalloc + zero fill (every buffer has defined contents from its first use).
// Guard mode (ORBX_GUARD=1|2, debugging aid, see tools/repro_fault.sh): the buffer is placed through the HIP virtual-memory
// API so that its END (1) or its START (2) touches an unmapped address range, and it is filled with a poison byte: any kernel or copy
// that runs past (before) the buffer, or indexes with a value it never wrote, faults deterministically instead of silently
// touching a neighbouring allocation.
int guard_mode();
int guard_fill();   // ORBX_GUARD_FILL=<byte> (default 0xCB)
// Optional roctx ranges around the host-side phases (the reference instruments the same two spots with its REGISTER_TIMES timers:
// ORB extraction and stereo / projection matching, Tracking.cc).  ORBX_ROCTX=1 resolves libroctx64.so / librocprofiler-sdk-roctx.so at
// run time (no link dependency); rocprofv3 --marker-trace then shows orbx:extract / orbx:match / orbx:download ranges.
struct RoctxRange {
    explicit RoctxRange(const char *name);
    ~RoctxRange();
    bool on;
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    // guard-mode bookkeeping
    void *va = nullptr; size_t va_bytes = 0, map_bytes = 0, gran = 0;
    hipMemGenericAllocationHandle_t handle{};
    bool guarded = false;
    int ensure(size_t need) {
        if (need <= bytes) return ORBX_OK;
        release();
        if (guard_mode() == 0) {
            ORBX_HIP(hipMalloc(&p, need));
            bytes = need;
            // the clear must have RUN before anybody uses the buffer: hipMemset may return before the fill executes (it is queued on the
            // null stream), and the library's non-blocking streams do not wait for the null stream -- a copy or kernel enqueued right
            // after ensure() could otherwise be overtaken by the zeros (seen once: the first host-input batch into a fresh slab)
            ORBX_HIP(hipMemset(p, 0, need));
            ORBX_HIP(hipDeviceSynchronize());   // control run without this wait: 5 of 5 short-batch loops differ (profiles/r03_a_open_item_control.log)
            if (getenv("ORBX_DEBUG_ALLOC")) fprintf(stderr, "[orbx alloc] %p .. %p  %zu bytes\n", p, (char *)p + need, need);
            return ORBX_OK;
        }
        int dev = 0;
        ORBX_HIP(hipGetDevice(&dev));
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        ORBX_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
        const size_t need16 = (need + 15) & ~(size_t)15;
        map_bytes = (need16 + gran - 1) / gran * gran;
        va_bytes = map_bytes + 2 * gran;
        ORBX_HIP(hipMemAddressReserve(&va, va_bytes, gran, nullptr, 0));
        ORBX_HIP(hipMemCreate(&handle, map_bytes, &prop, 0));
        ORBX_HIP(hipMemMap((char *)va + gran, map_bytes, 0, handle, 0));
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ORBX_HIP(hipMemSetAccess((char *)va + gran, map_bytes, &acc, 1));
        guarded = true;
        ORBX_HIP(hipMemset((char *)va + gran, guard_fill(), map_bytes));
        ORBX_HIP(hipDeviceSynchronize());   // as above
        p = (char *)va + gran + (guard_mode() == 1 ? map_bytes - need16 : 0);
        bytes = need;
        return ORBX_OK;
    }
    void release() {
        if (guarded) {
            // the address range stays reserved (debug mode; re-using freed ranges crashed hipMemMap in ROCm 7.0's runtime)
            (void)hipMemUnmap((char *)va + gran, map_bytes);
            (void)hipMemRelease(handle);
            guarded = false; va = nullptr;
        } else if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
    }
};

enum { K_PYR_BASE, K_PYR_RESIZE, K_FAST, K_OCTREE, K_FINALIZE, K_BLUR, K_DESCRIBE, K_MATCH_SCAN, K_MATCH_RESOLVE, K_COUNT };

}  // namespace orbx

extern "C" int orbx_materialize_level0(struct orbx_extractor *ex);   // orbx_extractor.hip (internal: not part of include/orbx.h)

struct orbx_extractor {
    typedef orbx::DevBuf DevBuf;
    typedef orbx::LevelInfo LevelInfo;
    enum { K_COUNT = orbx::K_COUNT };
    orbx_params prm;
    int device = 0;
    hipStream_t stream = nullptr;
    // T1 tables (ORBextractor.cc:414-468)
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> quota;
    int umax[16];
    // geometry of the current configuration
    int width = 0, height = 0, batch_cap = 0, last_batch = 0;
    std::vector<LevelInfo> lv;
    size_t pyr_frame = 0, blur_frame = 0, cand_frame = 0, lvl_frame = 0;
    int total_cells = 0, cap = 0, max_pool = 0;
    size_t fast_lds = 0;
    bool resize_march_ok[orbx::kMaxLevels] = {};   // every tap pair of a dword column of level l within 8 source bytes (k_pyr_resize_march)
    bool fast_strip = false;    // k_fast_strip applies (cells at most 57 px wide, 63 px high)
    static constexpr int kStripGcap0 = 512, kStripQcap0 = 816;   // the default queue sizes (sparse scenes); orbx_tune_fast_queues
    int strip_gmax = 1024, strip_qmax = 4096;                     // a wave's band in groups / pixels, over the geometry's strips (configure)
    int n_strips = 0, strip_pix_bytes = 0, strip_gcap = kStripGcap0, strip_qcap = kStripQcap0;   // k_fast_strip: strips per frame, tile rows, per-wave group / pixel queue capacities
    int fast_wave_pitch = 64, fast_wave_rows = 0, fast_wave_qfull = 16;   // list pass (fast_wave_cell): LDS tile pitch (48 / 64), max sub-image rows, whole-cell queue
    DevBuf d_fast_ovf;          // [1 + n_fast_tiles * batch] overflow counter + list of k_fast_wave
    bool oct_par = true;        // wave-parallel quad-tree kernel (k_octree_par); false: sequential emulation (k_octree)
    bool oct_single_wave = false;   // diagnostic: levels beyond the LDS tiers on one wave (round 5) instead of the whole workgroup
    bool fast_wave = true;      // every level's cell fits k_fast_wave's fixed LDS pitch
    int n_fast_tiles = 0, n_blur_items = 0;
    int blur_waves = 2048;      // single-wave workgroups of k_blur_stream (measured: 512 / 1024 / 2048 / 4096 -> EuRoC step 1.25 / 1.115 / 1.09 / 1.10 ms)
    // device memory
    DevBuf d_lv, d_xtab, d_ytab, d_xgtab, d_fast_tiles, d_strips, d_blur_items, d_dc;
    // k_pyr_stream (levels 1 .. n-1 in one launch): per-geometry tables; ps_ok false = the geometry does not fit, the per-level launches run
    // one plan per band count (1, 2, 4, 8 bands per frame): the launch takes the one that puts about two workgroups on every CU for the batch at hand
    struct PyrPlanDev { DevBuf cols, steps, tasks, band0; orbx::PyrStreamGeom geom; int bands = 0; size_t lds = 0; bool ok = false; };
    PyrPlanDev ps_plan[4];
    bool ps_ok = false;         // at least one plan exists
    int ps_wg_target = 512;     // workgroups a launch should have (2 per CU): 256 frames -> 2 bands, 128 -> 4 (KITTI 150 -> 104 us, TUM-VI 328 -> 222 us against 2 bands)
    int ps_min_frames = 48;   // batches smaller than ps_min_frames keep the per-level launches (a band is one workgroup: too few to fill the device; measured equal at 16 frames, 3 % slower at 32, 5 % faster at 64, 14 % at 128)
    // Level 0 in place: a batch extracted by k_pyr_stream + k_fast_strip + k_describe_fused reads level 0 from the caller's frames (which orbx.h
    // keeps untouched until orbx_sync / orbx_download_wait) and leaves the slab's level 0 unwritten; the few readers of the padded level 0
    // (orbx_get_level, the stereo rig's SAD stage, the blurred-level debug readout) materialise it first: materialize_level0()
    bool lvl0_inplace = false;
    // level 0 of an in-place batch is written on REQUEST from the caller's frames -- which include/orbx.h lets the caller overwrite after orbx_sync /
    // orbx_download_wait.  Those two calls therefore mark the frames released; a request for level 0 after that is refused (ORBX_E_STALE) instead of
    // answered from whatever the buffer holds by then.  in0_copy_seq = downloads issued when the batch was enqueued (a download issued later covers it).
    bool in0_released = false;
    unsigned in0_copy_seq = 0;
    hipEvent_t in0_event = nullptr;   // "input consumed" event of the in-place batch (upload slab of orbx_extract_batch_host): re-recorded behind the materialisation
    const uint8_t *in0_images = nullptr;
    size_t in0_row_stride = 0, in0_frame_stride = 0;
    int n_strips0 = 0;          // level-0 strips of k_fast_strip (the first of a frame)
    // Stereo rig (orbx_stereo_batch_device): the SAD stage reads both extractors' pyramids on the match stream while the NEXT pair of batches is
    // extracted, so an extractor that has been part of a rig alternates between two pyramid slabs (allocated at the first stereo call)
    DevBuf d_pyr2;
    bool pyr_double = false;
    int pyr_slot = 0;
    uint8_t *pyr_cur() const { return (uint8_t *)(pyr_slot ? d_pyr2.p : d_pyr.p); }
    DevBuf d_pyr, d_blur, d_cellcnt, d_cellent, d_keys0, d_keys1, d_nof0, d_nof1, d_lvlkp, d_lvlcnt, d_candtot, d_work;
    DevBuf d_kps, d_desc, d_count, d_mono, d_err;
    DevBuf d_mkey1, d_mkey2, d_mocc, d_mentries, d_mprobs, d_mres, d_mscale, d_mgrid;  // batched frame-to-frame matcher scratch
    // pinned host staging
    void *h_stage = nullptr;
    size_t h_stage_bytes = 0;
    // profiling
    bool profile = false;
    double prof_ms[K_COUNT] = {0};
    int64_t prof_n[K_COUNT] = {0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // asynchronous result download (copy stream overlaps the next batch's kernels)
    hipStream_t copy_stream = nullptr;
    // concurrent branches of one batch: blur runs beside FAST/quad-tree (aux), the frame-to-frame matcher of batch i
    // runs beside the pyramid/FAST/quad-tree of batch i+1 (match)
    hipStream_t aux_stream = nullptr, match_stream = nullptr;
    // host-resident input (orbx_extract_batch_host): upload stream + two input slabs; slot s may be overwritten once the
    // k_pyr_base that read it has finished (ev_in_free), the extraction may start once the upload has landed (ev_in_ready)
    hipStream_t in_stream = nullptr;
    hipStream_t spare_stream = nullptr;   // never used: spaces the hardware queues of a second extractor (orbx_create)
    DevBuf d_in[2];
    hipEvent_t ev_in_free[2] = {nullptr, nullptr}, ev_in_ready[2] = {nullptr, nullptr};
    bool in_used[2] = {false, false};
    unsigned in_issued = 0;
    hipEvent_t ev_pyr = nullptr, ev_blur = nullptr, ev_describe = nullptr, ev_match = nullptr;
    bool match_pending = false;
    bool copy_covers_match = false;   // the most recent download waited for ev_match on the copy stream: its ev_copy_done implies the matcher is done
    bool side_streams = true;  // the blur pass / the matcher / the downloads on their own streams (profile mode keeps every kernel on the main stream)
    bool fused_blur = false;   // k_describe_fused (the blur on demand around the keypoints) instead of k_blur_stream + k_describe: chosen per geometry in
                               // configure(), ORBX_FUSED_BLUR=0 / 1 forces
    hipEvent_t ev_stereo_copy[2] = {nullptr, nullptr};   // ends of the last two orbx_stereo_batch_download_async
    unsigned stereo_copy_issued = 0, stereo_copy_waited = 0;
    hipEvent_t ev_copy_done[2] = {nullptr, nullptr};  // ring: up to two downloads in flight
    unsigned copy_issued = 0, copy_waited = 0;         // copy_issued - copy_waited = downloads in flight
    bool copy_pending = false;                         // a download was issued since the last (re)configuration
    int32_t *h_err = nullptr;  // pinned, 2 slots
    DevBuf d_st_bidx, d_st_bdist, d_st_ur, d_st_depth, d_st_sad, d_st_nm, d_st_scales, d_st_rowptr, d_st_rowidx;  // device stereo matcher (left extractor)
    DevBuf d_match, d_nmatch;  // internal match outputs [B][cap], [B] (one matcher per batch: orbx.h)
    int internal_match_owner = 0;   // which batched matcher wrote them for the current batch: 0 none, 1 frame-to-frame, 2 map points
    // cached problem descriptors of orbx_match_consecutive_device
    struct MatchKey { int n = 0, cap = 0; const void *match = nullptr, *nm = nullptr; float th = 0, du = 0, dv = 0; int ori = 0; const void *kps = nullptr; } mkey;

    // batched SearchByProjection(Frame, MapPoints) on the resident batch (orbx_search_mappoints_batch_device)
    DevBuf d_mp_qr, d_mp_qmin, d_mp_qmax, d_mp_valid, d_mp_keys, d_mp_meta, d_mp_grid, d_mp_probs, d_mp_res, d_mp_misc, d_mp_entries;
    struct MpKey { int n = 0, cap = 0, n_mp = 0; const void *px = nullptr, *py = nullptr, *lvl = nullptr, *vc = nullptr, *iv = nullptr, *desc = nullptr,
                   *match = nullptr, *nm = nullptr, *kps = nullptr; size_t dstride = 0; float th = 0, ratio = 0; } mpkey;

    // camera of the batch path (orbx_set_camera): undistorted keypoints + undistorted image bounds for the batched matchers
    bool has_camera = false;
    float cam_params[9] = {0};           // fx fy cx cy k1 k2 p1 p2 k3
    float cam_bf = 0;
    float bounds[4] = {0, 0, 0, 0};      // mnMinX, mnMaxX, mnMinY, mnMaxY of the current geometry
    DevBuf d_kps_un, d_frustum_frames;
    void *h_frustum[3] = {nullptr, nullptr, nullptr};   // pinned ring of orbx_frustum_batch_device's pose uploads
    size_t h_frustum_bytes[3] = {0, 0, 0};
    hipEvent_t ev_frustum[3] = {nullptr, nullptr, nullptr};
    bool frustum_used[3] = {false, false, false};
    unsigned frustum_issued = 0;
    const void *match_kps() const { return has_camera ? d_kps_un.p : d_kps.p; }

    // Synchronous device -> caller copy through the pinned staging buffer (no caller pointer is ever handed to the HIP runtime: see
    // PinnedArena in orbx_matcher.hip).  `off` = byte offset inside the staging buffer, so that several copies share one sync.
    int d2h_staged_begin(size_t total_bytes) { return ensure_stage(total_bytes); }
    int d2h_staged(size_t off, const void *d_src, size_t bytes) {
        if (bytes == 0) return ORBX_OK;
        ORBX_HIP(hipMemcpyAsync((uint8_t *)h_stage + off, d_src, bytes, hipMemcpyDeviceToHost, stream));
        return ORBX_OK;
    }
    const uint8_t *staged(size_t off) const { return (const uint8_t *)h_stage + off; }

    int ensure_stage(size_t bytes) {
        if (bytes <= h_stage_bytes) return ORBX_OK;
        if (h_stage) (void)hipHostFree(h_stage);
        h_stage = nullptr; h_stage_bytes = 0;
        // coherent (fine-grained) whatever HIP_HOST_COHERENT says: orbx_extract's kernels read the image from and write the results into this
        // block themselves, and the host reads them right after the stream synchronisation
        ORBX_HIP(hipHostMalloc(&h_stage, bytes, hipHostMallocCoherent));
        h_stage_bytes = bytes;
        return ORBX_OK;
    }
};

