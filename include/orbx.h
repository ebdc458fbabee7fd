/*
 * orbx.h -- C ABI of liborbx.so: the MI355X (gfx950) ORB front-end for ORB-SLAM3.
 *
 * This is the drop-in boundary.  The reference has no FFI seam: ORBextractor / ORBmatcher are ordinary C++
 * classes (/root/reference/include/ORBextractor.h:43-109, include/ORBmatcher.h:36-103).  The replacement is
 * a same-named C++ adapter (orb_slam3_amd/cpp/ORBextractor.h, ORBmatcher.h) whose methods forward to the
 * entry points below; INTEGRATION.md shows the binding a maintainer adds.  Plain pointers and sizes only --
 * no C++ types, no OpenCV types, no torch types, no exceptions cross this ABI.
 *
 * All compute entry points run hand-written HIP kernels on the selected GPU.  There is no CPU fallback:
 * if no HIP device is usable the create functions fail with ORBX_E_NO_DEVICE.
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_VERSION 100

/* status codes: >= 0 success, < 0 error.  ORBX_E_EMPTY mirrors the reference's "return -1" for an empty
 * image (ORBextractor.cc:1090). */
enum {
    ORBX_OK = 0,
    ORBX_E_EMPTY = -1,
    ORBX_E_BAD_ARG = -2,
    ORBX_E_TOO_SMALL = -3,   /* a pyramid level would have no 35-px FAST cell, or its detection window is narrower than half its
                              * height (nIni = round(w/h) = 0, src/ORBextractor.cc:559-580): the reference divides by zero there */
    ORBX_E_CAPACITY = -4,    /* output capacity too small */
    ORBX_E_NO_DEVICE = -5,
    ORBX_E_HIP = -6,         /* a HIP runtime call failed; see orbx_last_error() */
    ORBX_E_TOO_LARGE = -7,   /* image / batch exceeds the limits given at creation, or a dimension > 4095 px */
    ORBX_E_INTERNAL = -8,    /* device-side consistency check failed; orbx_last_error() names the code: 1 more keypoints than the output
                              * capacity, 2 / 3 quad-tree node pool / level list overflow */
    ORBX_E_STALE = -9        /* level 0 of a batch that was extracted in place was requested after orbx_sync / orbx_download_wait released the
                              * caller's frames (see orbx_get_level) */
};

/* 28-byte POD with the field layout of cv::KeyPoint {Point2f pt; float size, angle, response; int octave,
 * class_id} -- what ORBextractor::operator() fills (ORBextractor.cc:861-869, 884-890). */
typedef struct orbx_keypoint {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} orbx_keypoint;

/* flags in orbx_params.flags */
enum {
    /* Round descriptor sample coordinates with separate multiply and add (strict ISO evaluation of
     * ORBextractor.cc:117-119).  Default (flag clear): the fused form GCC emits for the reference's own
     * build flags (-O3 -march=native, CMakeLists.txt:10-13) on an FMA-capable x86. */
    ORBX_FLAG_DESC_STRICT = 1u,
    /* 7x7 Gaussian taps of OpenCV <= 4.5.0 ([18,34,49,55,49,34,18]/256) instead of >= 4.5.1
     * ([18,34,48,56,48,34,18]/256). */
    ORBX_FLAG_BLUR_OCV440 = 2u,
    /* cv::fastAtan2 (ORBextractor.cc:102) with its polynomial contracted to fused multiply-adds:
     * fma(fma(fma(p7,c2,p5),c2,p3),c2,p1) * c, and 90 - q*c as ONE fnmadd.  That is what the scalar atan_f32 of
     * OpenCV's core/src/mathfuncs.cpp compiles to where FMA is part of the library's BASELINE instruction set and the
     * compiler contracts by default (GCC: -ffp-contract=fast): aarch64 builds (NEON FMA is baseline: Jetson, Apple),
     * and x86 builds configured with CPU_BASELINE >= FMA3 / AVX2 or -march=native.  Default (flag clear): every
     * operation rounded separately -- the stock x86-64 packages (baseline SSE3; FMA only in the dispatched array
     * kernels, which the scalar cv::fastAtan2 does not go through).  The two differ by 1 ulp on a few percent of the
     * angles (up to 3e-5 degrees near 360). */
    ORBX_FLAG_ATAN_FMA = 4u
};

/* ORBextractor constructor arguments (ORBextractor.h:50-51; values from Settings.cc:443-451) */
typedef struct orbx_params {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    uint32_t flags;
} orbx_params;

typedef struct orbx_extractor orbx_extractor;

/* ---------------------------------------------------------------------------------------------------
 * Extractor  (replaces ORBextractor, include/ORBextractor.h:43-109)
 * ------------------------------------------------------------------------------------------------- */

/* ORBextractor::ORBextractor (ORBextractor.cc:409-469).  `device` is the HIP device ordinal.  Workspaces are
 * sized lazily for the largest (width, height, batch) seen; max_* are optional pre-allocation hints (0 = lazy). */
int orbx_create(const orbx_params *params, int device, int max_width, int max_height, int max_batch,
                orbx_extractor **out);
void orbx_destroy(orbx_extractor *ex);

/* ORBextractor::operator() (ORBextractor.cc:1086-1168) on a host image (8-bit, 1 channel, `stride` bytes per row).
 * vLappingArea = {lap0, lap1}.  Writes up to `cap` keypoints / 32-byte descriptors; *n_out = number of keypoints,
 * *mono_index = the reference's return value.  Synchronous (H2D, kernels, D2H on the extractor's stream).
 * Returns ORBX_OK, ORBX_E_EMPTY (image NULL or 0-sized: the reference returns -1), or another error. */
int orbx_extract(orbx_extractor *ex, const uint8_t *image, int width, int height, size_t stride, int lap0, int lap1,
                 orbx_keypoint *keypoints, uint8_t *descriptors, int cap, int *n_out, int *mono_index);

/* Batched form over device-resident frames (one camera stream = one extractor = one HIP stream).
 * d_images: device pointer; frame f row y starts at d_images + f*frame_stride + y*row_stride.
 * Enqueues all kernels asynchronously on the extractor's stream; results stay in HBM until downloaded.
 * The extractor's streams are non-blocking: d_images (like every device buffer handed to an orbx_*_device call) must be complete
 * when the call is made, and must stay untouched until orbx_sync / orbx_download_wait. */
int orbx_extract_batch_device(orbx_extractor *ex, const uint8_t *d_images, int n_frames, int width, int height,
                              size_t row_stride, size_t frame_stride, int lap0, int lap1);

/* The same over HOST-resident frames (what the reference hands to operator(): cv::Mat data in host memory, ORBextractor.cc:1086):
 * the frames are copied to the device on the extractor's upload stream into one of two internal input slabs, so the upload
 * of batch i+1 overlaps the kernels of batch i; everything else as orbx_extract_batch_device.  h_images MUST be pinned
 * (hipHostMalloc / hipHostRegister; ORBX_E_BAD_ARG otherwise); it may be reused by the caller as soon as the NEXT call
 * to this function (or orbx_sync) has returned. */
int orbx_extract_batch_host(orbx_extractor *ex, const uint8_t *h_images, int n_frames, int width, int height,
                            size_t row_stride, size_t frame_stride, int lap0, int lap1);

/* Device-side view of the last batch (valid until the next extract call on this extractor). */
typedef struct orbx_batch_view {
    int32_t n_frames;
    int32_t cap;                    /* per-frame capacity of keypoints / descriptors */
    const orbx_keypoint *d_keypoints; /* [n_frames][cap] */
    const uint8_t *d_descriptors;     /* [n_frames][cap][32] */
    const int32_t *d_count;           /* [n_frames] keypoints per frame */
    const int32_t *d_mono_index;      /* [n_frames] */
    const orbx_keypoint *d_keypoints_un;  /* [n_frames][cap] mvKeysUn: == d_keypoints unless orbx_set_camera gave a distortion model */
} orbx_batch_view;
int orbx_batch_view_get(orbx_extractor *ex, orbx_batch_view *view);

/* Wait for the extractor's stream. */
int orbx_sync(orbx_extractor *ex);
/* D2H of one frame of the last batch (synchronises the stream). */
int orbx_batch_download(orbx_extractor *ex, int frame, orbx_keypoint *keypoints, uint8_t *descriptors, int cap,
                        int *n_out, int *mono_index);
/* D2H of all frames of the last batch into packed host arrays: counts[n_frames], mono[n_frames], keypoints and
 * descriptors laid out [n_frames][cap_per_frame] (cap_per_frame from orbx_output_capacity). Synchronous. */
int orbx_batch_download_all(orbx_extractor *ex, orbx_keypoint *keypoints, uint8_t *descriptors, int32_t *counts,
                            int32_t *mono_index);
int orbx_output_capacity(orbx_extractor *ex, int width, int height);
/* Asynchronous form: the D2H copies run on the extractor's copy stream behind the kernels of the last batch and
 * overlap the kernels of the NEXT batch (which wait for the copy before overwriting the device outputs).
 * Host buffers MUST be pinned (hipHostMalloc / hipHostRegister; ORBX_E_BAD_ARG otherwise).  match / nmatches (optional) receive the internal results of
 * orbx_match_consecutive_device(..., NULL, NULL): [n_frames][cap] / [n_frames].  Up to TWO downloads may be in flight
 * (enqueue batch i+1 before waiting for batch i, so that the matcher of batch i overlaps the pyramid / FAST of batch
 * i+1); orbx_download_wait blocks until the OLDEST one has landed and reports device-side errors. */
int orbx_batch_download_async(orbx_extractor *ex, orbx_keypoint *keypoints, uint8_t *descriptors, int32_t *counts,
                              int32_t *mono_index, int32_t *match, int32_t *nmatches);
int orbx_download_wait(orbx_extractor *ex);

/* mvImagePyramid[level] (public member read by Frame::ComputeStereoMatches, Frame.cc:818,908,923): copies the
 * padded level (19-px REFLECT_101 ring included) of `frame` of the last batch to host memory.
 * dst must hold (h+38) rows of dst_stride >= w+38 bytes; the ROI origin is dst + 19*dst_stride + 19.
 * LEVEL 0 of a batched extraction of ps_min_frames (48) frames or more is not copied into the library's pyramid (its kernels read the caller's
 * frames in place): the first orbx_get_level / orbx_get_level_device of level 0 after such a batch writes the padded level from those frames.
 * That is only possible while the frames are still the library's to read, i.e. BEFORE the orbx_sync / orbx_download_wait that hands them back to
 * the caller: afterwards the request fails with ORBX_E_STALE rather than return a level built from whatever the buffer holds by then.  (The frames
 * of orbx_extract_batch_host live in the library's upload slab, which is kept until the call after the next.)  orbx_get_level_device(0) of such a
 * batch synchronises the extractor's stream before it returns the pointer; for every other level and batch it is a pure getter. */
int orbx_get_level(orbx_extractor *ex, int frame, int level, uint8_t *dst, size_t dst_stride);
int orbx_level_size(orbx_extractor *ex, int width, int height, int level, int *w, int *h);
/* Device pointer to the padded level (for device-resident consumers such as the stereo matcher). */
int orbx_get_level_device(orbx_extractor *ex, int frame, int level, const uint8_t **d_padded, size_t *pitch);

/* GetLevels / GetScaleFactor(s) / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (ORBextractor.h:62-83).  Arrays hold nlevels floats; any pointer may be NULL. */
int orbx_get_levels(const orbx_extractor *ex);
float orbx_get_scale_factor(const orbx_extractor *ex);
int orbx_get_scale_tables(const orbx_extractor *ex, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2);
/* mnFeaturesPerLevel (nlevels ints) and umax (16 ints) -- for known-answer tests */
int orbx_get_feature_tables(const orbx_extractor *ex, int32_t *features_per_level, int32_t *umax16);

/* Stage introspection of the last batch (parity tests): FAST candidates of a level in reference order
 * (x, y relative to the 16-px border; response = score) and the level's keypoints after the quad-tree cull
 * (level coordinates).  Return the count, or < 0. */
int orbx_debug_level_candidates(orbx_extractor *ex, int frame, int level, orbx_keypoint *out, int cap);
int orbx_debug_level_keypoints(orbx_extractor *ex, int frame, int level, orbx_keypoint *out, int cap);
int orbx_debug_level_blurred(orbx_extractor *ex, int frame, int level, uint8_t *dst, size_t dst_stride);
/* The blur on demand of k_describe_fused (no blurred pyramid exists then): the 37 x 37 blurred pixels around keypoint k of `frame` of the last batch,
 * dst[k][37][37] for the first min(count, cap_keypoints) keypoints in OUTPUT order, centre = the keypoint's level pixel; pixels outside the level are
 * its BORDER_REFLECT_101 extension.  Re-runs the descriptor kernel of the last batch (its inputs must still be valid).  Returns the number of
 * patches, or < 0 (ORBX_E_BAD_ARG when the extractor uses the blurred slab: orbx_debug_level_blurred). */
int orbx_debug_fused_patches(orbx_extractor *ex, int frame, uint8_t *dst, int cap_keypoints);
/* Which paths the last batch took (bench / stress tests): out[0] = cells that went to the FAST list pass (k_fast_wave_list: corners at iniThFAST
 * that all lost the NMS, or a strip / cell whose candidate queue overflowed), out[1] = cells of the batch, out[2..4] = (frame, level) quad-trees with
 * <= 1792 / <= 4096 / more candidates (the two 256-thread tiers and the single-wave chunked form), out[5] = FAST candidates of the batch,
 * out[6] = the largest candidate count of a level.  Returns the number of entries written (7), or < 0. */
int orbx_debug_stage_stats(orbx_extractor *ex, int64_t *out, int cap);

/* The FAST stage's candidate queues.  k_fast_strip keeps the pixels that pass its pre-tests in per-wave LDS queues sized for sparse scenes (the default
 * leaves seven workgroups on a CU); a strip whose queue overflows is finished by the list pass (k_fast_wave_list) -- the RESULTS never depend on the
 * queue size, only the time does (dense texture: two thirds of the cells take the list pass, the stage is 4 x slower).  This call looks at the LAST batch
 * (it waits for it): mode 0 reports only; mode 1 doubles the queues (up to "every pixel of a wave's band") when more than a tenth of the batch's cells
 * went to the list pass, and halves them again (never below the default) when fewer than 1 in 200 did while the queues are enlarged; mode 2 restores the
 * default.  A caller with unknown imagery runs it after each of its first few batches (orb_slam3_amd.ORBextractor.tune_fast_queues, bench.py --scene
 * texture).  info[0] = cells of the last batch that took the list pass, info[1] = cells of the batch, info[2] / info[3] = group / pixel queue entries per
 * wave now in force.  Returns 1 if the sizes changed, 0 if not, < 0 on error. */
int orbx_tune_fast_queues(orbx_extractor *ex, int mode, int32_t info[4]);

/* Average GPU time (ms) per launch of each extractor kernel over the calls since the last reset, measured with
 * HIP events on the extractor's stream when profiling is enabled.  names/ms arrays of `cap` entries; returns the
 * number of kernels. */
int orbx_profile_enable(orbx_extractor *ex, int enable);
int orbx_profile_read(orbx_extractor *ex, const char **names, double *avg_ms, int64_t *launches, int cap);

/* ---------------------------------------------------------------------------------------------------
 * Matcher  (replaces ORBmatcher, include/ORBmatcher.h:36-103, and the Hamming stages of Frame.cc)
 * One context per host thread (own HIP stream): re-entrant across Tracking / LocalMapping / LoopClosing.
 * ------------------------------------------------------------------------------------------------- */
typedef struct orbx_matcher orbx_matcher;
int orbx_matcher_create(int device, orbx_matcher **out);
void orbx_matcher_destroy(orbx_matcher *m);
/* Transfers of the context's LAST call (test hook; host-pointer entry points): out[0] = host-to-device transfers submitted (one per run of adjacent
 * buffers), out[1] = device-to-host ones, out[2] / out[3] = their bytes; with cap >= 6 also out[4] = how many of them a DMA engine carried
 * (hipMemcpyAsync) and out[5] = k_xfer launches.  A call stages its inputs in a pinned, device-visible mirror of its device arena; runs up to 1 MiB are
 * moved by the lanes of a k_xfer launch in the call's own queue (a projection-matcher call: 1 run up, 1 down, 2 launches, no DMA submission), larger
 * ones by the DMA engine; ORBX_MATCHER_DMA=1 sends everything through the DMA engine (round 5's transport).  Returns the entries written, or < 0. */
int orbx_matcher_debug_transfers(const orbx_matcher *m, int64_t *out, int cap);
/* Test hook: the replay of the context's last orbx_search_for_initialization (k_replay_init_lists): out3[0] = rounds of its chunk loop, out3[1] = queries whose
 * candidate list ran dry and were re-scanned by the whole wave, out3[2] = queries (level-0 keypoints of F1).  Returns 3, or < 0. */
int orbx_matcher_debug_replay_stats(const orbx_matcher *m, int32_t *out3);

/* ORBmatcher::TH_LOW / TH_HIGH / HISTO_LENGTH (ORBmatcher.cc:35-37) */
#define ORBX_TH_LOW 50
#define ORBX_TH_HIGH 100
#define ORBX_HISTO_LENGTH 30

/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:2058-2074) for every candidate of a CSR candidate list:
 * query i is compared with train rows cand[row_ptr[i] .. row_ptr[i+1]); dist_out[k] = Hamming(q_i, t_cand[k]).
 * Host pointers; synchronous.  The caller replays best / second-best / ratio / taken-mask logic in reference order. */
int orbx_hamming_csr(orbx_matcher *m, const uint8_t *q_desc, int n_q, const uint8_t *t_desc, int n_t,
                     const int32_t *row_ptr, const int32_t *cand, uint16_t *dist_out);

/* Best and second-best candidate per query, ties resolved to the EARLIEST candidate position (strict '<' scan,
 * as every matcher except SearchForTriangulation does).  best_pos/second_pos are positions inside the query's
 * candidate list (-1 if none); distances are 256 if none. */
int orbx_hamming_best2_csr(orbx_matcher *m, const uint8_t *q_desc, int n_q, const uint8_t *t_desc, int n_t,
                           const int32_t *row_ptr, const int32_t *cand, int32_t *best_pos, int32_t *best_dist,
                           int32_t *second_pos, int32_t *second_dist);

/* cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, k=2) as used by Frame::ComputeStereoFishEyeMatches
 * (Frame.cc:1144): idx[2*i], idx[2*i+1] = best / second-best train row (lower index wins ties), -1 if absent. */
int orbx_knn2(orbx_matcher *m, const uint8_t *q_desc, int n_q, const uint8_t *t_desc, int n_t, int32_t *idx,
              int32_t *dist);

/* Hamming stage of Frame::ComputeStereoMatches (Frame.cc:849-894): for each left keypoint the best right keypoint
 * among those registered on row (int)yL with |octave difference| <= 1 and uL-maxD <= uR <= uL-minD.
 * best_idx_r = -1 / best_dist = ORBX_TH_HIGH when no candidate is closer than TH_HIGH. */
int orbx_stereo_rowband(orbx_matcher *m, const orbx_keypoint *kp_left, const uint8_t *desc_left, int n_left,
                        const orbx_keypoint *kp_right, const uint8_t *desc_right, int n_right,
                        const float *scale_factors, int nlevels, int n_rows, float min_d, float max_d,
                        int32_t *best_idx_r, int32_t *best_dist);

/* Frame::ComputeStereoMatches complete (Frame.cc:811-981): row-band Hamming + 11x11 SAD sub-pixel refinement on the
 * pyramid levels + median outlier rejection.  pyr_left/right[l] point at the level ROI origin (host memory),
 * as mvImagePyramid[l] does.  Fills u_right[n_left], depth[n_left] (-1 where unmatched); returns #matches. */
int orbx_compute_stereo_matches(orbx_matcher *m, const orbx_keypoint *kp_left, const uint8_t *desc_left, int n_left,
                                const orbx_keypoint *kp_right, const uint8_t *desc_right, int n_right,
                                const float *scale_factors, const float *inv_scale_factors, int nlevels,
                                const uint8_t *const *pyr_left, const uint8_t *const *pyr_right, const int32_t *pyr_w,
                                const int32_t *pyr_h, const size_t *pyr_stride, float bf, float b, float *u_right,
                                float *depth);

/* Frame description for the projection matchers: undistorted keypoints (mvKeysUn), descriptors, image bounds
 * (mnMinX..mnMaxY) from which the 64x48 grid (Frame.h:44-45, Frame.cc:385-416) is built, per-level scale factors,
 * optional right coordinates (mvuRight, NULL for mono).  n <= ORBX_MAX_FRAME_FEATURES for the projection matchers
 * (ORBX_E_TOO_LARGE otherwise, checked before any work is enqueued). */
#define ORBX_MAX_FRAME_FEATURES 16000
typedef struct orbx_frame_desc {
    const orbx_keypoint *keypoints_un;
    const uint8_t *descriptors;
    int32_t n;
    float min_x, max_x, min_y, max_y;
    const float *scale_factors;
    int32_t nlevels;
    const float *u_right;
} orbx_frame_desc;

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)
 * (ORBmatcher.cc:43-213), monocular / rectified-stereo form (Nleft == -1).  Map points are passed flattened:
 * the MapPoint scratch fields the loop reads (MapPoint.h:171-179) and a 32-byte descriptor each.
 * mp_in_view[j] != 0  <=> mbTrackInView && !isBad() && !(bFarPoints && mTrackDepth > thFarPoints).
 * frame_occupied[i] != 0 <=> F.mvpMapPoints[i] != NULL with Observations() > 0 on entry (may be NULL).
 * mp_has_obs[j] = Observations() > 0 of map point j (may be NULL = all true).
 * Output frame_match[i] = j if map point j was assigned to feature i, else -1.  Returns nmatches (>= 0). */
int orbx_search_by_projection_mappoints(orbx_matcher *m, const orbx_frame_desc *frame, const uint8_t *frame_occupied,
                                        int n_mp, const float *proj_x, const float *proj_y, const float *proj_xr,
                                        const int32_t *pred_level, const float *view_cos, const uint8_t *mp_desc,
                                        const uint8_t *mp_in_view, const uint8_t *mp_has_obs, float th, float nnratio,
                                        int32_t *frame_match);

/* ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) (ORBmatcher.cc:1676-1887) after the
 * adapter has projected the last frame's map points (one query per point that passed the projection gates).
 * level_mode: 0 = [o-1, o+1], 1 = forward [o, inf), 2 = backward [0, o].  Rotation-histogram filter applied when
 * check_orientation != 0.  cur_match[i] = query index, -1 (never assigned: the slot keeps what it held), or -2 (assigned by the
 * loop and then cleared by the rotation check: the reference leaves NULL there, ORBmatcher.cc:1871-1881).  Returns nmatches. */
int orbx_search_by_projection_frame(orbx_matcher *m, const orbx_frame_desc *cur, const uint8_t *cur_occupied, int n_q,
                                    const float *q_u, const float *q_v, const float *q_ur, const int32_t *q_octave,
                                    const float *q_angle, const uint8_t *q_desc, const uint8_t *q_has_obs, float th,
                                    int level_mode, int check_orientation, int32_t *cur_match);

/* General window form of the projection matchers: one query per projected map point with explicit window half-size
 * and level range; accept iff (float)bestDist <= max_dist; a matched feature becomes occupied (q_has_obs NULL = always).
 *   SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)  (ORBmatcher.cc:1889-2010):
 *       r = th*scale[lvl], levels [lvl-1, lvl+1], max_dist = ORBdist, rotation check, occupied = mvpMapPoints[i] != NULL
 *   SearchByProjection(KeyFrame*, Sim3f&, vpPoints[, vpPointsKFs], vpMatched[, vpMatchedKF], th, ratioHamming)
 *       (ORBmatcher.cc:427-532, 534-646): r = th*scale[lvl], levels [lvl-1, lvl], max_dist = TH_LOW*ratioHamming,
 *       no rotation check, occupied = vpMatched[i] != NULL
 * match[i] = query index or -1.  Returns nmatches. */
int orbx_search_by_projection_window(orbx_matcher *m, const orbx_frame_desc *frame, const uint8_t *occupied, int n_q,
                                     const float *q_x, const float *q_y, const float *q_r, const int32_t *q_min_level,
                                     const int32_t *q_max_level, const float *q_angle, const uint8_t *q_desc,
                                     const uint8_t *q_has_obs, float max_dist, int check_orientation, int32_t *match);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:648-763).
 * prev_matched: n1 (x, y) pairs, updated in place (:757-760).  matches12[i1] = index in F2 or -1. */
int orbx_search_for_initialization(orbx_matcher *m, const orbx_keypoint *kps1_un, const uint8_t *desc1, int n1,
                                   const orbx_frame_desc *F2, float *prev_matched, int window_size, float nnratio,
                                   int check_orientation, int32_t *matches12);

/* DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) flattened: node ids ascending + CSR of indices */
typedef struct orbx_featvec {
    const uint32_t *node_id;
    const int32_t *node_ptr;   /* n_nodes + 1 */
    const int32_t *index;
    int32_t n_nodes;
} orbx_featvec;

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (ORBmatcher.cc:223-425, monocular).
 * kf_valid[i] != 0 <=> KF feature i has a good map point.  f_match[iF] = KF feature index or -1. */
int orbx_search_by_bow_frame(orbx_matcher *m, const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid,
                             int n_kf, const orbx_featvec *kf_fv, const uint8_t *f_desc, const float *f_angle, int n_f,
                             const orbx_featvec *f_fv, float nnratio, int check_orientation, int32_t *f_match);
/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (ORBmatcher.cc:765-905).  match12[i1] = i2 or -1 */
int orbx_search_by_bow_keyframes(orbx_matcher *m, const uint8_t *desc1, const float *angle1, const uint8_t *valid1,
                                 int n1, const orbx_featvec *fv1, const uint8_t *desc2, const float *angle2,
                                 const uint8_t *valid2, int n2, const orbx_featvec *fv2, float nnratio,
                                 int check_orientation, int32_t *match12);
/* ORBmatcher::SearchForTriangulation (ORBmatcher.cc:907-1146).  skipN[i] != 0 <=> feature i already has a map point
 * (or fails bOnlyStereo).  pair_ok(user, idx1, idx2) evaluates the geometric gates of :1026-1072 (epipole distance,
 * epipolarConstrain unless bCoarse) -- host float math that stays in the adapter; NULL = always true.
 * matches12[i1] = idx2 or -1 (vMatchedPairs = the pairs with matches12[i1] >= 0 in i1 order). */
typedef int (*orbx_pair_predicate)(void *user, int idx1, int idx2);
int orbx_search_for_triangulation(orbx_matcher *m, const uint8_t *desc1, const float *angle1, const uint8_t *skip1, int n1,
                                  const orbx_featvec *fv1, const uint8_t *desc2, const float *angle2, const uint8_t *skip2,
                                  int n2, const orbx_featvec *fv2, int check_orientation, orbx_pair_predicate pair_ok,
                                  void *user, int32_t *matches12);

/* SearchForTriangulation for PINHOLE key frames with both geometric gates evaluated on the device -- no callback:
 *  - epipole distance, src/ORBmatcher.cc:1026-1034 (pairs where neither feature is stereo): (ex-x2)^2 + (ey-y2)^2 <
 *    100 * pKF2->mvScaleFactors[kp2.octave] rejects; (ep_x, ep_y) = pKF2->mpCamera->project(T2w * Cw) (:919-921);
 *  - Pinhole::epipolarConstrain, src/CameraModels/Pinhole.cpp:107-129, on the caller's F12 = K1^-T [t12]x R12 K2^-1 (row-major;
 *    the 3x3 algebra stays with the caller's Eigen): dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave]; skipped when coarse (bCoarse).
 * strict_fp = 0 applies the FMA contraction GCC -O3 -march=native gives the reference's text, 1 rounds
 * every operation separately.  skip1/skip2, feature vectors, check_orientation, matches12 and the return value as in
 * orbx_search_for_triangulation; rotation uses kps*_un[i].angle (:1086-1094). */
typedef struct orbx_pinhole_gate {
    const orbx_keypoint *kps1_un, *kps2_un; /* pKF1/pKF2->mvKeysUn */
    const float *u_right1, *u_right2;       /* mvuRight, NULL = monocular (all < 0) */
    const float *scale_factors2;            /* pKF2->mvScaleFactors [nlevels] */
    const float *level_sigma2_2;            /* pKF2->mvLevelSigma2 [nlevels] */
    int nlevels;
    float F12[9];
    float ep_x, ep_y;
    int coarse;
    int strict_fp;
} orbx_pinhole_gate;
int orbx_search_for_triangulation_pinhole(orbx_matcher *m, const uint8_t *desc1, const uint8_t *skip1, int n1, const orbx_featvec *fv1,
                                          const uint8_t *desc2, const uint8_t *skip2, int n2, const orbx_featvec *fv2,
                                          int check_orientation, const orbx_pinhole_gate *gate, int32_t *matches12);

/* SearchForTriangulation between two key frames of a FISHEYE rig (pKF1->mpCamera2 && pKF2->mpCamera2) with the geometric gate on the device -- no callback:
 * src/ORBmatcher.cc:1036-1072 picks, per candidate pair, the cameras the two features were seen by (idx < NLeft: left) and the relative pose of that camera pair,
 * then KannalaBrandt8::epipolarConstrain (src/CameraModels/KannalaBrandt8.cpp:216-221 = TriangulateMatches :305-368 > 0.0001: unproject both keypoints,
 * parallax test, Triangulate :387-400 (Eigen::JacobiSVD of the 4x4 system), depth tests, reprojection errors against 5.991 * mvLevelSigma2); skipped when coarse.
 * There is no epipole-distance test for such key frames (:1026).  Float operations round as the reference text writes them; atan2f / tanf are glibc's restated
 * bit for bit, cos / sin evaluate in double as the reference's translation unit does; Eigen's JacobiSVD is restated from its published source (Eigen is not
 * vendored by the reference: that step's parity is UNPINNED -- DESIGN.md section 5).
 * kps1 / kps2: ALL features of the key frame, [0, n_left) = mvKeys, [n_left, N) = mvKeysRight (pt, octave, angle are read).
 * cam1[0] / cam1[1] = mvParameters (fx, fy, cx, cy, k0..k3) of pKF1->mpCamera / mpCamera2, cam2 likewise for pKF2.
 * R12 / t12 [2 * right1 + right2] = Rll tll, Rlr tlr, Rrl trl, Rrr trr of :934-944 (row-major), computed by the caller's Sophus as the reference does. */
typedef struct orbx_kb8_gate {
    const orbx_keypoint *kps1, *kps2;
    int n_left1, n_left2;
    const float *level_sigma2_1, *level_sigma2_2; /* mvLevelSigma2 of pKF1 / pKF2 [nlevels] */
    int nlevels;
    float cam1[2][8], cam2[2][8];
    float R12[4][9], t12[4][3];
    int coarse;
} orbx_kb8_gate;
int orbx_search_for_triangulation_kb8(orbx_matcher *m, const uint8_t *desc1, const uint8_t *skip1, int n1, const orbx_featvec *fv1, const uint8_t *desc2,
                                      const uint8_t *skip2, int n2, const orbx_featvec *fv2, int check_orientation, const orbx_kb8_gate *gate,
                                      int32_t *matches12);

/* Test hook: KannalaBrandt8::epipolarConstrain (src/CameraModels/KannalaBrandt8.cpp:216-221) of n independent keypoint pairs evaluated on the device, one verdict
 * per pair -- the gate of orbx_search_for_triangulation_kb8 on its own.  cam1 / cam2: [2][8] parameters of (mpCamera, mpCamera2); R12 [4][9] / t12 [4][3] as in
 * orbx_kb8_gate; sel[i] = 2 * right1 + right2 picks the cameras and the pose of pair i; sigma1 / sigma2: the two level variances per pair. */
int orbx_debug_kb8_epipolar(orbx_matcher *m, const float *cam1_2x8, const float *cam2_2x8, const float *R12_4x9, const float *t12_4x3, int n, const float *xy1,
                            const float *xy2, const float *sigma1, const float *sigma2, const uint8_t *sel, uint8_t *ok);

/* ---- fisheye-stereo forms (F.Nleft != -1: KannalaBrandt8 stereo rigs, both cameras' features in one Frame) ----
 * Feature indices follow the reference: [0, n_left) = left camera (F.mvKeys), [n_left, n_left + n_right) = right camera
 * (F.mvKeysRight); `left` describes the left camera (keypoints_un = mvKeys, n = n_left, image bounds, scale factors) and its
 * `descriptors` field points at ALL n_left + n_right rows of F.mDescriptors; occupied / match arrays cover all features.
 *
 * SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) ORBmatcher.cc:43-213 whole: per map point the left search and then
 * the right-camera twin (:144-210) with mbTrackInViewR / mTrackProjXR,YR / mnTrackScaleLevelR / mTrackViewCosR; an accepted match
 * is also written to the stereo partner's slot (mvLeftToRightMatch / mvRightToLeftMatch, -1 = none) and counts twice. */
int orbx_search_by_projection_mappoints_fisheye(orbx_matcher *m, const orbx_frame_desc *left, const orbx_keypoint *kps_right, int n_right,
                                                const int32_t *left_to_right, const int32_t *right_to_left, const uint8_t *frame_occupied,
                                                int n_mp, const uint8_t *in_view, const float *proj_x, const float *proj_y,
                                                const int32_t *pred_level, const float *view_cos, const uint8_t *in_view_r,
                                                const float *proj_xr, const float *proj_yr, const int32_t *pred_level_r,
                                                const float *view_cos_r, const uint8_t *mp_desc, const uint8_t *mp_has_obs, float th,
                                                float nnratio, int32_t *frame_match);
/* SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) ORBmatcher.cc:1676-1887 incl. the right-camera twin :1794-1863:
 * (q_u, q_v) = projection into the left camera, (q_ur, q_vr) = projection of Trl * x3Dc into the right camera.  cur_match as in
 * orbx_search_by_projection_frame (-2 = assigned, then cleared by the rotation check). */
int orbx_search_by_projection_frame_fisheye(orbx_matcher *m, const orbx_frame_desc *left, const orbx_keypoint *kps_right, int n_right,
                                            const uint8_t *cur_occupied, int n_q, const float *q_u, const float *q_v, const float *q_ur,
                                            const float *q_vr, const int32_t *q_octave, const float *q_angle, const uint8_t *q_desc,
                                            const uint8_t *q_has_obs, float th, int level_mode, int check_orientation, int32_t *cur_match);
/* SearchByBoW(KeyFrame*, Frame&, ...) for a fisheye-stereo frame, ORBmatcher.cc:283-392: frame features >= n_f_left are the right
 * camera's; best / second best are kept per camera, the right match is taken only inside the left `bestDist1 <= TH_LOW` branch and
 * its ratio test is disabled by the reference's `|| true` (:359).  f_angle covers all n_f features (mvKeys then mvKeysRight). */
int orbx_search_by_bow_frame_fisheye(orbx_matcher *m, const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid, int n_kf,
                                     const orbx_featvec *kf_fv, const uint8_t *f_desc, const float *f_angle, int n_f, int n_f_left,
                                     const orbx_featvec *f_fv, float nnratio, int check_orientation, int32_t *f_match);

/* Device-resident, batched frame-to-frame matcher used by the throughput path: for every frame f >= 1 of the
 * extractor's last batch, the keypoints of frame f-1 (queries, at their own position shifted by (du, dv)) are
 * matched against frame f exactly as orbx_search_by_projection_frame does with level_mode 0, all features free on
 * entry and every query "has observations".  d_match: device int32 [n_frames][cap] (query index in frame f-1 or -1),
 * d_nmatches: device int32 [n_frames]; pass NULL for both to use internal buffers (fetched with
 * orbx_batch_download_async).  There is ONE set of internal buffers per extractor: of the two batched matchers
 * (this one and orbx_search_mappoints_batch_device) only the first to be called on a batch may use them, the other gets
 * ORBX_E_BAD_ARG and has to be given result buffers of its own.  Asynchronous on the extractor's stream. */
int orbx_match_consecutive_device(orbx_extractor *ex, float th, float du, float dv, int check_orientation,
                                  int32_t *d_match, int32_t *d_nmatches);

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, ...) (ORBmatcher.cc:39-141; called by
 * Tracking::SearchLocalPoints, Tracking.cc:3390-3413) for EVERY frame of the extractor's last batch, device-resident: frame f is
 * searched against n_mp map points whose per-frame projection data (what Frame::isInFrustum stores in the MapPoint: mTrackProjX/Y,
 * mnTrackScaleLevel, mTrackViewCos, mbTrackInView) lie in device arrays [n_frames][n_mp]; d_mp_desc holds the map points'
 * descriptors, frame f's at d_mp_desc + f*desc_frame_stride (0 = one shared set).  Monocular form (Nleft == -1, no mvuRight),
 * all features free on entry, every map point "has observations".  d_match: device int32 [n_frames][cap] (map-point index
 * per feature or -1), d_nmatches [n_frames]; NULL for both = internal buffers (orbx_batch_download_async).  Asynchronous. */
int orbx_search_mappoints_batch_device(orbx_extractor *ex, int n_mp, const float *d_proj_x, const float *d_proj_y,
                                       const int32_t *d_level, const float *d_view_cos, const uint8_t *d_in_view,
                                       const uint8_t *d_mp_desc, size_t desc_frame_stride, float th, float nnratio,
                                       int32_t *d_match, int32_t *d_nmatches);

/* ---- candidate generation for the projection matchers: Frame::UndistortKeyPoints, ComputeImageBounds, isInFrustum ----
 * Pinhole intrinsics (Frame::mK / fx, fy, cx, cy), radial-tangential distortion (Frame::mDistCoef: k1, k2, p1, p2[, k3]) and mbf. */
typedef struct orbx_camera {
    float fx, fy, cx, cy;
    float k1, k2, p1, p2, k3;
    float bf;
} orbx_camera;
/* Frame::mRcw (row-major), mtcw, mOw */
typedef struct orbx_frame_pose {
    float Rcw[9], tcw[3], Ow[3];
} orbx_frame_pose;

/* Frame::UndistortKeyPoints (Frame.cc:747-780): kps_un[i] = kps[i] with the point run through cv::undistortPoints(K, distCoef, R = I,
 * P = K) (published algorithm: 5 fixed-point iterations of the radial-tangential model in double); a plain copy when k1 == 0. */
int orbx_undistort_keypoints(orbx_matcher *m, const orbx_camera *cam, const orbx_keypoint *kps, int n, orbx_keypoint *kps_un);
/* Frame::ComputeImageBounds (Frame.cc:782-810): bounds4 = {mnMinX, mnMaxX, mnMinY, mnMaxY} (host arithmetic, no device needed) */
int orbx_image_bounds(const orbx_camera *cam, int width, int height, float *bounds4);
/* Frame::isInFrustum(pMP, viewingCosLimit) (Frame.cc:512-575, Nleft == -1) for n_mp map points given flat: world position and normal
 * (3 floats each), mfMinDistance / mfMaxDistance.  Outputs are the MapPoint fields the function writes: in_view (mbTrackInView),
 * proj_x / proj_y (mTrackProjX/Y; -1 unless the projection lies inside the image bounds), and where in_view: proj_xr, depth
 * (mTrackDepth), level (mnTrackScaleLevel = PredictScale, MapPoint.cc:531-546), view_cos.  Every float operation rounds as the
 * reference text writes it. */
int orbx_is_in_frustum(orbx_matcher *m, const orbx_camera *cam, const orbx_frame_pose *pose, const float *bounds4, float log_scale_factor,
                       int nlevels, float viewing_cos_limit, int n_mp, const float *pos, const float *normal, const float *min_dist,
                       const float *max_dist, uint8_t *in_view, float *proj_x, float *proj_y, float *proj_xr, float *depth, int32_t *level,
                       float *view_cos);
/* One camera of a fisheye rig as Frame::isInFrustumChecks (Frame.cc:1168-1240) sees it.  The caller evaluates the reference's expressions (:1172-1186):
 * left camera R = mRcw, t = mtcw, twc = mOw; right camera (bRight) R = Rrl * mRcw, t = Rrl * mtcw + trl, twc = mRwc * mTlr.translation() + mOw.
 * params = KannalaBrandt8::mvParameters (fx, fy, cx, cy, k0 .. k3). */
typedef struct orbx_fisheye_view {
    float R[9], t[3], twc[3];
    float params[8];
} orbx_fisheye_view;
/* Frame::isInFrustumChecks(pMP, viewingCosLimit, bRight) (Frame.cc:1168-1240, called from Frame::isInFrustum :577-590 when Nleft != -1) with
 * KannalaBrandt8::project (CameraModels/KannalaBrandt8.cpp:67-85) for n_mp map points and n_views (1 or 2: left, right) cameras of the rig.
 * Outputs [n_views][n_mp]: in_view (mbTrackInView / mbTrackInViewR) and, where in_view, proj_x / proj_y (mTrackProjX/Y[R]), depth (mTrackDepth[R]),
 * level (mnTrackScaleLevel[R]), view_cos (mTrackViewCos[R]); elsewhere level = -1 (:579-580) and zeros (the reference leaves those fields untouched).
 * atan2f is glibc's, restated bit for bit; cos / sin of the azimuth are evaluated in double as the reference's translation unit does (no float overload
 * in scope): proj_x / proj_y may differ from an x86-64 glibc build in the last float ulp where the double result lies on a rounding boundary. */
int orbx_is_in_frustum_checks(orbx_matcher *m, const orbx_fisheye_view *views, int n_views, const float *bounds4, float log_scale_factor, int nlevels,
                              float viewing_cos_limit, int n_mp, const float *pos, const float *normal, const float *min_dist, const float *max_dist,
                              uint8_t *in_view, float *proj_x, float *proj_y, float *depth, int32_t *level, float *view_cos);
/* The same for n_frames poses at once on device-resident map-point data (shared by all frames); outputs [n_frames][n_mp] in device
 * memory -- exactly the arrays orbx_search_mappoints_batch_device consumes (Tracking::SearchLocalPoints, Tracking.cc:3339-3413: frustum
 * test, then SearchByProjection).  bounds4 = NULL uses the extractor's camera (orbx_set_camera) or the plain image rectangle.
 * Asynchronous, ordered before a following orbx_search_mappoints_batch_device on the same extractor. */
int orbx_frustum_batch_device(orbx_extractor *ex, const orbx_camera *cam, const orbx_frame_pose *poses, int n_frames, const float *bounds4,
                              float viewing_cos_limit, int n_mp, const float *d_pos, const float *d_normal, const float *d_min_dist,
                              const float *d_max_dist, uint8_t *d_in_view, float *d_proj_x, float *d_proj_y, float *d_proj_xr, float *d_depth,
                              int32_t *d_level, float *d_view_cos);
/* Gives the extractor's batch path a camera: after every batched extraction the keypoints are undistorted on the device (mvKeysUn,
 * orbx_batch_view.d_keypoints_un) and the batched matchers use them together with the undistorted image bounds.  cam = NULL removes
 * it (mvKeysUn = mvKeys, bounds = image rectangle: the distortion-free default). */
int orbx_set_camera(orbx_extractor *ex, const orbx_camera *cam);
/* mvKeysUn of one frame of the last batch (equals the keypoints when no camera is set) */
int orbx_batch_download_keypoints_un(orbx_extractor *ex, int frame, orbx_keypoint *kps_un, int cap, int *n_out);

/* The matching core of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th, bRight) (ORBmatcher.cc:1148-1337, candidate loop
 * :1246-1306) and Fuse(KeyFrame*, Sim3f&, vpPoints, th, vpReplacePoint) (:1339-1455, loop :1405-1433): for each projected
 * map point (u, v[, ur], radius = th*scale[lvl], predicted level) the best feature of the key frame among
 * KeyFrame::GetFeaturesInArea(u, v, r) with octave in [lvl-1, lvl] and -- when inv_level_sigma2 != NULL (first overload) --
 * reprojection chi2 <= 5.99 (mono) / 7.8 (kf->u_right[idx] >= 0).  Queries do not interact; best_idx = -1 / best_dist = 256
 * when nothing qualifies.  The caller accepts best_dist <= ORBX_TH_LOW and performs Replace / AddObservation / AddMapPoint.
 * strict_fp = 0: e2 summed with the fused multiply-adds GCC emits for the reference's flags; 1: separate mul/add. */
int orbx_fuse_search(orbx_matcher *m, const orbx_frame_desc *kf, const float *inv_level_sigma2, int n_q, const float *q_u,
                     const float *q_v, const float *q_ur, const float *q_r, const int32_t *q_level, const uint8_t *q_desc,
                     int strict_fp, int32_t *best_idx, int32_t *best_dist);

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:329-403), batched over map points: set s = the descriptors of the
 * observations of map point s, descriptors[set_ptr[s] .. set_ptr[s+1]) (gathered by the adapter from
 * pKF->mDescriptors.row(leftIndex/rightIndex) in std::map order).  best_idx[s] = index inside the set of the descriptor
 * with the least median Hamming distance to the others (first minimum wins), -1 for an empty set. */
int orbx_distinctive_descriptors(orbx_matcher *m, const uint8_t *descriptors, const int32_t *set_ptr, int n_sets,
                                 int32_t *best_idx);

/* ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> on the device (include/ORBVocabulary.h;
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h).  The tree crosses the ABI flattened: node 0 is the root, the children
 * of node i are child_idx[child_ptr[i] .. child_ptr[i+1]) in m_nodes[i].children order, node_desc = n_nodes x 32 bytes,
 * word_id[i] >= 0 iff node i is a leaf (its WordId).  L = depth (m_L). */
typedef struct orbx_vocabulary orbx_vocabulary;
int orbx_vocabulary_create(int device, int L, int n_nodes, const int32_t *child_ptr, const int32_t *child_idx,
                           const uint8_t *node_desc, const int32_t *word_id, orbx_vocabulary **out);
void orbx_vocabulary_destroy(orbx_vocabulary *voc);
/* TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (TemplatedVocabulary.h:1206-1250) for the n
 * descriptors of a frame, as Frame::ComputeBoW needs (Frame.cc:738-745, levelsup = 4): word_id[i] = WordId of the leaf
 * reached, node_id[i] = NodeId at level L - levelsup (the FeatureVector key).  The adapter builds BowVector (weights from
 * its own vocabulary copy, tf-idf + L1 in double) and FeatureVector from these ids. */
int orbx_bow_transform(orbx_matcher *m, const orbx_vocabulary *voc, const uint8_t *descriptors, int n, int levelsup,
                       int32_t *word_id, int32_t *node_id);

/* Frame::ComputeStereoMatches (Frame.cc:811-981) for every frame of two resident batches: `left` and `right` must have
 * extracted batches of the same size and image shape (rectified stereo, lapping {0,0}).  Row-band Hamming match, 11x11
 * SAD sub-pixel refinement on the device-resident pyramids and the median outlier rejection all run on the device, on
 * the left extractor's MATCH stream: the next pair of batches may be extracted meanwhile (from the first call on both
 * extractors alternate between two pyramid slabs, and their next k_finalize waits for this stage as for a matcher of
 * their own).  Results (mvuRight, mvDepth; -1 = no match) per frame via orbx_stereo_batch_download[_all | _async]. */
int orbx_stereo_batch_device(orbx_extractor *left, orbx_extractor *right, float bf, float b);
int orbx_stereo_batch_download(orbx_extractor *left, int frame, float *u_right, float *depth, int *n_left, int *n_matches);
/* all frames at once: u_right / depth [n_frames][cap] (entries beyond a frame's keypoint count unspecified), n_matches [n_frames] */
int orbx_stereo_batch_download_all(orbx_extractor *left, float *u_right, float *depth, int32_t *n_matches);
/* the same into PINNED host buffers, asynchronously behind the stereo kernels (at most two such downloads in flight): the next pair of batches
 * can be extracted meanwhile; orbx_stereo_download_wait returns when the buffers of the OLDER one are complete */
int orbx_stereo_batch_download_async(orbx_extractor *left, float *u_right, float *depth, int32_t *n_matches);
int orbx_stereo_download_wait(orbx_extractor *left);

const char *orbx_last_error(void);
const char *orbx_status_string(int status);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
