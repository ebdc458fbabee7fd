// matcher_kernels.hip.h -- hand-written HIP kernels (gfx950, wave64) for the Hamming matchers.
//
//   k_hamming_csr        DescriptorDistance over a CSR candidate list     /root/reference/src/ORBmatcher.cc:2058-2074
//   k_hamming_best2_csr  best / second-best per query (first minimum wins) e.g. ORBmatcher.cc:103-119
//   k_knn2               cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)         Frame.cc:1144
//   k_stereo_rowband     Hamming stage of Frame::ComputeStereoMatches      Frame.cc:849-894
//   k_grid_build         Frame::AssignFeaturesToGrid as a counting sort by grid cell (Frame.cc:385-416)
//   k_window_best2       GetFeaturesInArea + Hamming scan of the projection matchers (Frame.cc:657-723,
//                        ORBmatcher.cc:71-119, 1728-1768): per grid column one contiguous candidate slice, ties
//                        resolved in the reference's candidate order through a packed (dist, seq, idx) key
//   k_greedy_resolve     the sequential part of SearchByProjection (taken-mask read-after-write, ratio test,
//                        rotation histogram) replayed in query order by one wave per frame
//
// A descriptor is 4 x u64; distance = 4 x __popcll(a ^ b).  No MFMA: this is bitwise work.
#pragma once

#include "orbx_internal.h"

#include "geometry_kernels.hip.h"   // kb8_epipolar_constrain: the fisheye gate of SearchForTriangulation

namespace orbx {

typedef unsigned long long u64;

struct Desc { u64 w[4]; };

__device__ __forceinline__ Desc load_desc(const uint8_t *p) {
    const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p);  // rows are 32-byte aligned
    const ulonglong2 a = q[0], b = q[1];
    Desc d;
    d.w[0] = a.x; d.w[1] = a.y; d.w[2] = b.x; d.w[3] = b.y;
    return d;
}
__device__ __forceinline__ int hamming(const Desc &a, const Desc &b) {
    return __popcll(a.w[0] ^ b.w[0]) + __popcll(a.w[1] ^ b.w[1]) + __popcll(a.w[2] ^ b.w[2]) + __popcll(a.w[3] ^ b.w[3]);
}

constexpr u64 kNoKey = ~0ull;
// candidates a query's list holds.  Round 6: 4 -> 6.  A list that runs dry (every entry taken, more candidates in the window) costs the replay a wave-wide
// re-scan of the window, ~3.5 us each on its critical path: 6 per 1000 queries of the bench's frame pairs with 4 entries, under 1 with 6 (k_greedy_resolve_t
// 100 -> 75 us per 255 pairs, k_window_best2_t 53 -> 57; 8 entries: 78 / 59 -- the replay's rounds read every entry)
constexpr int kTopK = 6;

// keep the two smallest keys
__device__ __forceinline__ void push2(u64 &k1, u64 &k2, u64 k) {
    // branch-free on purpose: written as "if (k < k1) {k2 = k1; k1 = k;} else if (k < k2) k2 = k;" the compiler turns the two
    // destinations into a selected ADDRESS and keeps (k1, k2) in scratch memory -- a global-memory round trip per candidate
    const u64 lo = k < k1 ? k : k1;
    const u64 other = k < k1 ? k1 : k;
    k2 = other < k2 ? other : k2;
    k1 = lo;
}
__device__ __forceinline__ void wave_min2(u64 &k1, u64 &k2) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const u64 o1 = __shfl_xor(k1, s), o2 = __shfl_xor(k2, s);
        // merge (k1<=k2) with (o1<=o2): two smallest of the four
        const u64 lo = k1 < o1 ? k1 : o1;
        const u64 hi = k1 < o1 ? (k2 < o1 ? k2 : o1) : (o2 < k1 ? o2 : k1);
        k1 = lo; k2 = hi;
    }
}
__device__ __forceinline__ u64 wave_min1(u64 k) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const u64 o = __shfl_xor(k, s);
        k = o < k ? o : k;
    }
    return k;
}

// Ordering point of a ONE-WAVE workgroup whose lanes hand data to each other through LDS only: the wave's LDS instructions execute in program order, so
// the compiler must not move LDS accesses across it and nothing else is needed.  (__syncthreads() is also a fence for global memory: s_waitcnt vmcnt(0)
// on gfx950, i.e. the wave waits for the acknowledgement of every store it has issued.)
__device__ __forceinline__ void one_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A pointer a kernel finds INSIDE a problem record is a generic ("flat") pointer to the compiler: it emits flat_load / flat_store, which count on BOTH wait
// counters.  In a latency chain that matters: a wait for an LDS result (lgkmcnt) then also waits for every flat load in flight -- the candidate lists
// k_greedy_resolve requests one chunk ahead arrived synchronously, a memory round trip per 64 queries.  gld / gst state the address space (scalar types):
// global_load / global_store, vmcnt alone.
template <class T> __device__ __forceinline__ T gld(const T *p) { return *(const __attribute__((address_space(1))) T *)p; }
template <class T> __device__ __forceinline__ void gst(T *p, T v) { *(__attribute__((address_space(1))) T *)p = v; }
__device__ __forceinline__ Desc gld_desc(const uint8_t *p) {   // rows are 32-byte aligned
    const u64 *q = reinterpret_cast<const u64 *>(p);
    Desc d;
    d.w[0] = gld(q); d.w[1] = gld(q + 1); d.w[2] = gld(q + 2); d.w[3] = gld(q + 3);
    return d;
}
__device__ __forceinline__ orbx_keypoint gld_kp(const orbx_keypoint *p) {   // the fields the matchers look at (position, angle, octave)
    orbx_keypoint k;
    k.x = gld(&p->x); k.y = gld(&p->y); k.angle = gld(&p->angle); k.octave = gld(&p->octave);
    k.size = 0.f; k.response = 0.f; k.class_id = -1;
    return k;
}

// ---------------------------------------------------------------------------------------------------------
// one wave per query; lanes stride the query's candidate list
// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// k_xfer: the transfers and fills of ONE host-pointer matcher call as a launch in the call's own queue (round 6).  A call's inputs are staged in a
// pinned, device-visible mirror of its arena (orbx_matcher); until round 5 every run of adjacent buffers went up with a hipMemcpyAsync and the
// results came back with one -- each a hand-over between the compute queue and a DMA engine (~ 8-10 us apiece on the critical path of a call that
// computes for 10-20 us), plus a launch per hipMemsetAsync.  Here the lanes read the mirror (host memory) / write it themselves, 16 bytes each, and the
// fills ride in the same launch: a call is kernel launches only, in ONE in-order queue, and ends with one stream synchronisation.  Runs above
// kKernelXferMax (the stereo matcher's 4.4 MB of pyramids) stay with the DMA engine.
// op: dst / src device-visible addresses, 16-byte aligned; units = 16-byte units; src == nullptr: fill with the byte `fill`.
// grid: ceil(largest op / 256) capped at 1024, block 256
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxXferOps = 12;
struct XferOp { uint8_t *dst; const uint8_t *src; uint32_t units; uint32_t fill; };
struct XferOps { XferOp op[kMaxXferOps]; int n; };
__global__ __launch_bounds__(256) void k_xfer(const XferOps X) {
    const uint32_t gtid = blockIdx.x * 256 + threadIdx.x, gstride = gridDim.x * 256;
    for (int k = 0; k < X.n; k++) {
        uint4 *d = reinterpret_cast<uint4 *>(X.op[k].dst);
        const uint4 *s = reinterpret_cast<const uint4 *>(X.op[k].src);
        const uint32_t units = X.op[k].units;
        if (s) {
            for (uint32_t u = gtid; u < units; u += gstride) d[u] = s[u];
        } else {
            const uint32_t f = X.op[k].fill * 0x01010101u;
            const uint4 v = make_uint4(f, f, f, f);
            for (uint32_t u = gtid; u < units; u += gstride) d[u] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_hamming_csr(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t,
                                                     const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ cand,
                                                     uint16_t *__restrict__ dist) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= nq) return;
    const int b = row_ptr[qi], e = row_ptr[qi + 1];
    if (b >= e) return;
    const Desc dq = load_desc(q + (size_t)qi * 32);
    for (int k = b + lane; k < e; k += 64) dist[k] = (uint16_t)hamming(dq, load_desc(t + (size_t)cand[k] * 32));
}

__global__ __launch_bounds__(256) void k_hamming_best2_csr(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t,
                                                           const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ cand,
                                                           int32_t *__restrict__ best_pos, int32_t *__restrict__ best_dist,
                                                           int32_t *__restrict__ second_pos, int32_t *__restrict__ second_dist) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= nq) return;
    const int b = row_ptr[qi], e = row_ptr[qi + 1];
    u64 k1 = kNoKey, k2 = kNoKey;
    if (b < e) {
        const Desc dq = load_desc(q + (size_t)qi * 32);
        for (int k = b + lane; k < e; k += 64) {
            const int d = hamming(dq, load_desc(t + (size_t)cand[k] * 32));
            push2(k1, k2, ((u64)d << 32) | (u64)(uint32_t)(k - b));
        }
    }
    wave_min2(k1, k2);
    if (lane == 0) {
        best_pos[qi] = k1 == kNoKey ? -1 : (int32_t)(k1 & 0xffffffffu);
        best_dist[qi] = k1 == kNoKey ? 256 : (int32_t)(k1 >> 32);
        second_pos[qi] = k2 == kNoKey ? -1 : (int32_t)(k2 & 0xffffffffu);
        second_dist[qi] = k2 == kNoKey ? 256 : (int32_t)(k2 >> 32);
    }
}

// ---------------------------------------------------------------------------------------------------------
// brute-force kNN-2: one wave per query, train rows staged through LDS in tiles of kKnnTile rows
// ---------------------------------------------------------------------------------------------------------
constexpr int kKnnTile = 1024;  // 32 KiB

__global__ __launch_bounds__(256) void k_knn2(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t, int nt,
                                              int32_t *__restrict__ idx, int32_t *__restrict__ dist) {
    __shared__ ulonglong2 tile[kKnnTile * 2];
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    Desc dq;
    if (qi < nq) dq = load_desc(q + (size_t)qi * 32);
    u64 k1 = kNoKey, k2 = kNoKey;
    for (int t0 = 0; t0 < nt; t0 += kKnnTile) {
        const int n = min(kKnnTile, nt - t0);
        __syncthreads();
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(t + (size_t)t0 * 32);
        for (int i = threadIdx.x; i < n * 2; i += 256) tile[i] = src[i];
        __syncthreads();
        if (qi < nq) {
            for (int j = lane; j < n; j += 64) {
                const ulonglong2 a = tile[2 * j], b = tile[2 * j + 1];
                const int d = __popcll(dq.w[0] ^ a.x) + __popcll(dq.w[1] ^ a.y) + __popcll(dq.w[2] ^ b.x) + __popcll(dq.w[3] ^ b.y);
                push2(k1, k2, ((u64)d << 32) | (u64)(uint32_t)(t0 + j));
            }
        }
    }
    if (qi >= nq) return;
    wave_min2(k1, k2);
    if (lane == 0) {
        idx[2 * qi] = k1 == kNoKey ? -1 : (int32_t)(k1 & 0xffffffffu);
        dist[2 * qi] = k1 == kNoKey ? -1 : (int32_t)(k1 >> 32);
        idx[2 * qi + 1] = k2 == kNoKey ? -1 : (int32_t)(k2 & 0xffffffffu);
        dist[2 * qi + 1] = k2 == kNoKey ? -1 : (int32_t)(k2 >> 32);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches, Hamming stage (Frame.cc:824-894).  Right keypoint iR is a candidate of left
// keypoint iL iff floor(yR - r) <= (int)yL <= ceil(yR + r), r = 2*scale[octaveR] (the row table :828-838),
// |octaveR - octaveL| <= 1 and uL - maxD <= uR <= uL - minD.  Candidates are visited in iR order with a strict
// '<' against bestDist = TH_HIGH, so the packed key (dist << 32 | iR) minimum reproduces the result.
// One wave per left keypoint.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stereo_rowband(const orbx_keypoint *__restrict__ kl, const uint8_t *__restrict__ dl, int nl,
                                                        const orbx_keypoint *__restrict__ kr, const uint8_t *__restrict__ dr, int nr,
                                                        const float *__restrict__ scale, int n_rows, float minD, float maxD,
                                                        int32_t *__restrict__ best_idx, int32_t *__restrict__ best_dist) {
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (iL >= nl) return;
    const orbx_keypoint kpL = kl[iL];
    const int row = (int)kpL.y;  // vRowIndices[vL]: float -> index truncation (:856)
    const float minU = kpL.x - maxD, maxU = kpL.x - minD;
    u64 best = kNoKey;
    if (!(maxU < 0) && row >= 0 && row < n_rows) {
        const Desc dq = load_desc(dl + (size_t)iL * 32);
        for (int iR = lane; iR < nr; iR += 64) {
            const orbx_keypoint kpR = kr[iR];
            const float r = 2.0f * scale[kpR.octave];
            const int maxr = (int)ceilf(kpR.y + r), minr = (int)floorf(kpR.y - r);
            if (row < minr || row > maxr) continue;
            if (kpR.octave < kpL.octave - 1 || kpR.octave > kpL.octave + 1) continue;
            if (kpR.x >= minU && kpR.x <= maxU) {
                const int d = hamming(dq, load_desc(dr + (size_t)iR * 32));
                const u64 k = ((u64)d << 32) | (u64)(uint32_t)iR;
                best = k < best ? k : best;
            }
        }
    }
    best = wave_min1(best);
    if (lane == 0) {
        const int d = best == kNoKey ? 256 : (int)(best >> 32);
        if (d < ORBX_TH_HIGH) { best_idx[iL] = (int32_t)(best & 0xffffffffu); best_dist[iL] = d; }
        else { best_idx[iL] = -1; best_dist[iL] = ORBX_TH_HIGH; }
    }
}

// ---------------------------------------------------------------------------------------------------------
// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:329-403), batched: one wave per observation set.  For every row
// i of the N x N Hamming matrix the median (sorted row element floor(0.5*(N-1))) is found with a 257-bin LDS histogram
// and a wave prefix scan; the first row with the smallest median wins.
// grid (ceil(n_sets/4)), block 256
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_distinctive(const uint8_t *__restrict__ desc, const int32_t *__restrict__ set_ptr, int n_sets,
                                                     int32_t *__restrict__ best_idx) {
    __shared__ int hist_all[4][320];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + wv;
    if (s >= n_sets) return;  // wave-uniform; no block barrier below
    int *hist = hist_all[wv];
    const int b = set_ptr[s], N = set_ptr[s + 1] - b;
    if (N <= 0) { if (lane == 0) best_idx[s] = -1; return; }
    const uint8_t *D = desc + (size_t)b * 32;
    const int kth = (int)(0.5 * (N - 1));
    int bestMedian = 0x7fffffff, bestIdx = 0;
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int q = 0; q < 5; q++) hist[lane * 5 + q] = 0;
        __builtin_amdgcn_wave_barrier();
        const Desc di = load_desc(D + (size_t)i * 32);
        for (int j = lane; j < N; j += 64) atomicAdd(&hist[hamming(di, load_desc(D + (size_t)j * 32))], 1);
        __builtin_amdgcn_wave_barrier();
        int c[5], tot = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) { c[q] = hist[lane * 5 + q]; tot += c[q]; }
        int incl = tot;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int t = __shfl_up(incl, sft);
            if (lane >= sft) incl += t;
        }
        int excl = incl - tot, med = -1;
        if (excl <= kth && kth < incl) {  // exactly one lane
            int acc = excl;
#pragma unroll
            for (int q = 0; q < 5; q++) { if (med < 0 && kth < acc + c[q]) med = lane * 5 + q; acc += c[q]; }
        }
        const unsigned long long owner = __ballot(med >= 0);
        med = __shfl(med, __ffsll((long long)owner) - 1);
        if (med < bestMedian) { bestMedian = med; bestIdx = i; }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) best_idx[s] = bestIdx;
}

// ---------------------------------------------------------------------------------------------------------
// DBoW2 TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (TemplatedVocabulary.h:1206-1250):
// descend the k-ary vocabulary tree, at every level to the child with the smallest Hamming distance (strict '<': the
// first minimum in m_nodes[i].children order wins).  16 lanes per feature; lanes stride the children of the current
// node, the packed key (dist << 16 | child position) is min-reduced inside the 16-lane group.
// grid (ceil(n/16)), block 256
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bow_transform(const int32_t *__restrict__ child_ptr, const int32_t *__restrict__ child_idx,
                                                       const uint8_t *__restrict__ node_desc, const int32_t *__restrict__ word_id, int L,
                                                       int levelsup, const uint8_t *__restrict__ desc, int n,
                                                       int32_t *__restrict__ word_out, int32_t *__restrict__ node_out) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sl = threadIdx.x & 15;
    const bool valid = i < n;
    Desc dq;
    if (valid) dq = load_desc(desc + (size_t)i * 32);
    const int nid_level = L - levelsup;
    int node = 0, level = 0, nid = 0;
    bool done = !valid;
    // all 16 lanes of a group follow the same path; groups of one wave may need a different number of levels
    while (__ballot(!done) != 0ull) {
        uint32_t best = 0xffffffffu;
        int b = 0, e = 0;
        if (!done) {
            b = child_ptr[node]; e = child_ptr[node + 1];
            for (int c = b + sl; c < e; c += 16) {
                const int d = hamming(dq, load_desc(node_desc + (size_t)child_idx[c] * 32));
                const uint32_t k = ((uint32_t)d << 16) | (uint32_t)(c - b);
                best = k < best ? k : best;
            }
        }
#pragma unroll
        for (int s = 8; s > 0; s >>= 1) {
            const uint32_t o = __shfl_xor(best, s);
            best = o < best ? o : best;
        }
        if (!done) {
            ++level;
            node = child_idx[b + (int)(best & 0xffffu)];
            if (level == nid_level) nid = node;
            done = (child_ptr[node + 1] <= child_ptr[node]);  // isLeaf()
        }
    }
    if (valid && sl == 0) { word_out[i] = word_id[node]; node_out[i] = (nid_level <= 0) ? 0 : nid; }
}

// ---------------------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches fully on the device for a batch of rectified stereo frames whose left / right
// extractions are resident (keypoints, descriptors, padded pyramids).
//   k_stereo_rowband_batch : Hamming stage (:849-894), one wave per left keypoint
//   k_stereo_sad           : 11x11 SAD over 11 shifts on the keypoint's pyramid level, parabola sub-pixel fit,
//                            disparity -> depth (:896-964), one wave per left keypoint
//   k_stereo_reject        : median of the SAD distances per frame by a 2-pass LDS radix select, then removal of
//                            matches with distance >= 1.5*1.4*median (:966-980), one workgroup per frame
// ---------------------------------------------------------------------------------------------------------
struct StereoBatch {
    const orbx_keypoint *kl, *kr;      // [B][capL], [B][capR]
    const uint8_t *dl, *dr;            // [B][cap][32]
    const int32_t *nl, *nr;            // [B]
    int capL, capR;
    const uint8_t *pyrL, *pyrR;        // pyramid slabs of the two extractors
    size_t pyr_frame_L, pyr_frame_R;
    const LevelInfo *lvL, *lvR;        // level tables (identical geometry)
    const float *scale, *inv_scale;    // mvScaleFactors / mvInvScaleFactors
    int n_rows;                        // mvImagePyramid[0].rows
    float bf, b;
    int32_t *best_idx, *best_dist;     // [B][capL] Hamming stage result
    float *u_right, *depth;            // [B][capL] outputs (-1 = none)
    int32_t *sad;                      // [B][capL] SAD distance of accepted matches, -1 otherwise
    int32_t *nmatches;                 // [B]
    // right keypoints bucketed by floor(y) >> row_shift (k_stereo_row_index): row_ptr [B][n_buckets + 1], row_ent [B][capR] in bucket order.
    // band = ceil(2 * largest scale factor) + 1: a right keypoint whose row band (:828-838) contains row v has floor(y) in [v - band, v + band]
    const int32_t *row_ptr;
    const uint4 *row_ent;              // x (float bits), minr, maxr, index << 8 | octave
    int band, row_shift, n_buckets;
};

// vRowIndices (Frame.cc:822-838) as an index instead of 2r + 1 registrations per keypoint: the right keypoints of a frame counted and
// listed by the bucket of floor(y), each with its row band [minr, maxr] (:830-832) precomputed; k_stereo_rowband_batch then looks at the
// ~5 % of the right keypoints whose bucket lies within `band` rows of the left keypoint's row instead of at all of them.  The order inside
// a bucket is arbitrary (atomics): the search keeps the minimum of (distance, index), which does not depend on it.
// grid (B), block 256, LDS 4 * (n_buckets + 1) + 1024
constexpr int kStereoIndexMaxBuckets = 8192;
__global__ __launch_bounds__(256) void k_stereo_row_index(StereoBatch S, int32_t *row_ptr, uint4 *row_ent) {
    extern __shared__ int32_t rows_lds[];   // [n_buckets + 1] counts -> start offsets -> cursors, then 256 partial sums
    const int f = blockIdx.x, tid = threadIdx.x, nb = S.n_buckets, nr = S.nr[f];
    int32_t *part = rows_lds + nb + 1;
    const orbx_keypoint *kr = S.kr + (size_t)f * S.capR;
    for (int i = tid; i <= nb; i += 256) rows_lds[i] = 0;
    __syncthreads();
    for (int i = tid; i < nr; i += 256) atomicAdd(&rows_lds[min(max((int)floorf(kr[i].y), 0) >> S.row_shift, nb - 1)], 1);
    __syncthreads();
    const int chunk = (nb + 256) / 256, c0 = tid * chunk, c1 = min(c0 + chunk, nb + 1);
    int sum = 0;
    for (int i = c0; i < c1; i++) sum += rows_lds[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    int run = part[tid];
    int32_t *gp = row_ptr + (size_t)f * (nb + 1);
    for (int i = c0; i < c1; i++) { const int v = rows_lds[i]; rows_lds[i] = run; gp[i] = run; run += v; }
    __syncthreads();
    uint4 *ge = row_ent + (size_t)f * S.capR;
    for (int i = tid; i < nr; i += 256) {
        const orbx_keypoint kp = kr[i];
        const float r = 2.0f * S.scale[kp.octave];                                   // :829
        const int maxr = (int)ceilf(kp.y + r), minr = (int)floorf(kp.y - r);         // :830-831
        const int slot = atomicAdd(&rows_lds[min(max((int)floorf(kp.y), 0) >> S.row_shift, nb - 1)], 1);
        ge[slot] = make_uint4(__float_as_uint(kp.x), (uint32_t)minr, (uint32_t)maxr, ((uint32_t)i << 8) | (uint32_t)(kp.octave & 0xff));
    }
}

// Hamming stage (:841-883): 16 lanes per left keypoint (4 keypoints per wave) stride the bucketed right keypoints near its row
// grid (ceil(capL / 16), B), block 256
__global__ __launch_bounds__(256) void k_stereo_rowband_batch(StereoBatch S) {
    const int f = blockIdx.y;
    const int iL = blockIdx.x * 16 + (threadIdx.x >> 4), sl = threadIdx.x & 15;
    const int nl = S.nl[f];
    if (blockIdx.x * 16 >= nl) return;
    const bool live = iL < nl;
    u64 best = kNoKey;
    if (live) {
        const uint8_t *dr = S.dr + (size_t)f * S.capR * 32;
        const orbx_keypoint kpL = S.kl[(size_t)f * S.capL + iL];
        const float minD = 0.f, maxD = S.bf / S.b;  // :841-843 (minZ = mb)
        const int row = (int)kpL.y;
        const float minU = kpL.x - maxD, maxU = kpL.x - minD;
        if (!(maxU < 0) && row >= 0 && row < S.n_rows) {
            const Desc dq = load_desc(S.dl + ((size_t)f * S.capL + iL) * 32);
            const int32_t *rp = S.row_ptr + (size_t)f * (S.n_buckets + 1);
            const int j0 = rp[max(row - S.band, 0) >> S.row_shift];
            const int j1 = rp[(min(row + S.band, S.n_rows - 1) >> S.row_shift) + 1];
            const uint4 *ent = S.row_ent + (size_t)f * S.capR;
            for (int j = j0 + sl; j < j1; j += 16) {
                const uint4 e = ent[j];
                const int oct = (int)(e.w & 0xffu), iR = (int)(e.w >> 8);
                const float xr = __uint_as_float(e.x);
                if (row < (int)e.y || row > (int)e.z) continue;
                if (oct < kpL.octave - 1 || oct > kpL.octave + 1) continue;
                if (xr >= minU && xr <= maxU) {
                    const int d = hamming(dq, load_desc(dr + (size_t)iR * 32));
                    const u64 k = ((u64)d << 32) | (u64)(uint32_t)iR;
                    best = k < best ? k : best;
                }
            }
        }
    }
#pragma unroll
    for (int s = 8; s > 0; s >>= 1) {   // minimum inside the 16-lane group
        const u64 o = __shfl_xor(best, s);
        best = o < best ? o : best;
    }
    if (live && sl == 0) {
        const int d = best == kNoKey ? 256 : (int)(best >> 32);
        const size_t o = (size_t)f * S.capL + iL;
        if (d < ORBX_TH_HIGH) { S.best_idx[o] = (int32_t)(best & 0xffffffffu); S.best_dist[o] = d; }
        else { S.best_idx[o] = -1; S.best_dist[o] = ORBX_TH_HIGH; }
    }
}

// SAD refinement + parabola (:885-964): 16 lanes per left keypoint, lane yy < 11 = one ROW of the 11 x 11 window: it fetches the row of the
// left patch (12 bytes) and the 21 + 3 bytes of the right image that the eleven shifts incR = -5 .. 5 slide over with two loads, forms the
// eleven row sums with v_alignbyte + v_sad_u8 and the group adds them up (xor butterflies: every lane ends with all eleven totals and
// evaluates the same scalar tail).  Three load instructions per wave of four keypoints.
// grid (ceil(capL / 16), B), block 256
__global__ __launch_bounds__(256) void k_stereo_sad(StereoBatch S) {
    const int f = blockIdx.y;
    const int iL = blockIdx.x * 16 + (threadIdx.x >> 4), sl = threadIdx.x & 15;
    const int nl = S.nl[f];
    if (blockIdx.x * 16 >= nl) return;
    const bool live = iL < nl;
    const size_t o = (size_t)f * S.capL + min(iL, nl - 1);
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;
    const int bidx = S.best_idx[o];
    const int thOrbDist = (ORBX_TH_HIGH + ORBX_TH_LOW) / 2;
    const bool cand = live && bidx >= 0 && S.best_dist[o] < thOrbDist;   // uniform inside a 16-lane group
    const orbx_keypoint kpL = S.kl[o];
    const float uL = kpL.x;
    const float uR0 = S.kr[(size_t)f * S.capR + max(bidx, 0)].x;
    const int lvl = kpL.octave;
    const float sf = S.inv_scale[lvl];
    const float scaleduL = roundf(__fmul_rn(kpL.x, sf)), scaledvL = roundf(__fmul_rn(kpL.y, sf)), scaleduR0 = roundf(__fmul_rn(uR0, sf));
    const int w = 5, Lh = 5;
    const int lw = S.lvR[lvl].w;
    const float iniu = scaleduR0 + Lh - w, endu = scaleduR0 + Lh + w + 1;
    const bool inside = cand && !(iniu < 0 || endu >= (float)lw);
    int sums[11];
#pragma unroll
    for (int k = 0; k < 11; k++) sums[k] = 0;
    if (inside && sl < 11) {
        const int pitchL = S.lvL[lvl].pitch, pitchR = S.lvR[lvl].pitch;
        const uint8_t *pl = S.pyrL + (size_t)f * S.pyr_frame_L + S.lvL[lvl].off + (size_t)(kEdge + (int)(scaledvL - w) + sl) * pitchL + kRoiX + (int)(scaleduL - w);
        const uint8_t *pr = S.pyrR + (size_t)f * S.pyr_frame_R + S.lvR[lvl].off + (size_t)(kEdge + (int)(scaledvL - w) + sl) * pitchR + kRoiX + (int)(scaleduR0 - w) - 5;
        uint32_t a[3], r[6];
        __builtin_memcpy(a, pl, 12);   // 11 bytes of the row + one that is masked off
        __builtin_memcpy(r, pr, 24);   // columns -10 .. +10 around scaleduR0 + three that are masked off
        a[2] &= 0x00ffffffu;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const int q = k >> 2, sh = k & 3;
            const uint32_t w0 = sh ? __builtin_amdgcn_alignbyte(r[q + 1], r[q], sh) : r[q];
            const uint32_t w1 = sh ? __builtin_amdgcn_alignbyte(r[q + 2], r[q + 1], sh) : r[q + 1];
            const uint32_t w2 = (sh ? __builtin_amdgcn_alignbyte(r[q + 3], r[q + 2], sh) : r[q + 2]) & 0x00ffffffu;
            sums[k] = (int)__builtin_amdgcn_sad_u8(a[0], w0, __builtin_amdgcn_sad_u8(a[1], w1, __builtin_amdgcn_sad_u8(a[2], w2, 0u)));
        }
    }
#pragma unroll
    for (int k = 0; k < 11; k++) {
#pragma unroll
        for (int x = 8; x > 0; x >>= 1) sums[k] += __shfl_xor(sums[k], x);
    }
    if (inside) {
        int bestDist = 0x7fffffff, bestinc = 0;
#pragma unroll
        for (int k = 0; k < 11; k++)
            if ((float)sums[k] < (float)bestDist) { bestDist = sums[k]; bestinc = k - 5; }  // :926 (float dist vs int best)
        if (!(bestinc == -Lh || bestinc == Lh)) {
            float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
            for (int k = 1; k < 10; k++)
                if (k - 5 == bestinc) { d1 = (float)sums[k - 1]; d2 = (float)sums[k]; d3 = (float)sums[k + 1]; }
            // :944  deltaR = (dist1-dist3)/(2.0f*(dist1+dist3-2.0f*dist2))
            const float den = __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2)));
            const float deltaR = __fdiv_rn(__fsub_rn(d1, d3), den);
            if (!(deltaR < -1 || deltaR > 1)) {
                float bestuR = __fmul_rn(S.scale[lvl], __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));  // :950
                float disparity = __fsub_rn(uL, bestuR);
                const float minD = 0.f, maxD = S.bf / S.b;
                if (disparity >= minD && disparity < maxD) {
                    if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }  // :956-959
                    out_d = __fdiv_rn(S.bf, disparity);
                    out_u = bestuR;
                    out_sad = bestDist;
                }
            }
        }
    }
    if (live && sl == 0) { S.u_right[o] = out_u; S.depth[o] = out_d; S.sad[o] = out_sad; }
}

// one workgroup per frame: median SAD distance (element size/2 of the ascending order, :967-968) via radix select,
// then reject distances >= 1.5f*1.4f*median (:969-980)
__global__ __launch_bounds__(256) void k_stereo_reject(StereoBatch S) {
    __shared__ int hist[256];
    __shared__ int sh_sel, sh_rank, sh_count;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int nl = S.nl[f];
    const size_t base = (size_t)f * S.capL;
    // count accepted matches
    int c = 0;
    for (int i = tid; i < nl; i += 256) c += (S.sad[base + i] >= 0);
    if (tid == 0) sh_count = 0;
    __syncthreads();
    atomicAdd(&sh_count, c);
    __syncthreads();
    const int M = sh_count;
    if (M == 0) { if (tid == 0) S.nmatches[f] = 0; return; }
    int k = M / 2;  // rank of the median in ascending order
    // pass 1: high byte (SAD <= 121*255 < 2^15), pass 2: low byte
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nl; i += 256) { const int v = S.sad[base + i]; if (v >= 0) atomicAdd(&hist[(v >> 8) & 0xff], 1); }
    __syncthreads();
    if (tid == 0) {
        int acc = 0, b = 0;
        for (b = 0; b < 256; b++) { if (acc + hist[b] > k) break; acc += hist[b]; }
        sh_sel = b; sh_rank = k - acc;
    }
    __syncthreads();
    const int hi = sh_sel;
    k = sh_rank;
    __syncthreads();
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nl; i += 256) { const int v = S.sad[base + i]; if (v >= 0 && ((v >> 8) & 0xff) == hi) atomicAdd(&hist[v & 0xff], 1); }
    __syncthreads();
    if (tid == 0) {
        int acc = 0, b = 0;
        for (b = 0; b < 256; b++) { if (acc + hist[b] > k) break; acc += hist[b]; }
        sh_sel = (hi << 8) | b;
    }
    __syncthreads();
    const float median = (float)sh_sel;
    const float thDist = __fmul_rn(1.5f * 1.4f, median);
    int kept = 0;
    for (int i = tid; i < nl; i += 256) {
        const int v = S.sad[base + i];
        if (v >= 0) {
            if ((float)v < thDist) kept++;
            else { S.u_right[base + i] = -1.0f; S.depth[base + i] = -1.0f; }
        }
    }
    if (tid == 0) sh_count = 0;
    __syncthreads();
    atomicAdd(&sh_count, kept);
    __syncthreads();
    if (tid == 0) S.nmatches[f] = sh_count;
}

// ---------------------------------------------------------------------------------------------------------
// Projection matchers.  A "problem" p is one (query set, current frame) pair; problems are batched along
// blockIdx.y so that a whole batch of frames is matched in one launch.
// ---------------------------------------------------------------------------------------------------------
struct GridParams {  // Frame::mnMinX.., mfGridElementWidthInv.. (Frame.cc:342-343)
    float minx, miny, inv_w, inv_h;
};

struct WindowProblem {
    // current frame
    const orbx_keypoint *kps;   // undistorted keypoints (mvKeysUn)
    const uint8_t *desc;
    const int32_t *n_ptr;       // number of features (device scalar; batches have per-frame counts)
    const float *u_right;       // mvuRight or NULL
    const uint8_t *occupied0;   // taken on entry or NULL
    // queries
    const float *qx, *qy, *qr;  // window centre and half size
    const int32_t *qmin, *qmax; // level range passed to GetFeaturesInArea
    const float *qxr;           // right-coordinate prediction (with u_right) or NULL
    const uint8_t *qdesc;
    const uint8_t *qvalid;      // skip query when 0 (NULL = all valid)
    const int32_t *nq_ptr;
    // derived-query mode (consecutive frames): when q_from_kps != NULL the query i is keypoint i of that frame:
    // centre = kp + (du,dv), radius = th*scale[octave], levels [o-1,o+1], descriptor q_from_desc
    const orbx_keypoint *q_from_kps;
    const float *scale;         // mvScaleFactors
    float th, du, dv;
    // Fuse reprojection gate (ORBmatcher.cc:1266-1289): active when inv_sigma2 != NULL; u_right is then the KeyFrame's
    // mvuRight (>= 0 = stereo observation) and qxr the predicted right coordinate
    const float *inv_sigma2;
    int chi2_fma;
    // 64x48 grid of the current frame built by k_grid_build: features sorted by (cell x, cell y, index)
    uint16_t *gstart;           // [64*48 + 1] offsets into gorder, cell id = x * 48 + y
    uint16_t *gorder;           // [n] feature indices
    // outputs
    u64 *keys;                  // per query: the kTopK smallest candidate keys, ascending (kNoKey = none)
    int32_t *meta;              // per query: valid_len | exhaustive << 8 | empty_window << 9  (see k_window_best2)
    // k_window_best2_t<64> only (SearchForInitialization): EVERY candidate key of a query, unordered, up to all_cap per query (NULL: not wanted); all_cnt[q] = how
    // many the query has (> all_cap: the list is incomplete).  k_replay_init_lists re-evaluates a query whose short list ran dry from this one -- a coalesced read
    // and LDS look-ups instead of a walk through the grid (bounds -> indices -> keypoints -> descriptors, ten times over for a 100-px window)
    u64 *all_keys;
    int32_t *all_cnt;
    int32_t all_cap;
};

// candidate key: dist << 32 | cellx << 24 | celly << 16 | idx   (candidate order of GetFeaturesInArea: ix outer,
// iy inner, insertion (= ascending idx) inside a cell; first minimum wins)
__device__ __forceinline__ u64 cand_key(int dist, int cx, int cy, int idx) {
    return ((u64)(uint32_t)dist << 32) | ((u64)(uint32_t)cx << 24) | ((u64)(uint32_t)cy << 16) | (u64)(uint32_t)idx;
}

struct QueryWin {
    float x, y, r, xr;
    int minL, maxL, cx0, cx1, cy0, cy1;
    bool check_levels, empty;
};

__device__ __forceinline__ QueryWin make_window(const GridParams &g, float x, float y, float r, int minL, int maxL, float xr) {
    QueryWin w;
    w.x = x; w.y = y; w.r = r; w.xr = xr; w.minL = minL; w.maxL = maxL;
    // Frame::GetFeaturesInArea :665-687
    w.cx0 = max(0, (int)floorf((x - g.minx - r) * g.inv_w));
    w.cx1 = min(63, (int)ceilf((x - g.minx + r) * g.inv_w));
    w.cy0 = max(0, (int)floorf((y - g.miny - r) * g.inv_h));
    w.cy1 = min(47, (int)ceilf((y - g.miny + r) * g.inv_h));
    w.empty = (w.cx0 >= 64) || (w.cx1 < 0) || (w.cy0 >= 48) || (w.cy1 < 0);
    w.check_levels = (minL > 0) || (maxL >= 0);
    return w;
}

// returns true and the candidate's grid cell if keypoint kp is returned by GetFeaturesInArea for window w
__device__ __forceinline__ bool in_window(const GridParams &g, const QueryWin &w, const orbx_keypoint &kp, int *cx, int *cy) {
    // AssignFeaturesToGrid / PosInGrid :725-735
    const int px = (int)roundf((kp.x - g.minx) * g.inv_w), py = (int)roundf((kp.y - g.miny) * g.inv_h);
    if (px < 0 || px >= 64 || py < 0 || py >= 48) return false;
    if (px < w.cx0 || px > w.cx1 || py < w.cy0 || py > w.cy1) return false;
    if (w.check_levels) {
        if (kp.octave < w.minL) return false;
        if (w.maxL >= 0 && kp.octave > w.maxL) return false;
    }
    const float dx = kp.x - w.x, dy = kp.y - w.y;
    if (!(fabsf(dx) < w.r && fabsf(dy) < w.r)) return false;
    *cx = px; *cy = py;
    return true;
}

__device__ __forceinline__ bool load_query(const WindowProblem &P, int qi, QueryWin *w, const GridParams &g, Desc *dq) {
    if (P.qvalid && !gld(P.qvalid + qi)) return false;
    if (P.q_from_kps) {
        const orbx_keypoint k = gld_kp(P.q_from_kps + qi);
        const float radius = P.th * gld(P.scale + k.octave);
        *w = make_window(g, k.x + P.du, k.y + P.dv, radius, k.octave - 1, k.octave + 1, 0.f);
    } else {
        *w = make_window(g, gld(P.qx + qi), gld(P.qy + qi), gld(P.qr + qi), gld(P.qmin + qi), gld(P.qmax + qi), P.qxr ? gld(P.qxr + qi) : 0.f);
    }
    *dq = gld_desc(P.qdesc + (size_t)qi * 32);
    return !w->empty;
}

// load_query for ONE query of the whole wave (the re-scan of k_greedy_resolve, a latency chain): the validity flag, the query's record and its descriptor
// are requested together -- two dependent round trips (record, scale of its octave) instead of three
__device__ __forceinline__ bool load_query_eager(const WindowProblem &P, int qi, QueryWin *w, const GridParams &g, Desc *dq) {
    const uint8_t valid = P.qvalid ? gld(P.qvalid + qi) : (uint8_t)1;
    *dq = gld_desc(P.qdesc + (size_t)qi * 32);
    if (P.q_from_kps) {
        const orbx_keypoint k = gld_kp(P.q_from_kps + qi);
        const float radius = P.th * gld(P.scale + (valid ? k.octave : 0));   // (the record of a query flagged invalid is not looked at)
        *w = make_window(g, k.x + P.du, k.y + P.dv, radius, k.octave - 1, k.octave + 1, 0.f);
    } else {
        *w = make_window(g, gld(P.qx + qi), gld(P.qy + qi), gld(P.qr + qi), gld(P.qmin + qi), gld(P.qmax + qi), P.qxr ? gld(P.qxr + qi) : 0.f);
    }
    return valid && !w->empty;
}

// scan all features of the current frame for query window w; skip features flagged in `occ` (may be NULL).  Four features per lane are in flight at once
// (keypoint, right coordinate and descriptor requested together before any test: the scan is a latency chain of ONE wave -- round 6: the re-scan of
// k_greedy_resolve when k_window_brute made the lists, 16 dependent round trips for 1000 features in the one-at-a-time form, now four; eight in flight cost the kernel 80 more VGPRs)
template <int OS = 1>   // byte stride of the taken flags (k_resolve_wide_t keeps them in the low bytes of 16-bit words)
__device__ __forceinline__ int scan_window(const WindowProblem &P, const GridParams &g, const QueryWin &w, const Desc &dq, int n,
                                           const uint8_t *occ, int lane, u64 &k1, u64 &k2) {
    int cnt = 0;
    for (int i0 = 0; i0 < n; i0 += 4 * 64) {
        orbx_keypoint kp[4];
        Desc dc[4];
        float ur[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = min(i0 + 64 * u + lane, n - 1);
            kp[u] = gld_kp(P.kps + i);
            dc[u] = gld_desc(P.desc + (size_t)i * 32);
            ur[u] = P.u_right ? gld(P.u_right + i) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + 64 * u + lane;
            if (i >= n || (occ && occ[i * OS])) continue;
            int cx, cy;
            if (!in_window(g, w, kp[u], &cx, &cy)) continue;
            if (P.u_right && ur[u] > 0) {  // ORBmatcher.cc:92-97 / 1751-1757
                const float er = fabsf(w.xr - ur[u]);
                if (er > w.r) continue;
            }
            const int d = hamming(dq, dc[u]);
            push2(k1, k2, cand_key(d, cx, cy, i));
            cnt++;
        }
    }
    return cnt;
}

// The same scan through the grid k_grid_build left in P.gstart / P.gorder (whole wave, one query): lane c fetches the slice bounds of
// grid column cx0 + c, a wave prefix sum concatenates the slices, and the lanes walk the concatenated candidate sequence -- four
// dependent memory round trips (bounds, index, keypoint, descriptor) instead of one per 64 features of the frame (scan_window reads
// every keypoint of the frame: 16 serial round trips for 1000 features, and the re-scans were most of k_greedy_resolve's time).
// The candidates are a superset (whole grid cells); in_window() applies GetFeaturesInArea's tests and supplies the cell for the key,
// so keys -- and with them the tie-break order -- are those of scan_window.
template <int OS = 1>
__device__ __forceinline__ int scan_window_grid(const WindowProblem &P, const GridParams &g, const QueryWin &w, const Desc &dq, int n,
                                                const uint8_t *occ, int lane, u64 &k1, u64 &k2) {
    const int ncol = w.cx1 - w.cx0 + 1;   // 1..64 (make_window clamps to the grid)
    int cs = 0, len = 0;
    if (lane < ncol) {
        const int base = (w.cx0 + lane) * 48;
        cs = gld(P.gstart + base + w.cy0);
        len = (int)gld(P.gstart + base + w.cy1 + 1) - cs;
    }
    int incl = len;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int t = __shfl_up(incl, s);
        if (lane >= s) incl += t;
    }
    const int pre = incl - len, total = __shfl(incl, 63);
    int cnt = 0;
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        int j = -1;
        for (int c = 0; c < ncol; c++) {   // column of the t-th candidate: the last c with pre[c] <= t
            const int pc = __shfl(pre, c), sc = __shfl(cs, c);
            if (t >= pc) j = sc + (t - pc);
        }
        // the re-scan is a latency chain of the one wave: the keypoint, the descriptor and the right coordinate of a candidate are requested TOGETHER, before
        // any of the window tests (a lane without a candidate reads feature 0: total > 0 implies n > 0) -- three dependent round trips (bounds, index,
        // record) instead of four
        const bool v0 = t < total && j >= 0;
        const int i0 = v0 ? (int)gld(P.gorder + j) : 0;
        const bool v1 = v0 && i0 < n;
        const int i = v1 ? i0 : 0;
        const orbx_keypoint kp = gld_kp(P.kps + i);
        const Desc dc = gld_desc(P.desc + (size_t)i * 32);
        const float ur = P.u_right ? gld(P.u_right + i) : 0.f;
        if (!v1 || (occ && occ[i * OS])) continue;
        int cx, cy;
        if (!in_window(g, w, kp, &cx, &cy)) continue;
        if (P.u_right && ur > 0) {  // ORBmatcher.cc:92-97 / 1751-1757
            const float er = fabsf(w.xr - ur);
            if (er > w.r) continue;
        }
        const int d = hamming(dq, dc);
        push2(k1, k2, cand_key(d, cx, cy, i));
        cnt++;
    }
    return cnt;
}

// ---------------------------------------------------------------------------------------------------------
// Frame::AssignFeaturesToGrid (Frame.cc:385-416): one wave per frame builds the 64x48 grid as a counting sort of
// the feature indices by cell id (x * 48 + y).  Inside a cell the indices stay ascending (= insertion order), and
// cells of one grid column are contiguous, so the candidates of GetFeaturesInArea for a column range [y0,y1] are
// ONE contiguous slice of gorder -- already in the reference's enumeration order (x outer, y inner, insertion).
// grid (n_problems), block 64
// ---------------------------------------------------------------------------------------------------------
constexpr int kGridCells = 64 * 48;

// Eight keypoints per lane are in flight at once (two round trips per 512 features and pass) and the lanes of a chunk that share
// a cell are placed by claim rounds (LDS atomicMin of the lane id: the lowest lane wins, takes the cell's cursor and drops out;
// as many rounds as the largest multiplicity inside the chunk, usually one or two), which keeps insertion order.  The first form (one
// round trip per 64 features and pass, a 64-step shuffle loop per chunk for the rank inside a cell) took 51 us per 256 frames, this one
// 25 us (round 3, profiles/r03_a_chain_grid2_dpp_kernel_stats.csv).
// grid (n_problems), block 64
__global__ __launch_bounds__(64) void k_grid_build(const WindowProblem *__restrict__ probs, GridParams g) {
    __shared__ uint16_t cnt[kGridCells];
    __shared__ uint16_t start[kGridCells];
    __shared__ uint32_t claim[kGridCells];
    const WindowProblem P = probs[blockIdx.x];
    const int lane = threadIdx.x;
    const int n = gld(P.n_ptr);   // (gld / gst: the record's pointers are device memory, see above)
    for (int i = lane; i < kGridCells; i += 64) { cnt[i] = 0; claim[i] = 0xffffffffu; }
    __syncthreads();
    auto cell_of = [&](float x, float y) -> int {   // PosInGrid (Frame.cc:725-735); -1 = outside the grid
        const int px = (int)roundf((x - g.minx) * g.inv_w), py = (int)roundf((y - g.miny) * g.inv_h);
        return (px >= 0 && px < 64 && py >= 0 && py < 48) ? px * 48 + py : -1;
    };
    // pass 1: cell histogram (16-bit LDS counters packed in pairs: 32-bit atomics on the containing word)
    uint32_t *cnt32 = reinterpret_cast<uint32_t *>(cnt);
    for (int i0 = 0; i0 < n; i0 += 8 * 64) {
        float2 xy[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + 64 * k + lane;
            xy[k] = float2{-1e8f, -1e8f};   // (outside every grid; small enough that the float -> int conversion of its cell is defined)
            if (i < n) { xy[k].x = gld(&P.kps[i].x); xy[k].y = gld(&P.kps[i].y); }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = cell_of(xy[k].x, xy[k].y);
            if (c >= 0) atomicAdd(&cnt32[c >> 1], (c & 1) ? 0x10000u : 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the 3072 counters (48 per lane, then a wave scan of the lane totals)
    int lane_tot = 0;
    for (int k = 0; k < kGridCells / 64; k++) lane_tot += cnt[lane * (kGridCells / 64) + k];
    int incl = lane_tot;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int t = __shfl_up(incl, s);
        if (lane >= s) incl += t;
    }
    int run = incl - lane_tot;
    for (int k = 0; k < kGridCells / 64; k++) {
        const int c = lane * (kGridCells / 64) + k;
        start[c] = (uint16_t)run;
        gst(P.gstart + c, (uint16_t)run);
        run += cnt[c];
    }
    if (lane == 63) gst(P.gstart + kGridCells, (uint16_t)run);
    __syncthreads();
    // pass 2: stable fill, 64 features at a time in index order (eight chunks loaded per round trip)
    for (int i0 = 0; i0 < n; i0 += 8 * 64) {
        float2 xy[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + 64 * k + lane;
            xy[k] = float2{-1e8f, -1e8f};   // (outside every grid; small enough that the float -> int conversion of its cell is defined)
            if (i < n) { xy[k].x = gld(&P.kps[i].x); xy[k].y = gld(&P.kps[i].y); }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + 64 * k + lane;
            const int c = cell_of(xy[k].x, xy[k].y);
            bool todo = c >= 0;
            while (__ballot(todo)) {   // wave-uniform: one round per multiplicity of the most crowded cell of this chunk
                if (todo) atomicMin(&claim[c], (uint32_t)lane);
                one_wave_sync();   // LDS-only hand-overs of the one wave: a __syncthreads() here waits for the gorder stores' acknowledgement in every round
                const bool won = todo && claim[c] == (uint32_t)lane;
                one_wave_sync();
                if (won) {
                    gst(P.gorder + start[c], (uint16_t)i);
                    start[c] = (uint16_t)(start[c] + 1);
                    claim[c] = 0xffffffffu;
                    todo = false;
                }
                one_wave_sync();
            }
        }
    }
}

// Search windows of SearchByProjection(Frame, MapPoints) (ORBmatcher.cc:53-72) for every (frame, map point) of a batch:
// r = RadiusByViewingCos(viewCos) [* th if th != 1] * mvScaleFactors[nPredictedLevel], levels [level-1, level]; a map point
// that is not in view (mbTrackInView false) or whose predicted level is out of range is skipped.
// grid (ceil(n_mp/256), n_frames), block 256
__global__ __launch_bounds__(256) void k_mappoint_windows(int n_mp, const int32_t *__restrict__ level, const float *__restrict__ view_cos,
                                                          const uint8_t *__restrict__ in_view, const float *__restrict__ scale, int nlevels,
                                                          float th, float *__restrict__ qr, int32_t *__restrict__ qmin,
                                                          int32_t *__restrict__ qmax, uint8_t *__restrict__ qvalid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_mp) return;
    const size_t k = (size_t)blockIdx.y * n_mp + i;
    const int lvl = level[k];
    const bool ok = (!in_view || in_view[k]) && lvl >= 0 && lvl < nlevels;
    float r = (view_cos[k] > 0.998f) ? 2.5f : 4.0f;   // :146 RadiusByViewingCos
    if (th != 1.0f) r = __fmul_rn(r, th);
    qr[k] = ok ? __fmul_rn(r, scale[lvl]) : 0.f;
    qmin[k] = lvl - 1;
    qmax[k] = lvl;
    qvalid[k] = ok ? 1 : 0;
}

// candidate key of the grid scan: dist << 32 | seq << 16 | idx, seq = position in the reference's candidate enumeration
__device__ __forceinline__ u64 seq_key(int dist, int seq, int idx) {
    return ((u64)(uint32_t)dist << 32) | ((u64)(uint32_t)seq << 16) | (u64)(uint32_t)idx;
}

// grid (8 | 1, ceil(max_q / (256 / LQ)), ceil(n_problems / 8) | n_problems), block 256: LQ = 16 lanes per query (4 queries per wave) or 8 (8 queries per wave).
// (Round 4: 8 is the default -- a window of the bench's matchers holds 1 - 10 candidates, most of 16 lanes idle through the scan and the
// round-by-round extraction: kernel 74 -> 59 us alone, TUM-VI step 1.33 -> 1.26 ms, EuRoC level; 4 lanes measured like 8:
// profiles/r04_u_*, r04_v_*.  ORBX_MATCH_LANES=16 / 8 / 4.)
// For every grid column of the window the candidates are one contiguous slice of gorder; the LQ lanes stride it,
// apply GetFeaturesInArea's level / distance tests (Frame.cc:697-716) and the callers' gates, and keep their two best
// keys.  Output: the kTopK smallest candidate keys in ascending order, extracted round by round across the LQ lanes.
// If a lane that saw more than two candidates has both of its entries extracted, later rounds could miss that lane's
// third candidate, so the list is cut there (valid_len); `exhaustive` says the list holds every candidate of the
// query.  k_greedy_resolve falls back to a re-scan when it needs more than the valid part of a non-exhaustive list.
template <int LQ>   // lanes per query: 16, 8 or 4
__global__ __launch_bounds__(256) void k_window_best2_t(const WindowProblem *__restrict__ probs, GridParams g, int n_problems) {
    static_assert(LQ == 64 || LQ == 16 || LQ == 8 || LQ == 4, "lanes per query");
    const int prob = (int)(blockIdx.z * gridDim.x + blockIdx.x);   // grid (8, query blocks, problems / 8): x is the XCD, a problem's blocks share one L2
    if (prob >= n_problems) return;
    const WindowProblem P = probs[prob];
    const int sub = threadIdx.x / LQ, sl = threadIdx.x & (LQ - 1);
    const int qi = blockIdx.y * (256 / LQ) + sub;
    __shared__ int all_n[LQ == 64 ? 4 : 1];
    if (LQ == 64 && P.all_keys) {   // workgroup-uniform
        if (threadIdx.x < 4) all_n[threadIdx.x] = 0;
        __syncthreads();
    }
    const int nq = gld(P.nq_ptr);
    const bool qvalid = qi < nq;
    QueryWin w;
    Desc dq;
    u64 k1 = kNoKey, k2 = kNoKey;
    int cnt = 0;
    bool go = false;
    if (qvalid) go = load_query(P, qi, &w, g, &dq);
    if (go) {
        // The window's grid columns are contiguous slices of gorder (one per column).  Their bounds are fetched for up to eight
        // columns at once and the 16 lanes walk the CONCATENATED candidate sequence (= the reference's enumeration order, which
        // is also the tie-break order): one dependent load chain per query instead of one per column.
        int seq0 = 0;
        for (int cx = w.cx0; cx <= w.cx1; cx += 8) {
            int cs[8], ce[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int ix = min(cx + c, w.cx1);
                cs[c] = gld(P.gstart + ix * 48 + w.cy0);
                ce[c] = gld(P.gstart + ix * 48 + w.cy1 + 1);
            }
            int pre[9];
            pre[0] = 0;
#pragma unroll
            for (int c = 0; c < 8; c++) pre[c + 1] = pre[c] + ((cx + c <= w.cx1) ? ce[c] - cs[c] : 0);
            const int tot = pre[8];
            for (int t = sl; t < tot; t += LQ) {
                int j = cs[0] + t;  // column of the t-th candidate: the last c with pre[c] <= t
#pragma unroll
                for (int c = 1; c < 8; c++) j = (t >= pre[c]) ? cs[c] + (t - pre[c]) : j;
                const int s = j - t;  // so that seq0 + (j - s) = seq0 + t below
                const int i = gld(P.gorder + j);
                if (P.occupied0 && gld(P.occupied0 + i)) continue;
                const orbx_keypoint kp = gld_kp(P.kps + i);
                if (w.check_levels) {
                    if (kp.octave < w.minL) continue;
                    if (w.maxL >= 0 && kp.octave > w.maxL) continue;
                }
                const float dx = kp.x - w.x, dy = kp.y - w.y;
                if (!(fabsf(dx) < w.r && fabsf(dy) < w.r)) continue;
                if (P.inv_sigma2) {  // Fuse: chi2 gate on the reprojection error
                    const float ex = __fsub_rn(w.x, kp.x), ey = __fsub_rn(w.y, kp.y);
                    if (P.u_right && gld(P.u_right + i) >= 0) {
                        const float er = __fsub_rn(w.xr, gld(P.u_right + i));
                        const float e2 = P.chi2_fma ? __fmaf_rn(er, er, __fmaf_rn(ex, ex, __fmul_rn(ey, ey)))
                                                    : __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                        if ((double)__fmul_rn(e2, gld(P.inv_sigma2 + kp.octave)) > 7.8) continue;
                    } else {
                        const float e2 = P.chi2_fma ? __fmaf_rn(ex, ex, __fmul_rn(ey, ey)) : __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                        if ((double)__fmul_rn(e2, gld(P.inv_sigma2 + kp.octave)) > 5.99) continue;
                    }
                } else if (P.u_right && gld(P.u_right + i) > 0) {  // ORBmatcher.cc:92-97 / 1751-1757
                    const float er = fabsf(w.xr - gld(P.u_right + i));
                    if (er > w.r) continue;
                }
                const int d = hamming(dq, gld_desc(P.desc + (size_t)i * 32));
                const u64 key = seq_key(d, seq0 + (j - s), i);
                push2(k1, k2, key);
                if (LQ == 64 && P.all_keys) {   // (compile-time false for the hot forms)
                    const int slot = atomicAdd(&all_n[sub], 1);
                    if (slot < P.all_cap) gst(P.all_keys + (size_t)qi * P.all_cap + slot, key);
                }
                cnt++;
            }
            seq0 += tot;
        }
    }
    // reductions inside the LQ-lane group (xor masks LQ/2 .. 1 stay inside the group)
    int total = cnt;
#pragma unroll
    for (int s = LQ / 2; s > 0; s >>= 1) total += __shfl_xor(total, s);
    u64 out[kTopK];
    int valid_len = 0, npop = 0;
    bool cut = false;
#pragma unroll
    for (int r = 0; r < kTopK; r++) {
        u64 m = k1;
#pragma unroll
        for (int s = LQ / 2; s > 0; s >>= 1) {
            const u64 o = __shfl_xor(m, s);
            m = o < m ? o : m;
        }
        out[r] = m;
        if (m != kNoKey && !cut) valid_len = r + 1;
        const bool mine = (m != kNoKey) && (k1 == m);
        if (mine) { k1 = k2; k2 = kNoKey; npop++; }
        // a lane that ran dry while it had seen more than two candidates invalidates everything after this round
        int dry = (mine && npop == 2 && cnt > 2) ? 1 : 0;
#pragma unroll
        for (int s = LQ / 2; s > 0; s >>= 1) dry |= __shfl_xor(dry, s);
        cut = cut || (dry != 0);
    }
    if (qvalid && sl == 0) {
#pragma unroll
        for (int r = 0; r < kTopK; r++) gst(P.keys + (size_t)qi * kTopK + r, out[r]);
        gst(P.meta + qi, valid_len | ((total <= valid_len) ? 256 : 0) | ((total == 0) ? 512 : 0));   // bit 9: GetFeaturesInArea returned nothing
        if (LQ == 64 && P.all_keys) gst(P.all_cnt + qi, total);
    }
}

// k_window_brute (round 6): the same lists as k_window_best2_t WITHOUT the frame's grid, for ONE small problem (a single host-pointer call with a
// thousand queries into a thousand features): a wave per query walks every feature of the frame, applies PosInGrid + GetFeaturesInArea's tests itself
// (in_window: the candidate's grid cell comes out of that, so the key orders candidates exactly as the grid enumeration does: cell x, cell y, index) and
// the caller's gates, and extracts the kTopK smallest keys.  It replaces k_grid_build (one wave, 12 us: a counting sort the call then uses once) +
// k_window_best2_t (7 - 9 us) by one launch of ~5 us; k_greedy_resolve's re-scan takes the grid-less scan_window when the record has no grid.
// grid (ceil(nq / 4)), block 256
__global__ __launch_bounds__(256) void k_window_brute(const WindowProblem *__restrict__ probs, GridParams g) {
    const WindowProblem P = probs[0];
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nq = gld(P.nq_ptr), n = gld(P.n_ptr);
    if (qi >= nq) return;   // wave-uniform
    QueryWin w;
    Desc dq;
    u64 k1 = kNoKey, k2 = kNoKey;
    int cnt = 0;
    if (load_query(P, qi, &w, g, &dq)) {
        for (int i = lane; i < n; i += 64) {
            if (P.occupied0 && gld(P.occupied0 + i)) continue;
            const orbx_keypoint kp = gld_kp(P.kps + i);
            int cx, cy;
            if (!in_window(g, w, kp, &cx, &cy)) continue;
            if (P.inv_sigma2) {  // Fuse: chi2 gate on the reprojection error (as k_window_best2_t)
                const float ex = __fsub_rn(w.x, kp.x), ey = __fsub_rn(w.y, kp.y);
                if (P.u_right && gld(P.u_right + i) >= 0) {
                    const float er = __fsub_rn(w.xr, gld(P.u_right + i));
                    const float e2 = P.chi2_fma ? __fmaf_rn(er, er, __fmaf_rn(ex, ex, __fmul_rn(ey, ey)))
                                                : __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                    if ((double)__fmul_rn(e2, gld(P.inv_sigma2 + kp.octave)) > 7.8) continue;
                } else {
                    const float e2 = P.chi2_fma ? __fmaf_rn(ex, ex, __fmul_rn(ey, ey)) : __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                    if ((double)__fmul_rn(e2, gld(P.inv_sigma2 + kp.octave)) > 5.99) continue;
                }
            } else if (P.u_right && gld(P.u_right + i) > 0) {  // ORBmatcher.cc:92-97 / 1751-1757
                const float er = fabsf(w.xr - gld(P.u_right + i));
                if (er > w.r) continue;
            }
            const int d = hamming(dq, gld_desc(P.desc + (size_t)i * 32));
            push2(k1, k2, cand_key(d, cx, cy, i));
            cnt++;
        }
    }
    int total = cnt;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) total += __shfl_xor(total, s);
    u64 out[kTopK];
    int valid_len = 0, npop = 0;
    bool cut = false;
#pragma unroll
    for (int r = 0; r < kTopK; r++) {
        const u64 m = wave_min1(k1);
        out[r] = m;
        if (m != kNoKey && !cut) valid_len = r + 1;
        const bool mine = (m != kNoKey) && (k1 == m);
        if (mine) { k1 = k2; k2 = kNoKey; npop++; }
        // a lane that ran dry while it had seen more than two candidates invalidates everything after this round (as k_window_best2_t)
        const bool dry = mine && npop == 2 && cnt > 2;
        cut = cut || (__ballot(dry) != 0ull);
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < kTopK; r++) gst(P.keys + (size_t)qi * kTopK + r, out[r]);
        gst(P.meta + qi, valid_len | ((total <= valid_len) ? 256 : 0) | ((total == 0) ? 512 : 0));
    }
}
// the largest (queries x features) a single call hands to k_window_brute (beyond it the grid pays for itself)
constexpr size_t kBruteMaxPairs = (size_t)5 << 19;   // 2.6 M

struct ResolveProblem {
    int mode;                 // 1 = SearchByProjection(Frame, MapPoints) (M1), 2 = SearchByProjection(Cur, Last) (M2)
    float nnratio;            // M1
    float max_dist;           // accept threshold on the best distance (TH_HIGH, ORBdist, TH_LOW*ratioHamming ...)
    int check_orientation;    // M2
    int cleared_value;        // what match[i] becomes when the rotation check clears an assignment: -1, or -2 so that the caller can tell
                              // "never assigned" (slot keeps its old content) from "assigned, then set to NULL" (ORBmatcher.cc:1871-1881)
    const float *q_angle;     // M2 (NULL with q_from_kps)
    const uint8_t *q_has_obs; // NULL = all true
    int32_t *match;           // [n] query index per feature or -1
    int32_t *nmatches;        // scalar out
    int32_t *entries;         // [nq] scratch: rotation histogram entries bin << 16 | feature (M2)
};

// One wave per problem: exact replay of the reference's sequential query loop.
// The dependency between queries is only the taken-mask (occ).  Queries are processed 64 at a time: every lane
// holds its query's sorted candidate list and picks the first (M1: first two) candidates that are still free.
// A lane "conflicts" when an earlier lane of the chunk currently wants the same feature.  All lanes before the
// first conflicting lane commit in parallel (their choices cannot influence each other); the conflicting lane simply
// re-evaluates in the next round against the updated mask.  Only when a lane exhausts the valid part of a
// non-exhaustive list does the whole wave re-scan that query's window against the current mask -- exactly what the
// sequential loop would have seen.  Dynamic LDS: claim[n_alloc] (u32) + angle[n_alloc] (f32) + occ[n_alloc] (u8) + octave[n_alloc] (u8);
// nothing inside the round loop touches global memory except fire-and-forget result stores: the octaves of the ratio test (M1) are staged in
// LDS with the angles, a query's has-observations flag arrives with its candidate list (one chunk ahead), and the hand-overs between the rounds
// are LDS-only orderings of the ONE wave (one_wave_sync) -- a __syncthreads() there is s_waitcnt vmcnt(0): every round waited for the
// acknowledgement of its result stores (round 5: ~45 rounds per 1000 queries of the bench's frame pairs, 170 per 10 000 map points).
// BRUTE: the problems may come without a grid (k_window_brute made the lists: single small host-pointer calls) -- the re-scan then walks all features
// (scan_window).  An instantiation of its own because that scan keeps four features per lane in flight: 31 more VGPRs than the batched pipeline's form,
// which shares the machine with the next batch's extraction, has to carry.
template <bool BRUTE>
__global__ __launch_bounds__(64) void k_greedy_resolve_t(const WindowProblem *__restrict__ probs, const ResolveProblem *__restrict__ res,
                                                       GridParams g, int n_alloc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ int hist[ORBX_HISTO_LENGTH + 2];
    uint32_t *claim = reinterpret_cast<uint32_t *>(lds);
    float *ang = reinterpret_cast<float *>(lds + (size_t)n_alloc * 4);
    uint8_t *occ = lds + (size_t)n_alloc * 8;
    uint8_t *oct = occ + n_alloc;
    const WindowProblem P = probs[blockIdx.x];
    const ResolveProblem R = res[blockIdx.x];
    const int lane = threadIdx.x;
    const u64 lt_mask = (1ull << lane) - 1ull;
    const int n = min(gld(P.n_ptr), n_alloc), nq = gld(P.nq_ptr);
    for (int i0 = 0; i0 < n; i0 += 8 * 64) {   // eight loads in flight per lane: two memory round trips for 1000 features, not sixteen
        float a[8];
        int l[8];
        uint8_t o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + 64 * k + lane;
            a[k] = i < n ? gld(&P.kps[i].angle) : 0.f;
            l[k] = i < n ? gld(&P.kps[i].octave) : 0;
            o[k] = (i < n && P.occupied0) ? gld(P.occupied0 + i) : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + 64 * k + lane;
            if (i < n) {
                occ[i] = o[k];
                oct[i] = (uint8_t)l[k];   // octaves are 0 .. nlevels - 1 < 256; compared for equality only
                claim[i] = 0xffffffffu;
                ang[i] = a[k];
                gst(R.match + i, -1);
            }
        }
    }
    if (lane < ORBX_HISTO_LENGTH) hist[lane] = 0;
    __syncthreads();
    int nmatches = 0, n_entries = 0;
    const float factor = 1.0f / ORBX_HISTO_LENGTH;
    const bool ori = (R.mode == 2 && R.check_orientation);
    const bool two = (R.mode == 1);

    // decide whether (k1,k2) yields a match; M1 applies the ratio test (ORBmatcher.cc:123-139)
    auto accept = [&](u64 k1, u64 k2) -> bool {
        if (k1 == kNoKey) return false;
        const int bestDist = (int)(k1 >> 32);
        if ((float)bestDist > R.max_dist) return false;
        if (two) {
            const int bestDist2 = (k2 == kNoKey) ? 256 : (int)(k2 >> 32);
            const int bestLevel = oct[(int)(k1 & 0xffff)];
            const int bestLevel2 = (k2 == kNoKey) ? -1 : (int)oct[(int)(k2 & 0xffff)];
            if (bestLevel == bestLevel2 && (float)bestDist > R.nnratio * (float)bestDist2) return false;
            if (!(bestLevel != bestLevel2 || (float)bestDist <= R.nnratio * (float)bestDist2)) return false;
        }
        return true;
    };
    auto rot_bin = [&](float qa, int idx) -> int {  // :1775-1792
        float rot = qa - ang[idx];
        if (rot < 0.0f) rot += 360.0f;
        int b = (int)roundf(rot * factor);
        if (b == ORBX_HISTO_LENGTH) b = 0;
        return b;
    };

    // candidate lists of a chunk of 64 queries; the next chunk's are requested before the current chunk is replayed
    struct Chunk { u64 L[kTopK]; int meta; float q_ang; uint8_t obs; };
    auto fetch = [&](int q0) -> Chunk {
        Chunk c;   // inactive lane: empty exhaustive list
#pragma unroll
        for (int e = 0; e < kTopK; e++) c.L[e] = kNoKey;
        c.meta = 256; c.q_ang = 0.f; c.obs = 1;
        const int qi = q0 + lane;
        if (qi < nq) {
            const u64 *kp = P.keys + (size_t)qi * kTopK;
#pragma unroll
            for (int e = 0; e < kTopK; e++) c.L[e] = gld(kp + e);
            c.meta = gld(P.meta + qi);
            if (ori) c.q_ang = P.q_from_kps ? gld(&P.q_from_kps[qi].angle) : gld(R.q_angle + qi);
            if (R.q_has_obs) c.obs = gld(R.q_has_obs + qi);
        }
        return c;
    };
    Chunk nxt = fetch(0);
    for (int q0 = 0; q0 < nq; q0 += 64) {
        const int qi = q0 + lane;
        const bool active = qi < nq;
        const Chunk C = nxt;
        // the wait for this chunk's lists goes HERE, before the next chunk's are requested: placed at their first use further down it is a vmcnt(0) across the
        // loop's back edge, which also waits for the requests just made -- the lists would arrive synchronously again
#pragma unroll
        for (int e = 0; e < kTopK; e++) asm volatile("" :: "v"(C.L[e]) : "memory");
        asm volatile("" :: "v"(C.meta), "v"(C.q_ang), "v"((int)C.obs) : "memory");
        nxt = fetch(q0 + 64);
        const int valid_len = C.meta & 0xff;
        const bool exhaustive = (C.meta & 256) != 0;
        const float q_ang = C.q_ang;
        const uint8_t q_obs = C.obs;
        int pos = 0;
        while (pos < 64) {
            const bool live = active && lane >= pos;
            // first (two) still-free candidates of the valid list
            u64 c1 = kNoKey, c2 = kNoKey;
            bool need_slow = false;
            if (live) {
#pragma unroll
                for (int e = 0; e < kTopK; e++) {
                    const u64 k = C.L[e];
                    if (e < valid_len && !occ[(int)(k & 0xffff)]) {
                        if (c1 == kNoKey) c1 = k;
                        else if (c2 == kNoKey) c2 = k;
                    }
                }
                if (!exhaustive) {
                    if (c1 == kNoKey) need_slow = true;                                   // best unknown
                    else if (two && c2 == kNoKey && (float)(int)(c1 >> 32) <= R.max_dist) need_slow = true;  // second-best unknown
                }
            }
            const bool has = live && !need_slow && c1 != kNoKey && (float)(int)(c1 >> 32) <= R.max_dist;
            const int t1 = has ? (int)(c1 & 0xffff) : -1;
            const int t2 = (has && two && c2 != kNoKey) ? (int)(c2 & 0xffff) : -1;
            if (has) atomicMin(&claim[t1], (uint32_t)lane);
            one_wave_sync();
            const bool conflict = need_slow || (has && (claim[t1] != (uint32_t)lane || (t2 >= 0 && claim[t2] < (uint32_t)lane)));
            one_wave_sync();
            if (has) claim[t1] = 0xffffffffu;
            const u64 cb = __ballot(conflict);
            const int c = cb ? (__ffsll((long long)cb) - 1) : 64;
            // parallel commit of the conflict-free prefix [pos, c)
            const bool ok = has && lane < c && accept(c1, c2);
            const u64 okb = __ballot(ok);
            if (ok) {
                gst(R.match + t1, qi);
                occ[t1] = q_obs;
                if (ori) {
                    const int b = rot_bin(q_ang, t1);
                    gst(R.entries + n_entries + __popcll(okb & lt_mask), (b << 16) | t1);  // rotHist[bin].push_back(bestIdx2)
                    atomicAdd(&hist[b], 1);
                }
            }
            nmatches += __popcll(okb);
            if (ori) n_entries += __popcll(okb);
            one_wave_sync();
            if (c >= 64) break;
            const bool slow = (__shfl((int)need_slow, c) != 0);
            if (!slow) { pos = c; continue; }  // claim conflict: lane c re-evaluates against the updated mask
            {   // list exhausted: re-scan query q0+c against the occupancy the sequential loop sees at this point
                const int qc = q0 + c;
                const float qa_c = __shfl(q_ang, c);  // all lanes participate in the shuffle
                const int obs_c = __shfl((int)q_obs, c);
                QueryWin w;
                Desc dq;
                u64 r1 = kNoKey, r2 = kNoKey;
                if (load_query_eager(P, qc, &w, g, &dq)) {
                    if (!BRUTE || P.gstart) scan_window_grid(P, g, w, dq, n, occ, lane, r1, r2);
                    else scan_window(P, g, w, dq, n, occ, lane, r1, r2);   // a record without a grid (k_window_brute made its lists): wave-uniform
                    wave_min2(r1, r2);
                }
                if (accept(r1, r2)) {
                    const int idx = (int)(r1 & 0xffff);
                    if (lane == 0) {
                        gst(R.match + idx, qc);
                        occ[idx] = (uint8_t)obs_c;
                        if (ori) {
                            const int b = rot_bin(qa_c, idx);
                            gst(R.entries + n_entries, (b << 16) | idx);
                            hist[b]++;
                        }
                    }
                    nmatches++;
                    if (ori) n_entries++;
                }
                one_wave_sync();
            }
            pos = c + 1;
        }
    }
    __syncthreads();
    if (ori) {
        // ComputeThreeMaxima :2012-2053
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < ORBX_HISTO_LENGTH; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        // :1871-1881: every entry of a rejected bin clears its feature and decrements nmatches
        for (int e = lane; e < n_entries; e += 64) {
            const int v = gld(R.entries + e), b = v >> 16;
            if (b != ind1 && b != ind2 && b != ind3) gst(R.match + (v & 0xffff), R.cleared_value);
        }
        int dropped = 0;
        for (int i = 0; i < ORBX_HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3) dropped += hist[i];
        nmatches -= dropped;
    }
    if (lane == 0) gst(R.nmatches, nmatches);
}

// Workgroup barrier that orders LDS accesses ONLY (the address-space form of the fence: s_waitcnt lgkmcnt(0) + s_barrier).  __syncthreads() is also a
// release of global memory -- s_waitcnt vmcnt(0) -- so a round of the replay below would wait for the acknowledgement of its result stores.
__device__ __forceinline__ void wg_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// k_resolve_wide_t (round 6): the same replay of SearchByProjection's sequential query loop (ORBmatcher.cc:88-139, 1745-1800) as k_greedy_resolve_t, by a
// WORKGROUP of WAVES waves per problem: 64 * WAVES queries of a chunk in flight, one query per thread, and no restriction to a conflict-free PREFIX (the
// one-wave form commits the lanes before the first conflicting lane and starts a new round for every conflict: ~45 rounds + 6 wave-wide re-scans, one after
// the other, per 1000 queries of the bench's frame pairs -- 99 us of one wave's dependent instructions).  A query's decision reads the first (M1: first two)
// still-free entries of its candidate list, its PICKS p1 (, p2); it commits to p1.  A round:
//   claim     every unresolved query q with a decision to make: atomicMin(claim1[p1], q)                                                          | barrier
//   unstable  q is CLEAN when no earlier query picks its picks (claim1[p1] == q, claim1[p2] >= q).  A query that is not clean will move to another entry
//             of its list once the earlier one has committed: it is UNSTABLE and publishes every free entry of its list, atomicMin(claim2[entry], q) | barrier
//   closure   a clean query one of whose picks an EARLIER unstable query lists (claim2[pick] < q) may lose that pick: it becomes unstable too and publishes
//             its list; repeated until a pass adds nobody (one pass, rarely two)                                                                    | barrier per pass
//   fence     an unstable query must also be certain to stay INSIDE its list: its list is exhaustive, or it has as many free entries as its decision reads
//             that nobody earlier picks or lists.  The lowest unstable query that is not -- and the lowest query whose non-exhaustive list ran dry --
//             stops every later query for this round.
//   commit    all clean, never-threatened queries below the fence at once (what earlier queries commit to in the same round are their picks, which are
//             not this query's): ratio test, match[], the taken-mask, the rotation entries; claims cleared                                          | barrier
//   re-scan   queries whose non-exhaustive list ran dry are re-scanned against the CURRENT mask, one per wave, all waves at once: a fresh list of the kTopK
//             best FREE candidates (the mask only grows, so it stays a valid list).  The lowest unresolved query sees the mask the sequential loop sees
//             and is always served, so every round resolves at least that one.                                                                      | barrier
// "No match" outcomes that cannot change (exhaustive list all taken; best free distance above the threshold -- later states only have worse bests) resolve at
// once.  A round is bound by the issue rate of ONE wave's instruction stream (every wave runs the same code): the phases are branch-free, the list entries
// packed (distance << 16 | feature), taken flag and octave of a feature share a 16-bit LDS word.  Barriers order LDS only (wg_lds_sync); result stores are
// fire-and-forget.  The rotation histogram (:1775-1792) is built after the rounds from the committed (query, feature) pairs -- the angles never enter LDS.
// grid (n_problems), block 64 * WAVES, dynamic LDS resolve_lds_bytes(n_alloc): claim1 u32 + claim2 u32 + (taken | octave << 8) u16 per feature.
// With check_orientation the query index travels in 16 bits of an entry: nq <= 65535 (the host falls back to k_greedy_resolve_t beyond).
template <int WAVES, bool BRUTE>
__global__ __launch_bounds__(64 * WAVES) void k_resolve_wide_t(const WindowProblem *__restrict__ probs, const ResolveProblem *__restrict__ res,
                                                               GridParams g, int n_alloc) {
    constexpr int T = 64 * WAVES;
    constexpr uint32_t kNone = 0xffffffffu;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // static LDS stays small: with the largest frame (kMaxResolveFeatures) the dynamic part is 160 064 of the 163 840 bytes of a workgroup
    constexpr int kPool = 64;              // re-scans per round (slot 0 is kept for the LOWEST query that needs one)
    __shared__ int hist[ORBX_HISTO_LENGTH + 2];
    __shared__ uint32_t rl[kPool * kTopK]; // refreshed lists of the re-scanned queries (packed entries)
    __shared__ int rl_exh[kPool];
    __shared__ uint16_t rq[kPool];         // queries (thread ids) to re-scan this round
    // two sets of round counters, used by alternate rounds (a set is cleared in the round after the one that read it):
    //   0 fence | 1 unresolved after the round | 2 re-scan requests besides the lowest | 3 lowest query that needs a re-scan | 4 any unstable query
    //   5..7 "a closure pass added somebody" (pass i uses 5 + i % 3);  [16] nmatches, [17] rotation entries
    __shared__ int ctl[18];
    uint32_t *claim1 = reinterpret_cast<uint32_t *>(lds);
    uint32_t *claim2 = claim1 + n_alloc;
    uint16_t *oo = reinterpret_cast<uint16_t *>(lds + (size_t)n_alloc * 8);   // low byte: taken, high byte: octave
    const WindowProblem P = probs[blockIdx.x];
    const ResolveProblem R = res[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = min(gld(P.n_ptr), n_alloc), nq = gld(P.nq_ptr);
    for (int i0 = 0; i0 < n; i0 += 4 * T) {   // four loads in flight per thread
        int l[4];
        uint8_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + T * k + tid;
            l[k] = i < n ? gld(&P.kps[i].octave) : 0;
            o[k] = (i < n && P.occupied0) ? gld(P.occupied0 + i) : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + T * k + tid;
            if (i < n) {
                oo[i] = (uint16_t)((o[k] ? 1u : 0u) | ((uint32_t)(l[k] & 0xff) << 8));   // octaves are 0 .. nlevels - 1 < 256; compared for equality only
                claim1[i] = kNone;
                claim2[i] = kNone;
                gst(R.match + i, -1);
            }
        }
    }
    if (tid < ORBX_HISTO_LENGTH) hist[tid] = 0;
    if (tid < 18) ctl[tid] = ((tid & 7) == 0 || (tid & 7) == 3) && tid < 16 ? 0x7fffffff : 0;
    __syncthreads();   // (also: the -1 fills are acknowledged before any wave stores a match into the same array)
    const bool ori = (R.mode == 2 && R.check_orientation);
    const bool two = (R.mode == 1);
    const int need_own = two ? 2 : 1;

    // the whole wave re-scans the window of query qc against the current mask: kTopK best free candidates -> pool slot (as k_window_brute extracts them)
    auto rescan = [&](int qc, int slot) {
        QueryWin w;
        Desc dq;
        u64 k1 = kNoKey, k2 = kNoKey;
        int cnt = 0;
        const uint8_t *occ8 = reinterpret_cast<const uint8_t *>(oo);   // the scans look at occ[i]: byte 2 i of the shared words
        if (load_query_eager(P, qc, &w, g, &dq)) {
            if (!BRUTE || P.gstart) cnt = scan_window_grid<2>(P, g, w, dq, n, occ8, lane, k1, k2);
            else cnt = scan_window<2>(P, g, w, dq, n, occ8, lane, k1, k2);   // a record without a grid: wave-uniform
        }
        int total = cnt;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) total += __shfl_xor(total, s);
        int valid_len = 0, npop = 0;
        bool cut = false;
#pragma unroll
        for (int r = 0; r < kTopK; r++) {
            const u64 m = wave_min1(k1);
            const bool keep = m != kNoKey && !cut;
            if (lane == 0) rl[slot * kTopK + r] = keep ? ((uint32_t)(m >> 32) << 16) | (uint32_t)(m & 0xffff) : kNone;
            if (keep) valid_len = r + 1;
            const bool mine = (m != kNoKey) && (k1 == m);
            if (mine) { k1 = k2; k2 = kNoKey; npop++; }
            const bool dry = mine && npop == 2 && cnt > 2;   // a lane that ran dry while it had seen more: everything after this round is unknown
            cut = cut || (__ballot(dry) != 0ull);
        }
        if (lane == 0) rl_exh[slot] = total <= valid_len ? 1 : 0;
    };

    struct Chunk { u64 L[kTopK]; int meta; uint8_t obs; };
    auto fetch = [&](int q0) -> Chunk {
        Chunk c;
#pragma unroll
        for (int e = 0; e < kTopK; e++) c.L[e] = kNoKey;
        c.meta = 256; c.obs = 1;
        const int qi = q0 + tid;
        if (qi < nq) {
            const u64 *kp = P.keys + (size_t)qi * kTopK;
#pragma unroll
            for (int e = 0; e < kTopK; e++) c.L[e] = gld(kp + e);
            c.meta = gld(P.meta + qi);
            if (R.q_has_obs) c.obs = gld(R.q_has_obs + qi);
        }
        return c;
    };
    auto pack = [&](u64 k, bool valid) -> uint32_t { return (valid && k != kNoKey) ? ((uint32_t)(k >> 32) << 16) | (uint32_t)(k & 0xffff) : kNone; };
    Chunk nxt = fetch(0);
    int par = 0;   // counter set of the round
    for (int q0 = 0; q0 < nq; q0 += T) {
        const int qi = q0 + tid;
        const Chunk C = nxt;
#pragma unroll
        for (int e = 0; e < kTopK; e++) asm volatile("" :: "v"(C.L[e]) : "memory");   // wait here, not across the next request
        asm volatile("" :: "v"(C.meta), "v"((int)C.obs) : "memory");
        nxt = fetch(q0 + T);
        const int vl = C.meta & 0xff;
        uint32_t E[kTopK];   // the list: distance << 16 | feature, kNone = no entry
#pragma unroll
        for (int e = 0; e < kTopK; e++) E[e] = pack(C.L[e], vl > e);
        bool exhaustive = (C.meta & 256) != 0;
        bool resolved = qi >= nq;
        for (;; par ^= 8) {
            int *ct = ctl + par;
            // ---- claim: the picks under the current mask ----
            int I[kTopK];
            uint32_t O[kTopK];
#pragma unroll
            for (int e = 0; e < kTopK; e++) I[e] = E[e] != kNone ? (int)(E[e] & 0xffff) : 0;
#pragma unroll
            for (int e = 0; e < kTopK; e++) O[e] = oo[I[e]];
            uint32_t fm = 0u;   // bit e: entry e is free
#pragma unroll
            for (int e = 0; e < kTopK; e++) fm |= (E[e] != kNone && !(O[e] & 0xff)) ? 1u << e : 0u;
            if (resolved) fm = 0u;
            const uint32_t fm2 = fm & (fm - 1u);
            const int k1 = fm ? __builtin_ctz(fm) : kTopK, k2 = fm2 ? __builtin_ctz(fm2) : kTopK;
            uint32_t p1 = kNone, p2 = kNone, op1 = 0u, op2 = 0u;
#pragma unroll
            for (int e = 0; e < kTopK; e++) {
                p1 = k1 == e ? E[e] : p1; op1 = k1 == e ? O[e] : op1;
                p2 = k2 == e ? E[e] : p2; op2 = k2 == e ? O[e] : op2;
            }
            const bool best_ok = p1 != kNone && (float)(int)(p1 >> 16) <= R.max_dist;
            const bool decided_no = !resolved && (p1 == kNone ? exhaustive : !best_ok);   // cannot change: the best free distance only grows
            const bool dry = !resolved && !exhaustive && (p1 == kNone || (two && p2 == kNone && best_ok));   // best / second best unknown
            const bool has = !resolved && !decided_no && !dry;
            const int t1 = (int)(p1 & 0xffff), t2 = (two && p2 != kNone) ? (int)(p2 & 0xffff) : -1;
            if (has) atomicMin(&claim1[t1], (uint32_t)tid);
            if (dry) { atomicMin(&ct[0], tid); atomicMin(&ct[3], tid); }
            wg_lds_sync();
            if (tid == 0) {   // the other set: last read before this barrier, next written after the round's last
                int *o = ctl + (par ^ 8);
                o[0] = 0x7fffffff; o[1] = 0; o[2] = 0; o[3] = 0x7fffffff; o[4] = 0; o[5] = 0; o[6] = 0; o[7] = 0;
            }
            // ---- unstable: a pick that an earlier query picks ----
            uint32_t C1[kTopK];
#pragma unroll
            for (int e = 0; e < kTopK; e++) C1[e] = claim1[I[e]];
            uint32_t cp1 = 0u, cp2 = 0u, own1 = 0u;   // own1: free entries that no earlier query picks (for the fence)
#pragma unroll
            for (int e = 0; e < kTopK; e++) {
                cp1 = k1 == e ? C1[e] : cp1;
                cp2 = k2 == e ? C1[e] : cp2;
                own1 |= C1[e] >= (uint32_t)tid ? 1u << e : 0u;
            }
            own1 &= fm;
            bool unstable = has && !(cp1 == (uint32_t)tid && (t2 < 0 || cp2 >= (uint32_t)tid));
            bool published = false;
            auto publish = [&]() {
#pragma unroll
                for (int e = 0; e < kTopK; e++)
                    if ((fm >> e) & 1u) atomicMin(&claim2[I[e]], (uint32_t)tid);
                published = true;
            };
            if (unstable) { publish(); ct[4] = 1; }
            wg_lds_sync();
            if (ct[4]) {   // workgroup-uniform
                // ---- closure: clean queries whose picks an earlier unstable query lists; fence ----
                for (int pass = 0;; pass++) {
                    const int fl = 5 + pass % 3;
                    uint32_t C2[kTopK];
#pragma unroll
                    for (int e = 0; e < kTopK; e++) C2[e] = claim2[I[e]];
                    uint32_t d1 = kNone, d2 = kNone, own = 0u;
#pragma unroll
                    for (int e = 0; e < kTopK; e++) {
                        d1 = k1 == e ? C2[e] : d1;
                        d2 = k2 == e ? C2[e] : d2;
                        own |= C2[e] >= (uint32_t)tid ? 1u << e : 0u;
                    }
                    if (has && !unstable && (d1 < (uint32_t)tid || (t2 >= 0 && d2 < (uint32_t)tid))) { unstable = true; publish(); ct[fl] = 1; }
                    if (unstable && !exhaustive && __popc(own & own1) < need_own) atomicMin(&ct[0], tid);   // may leave its list: nothing later commits this round
                    if (tid == 0) ct[5 + (pass + 1) % 3] = 0;
                    wg_lds_sync();
                    if (!ct[fl]) break;   // nobody was added in this pass: every test above saw every published list
                }
            }
            // ---- commit ----
            const int fence = ct[0];
            int my_slot = -1;
            if (!resolved) {
                bool done = decided_no;
                if (has && !unstable && tid < fence) {
                    done = true;
                    const int bestDist = (int)(p1 >> 16);
                    bool ok = true;
                    if (two) {   // ORBmatcher.cc:123-139
                        const int bestDist2 = p2 == kNone ? 256 : (int)(p2 >> 16);
                        const int bestLevel = (int)(op1 >> 8), bestLevel2 = p2 == kNone ? -1 : (int)(op2 >> 8);
                        if (bestLevel == bestLevel2 && (float)bestDist > R.nnratio * (float)bestDist2) ok = false;
                        if (!(bestLevel != bestLevel2 || (float)bestDist <= R.nnratio * (float)bestDist2)) ok = false;
                    }
                    if (ok) {
                        gst(R.match + t1, qi);
                        reinterpret_cast<uint8_t *>(oo)[2 * t1] = C.obs;
                        atomicAdd(&ctl[16], 1);
                        if (ori) gst(R.entries + atomicAdd(&ctl[17], 1), (int32_t)(((uint32_t)qi << 16) | (uint32_t)t1));
                    }
                }
                if (has) claim1[t1] = kNone;
                if (published) {
#pragma unroll
                    for (int e = 0; e < kTopK; e++)
                        if ((fm >> e) & 1u) claim2[I[e]] = kNone;
                }
                if (!done) {
                    atomicAdd(&ct[1], 1);
                    if (dry) {
                        my_slot = tid == ct[3] ? 0 : 1 + atomicAdd(&ct[2], 1);
                        if (my_slot < kPool) rq[my_slot] = (uint16_t)tid;
                    }
                }
                resolved = done;
            }
            wg_lds_sync();
            const int n_unres = ct[1], rq_n = ct[3] == 0x7fffffff ? 0 : min(1 + ct[2], kPool);
            if (n_unres == 0) { par ^= 8; break; }   // workgroup-uniform
            if (rq_n > 0) {
                for (int i = wave; i < rq_n; i += WAVES) rescan(q0 + (int)rq[i], i);
                wg_lds_sync();
                if (my_slot >= 0 && my_slot < kPool) {   // (a request beyond the pool keeps its list: it asks again next round)
#pragma unroll
                    for (int e = 0; e < kTopK; e++) E[e] = rl[my_slot * kTopK + e];
                    exhaustive = rl_exh[my_slot] != 0;
                }
            }
        }
    }
    __syncthreads();
    int nmatches = ctl[16];
    if (ori) {
        // rotation histogram :1775-1792 from the committed pairs; entries become (bin, feature)
        const int n_entries = ctl[17];
        const float factor = 1.0f / ORBX_HISTO_LENGTH;
        for (int e = tid; e < n_entries; e += T) {
            const uint32_t v = (uint32_t)gld(R.entries + e);
            const int q = (int)(v >> 16), t1 = (int)(v & 0xffff);
            const float qa = P.q_from_kps ? gld(&P.q_from_kps[q].angle) : gld(R.q_angle + q);
            float rot = qa - gld(&P.kps[t1].angle);
            if (rot < 0.0f) rot += 360.0f;
            int b = (int)roundf(rot * factor);
            if (b == ORBX_HISTO_LENGTH) b = 0;
            atomicAdd(&hist[b], 1);
            gst(R.entries + e, (b << 16) | t1);
        }
        __syncthreads();
        // ComputeThreeMaxima :2012-2053
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < ORBX_HISTO_LENGTH; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        // :1871-1881: every entry of a rejected bin clears its feature and decrements nmatches (a thread re-reads the entries it wrote itself)
        for (int e = tid; e < n_entries; e += T) {
            const int v = gld(R.entries + e), b = v >> 16;
            if (b != ind1 && b != ind2 && b != ind3) gst(R.match + (v & 0xffff), R.cleared_value);
        }
        int dropped = 0;
        for (int i = 0; i < ORBX_HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3) dropped += hist[i];
        nmatches -= dropped;
    }
    if (tid == 0) gst(R.nmatches, nmatches);
}

// ---------------------------------------------------------------------------------------------------------
// Matchers whose inner loop carries more state than the candidate lists of k_window_best2 can encode
// (SearchForInitialization's vMatchedDistance, the BoW merge-joins): ONE WAVE replays the reference's query loop in its
// own order; the 64 lanes evaluate the candidates of the current query in parallel (Hamming distance on the fly) against
// the state left by the queries before it, and a packed-key wave minimum gives "first (or last) minimum in the
// reference's candidate order" plus the second-best distance.  Everything stays on the device.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int dev_rot_bin(float a1, float a2) {  // e.g. ORBmatcher.cc:337-343
    float rot = a1 - a2;
    if (rot < 0.0f) rot += 360.0f;
    int b = (int)roundf(rot * (1.0f / ORBX_HISTO_LENGTH));
    if (b == ORBX_HISTO_LENGTH) b = 0;
    return b;
}
// ComputeThreeMaxima (ORBmatcher.cc:2012-2053)
__device__ __forceinline__ void dev_three_maxima(const int *hist, int &ind1, int &ind2, int &ind3) {
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < ORBX_HISTO_LENGTH; i++) {  // selects instead of branches: the state stays in registers
        const int s = hist[i];
        const bool a = s > max1, b = !a && s > max2, c = !a && !b && s > max3;
        const int n3 = (a || b) ? max2 : (c ? s : max3), j3 = (a || b) ? i2 : (c ? i : i3);
        const int n2 = a ? max1 : (b ? s : max2), j2 = a ? i1 : (b ? i : i2);
        max1 = a ? s : max1; i1 = a ? i : i1;
        max2 = n2; i2 = j2; max3 = n3; i3 = j3;
    }
    if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
    ind1 = i1; ind2 = i2; ind3 = i3;
}

// ---------------------------------------------------------------------------------------------------------
// Fisheye-stereo twins (F.Nleft != -1) of the two frame projection matchers: every query searches the LEFT camera's grid and
// then, in the same loop iteration, the RIGHT camera's grid; both searches read and write ONE occupancy / result array over the
// combined feature index space [0, n_left) left, [n_left, n_left + n_right) right.
//   mode 1  SearchByProjection(Frame, MapPoints) ORBmatcher.cc:43-213: ratio test between the two best free candidates of one
//           level (:123-139, :187-195); an accepted match is also written to the stereo partner's slot through
//           mvLeftToRightMatch / mvRightToLeftMatch and counts twice (:131-135, :197-201); a LEFT ratio rejection `continue`s
//           past the right twin (:125-126); the twin needs mbTrackInViewR and mnTrackScaleLevelR != -1 (:144-146: folded into
//           the right problem's qvalid) and uses RadiusByViewingCos without the th factor (:147: folded into its qr)
//   mode 2  SearchByProjection(Frame, Frame) :1676-1887 with the twin :1794-1863: best free candidate <= TH_HIGH on either side,
//           rotation-histogram entries for both; an EMPTY left window `continue`s past the twin (:1738-1739)
// probs[0] / probs[1] = left / right WindowProblem (own grid, own key lists from k_window_best2, occupied0 = NULL: the lists are
// occupancy-free, occupancy lives here).  ONE wave replays the query loop in the reference's order; the (at most kTopK) listed
// candidates of the current sub-query are examined by lanes 0..kTopK-1, a list that runs dry before the answer is known makes
// the whole wave re-scan the window against the current occupancy -- exactly what the sequential loop sees at that point.
// grid (1), block 64, dynamic LDS: occ[n_left + n_right] bytes
// ---------------------------------------------------------------------------------------------------------
struct TwinProblem {
    int mode;
    int n_left, n_right, nq;
    float nnratio, max_dist;
    int check_orientation, cleared_value;
    const int32_t *l2r, *r2l;      // mode 1: stereo partners (-1 = none)
    const uint8_t *occupied0;      // [n_left + n_right] taken on entry, or NULL
    const uint8_t *q_has_obs;      // [nq] Observations() > 0 of the query's map point, NULL = all true
    const float *q_angle;          // mode 2: angle of the last frame's keypoint
    int32_t *match;                // out [n_left + n_right]: query index, -1, or cleared_value
    int32_t *nmatches;             // out
    int32_t *entries;              // scratch [2 * nq]: rotation histogram pushes bin << 16 | slot
};

__global__ __launch_bounds__(64) void k_replay_twin(const WindowProblem *__restrict__ probs, TwinProblem T, GridParams g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t occ[];
    __shared__ int hist[ORBX_HISTO_LENGTH + 2];
    const int lane = threadIdx.x;
    const WindowProblem PL = probs[0], PR = probs[1];
    const int N = T.n_left + T.n_right;
    for (int i = lane; i < N; i += 64) {
        occ[i] = T.occupied0 ? T.occupied0[i] : 0;
        T.match[i] = -1;
    }
    if (lane < ORBX_HISTO_LENGTH) hist[lane] = 0;
    __syncthreads();
    int nmatches = 0, n_entries = 0;
    const bool two = (T.mode == 1), ori = (T.mode == 2 && T.check_orientation);
    for (int iq = 0; iq < T.nq; iq++) {
        const uint8_t obs = T.q_has_obs ? T.q_has_obs[iq] : (uint8_t)1;
        bool skip_right = false;
#pragma unroll 1
        for (int side = 0; side < 2; side++) {
            if (side == 1 && skip_right) break;
            const WindowProblem &P = side ? PR : PL;
            const int off = side ? T.n_left : 0, nside = side ? T.n_right : T.n_left;
            if (P.qvalid && !P.qvalid[iq]) continue;
            const int meta = P.meta[iq];
            if (meta & 512) {                                   // vIndices.empty()
                if (T.mode == 2 && side == 0) skip_right = true;
                continue;
            }
            const int valid_len = meta & 0xff;
            const bool exhaustive = (meta & 256) != 0;
            u64 k = kNoKey;
            if (lane < valid_len) k = P.keys[(size_t)iq * kTopK + lane];
            const bool is_free = (k != kNoKey) && !occ[off + (int)(k & 0xffff)];
            const u64 fb = __ballot(is_free);
            u64 c1 = kNoKey, c2 = kNoKey;
            if (fb) {
                c1 = __shfl(k, __ffsll((long long)fb) - 1);
                const u64 rest = fb & (fb - 1ull);
                if (rest) c2 = __shfl(k, __ffsll((long long)rest) - 1);
            }
            const bool need_slow = !exhaustive && (c1 == kNoKey || (two && c2 == kNoKey && (float)(int)(c1 >> 32) <= T.max_dist));
            if (need_slow) {   // the list ran dry: scan the window against the occupancy of this very moment
                QueryWin w;
                Desc dq;
                u64 r1 = kNoKey, r2 = kNoKey;
                if (load_query(P, iq, &w, g, &dq)) {
                    scan_window(P, g, w, dq, nside, occ + off, lane, r1, r2);
                    wave_min2(r1, r2);
                }
                c1 = r1; c2 = r2;
            }
            if (c1 == kNoKey) continue;
            const int bestDist = (int)(c1 >> 32);
            if ((float)bestDist > T.max_dist) continue;
            const int t = (int)(c1 & 0xffff);
            if (two) {
                const int bestDist2 = (c2 == kNoKey) ? 256 : (int)(c2 >> 32);
                const int bestLevel = P.kps[t].octave;
                const int bestLevel2 = (c2 == kNoKey) ? -1 : P.kps[(int)(c2 & 0xffff)].octave;
                if (bestLevel == bestLevel2 && (float)bestDist > T.nnratio * (float)bestDist2) {
                    if (side == 0) skip_right = true;           // :125-126: `continue` leaves the whole iteration
                    continue;
                }
            }
            int np = 1;
            int partner = -1;
            if (two) {
                const int p = side ? T.r2l[t] : T.l2r[t];
                if (p != -1) { partner = side ? p : T.n_left + p; np = 2; }
            }
            if (lane == 0) {
                T.match[off + t] = iq;
                occ[off + t] = obs;
                if (partner >= 0) { T.match[partner] = iq; occ[partner] = obs; }
                if (ori) {
                    const int b = dev_rot_bin(T.q_angle[iq], P.kps[t].angle);
                    T.entries[n_entries] = (b << 16) | (off + t);
                    hist[b]++;
                }
            }
            nmatches += np;
            if (ori) n_entries++;
            one_wave_sync();   // single wave: orders lane 0's LDS writes (occ, hist) before the next sub-query's reads; match / entries are read after the loop's __syncthreads()
        }
    }
    __syncthreads();
    if (ori) {
        int ind1, ind2, ind3;
        dev_three_maxima(hist, ind1, ind2, ind3);
        int dropped = 0;
        for (int e = lane; e < n_entries; e += 64) {
            const int v = T.entries[e], b = v >> 16;
            if (b != ind1 && b != ind2 && b != ind3) { T.match[v & 0xffff] = T.cleared_value; dropped++; }
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) dropped += __shfl_xor(dropped, s);
        nmatches -= dropped;
    }
    if (lane == 0) *T.nmatches = nmatches;
}

struct InitProblem {  // ORBmatcher::SearchForInitialization (ORBmatcher.cc:648-763)
    const orbx_keypoint *kps1; const uint8_t *desc1; int n1;
    const orbx_keypoint *kps2; const uint8_t *desc2; int n2;
    float *prev_matched;       // vbPrevMatched (x, y) per F1 feature; updated at the end (:757-760)
    float window; float nnratio; int check_orientation;
    int32_t *matches12;        // out [n1]
    int32_t *matches21;        // scratch [n2]
    int32_t *matched_dist;     // scratch [n2]  (vMatchedDistance)
    int32_t *entries;          // scratch [n1]: rotation histogram pushes bin << 16 | i1
    int32_t *nmatches;         // out
};

__global__ __launch_bounds__(64) void k_replay_init(InitProblem P, GridParams g) {
    __shared__ int hist[ORBX_HISTO_LENGTH];
    const int lane = threadIdx.x;
    for (int i = lane; i < P.n1; i += 64) P.matches12[i] = -1;
    for (int i = lane; i < P.n2; i += 64) { P.matches21[i] = -1; P.matched_dist[i] = INT_MAX; }
    if (lane < ORBX_HISTO_LENGTH) hist[lane] = 0;
    __syncthreads();
    int nmatches = 0, n_entries = 0;
    for (int i1 = 0; i1 < P.n1; i1++) {
        const orbx_keypoint k1p = P.kps1[i1];
        if (k1p.octave > 0) continue;  // :665
        const QueryWin w = make_window(g, P.prev_matched[2 * i1], P.prev_matched[2 * i1 + 1], P.window, k1p.octave, k1p.octave, 0.f);
        if (w.empty) continue;
        const Desc dq = load_desc(P.desc1 + (size_t)i1 * 32);
        u64 k1 = kNoKey, k2 = kNoKey;
        for (int i2 = lane; i2 < P.n2; i2 += 64) {
            const orbx_keypoint kp = P.kps2[i2];
            int cx, cy;
            if (!in_window(g, w, kp, &cx, &cy)) continue;
            const int d = hamming(dq, load_desc(P.desc2 + (size_t)i2 * 32));
            if (P.matched_dist[i2] <= d) continue;  // :687-688
            push2(k1, k2, cand_key(d, cx, cy, i2));
        }
        wave_min2(k1, k2);
        if (k1 == kNoKey) continue;
        const int best = (int)(k1 >> 32), best2 = (int)(k1 & 0xffff);
        const float second = (k2 == kNoKey) ? (float)INT_MAX : (float)(int)(k2 >> 32);
        if (best <= ORBX_TH_LOW && (float)best < second * P.nnratio) {  // :707-711
            const int old = P.matches21[best2];
            if (old >= 0) nmatches--;
            nmatches++;
            if (lane == 0) {
                if (old >= 0) P.matches12[old] = -1;
                P.matches12[i1] = best2;
                P.matches21[best2] = i1;
                P.matched_dist[best2] = best;
                if (P.check_orientation) {
                    const int b = dev_rot_bin(k1p.angle, P.kps2[best2].angle);
                    hist[b]++;
                    P.entries[n_entries] = (b << 16) | i1;
                }
            }
            if (P.check_orientation) n_entries++;
            __threadfence_block();
            __syncthreads();  // single wave: orders lane 0's state updates before the next query's reads
        }
    }
    __syncthreads();
    if (P.check_orientation) {
        int ind1, ind2, ind3;
        dev_three_maxima(hist, ind1, ind2, ind3);
        if (lane == 0) {  // :742-755: sequential, an index may sit in the histogram after it was unmatched
            for (int e = 0; e < n_entries; e++) {
                const int v = P.entries[e], b = v >> 16, i1 = v & 0xffff;
                if (b != ind1 && b != ind2 && b != ind3 && P.matches12[i1] >= 0) { P.matches12[i1] = -1; nmatches--; }
            }
        }
        nmatches = __shfl(nmatches, 0);
        __syncthreads();
    }
    for (int i1 = lane; i1 < P.n1; i1 += 64) {  // :757-760
        const int m = P.matches12[i1];
        if (m >= 0) { P.prev_matched[2 * i1] = P.kps2[m].x; P.prev_matched[2 * i1 + 1] = P.kps2[m].y; }
    }
    if (lane == 0) *P.nmatches = nmatches;
}

// ---------------------------------------------------------------------------------------------------------
// SearchForInitialization, round 6.  k_replay_init above scans ALL of F2 per query on one wave: 27.6 ms for the 5 x nFeatures extractor of the
// monocular initialisation (4993 x 4993 keypoints, 1086 level-0 queries, 100-px windows) against 2.0 ms on one CPU core.  Here the work that does not
// depend on the loop's state runs wide, and only the state-dependent decisions are replayed:
//   k_grid_build + k_window_best2_t<64>   F2's grid; per query (= level-0 keypoint of F1, index order) the kTopK smallest candidate keys of its window in
//                                         the reference's (distance, enumeration order) order -- no state involved
//   k_replay_init_lists (one wave)        the loop of :661-730 over chunks of 64 queries.  What a query may take depends on earlier queries only through
//                                         vMatchedDistance (:687 skips a candidate already matched at a distance <= its own): every lane picks the first two
//                                         entries of its list that the CURRENT vMatchedDistance does not skip, lanes that would change a vMatchedDistance a
//                                         later lane has looked at make that lane wait (LDS claims, as k_greedy_resolve), the conflict-free prefix commits in
//                                         parallel (vnMatches12 / vnMatches21 / vMatchedDistance in LDS), and a query whose list runs dry before two
//                                         unskipped entries are known is re-scanned by the whole wave against the state of that moment -- exactly what the
//                                         sequential loop sees.  The rotation histogram is a sum (bins computed after the loop from the accepted pairs).
// dynamic LDS: claim u32[n2] | vMatchedDistance u16[n2] | vnMatches21 u16[n2] | vnMatches12 u16[n1]   (0xffff = INT_MAX / -1; n1, n2 <= kMaxResolveFeatures)
// ---------------------------------------------------------------------------------------------------------
struct InitReplay {
    const int32_t *q_index;         // [nq] F1 feature index of query q
    const orbx_keypoint *kps1;      // F1.mvKeysUn
    int n1, n2, nq;
    float nnratio; int check_orientation;
    float *prev_matched;            // [n1][2], updated at the end (:757-760)
    int32_t *matches12;             // out [n1]
    int32_t *entries;               // scratch [nq]: accepted pairs i1 << 16 | i2 in query order (rotHist pushes, :716-727)
    int32_t *nmatches;              // out
};

// the window of ONE query scanned by the whole wave against the current vMatchedDistance (scan_window_grid with the skip rule of :687-688)
__device__ __forceinline__ void scan_window_grid_md(const WindowProblem &P, const GridParams &g, const QueryWin &w, const Desc &dq, int n, const uint16_t *md,
                                                    int lane, u64 &k1, u64 &k2) {
    const int ncol = w.cx1 - w.cx0 + 1;
    int cs = 0, len = 0;
    if (lane < ncol) {
        const int base = (w.cx0 + lane) * 48;
        cs = gld(P.gstart + base + w.cy0);
        len = (int)gld(P.gstart + base + w.cy1 + 1) - cs;
    }
    int incl = len;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int t = __shfl_up(incl, s);
        if (lane >= s) incl += t;
    }
    const int pre = incl - len, total = __shfl(incl, 63);
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        int j = -1;
        for (int c = 0; c < ncol; c++) {
            const int pc = __shfl(pre, c), sc = __shfl(cs, c);
            if (t >= pc) j = sc + (t - pc);
        }
        const bool v0 = t < total && j >= 0;
        const int i0 = v0 ? (int)gld(P.gorder + j) : 0;
        const bool v1 = v0 && i0 < n;
        const int i = v1 ? i0 : 0;
        const orbx_keypoint kp = gld_kp(P.kps + i);
        const Desc dc = gld_desc(P.desc + (size_t)i * 32);
        if (!v1) continue;
        int cx, cy;
        if (!in_window(g, w, kp, &cx, &cy)) continue;
        const int d = hamming(dq, dc);
        if ((int)md[i] <= d) continue;   // :687-688 (0xffff = INT_MAX: never)
        push2(k1, k2, cand_key(d, cx, cy, i));
    }
}

__global__ __launch_bounds__(64) void k_replay_init_lists(const WindowProblem *__restrict__ prob, InitReplay R, GridParams g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ int hist[ORBX_HISTO_LENGTH + 2];
    const WindowProblem P = prob[0];
    const int lane = threadIdx.x, n1 = R.n1, n2 = R.n2, nq = R.nq;
    const u64 lt_mask = (1ull << lane) - 1ull;
    uint32_t *claim = reinterpret_cast<uint32_t *>(lds);
    uint16_t *md = reinterpret_cast<uint16_t *>(lds + (size_t)n2 * 4);
    uint16_t *m21 = md + n2;
    uint16_t *m12 = m21 + n2;
    for (int i = lane; i < n2; i += 64) { claim[i] = 0xffffffffu; md[i] = 0xffffu; m21[i] = 0xffffu; }
    for (int i = lane; i < n1; i += 64) m12[i] = 0xffffu;
    if (lane < ORBX_HISTO_LENGTH) hist[lane] = 0;
    __syncthreads();
    int nmatches = 0, n_entries = 0, n_rounds = 0, n_rescans = 0;
    struct Chunk { u64 L[kTopK]; int meta, i1; };
    auto fetch = [&](int q0) -> Chunk {
        Chunk c;   // inactive lane: empty exhaustive list
#pragma unroll
        for (int e = 0; e < kTopK; e++) c.L[e] = kNoKey;
        c.meta = 256; c.i1 = 0;
        const int qi = q0 + lane;
        if (qi < nq) {
            const u64 *kp = P.keys + (size_t)qi * kTopK;
#pragma unroll
            for (int e = 0; e < kTopK; e++) c.L[e] = gld(kp + e);
            c.meta = gld(P.meta + qi);
            c.i1 = gld(R.q_index + qi);
        }
        return c;
    };
    // best / second best of a lane's list under the current vMatchedDistance; *slow: the list cannot answer (ran dry before what decides the query is known)
    auto pick = [&](const u64 (&L)[kTopK], int valid_len, bool exhaustive, u64 *c1, u64 *c2, bool *slow) {
        u64 a = kNoKey, b = kNoKey;
#pragma unroll
        for (int e = 0; e < kTopK; e++) {
            const u64 k = L[e];
            if (e < valid_len && !((int)md[(int)(k & 0xffff)] <= (int)(k >> 32))) {
                if (a == kNoKey) a = k;
                else if (b == kNoKey) b = k;
            }
        }
        *c1 = a; *c2 = b;
        // what the list has not seen sorts behind its last valid entry: with that entry's distance above TH_LOW nothing unseen can become the best match
        const int floor_d = valid_len > 0 ? (int)(L[valid_len - 1] >> 32) : 0;
        bool s = false;
        if (!exhaustive) {
            if (a == kNoKey) s = floor_d <= ORBX_TH_LOW;                                 // best unknown
            else if (b == kNoKey && (int)(a >> 32) <= ORBX_TH_LOW) s = true;             // second best unknown, and the ratio test will ask for it
        }
        *slow = s;
    };
    auto accept = [&](u64 c1, u64 c2) -> bool {   // :707-711
        if (c1 == kNoKey) return false;
        const int best = (int)(c1 >> 32);
        const float second = (c2 == kNoKey) ? (float)INT_MAX : (float)(int)(c2 >> 32);
        return best <= ORBX_TH_LOW && (float)best < second * R.nnratio;
    };
    Chunk nxt = fetch(0);
    for (int q0 = 0; q0 < nq; q0 += 64) {
        const int qi = q0 + lane;
        const bool active = qi < nq;
        const Chunk C = nxt;
#pragma unroll
        for (int e = 0; e < kTopK; e++) asm volatile("" :: "v"(C.L[e]) : "memory");   // this chunk's lists have arrived before the next chunk's are requested
        asm volatile("" :: "v"(C.meta), "v"(C.i1) : "memory");
        nxt = fetch(q0 + 64);
        u64 L[kTopK];
#pragma unroll
        for (int e = 0; e < kTopK; e++) L[e] = C.L[e];
        const int valid_len = C.meta & 0xff, i1 = C.i1;
        const bool exhaustive = (C.meta & 256) != 0;
        int pos = 0;
        while (pos < 64) {
            const bool live = active && lane >= pos;
            u64 c1 = kNoKey, c2 = kNoKey;
            bool need_slow = false;
            n_rounds++;
            if (live) pick(L, valid_len, exhaustive, &c1, &c2, &need_slow);
            const bool has = live && !need_slow && accept(c1, c2);
            const int t1 = has ? (int)(c1 & 0xffff) : -1;
            if (has) atomicMin(&claim[t1], (uint32_t)lane);
            one_wave_sync();
            // a lane must wait for every earlier lane of this round that is about to change a vMatchedDistance it has looked at (all valid entries of its list)
            bool conflict = need_slow;
            if (live && !need_slow) {
#pragma unroll
                for (int e = 0; e < kTopK; e++)
                    if (e < valid_len && claim[(int)(L[e] & 0xffff)] < (uint32_t)lane) conflict = true;
            }
            one_wave_sync();
            if (has) claim[t1] = 0xffffffffu;
            const u64 cb = __ballot(conflict);
            const int c = cb ? (__ffsll((long long)cb) - 1) : 64;
            const bool ok = has && lane < c;
            const u64 okb = __ballot(ok);
            bool unmatched = false;
            if (ok) {   // :713-727; the targets of one round's commits are distinct (two lanes with one target: the later one conflicts), and so are their former owners
                const int old = m21[t1];
                if (old != 0xffff) { m12[old] = 0xffffu; unmatched = true; }
                m12[i1] = (uint16_t)t1;
                m21[t1] = (uint16_t)i1;
                md[t1] = (uint16_t)(c1 >> 32);
                gst(R.entries + n_entries + __popcll(okb & lt_mask), (i1 << 16) | t1);
            }
            nmatches += __popcll(okb) - __popcll(__ballot(unmatched));
            n_entries += __popcll(okb);
            one_wave_sync();
            if (c >= 64) break;
            const bool slow = (__shfl((int)need_slow, c) != 0);
            if (!slow) { pos = c; continue; }   // lane c re-evaluates against the updated state
            {   // query q0 + c: its window scanned by the whole wave against the state the sequential loop has at this point
                const int qc = q0 + c;
                const int i1c = __shfl(i1, c);
                n_rescans++;
                QueryWin w;
                Desc dq;
                u64 r1 = kNoKey, r2 = kNoKey;
                const int call = P.all_keys ? gld(P.all_cnt + qc) : 0x7fffffff;
                if (call <= P.all_cap) {   // the query's complete candidate list: the skip rule of :687-688 on every entry, the two smallest survivors
                    const u64 *lk = P.all_keys + (size_t)qc * P.all_cap;
                    for (int t0 = 0; t0 < call; t0 += 4 * 64) {
                        u64 kk[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) kk[u] = t0 + 64 * u + lane < call ? gld(lk + t0 + 64 * u + lane) : kNoKey;
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (kk[u] != kNoKey && !((int)md[(int)(kk[u] & 0xffff)] <= (int)(kk[u] >> 32))) push2(r1, r2, kk[u]);
                    }
                    wave_min2(r1, r2);
                } else if (load_query_eager(P, qc, &w, g, &dq)) {
                    scan_window_grid_md(P, g, w, dq, n2, md, lane, r1, r2);
                    wave_min2(r1, r2);
                }
                if (accept(r1, r2)) {
                    const int t = (int)(r1 & 0xffff);
                    const int old = m21[t];
                    one_wave_sync();
                    if (lane == 0) {
                        if (old != 0xffff) m12[old] = 0xffffu;
                        m12[i1c] = (uint16_t)t;
                        m21[t] = (uint16_t)i1c;
                        md[t] = (uint16_t)(r1 >> 32);
                        gst(R.entries + n_entries, (i1c << 16) | t);
                    }
                    nmatches += 1 - (old != 0xffff ? 1 : 0);
                    n_entries++;
                }
                one_wave_sync();
            }
            pos = c + 1;
        }
    }
    __syncthreads();   // the entries are in memory (and visible to the wave's own loads) from here on
    if (R.check_orientation) {   // :733-755: bins of the accepted pairs, ComputeThreeMaxima, pairs of the losing bins dropped if they still stand
        for (int e = lane; e < n_entries; e += 64) {
            const int v = gld(R.entries + e), a = (int)((uint32_t)v >> 16), b = v & 0xffff;
            atomicAdd(&hist[dev_rot_bin(gld(&R.kps1[a].angle), gld(&P.kps[b].angle))], 1);
        }
        __syncthreads();
        int ind1, ind2, ind3;
        dev_three_maxima(hist, ind1, ind2, ind3);
        int dropped = 0;
        for (int e = lane; e < n_entries; e += 64) {
            const int v = gld(R.entries + e), a = (int)((uint32_t)v >> 16), b = v & 0xffff;
            const int bin = dev_rot_bin(gld(&R.kps1[a].angle), gld(&P.kps[b].angle));
            // an index may sit in the histogram after it was unmatched (:746): only pairs that still stand are dropped and counted
            if (bin != ind1 && bin != ind2 && bin != ind3 && m12[a] != 0xffff) { m12[a] = 0xffffu; dropped++; }
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) dropped += __shfl_xor(dropped, s);
        nmatches -= dropped;
        __syncthreads();
    }
    for (int i = lane; i < n1; i += 64) {   // vnMatches12 out; :757-760 vbPrevMatched
        const int m = m12[i];
        gst(R.matches12 + i, m == 0xffff ? -1 : m);
        if (m != 0xffff) {
            const orbx_keypoint kp = gld_kp(P.kps + m);
            gst(R.prev_matched + 2 * i, kp.x);
            gst(R.prev_matched + 2 * i + 1, kp.y);
        }
    }
    if (lane == 0) { gst(R.nmatches, nmatches); gst(R.nmatches + 1, n_rounds); gst(R.nmatches + 2, n_rescans); gst(R.nmatches + 3, nq); }   // [1..3]: orbx_matcher_debug_replay_stats
}

struct FeatVecDev { const uint32_t *node_id; const int32_t *node_ptr; const int32_t *index; int n_nodes; };

// The geometric gates of SearchForTriangulation for pinhole key frames (ORBmatcher.cc:1026-1034 epipole distance,
// CameraModels/Pinhole.cpp:107-129 epipolarConstrain on a caller-supplied F12).  Both are pure functions of the pair, so the
// reference's lazy evaluation order does not matter: they filter the candidates of the best-distance scan.
// The fisheye rig's gate of SearchForTriangulation (ORBmatcher.cc:1036-1072): KannalaBrandt8::epipolarConstrain between the camera kp1 was seen by (left:
// idx1 < NLeft of KF1) and the one of kp2, with the relative pose of that camera pair.  Lives in device memory (indexed by the pair's cameras).
struct Kb8Gate {
    int n_left1, n_left2;      // pKF1->NLeft, pKF2->NLeft: features [0, NLeft) are mvKeys (left camera), the rest mvKeysRight
    const float *sigma2_1;     // pKF1->mvLevelSigma2
    float cam[4][8];           // KannalaBrandt8::mvParameters of pKF1->mpCamera, pKF1->mpCamera2, pKF2->mpCamera, pKF2->mpCamera2
    float R12[4][9], t12[4][3];   // [2 * right1 + right2]: Rll tll, Rlr tlr, Rrl trl, Rrr trr (:938-944), row-major
};
__device__ __attribute__((noinline)) bool kb8_gate(const Kb8Gate *g, float x1, float y1, float sg1, int right1, float x2, float y2, float sg2, int right2) {
    const int sel = 2 * right1 + right2;
    return kb8_epipolar_constrain(g->cam[right1], g->cam[2 + right2], x1, y1, x2, y2, g->R12[sel], g->t12[sel], sg1, sg2);
}

struct TriGate {
    int enabled;               // 0: no gate at all (table/callback-free bCoarse form without keypoints)
    int coarse;                // bCoarse: skip epipolarConstrain (the epipole-distance test still applies)
    int strict;                // 1: every float op rounds separately; 0: the FMA contraction of the reference's build flags
    const orbx_keypoint *k1, *k2;
    const float *ur1, *ur2;    // mvuRight (NULL: monocular)
    const float *scale2;       // pKF2->mvScaleFactors
    const float *sigma2_2;     // pKF2->mvLevelSigma2
    float F[9];                // F12 row-major
    float ex, ey;              // epipole of camera 1 in image 2 (:921)
    const Kb8Gate *kb8;        // k_tri_kb8 only (fisheye key frames): k1 / k2 are mvKeys | mvKeysRight, no epipole test (:1026), KannalaBrandt8::epipolarConstrain
};

__device__ __forceinline__ bool tri_gate(const TriGate &g, int i1, int i2) {
    const bool st1 = g.ur1 && g.ur1[i1] >= 0.f, st2 = g.ur2 && g.ur2[i2] >= 0.f;
    const float x2 = g.k2[i2].x, y2 = g.k2[i2].y;
    const int oct2 = g.k2[i2].octave;
    if (!st1 && !st2) {
        const float dx = __fsub_rn(g.ex, x2), dy = __fsub_rn(g.ey, y2);
        const float d2 = g.strict ? __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) : __fmaf_rn(dx, dx, __fmul_rn(dy, dy));
        if (d2 < __fmul_rn(100.0f, g.scale2[oct2])) return false;
    }
    if (g.coarse) return true;
    const float x1 = g.k1[i1].x, y1 = g.k1[i1].y;
    float a, b, c, num, den;
    if (g.strict) {
        a = __fadd_rn(__fadd_rn(__fmul_rn(x1, g.F[0]), __fmul_rn(y1, g.F[3])), g.F[6]);
        b = __fadd_rn(__fadd_rn(__fmul_rn(x1, g.F[1]), __fmul_rn(y1, g.F[4])), g.F[7]);
        c = __fadd_rn(__fadd_rn(__fmul_rn(x1, g.F[2]), __fmul_rn(y1, g.F[5])), g.F[8]);
        num = __fadd_rn(__fadd_rn(__fmul_rn(a, x2), __fmul_rn(b, y2)), c);
        den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
    } else {  // which product GCC fuses is not uniform: read off the compiled reference text
        a = __fadd_rn(__fmaf_rn(x1, g.F[0], __fmul_rn(y1, g.F[3])), g.F[6]);
        b = __fadd_rn(__fmaf_rn(x1, g.F[1], __fmul_rn(y1, g.F[4])), g.F[7]);
        c = __fadd_rn(__fmaf_rn(y1, g.F[5], __fmul_rn(x1, g.F[2])), g.F[8]);
        num = __fadd_rn(__fmaf_rn(b, y2, __fmul_rn(a, x2)), c);
        den = __fmaf_rn(a, a, __fmul_rn(b, b));
    }
    if (den == 0.f) return false;
    const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
    return (double)dsqr < __dmul_rn(3.84, (double)g.sigma2_2[oct2]);
}

struct BowProblem {
    int mode;   // 0: SearchByBoW(KeyFrame, Frame) :223-425   1: SearchByBoW(KeyFrame, KeyFrame) :765-905
                // 2: SearchForTriangulation :907-1146; gate.enabled selects the pinhole gates, else every pair passes (bCoarse)
    TriGate gate;
    FeatVecDev fa, fb;
    const uint8_t *desc_a; const float *angle_a; const uint8_t *skip_a; int na;   // skip_a[i] != 0: feature i of A is not a query
    const uint8_t *desc_b; const float *angle_b; const uint8_t *skip_b; int nb;   // skip_b[i] != 0: feature i of B is never a candidate
    float nnratio; int check_orientation;
    int nb_left;         // mode 3 = mode 0 for a fisheye-stereo frame (F.Nleft != -1, :283-392): B features >= nb_left are the right camera's
    int32_t *match;      // out: mode 0 [nb] (A index per B feature), modes 1, 2 [na] (B index per A feature)
    uint8_t *taken_b;    // scratch [nb] (mode 1: vbMatched2)
    int32_t *entries;    // scratch [max(na, nb)] (+ 1 in mode 3's worst case per query: [2 max(na, nb)])
    int32_t *hist;       // scratch [ORBX_HISTO_LENGTH], zeroed: rotation histogram over all nodes
    int32_t *counters;   // scratch [2], zeroed: entries appended, matches accepted
    int32_t *nmatches;
    const int32_t *pair_b;   // [fa.n_nodes] index of the node with the same id in fb, or -1 (merge-join of the two sorted node-id lists, done by the host)
    int debug_stop;          // diagnostic (ORBX_BOW_DEBUG): 1 return after the node's ranges, 2 after the candidates' loads, 3 after the first chunk's query loads, 4 skip the loop body's reductions
};

// k_replay_bow: the query loops of SearchByBoW x 2 / SearchForTriangulation, ONE WAVE PER VOCABULARY NODE of A's feature vector.  The reference walks
// the nodes the two feature vectors share in order and, inside a node, A's features in order, each against B's features of that node; what a query
// may take depends only on earlier queries OF THE SAME NODE (a feature lies in exactly one node of its feature vector, so vpMapPointMatches[idxF] /
// vbMatched2[idx2] of a candidate are only ever touched from its node) -- nodes are independent, their order is not observable.  Rounds 2-4 replayed
// all nodes on one wave (1.3 ms for 1000 x 1000 features, a chain of dependent global loads per query; the CPU oracle takes 0.03 ms); a wave per
// node overlaps those chains.  The rotation histogram and the match count are sums over the nodes (global atomics), the consistency filter
// (ComputeThreeMaxima) runs in k_replay_bow_finish.
// grid ceil(fa.n_nodes / 4), block 256; P.hist / P.counters zeroed, P.match = -1, P.taken_b = 0 by the host before the launch
// Round 6: (a) the node pairing (the merge-join of the two sorted maps) comes from the host as pair_b[ia] -- the binary search it replaces was seven
// dependent global loads per wave; (b) a node whose B list has at most 64 features runs REGISTER-RESIDENT (replay_bow_node64): lane l holds candidate l --
// its index, descriptor, skip flag, angle, and for the triangulation gates its keypoint and level constants -- and the taken-state of the candidate
// (vpMapPointMatches[idx] != NULL / vbMatched2[idx], which only this wave ever changes: a feature lies in exactly one node) in a register; the queries of the
// node are preloaded the same way, 64 at a time, and broadcast one by one.  The query loop then touches no memory at all except fire-and-forget result
// stores: 44 -> 9 us for 1000 x 1000 features in 100 nodes (every query of the first form walked index -> flags -> descriptor -> state, each a
// dependent round trip).  Larger B lists keep the first form (replay_bow_big_node).
// grid ceil(fa.n_nodes / 4), block 256; P.hist / P.counters zeroed, P.match = -1, P.taken_b = 0 before the launch (orbx_matcher::fill)
__device__ __attribute__((noinline)) void replay_bow_big_node(const BowProblem &P, const int ia, const int ib, const int lane) {
    const bool toB = (P.mode == 0 || P.mode == 3);   // results are indexed by B's features
    int nmatches = 0;
    const int a0 = P.fa.node_ptr[ia], a1 = P.fa.node_ptr[ia + 1], b0 = P.fb.node_ptr[ib], b1 = P.fb.node_ptr[ib + 1];
    auto record = [&](int a_idx, int b_idx, int out_idx) {   // lane 0: the rotation bin of an accepted match
        if (!P.check_orientation) return;
        const int bin = dev_rot_bin(P.angle_a[a_idx], P.angle_b[b_idx]);
        atomicAdd(&P.hist[bin], 1);
        P.entries[atomicAdd(&P.counters[0], 1)] = (bin << 16) | out_idx;
    };
    for (int a = a0; a < a1; a++) {
        const int i = P.fa.index[a];
        if (P.skip_a && P.skip_a[i]) continue;
        const Desc dq = load_desc(P.desc_a + (size_t)i * 32);
        u64 k1 = kNoKey, k2 = kNoKey, r1 = kNoKey, r2 = kNoKey;   // r*: right-camera candidates of mode 3
        for (int b = b0 + lane; b < b1; b += 64) {
            const int j = P.fb.index[b];
            if (P.skip_b && P.skip_b[j]) continue;
            if (toB && P.match[j] >= 0) continue;           // vpMapPointMatches[realIdxF] already set (:281)
            if (P.mode == 1 && P.taken_b[j]) continue;      // vbMatched2 (:826)
            const int d = hamming(dq, load_desc(P.desc_b + (size_t)j * 32));
            const uint32_t pos = (uint32_t)(b - b0);
            if (P.mode == 2) {
                if (d > ORBX_TH_LOW) continue;               // :1017: '>' twice, so a later equal candidate wins
                if (P.gate.enabled && !tri_gate(P.gate, i, j)) continue;
                push2(k1, k2, ((u64)(uint32_t)d << 32) | (u64)(0xffffffffu - pos));
            } else if (P.mode == 3 && j >= P.nb_left) {
                push2(r1, r2, ((u64)(uint32_t)d << 32) | (u64)pos);   // :302-315: best / second best kept per camera
            } else {
                push2(k1, k2, ((u64)(uint32_t)d << 32) | (u64)pos);
            }
        }
        wave_min2(k1, k2);
        if (P.mode == 3) {
            // :318-377: the left match needs bestDist1 <= TH_LOW and the ratio test; the right one is looked at only INSIDE the
            // bestDist1 <= TH_LOW branch and its ratio test is switched off by `|| true` (:359)
            wave_min2(r1, r2);
            if (k1 == kNoKey || (int)(k1 >> 32) > ORBX_TH_LOW) continue;
            const int bestL = (int)(k1 >> 32);
            const float secondL = (k2 == kNoKey) ? 256.0f : (float)(int)(k2 >> 32);
            const int jl = P.fb.index[b0 + (int)(uint32_t)(k1 & 0xffffffffu)];
            const bool okL = (float)bestL < P.nnratio * secondL;
            const bool okR = r1 != kNoKey && (int)(r1 >> 32) <= ORBX_TH_LOW;
            const int jr = okR ? P.fb.index[b0 + (int)(uint32_t)(r1 & 0xffffffffu)] : -1;
            if (lane == 0) {
                if (okL) { P.match[jl] = i; record(i, jl, jl); }
                if (okR) { P.match[jr] = i; record(i, jr, jr); }
            }
            nmatches += (okL ? 1 : 0) + (okR ? 1 : 0);
            __threadfence_block();   // the wave's next query reads what lane 0 wrote
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (k1 == kNoKey) continue;
        const int best = (int)(k1 >> 32);
        const uint32_t pos = P.mode == 2 ? 0xffffffffu - (uint32_t)(k1 & 0xffffffffu) : (uint32_t)(k1 & 0xffffffffu);
        const int j = P.fb.index[b0 + (int)pos];
        const float second = (k2 == kNoKey) ? 256.0f : (float)(int)(k2 >> 32);
        bool ok;
        if (P.mode == 0) ok = best <= ORBX_TH_LOW && (float)best < P.nnratio * second;      // :318-320
        else if (P.mode == 1) ok = best < ORBX_TH_LOW && (float)best < P.nnratio * second;  // :848-850 (strict)
        else ok = true;
        if (!ok) continue;
        const int out_idx = P.mode == 0 ? j : i, out_val = P.mode == 0 ? i : j;
        nmatches++;
        if (lane == 0) {
            P.match[out_idx] = out_val;
            if (P.mode == 1) P.taken_b[j] = 1;
            record(i, j, out_idx);
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && nmatches) atomicAdd(&P.counters[1], nmatches);
}


__device__ __forceinline__ Desc shfl_desc(const Desc &d, int src) {
    Desc r;
#pragma unroll
    for (int k = 0; k < 4; k++) r.w[k] = __shfl(d.w[k], src);
    return r;
}

// tri_gate with both keypoints' fields in registers (the same operations in the same order)
__device__ __forceinline__ bool tri_gate_regs(const TriGate &g, float x1, float y1, bool st1, float x2, float y2, bool st2, float scale2_o, float sigma2_o) {
    if (!st1 && !st2) {
        const float dx = __fsub_rn(g.ex, x2), dy = __fsub_rn(g.ey, y2);
        const float d2 = g.strict ? __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) : __fmaf_rn(dx, dx, __fmul_rn(dy, dy));
        if (d2 < __fmul_rn(100.0f, scale2_o)) return false;
    }
    if (g.coarse) return true;
    float a, b, c, num, den;
    if (g.strict) {
        a = __fadd_rn(__fadd_rn(__fmul_rn(x1, g.F[0]), __fmul_rn(y1, g.F[3])), g.F[6]);
        b = __fadd_rn(__fadd_rn(__fmul_rn(x1, g.F[1]), __fmul_rn(y1, g.F[4])), g.F[7]);
        c = __fadd_rn(__fadd_rn(__fmul_rn(x1, g.F[2]), __fmul_rn(y1, g.F[5])), g.F[8]);
        num = __fadd_rn(__fadd_rn(__fmul_rn(a, x2), __fmul_rn(b, y2)), c);
        den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
    } else {
        a = __fadd_rn(__fmaf_rn(x1, g.F[0], __fmul_rn(y1, g.F[3])), g.F[6]);
        b = __fadd_rn(__fmaf_rn(x1, g.F[1], __fmul_rn(y1, g.F[4])), g.F[7]);
        c = __fadd_rn(__fmaf_rn(y1, g.F[5], __fmul_rn(x1, g.F[2])), g.F[8]);
        num = __fadd_rn(__fmaf_rn(b, y2, __fmul_rn(a, x2)), c);
        den = __fmaf_rn(a, a, __fmul_rn(b, b));
    }
    if (den == 0.f) return false;
    const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
    return (double)dsqr < __dmul_rn(3.84, (double)sigma2_o);
}

// minimum of a 32-bit key over the wave: DPP scan (row_shr 1 / 2 / 4 / 8, row_bcast15, row_bcast31), the result read from lane 63 -- a dozen vector
// instructions; the 64-bit wave_min2 above goes through ds_bpermute (twelve LDS round trips, dependent)
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t dpp_min_u32(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, CTRL, ROWS, 0xf, false);
    return o < v ? o : v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_min_u32<0x111, 0xf>(v);
    v = dpp_min_u32<0x112, 0xf>(v);
    v = dpp_min_u32<0x114, 0xf>(v);
    v = dpp_min_u32<0x118, 0xf>(v);
    v = dpp_min_u32<0x142, 0xa>(v);
    v = dpp_min_u32<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
constexpr uint32_t kNoKey32 = 0xffffffffu;

__device__ __forceinline__ void replay_bow_node64(const BowProblem &P, const int a0, const int a1, const int b0, const int nbn, const int lane) {
    const int mode = P.mode;
    const bool gated = mode == 2 && P.gate.enabled;
    // ---- the index lists first (both requests in flight together), then everything that hangs off them ----
    const bool cv = lane < nbn;
    const int j = cv ? P.fb.index[b0 + lane] : 0;
    int i_next = a0 + lane < a1 ? P.fa.index[a0 + lane] : 0;
    // ---- candidate `lane` of B's list ----
    const bool cskip = cv && P.skip_b && P.skip_b[j];
    const Desc dc = load_desc(P.desc_b + (size_t)j * 32);
    const float ang_b = (P.check_orientation && cv) ? P.angle_b[j] : 0.f;
    float x2 = 0.f, y2 = 0.f, sc2 = 0.f, sg2 = 0.f;
    bool st2 = false;
    if (gated && cv) {
        x2 = P.gate.k2[j].x; y2 = P.gate.k2[j].y;
        const int o2 = P.gate.k2[j].octave;
        sc2 = P.gate.scale2[o2]; sg2 = P.gate.sigma2_2[o2];
        st2 = P.gate.ur2 && P.gate.ur2[j] >= 0.f;
    }
    const bool right = mode == 3 && j >= P.nb_left;
    if (P.debug_stop == 2) { if (dc.w[0] == 0x1234567ull && ang_b == 1.5f && cskip && x2 == 3.f) P.match[0] = 7; return; }
    bool taken = false;   // vpMapPointMatches[realIdxF] != NULL (:281) / vbMatched2[idx2] (:826): nothing but this wave's own matches sets them
    int nmatches = 0;
    // candidate key: distance (<= 256) << 6 | lane -- the first minimum in list order wins; SearchForTriangulation (:1017) lets a later equal one win: 63 - lane
    const uint32_t ktie = mode == 2 ? (uint32_t)(63 - lane) : (uint32_t)lane;
    for (int q0 = a0; q0 < a1; q0 += 64) {
        // ---- query `lane` of this chunk of A's list ----
        const bool qv = q0 + lane < a1;
        const int i = i_next;
        if (q0 + 64 < a1) i_next = q0 + 64 + lane < a1 ? P.fa.index[q0 + 64 + lane] : 0;
        const bool qskip = !qv || (P.skip_a && P.skip_a[i]);
        const Desc dqa = load_desc(P.desc_a + (size_t)i * 32);
        const float ang_a = (P.check_orientation && qv) ? P.angle_a[i] : 0.f;
        float x1 = 0.f, y1 = 0.f;
        bool st1 = false;
        if (gated && qv) { x1 = P.gate.k1[i].x; y1 = P.gate.k1[i].y; st1 = P.gate.ur1 && P.gate.ur1[i] >= 0.f; }
        int e0 = -1, e1 = -1;   // this lane's QUERY produced these histogram entries (bin << 16 | out index); mode 3 can produce two
        if (P.debug_stop == 3) { if (dqa.w[0] == 0x1234567ull && ang_a == 1.5f && qskip && x1 == 3.f && dc.w[0] == 0x1234567ull && ang_b == 1.5f && cskip) P.match[0] = 7; return; }
        const int nqc = min(64, a1 - q0);
        const u64 skipmask = __ballot(qskip);
        for (int qq = 0; qq < nqc; qq++) {
            const int q = __builtin_amdgcn_readfirstlane(qq);   // wave-uniform: the query's fields come by v_readlane (scalar operands from here on)
            if ((skipmask >> q) & 1ull) continue;
            const int iq = __builtin_amdgcn_readlane(i, q);
            Desc dq;
#pragma unroll
            for (int k = 0; k < 4; k++)
                dq.w[k] = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(dqa.w[k] >> 32), q) << 32) | (u64)(uint32_t)__builtin_amdgcn_readlane((int)dqa.w[k], q);
            const float aq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ang_a), q));
            float xq = 0.f, yq = 0.f;
            bool stq = false;
            if (gated) {
                xq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1), q));
                yq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y1), q));
                stq = __builtin_amdgcn_readlane((int)st1, q) != 0;
            }
            uint32_t k = kNoKey32, kr = kNoKey32;   // kr: right-camera candidates of mode 3 (:302-315: best / second best kept per camera)
            if (cv && !cskip && !(taken && mode != 2)) {
                const int d = hamming(dq, dc);
                const uint32_t key = ((uint32_t)d << 6) | ktie;
                if (mode == 2) {
                    bool ok = d <= ORBX_TH_LOW;   // :1017
                    if (ok && gated) ok = tri_gate_regs(P.gate, xq, yq, stq, x2, y2, st2, sc2, sg2);
                    if (ok) k = key;
                } else if (right) {
                    kr = key;
                } else {
                    k = key;
                }
            }
            const uint32_t m1 = wave_min_u32(k);
            if (mode == 3) {
                if (m1 == kNoKey32 || (int)(m1 >> 6) > ORBX_TH_LOW) continue;   // :318-377: the right match is looked at only inside the bestDist1 <= TH_LOW branch
                const uint32_t m2 = wave_min_u32(k == m1 ? kNoKey32 : k);
                const uint32_t mr = wave_min_u32(kr);
                const int bestL = (int)(m1 >> 6);
                const float secondL = (m2 == kNoKey32) ? 256.0f : (float)(int)(m2 >> 6);
                const int pl = (int)(m1 & 63u);
                const bool okL = (float)bestL < P.nnratio * secondL;
                const bool okR = mr != kNoKey32 && (int)(mr >> 6) <= ORBX_TH_LOW;   // its ratio test is switched off by `|| true` (:359)
                const int pr = okR ? (int)(mr & 63u) : 0;
                if (okL && lane == pl) { taken = true; P.match[j] = iq; if (P.check_orientation) atomicAdd(&P.hist[dev_rot_bin(aq, ang_b)], 1); }
                if (okR && lane == pr) { taken = true; P.match[j] = iq; if (P.check_orientation) atomicAdd(&P.hist[dev_rot_bin(aq, ang_b)], 1); }
                if (P.check_orientation) {
                    const int jl = __builtin_amdgcn_readlane(j, pl), jr = __builtin_amdgcn_readlane(j, pr);
                    const int binl = dev_rot_bin(aq, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ang_b), pl)));
                    const int binr = dev_rot_bin(aq, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ang_b), pr)));
                    if (lane == q) { if (okL) e0 = (binl << 16) | jl; if (okR) e1 = (binr << 16) | jr; }
                }
                nmatches += (okL ? 1 : 0) + (okR ? 1 : 0);
                continue;
            }
            if (m1 == kNoKey32) continue;
            const int best = (int)(m1 >> 6);
            const int pos = mode == 2 ? 63 - (int)(m1 & 63u) : (int)(m1 & 63u);
            bool ok = true;
            if (mode != 2) {   // the ratio test needs the second-best distance
                const uint32_t m2 = wave_min_u32(k == m1 ? kNoKey32 : k);
                const float second = (m2 == kNoKey32) ? 256.0f : (float)(int)(m2 >> 6);
                if (mode == 0) ok = best <= ORBX_TH_LOW && (float)best < P.nnratio * second;      // :318-320
                else ok = best < ORBX_TH_LOW && (float)best < P.nnratio * second;                 // :848-850 (strict)
            }
            if (!ok) continue;
            nmatches++;
            const int jw = __builtin_amdgcn_readlane(j, pos);
            const int out_idx = mode == 0 ? jw : iq, out_val = mode == 0 ? iq : jw;
            if (lane == pos) {
                if (mode != 2) taken = true;
                P.match[out_idx] = out_val;
                if (P.check_orientation) atomicAdd(&P.hist[dev_rot_bin(aq, ang_b)], 1);
            }
            if (P.check_orientation) {
                const int bin = dev_rot_bin(aq, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ang_b), pos)));
                if (lane == q) e0 = (bin << 16) | out_idx;
            }
        }
        if (P.check_orientation) {   // the chunk's histogram entries: one returned atomic for all of them (order is not observable: the finish kernel drops by bin)
            const u64 b0m = __ballot(e0 >= 0), b1m = __ballot(e1 >= 0);
            const int tot = __popcll(b0m) + __popcll(b1m);
            if (tot) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&P.counters[0], tot);
                base = __builtin_amdgcn_readlane(base, 0);
                const u64 lt = (1ull << lane) - 1ull;
                if (e0 >= 0) P.entries[base + __popcll(b0m & lt)] = e0;
                if (e1 >= 0) P.entries[base + __popcll(b0m) + __popcll(b1m & lt)] = e1;
            }
        }
    }
    if (lane == 0 && nmatches) atomicAdd(&P.counters[1], nmatches);
}

__global__ __launch_bounds__(256) void k_replay_bow(BowProblem P) {
    const int lane = threadIdx.x & 63;
    const int ia = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (ia >= P.fa.n_nodes) return;
    const int ib = P.pair_b[ia];   // the node of B with the same id (:246-250, :800-805, :961-965 merge-join the two sorted maps), -1: none
    if (ib < 0) return;
    const int a0 = P.fa.node_ptr[ia], a1 = P.fa.node_ptr[ia + 1], b0 = P.fb.node_ptr[ib], b1 = P.fb.node_ptr[ib + 1];
    if (P.debug_stop == 1) { if (a0 + a1 + b0 + b1 == -12345) P.match[0] = 7; return; }
    if (b1 - b0 <= 64) replay_bow_node64(P, a0, a1, b0, b1 - b0, lane);
    else replay_bow_big_node(P, ia, ib, lane);
}

// debug kernel (orbx_debug_kb8_epipolar): KannalaBrandt8::epipolarConstrain for n independent keypoint pairs, a lane per pair -- the device function k_tri_kb8 calls,
// on its own, so that a test can compare EVERY verdict with the oracle on the hardware (through the search only the winning candidate of a query shows)
__global__ __launch_bounds__(256) void k_debug_kb8_gate(const Kb8Gate *__restrict__ g, int n, const float *__restrict__ xy1, const float *__restrict__ xy2,
                                                        const float *__restrict__ sigma1, const float *__restrict__ sigma2, const uint8_t *__restrict__ sel,
                                                        uint8_t *__restrict__ ok) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    ok[i] = kb8_gate(g, xy1[2 * i], xy1[2 * i + 1], sigma1[i], sel[i] >> 1, xy2[2 * i], xy2[2 * i + 1], sigma2[i], sel[i] & 1) ? 1 : 0;
}

// k_tri_kb8: SearchForTriangulation between key frames of a FISHEYE rig (:1036-1072), a wave per vocabulary node.  Its gate -- KannalaBrandt8::epipolarConstrain: two
// Newton unprojections, a 4 x 4 JacobiSVD, two projections -- is some 10^4 instructions per pair, and SearchForTriangulation keeps no taken-state (:1011 reads
// vbMatched2, nothing sets it): the queries of a node are independent and the gate is a pure function of the pair, so the reference's lazy order is not observable.
// Evaluated inside k_replay_bow's query loop the gate ran once per QUERY on the few lanes whose distance passed (570 us for 1000 x 1000 features, the CPU takes 670).
// Here the pairs of a node with distance <= TH_LOW are first COLLECTED (register-resident Hamming tiles of 64 queries x 64 candidates, ballot + mbcnt appends to
// an LDS list), then the gate runs over the list 64 pairs at a time with every lane busy, and a pair that passes lowers its query's key with an LDS atomic
// (distance << 16 | 0xffff - position: the later of two equal candidates wins, :1017).  grid ceil(fa.n_nodes / 4), block 256; P as for k_replay_bow mode 2.
constexpr int kTriListCap = 2048;   // pairs per wave between two flushes (8 KB)
__global__ __launch_bounds__(256) void k_tri_kb8(BowProblem P) {
    __shared__ uint32_t list_s[4][kTriListCap];
    __shared__ uint32_t keymin_s[4][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ia = (int)(blockIdx.x * 4) + wv;
    if (ia >= P.fa.n_nodes) return;
    const int ib = P.pair_b[ia];
    if (ib < 0) return;
    const int a0 = P.fa.node_ptr[ia], a1 = P.fa.node_ptr[ia + 1], b0 = P.fb.node_ptr[ib], b1 = P.fb.node_ptr[ib + 1];
    uint32_t *list = list_s[wv], *keymin = keymin_s[wv];
    const Kb8Gate *kb8 = P.gate.kb8;
    int nmatches = 0;
    for (int q0 = a0; q0 < a1; q0 += 64) {
        const bool qv = q0 + lane < a1;
        const int i = qv ? P.fa.index[q0 + lane] : 0;
        const bool qskip = !qv || (P.skip_a && P.skip_a[i]);
        const Desc dqa = load_desc(P.desc_a + (size_t)i * 32);
        keymin[lane] = kNoKey32;
        int m = 0;   // wave-uniform: pairs on the list
        const u64 skipmask = __ballot(qskip);
        const int nqc = min(64, a1 - q0);
        auto flush = [&]() {
            one_wave_sync();
            for (int p0 = 0; p0 < m; p0 += 64) {
                const int p = p0 + lane;
                if (p < m) {
                    const uint32_t e = list[p];
                    const int q = (int)((e >> 16) & 63u), c = (int)(e & 0xffffu);
                    const int i1 = P.fa.index[q0 + q], j2 = P.fb.index[b0 + c];
                    const float x1 = P.gate.k1[i1].x, y1 = P.gate.k1[i1].y, x2 = P.gate.k2[j2].x, y2 = P.gate.k2[j2].y;
                    const float sg1 = kb8->sigma2_1[P.gate.k1[i1].octave], sg2 = P.gate.sigma2_2[P.gate.k2[j2].octave];
                    if (kb8_gate(kb8, x1, y1, sg1, i1 >= kb8->n_left1 ? 1 : 0, x2, y2, sg2, j2 >= kb8->n_left2 ? 1 : 0))
                        atomicMin(&keymin[q], ((e >> 22) << 16) | (0xffffu - (uint32_t)c));
                }
            }
            m = 0;
            one_wave_sync();
        };
        for (int c0 = b0; c0 < b1; c0 += 64) {
            const bool cv = c0 + lane < b1;
            const int j = cv ? P.fb.index[c0 + lane] : 0;
            const bool cok = cv && !(P.skip_b && P.skip_b[j]);
            const Desc dc = load_desc(P.desc_b + (size_t)j * 32);
            const uint32_t cpos = (uint32_t)(c0 - b0 + lane);
            for (int qq = 0; qq < nqc; qq++) {
                const int q = __builtin_amdgcn_readfirstlane(qq);
                if ((skipmask >> q) & 1ull) continue;
                Desc dq;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    dq.w[k] = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(dqa.w[k] >> 32), q) << 32) | (u64)(uint32_t)__builtin_amdgcn_readlane((int)dqa.w[k], q);
                const int d = hamming(dq, dc);
                const bool pass = cok && d <= ORBX_TH_LOW;   // :1017
                const u64 b = __ballot(pass);
                if (b == 0ull) continue;
                if (pass) list[m + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = ((uint32_t)d << 22) | ((uint32_t)q << 16) | cpos;
                m += __popcll(b);
                if (m > kTriListCap - 64) flush();
            }
        }
        flush();
        // ---- the chunk's results, a query per lane ----
        const uint32_t km = keymin[lane];
        const bool hit = qv && km != kNoKey32;   // a skipped query has no pair on the list
        int e0 = -1;
        if (hit) {
            const int jw = P.fb.index[b0 + (int)(0xffffu - (km & 0xffffu))];
            P.match[i] = jw;
            if (P.check_orientation) {
                const int bin = dev_rot_bin(P.angle_a[i], P.angle_b[jw]);
                atomicAdd(&P.hist[bin], 1);
                e0 = (bin << 16) | i;
            }
        }
        const u64 hm = __ballot(hit);
        nmatches += __popcll(hm);
        if (P.check_orientation && hm) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&P.counters[0], __popcll(hm));
            base = __builtin_amdgcn_readlane(base, 0);
            if (hit) P.entries[base + __popcll(hm & ((1ull << lane) - 1ull))] = e0;
        }
        one_wave_sync();   // keymin is rewritten by the next chunk
    }
    if (lane == 0 && nmatches) atomicAdd(&P.counters[1], nmatches);
}

// the rotation-consistency filter over the matches of all nodes (:401-416, :882-897, :1120-1137) and the match count
__global__ __launch_bounds__(64) void k_replay_bow_finish(BowProblem P) {
    __shared__ int hist[ORBX_HISTO_LENGTH];
    const int lane = threadIdx.x;
    int nmatches = P.counters[1];
    if (P.check_orientation) {
        if (lane < ORBX_HISTO_LENGTH) hist[lane] = P.hist[lane];
        __syncthreads();
        const int n_entries = P.counters[0];
        int ind1, ind2, ind3;
        dev_three_maxima(hist, ind1, ind2, ind3);
        int dropped = 0;
        for (int e = lane; e < n_entries; e += 64) {
            const int v = P.entries[e], b = v >> 16;
            if (b != ind1 && b != ind2 && b != ind3) { P.match[v & 0xffff] = -1; dropped++; }
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) dropped += __shfl_xor(dropped, s);
        nmatches -= dropped;
    }
    if (lane == 0) *P.nmatches = nmatches;
}

}  // namespace orbx
