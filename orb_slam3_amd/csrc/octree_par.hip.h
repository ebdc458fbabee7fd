// octree_par.hip.h -- k_octree_par: DistributeOctTree (/root/reference/src/ORBextractor.cc:555-779) with every pass of the
// reference's control flow evaluated data-parallel by one wave per (frame, level), bit-identical to the sequential
// emulation in octree.hip.h (which stays as the selectable reference, ORBX_OCTREE=seq).  One 256-thread workgroup per
// (frame, level): the four waves share the key sweeps (each thread owns ceil(C/256)|1 consecutive keys; ranks come from a
// workgroup-wide SEGMENTED prefix sum of four packed 16-bit child counters), wave 0 runs the node-level steps and the sort.
// Levels with more keys than the LDS buffers hold (or more than four roots) are processed by wave 0 alone with the
// chunked single-wave form of the same passes (global-memory key buffers).
//
// What makes the reference order sensitive, and how each piece is reproduced without walking a list:
//   * lNodes is a std::list; a breadth pass (:618-680) erases every node with more than one key and push_front's its
//     non-empty children n1..n4.  So after a pass the list is   reverse(children in creation order) ++ (the nodes that were
//     not divided, in their old order)   and the next pass meets the divisible nodes in list order.  The list is kept as an
//     ARRAY in list order; a pass computes every child's position with prefix sums (children counted in processing order).
//   * the size-ordered expansion (:686-752) sorts vSizeAndPointerToNode with std::sort and divides from the back until the
//     list holds N nodes.  std::sort's tie order is reproduced exactly: libstdc++ introsort = median-of-3 Hoare partitions
//     until every segment has <= 16 elements, then one insertion sort, which is a STABLE sort of whatever the partitions
//     left.  A Hoare partition's swaps are fully determined by the original values (k-th element from the left that is
//     not less than the pivot <-> k-th element from the right that is not greater, while they have not crossed), so the
//     wave evaluates one partition with ballots / ranks; the final pass is a rank sort.  The heap-sort fallback (depth
//     limit) is delegated to the sequential replica.  The break "list size >= N" is a prefix sum over the sorted order.
//   * DivideNode's four key vectors keep the parent's key order: all divided nodes are split at once by a segmented
//     stable scatter (per-node child counters in LDS, in-chunk rank by ballot), keys ping-pong between two buffers.
//   * the retained keypoint of a node (:757-776, first maximum) is an LDS atomic max over (response, -position).
#pragma once

#include "octree.hip.h"

namespace orbx {

constexpr int kOctParLdsKeys = 4096;  // keys (and node-of-key entries) kept in LDS
constexpr int kOctTier1Keys = 1792;   // first tier of the 256-thread form (k_octree_par_t<kOctTier1Keys, -1>): 7 key slots per thread
constexpr int kOctBlkE = 17;          // keys per thread of the 256-thread form: ceil(C / 256) | 1 (odd: conflict-free LDS stride)

// ordering point for code that only ONE wave executes (LDS operations of a wave are performed in issue order; the
// compiler must not move them across): no s_barrier, so it may sit in wave-divergent sections of a larger workgroup
#define OCT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

__host__ __device__ inline size_t oct_par_pool_bytes(int pool) {
    // stack + per node: bounds 2x8, segment 2x8, hist 16, cpos 8, sa/sb 16, npos/eof/nebe/cbe/abe/ls/rs 7x2
    return ((size_t)kOctStackInts * 4 + (size_t)pool * (16 + 16 + 16 + 8 + 16 + 14) + 160 + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t oct_par_lds_bytes(int pool) { return oct_par_pool_bytes(pool) + (size_t)kOctParLdsKeys * 6; }

struct OctBnd { int16_t x0, y0, x1, y1; };   // UL.x, UL.y, UR.x, BR.y
struct OctSeg { int32_t beg, cnt; };         // key range of the node

// ---- wave-parallel std::sort replica (see the header comment).  a: n entries (oct_less order), work arrays ls/rs (n u16),
// tmp (n u64), stack (kOctStackInts).  Returns false when the introsort depth limit was reached (nothing usable in a).
__device__ __forceinline__ uint64_t oct_readlane64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
// __move_median_to_first(first, first+1, mid, last-1): index of the median among (a, b, c)
__device__ __forceinline__ int oct_median3(uint64_t va, uint64_t vb, uint64_t vc, int ia, int ib, int ic) {
    if (oct_less(va, vb)) {
        if (oct_less(vb, vc)) return ib;
        if (oct_less(va, vc)) return ic;
        return ia;
    }
    if (oct_less(va, vc)) return ia;
    if (oct_less(vb, vc)) return ic;
    return ib;
}

__device__ inline bool oct_par_sort(uint64_t *a, uint64_t *tmp, int n, uint16_t *ls, uint16_t *rs, int *stack, int lane, long long *dbg = nullptr) {
    long long tm = dbg ? (long long)wall_clock64() : 0;
#define SORT_TICK(k) do { if (dbg) { const long long tn = (long long)wall_clock64(); if (lane == 0) dbg[k] += tn - tm; tm = tn; } } while (0)
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const unsigned long long gt_mask = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    uint16_t *posL = reinterpret_cast<uint16_t *>(stack + 128), *posR = posL + 64;  // partner tables of the register partition
    if (n > 16) {
        int sp = 0;
        int first = 0, last = n, depth = 2 * (31 - __clz(n));
        while (true) {
            // segments longer than a wave: stoppers ranked through LDS
            while (last - first > 64) {
                if (depth == 0) return false;
                --depth;
                const int mid = first + (last - first) / 2;
                const int ia = first + 1, ib = mid, ic = last - 1;
                const uint64_t vf = a[first], va = a[ia], vb = a[ib], vc = a[ic];
                const int m = oct_median3(va, vb, vc, ia, ib, ic);
                const uint64_t pivot = m == ia ? va : (m == ib ? vb : vc);
                OCT_WAVE_SYNC();
                if (lane == 0) { a[first] = pivot; a[m] = vf; }
                OCT_WAVE_SYNC();
                // __unguarded_partition(first+1, last, pivot): stoppers of the two scans, by rank
                int nL = 0, nR = 0;
                for (int i0 = first + 1; i0 < last; i0 += 64) {
                    const int i = i0 + lane;
                    const bool act = i < last;
                    const uint64_t v = act ? a[i] : 0;
                    const bool isL = act && !oct_less(v, pivot), isR = act && !oct_less(pivot, v);
                    const unsigned long long bL = __ballot(isL), bR = __ballot(isR);
                    if (isL) ls[nL + __popcll(bL & lt_mask)] = (uint16_t)i;
                    if (isR) rs[nR + __popcll(bR & lt_mask)] = (uint16_t)i;   // ascending; the k-th from the right is rs[nR-1-k]
                    nL += __popcll(bL);
                    nR += __popcll(bR);
                }
                OCT_WAVE_SYNC();
                const int kmax = min(nL, nR);
                int K = 0;  // pairs that are swapped: ls[k] < rs[nR-1-k] (a prefix, both sequences are monotone)
                for (int k0 = 0; k0 < kmax; k0 += 64) {
                    const int k = k0 + lane;
                    const bool sw = k < kmax && ls[k] < rs[nR - 1 - k];
                    const unsigned long long b = __ballot(sw);
                    K += __popcll(b);
                    if (__popcll(b) < min(64, kmax - k0)) break;
                }
                for (int k = lane; k < K; k += 64) {
                    const int i = ls[k], j = rs[nR - 1 - k];
                    const uint64_t vi = a[i], vj = a[j];
                    a[i] = vj; a[j] = vi;
                }
                int cut = last;
                if (K < nL) cut = min(cut, (int)ls[K]);
                if (K > 0) cut = min(cut, (int)rs[nR - K]);
                OCT_WAVE_SYNC();
                // __introsort_loop recurses on [cut, last) and loops on [first, cut)
                if (lane == 0) { stack[3 * sp] = cut; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth; }
                sp++;
                last = cut;
                SORT_TICK(12);
                if (dbg && lane == 0) dbg[15] += 1;
            }
            // at most one element per lane: the segment stays in registers while its left part keeps being partitioned
            if (last - first > 16) {
                uint64_t v = (first + lane < last) ? a[first + lane] : 0;
                while (last - first > 16) {
                    if (depth == 0) return false;
                    --depth;
                    const int len = last - first;
                    const int ra = 1, rb = len / 2, rc = len - 1;
                    const uint64_t vf = oct_readlane64(v, 0), va = oct_readlane64(v, ra), vb = oct_readlane64(v, rb), vc = oct_readlane64(v, rc);
                    const int m = oct_median3(va, vb, vc, ra, rb, rc);
                    const uint64_t pivot = m == ra ? va : (m == rb ? vb : vc);
                    if (lane == 0) v = pivot; else if (lane == m) v = vf;
                    const bool act = lane >= 1 && lane < len;
                    const bool isL = act && !oct_less(v, pivot), isR = act && !oct_less(pivot, v);
                    const unsigned long long bL = __ballot(isL), bR = __ballot(isR);
                    const int kL = __popcll(bL & lt_mask), kR = __popcll(bR & gt_mask);  // rank from the left / from the right
                    // the k-th stopper from the left is exchanged with the k-th from the right while they have not crossed
                    const bool swL = isL && __popcll(bR & gt_mask) > kL, swR = isR && __popcll(bL & lt_mask) > kR;
                    const unsigned long long bSL = __ballot(swL), bSR = __ballot(swR);
                    if (swL) posL[kL] = (uint16_t)lane;
                    if (swR) posR[kR] = (uint16_t)lane;
                    OCT_WAVE_SYNC();
                    int partner = lane;
                    if (swL) partner = posR[kL];
                    if (swR) partner = posL[kR];
                    v = __shfl((unsigned long long)v, partner);
                    const int K = __popcll(bSL), nL = __popcll(bL);
                    int cutr = len;
                    if (K < nL) cutr = min(cutr, __ffsll((long long)(bL & ~bSL)) - 1);
                    if (K > 0) cutr = min(cutr, __ffsll((long long)bSR) - 1);
                    if (lane < len) a[first + lane] = v;  // the right part is final for this round; the left part continues in v
                    const int cut = first + cutr;
                    if (lane == 0) { stack[3 * sp] = cut; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth; }
                    sp++;
                    last = cut;
                    OCT_WAVE_SYNC();
                    SORT_TICK(13);
                    if (dbg && lane == 0) dbg[15] += 1;
                }
            }
            if (sp == 0) break;
            sp--;
            OCT_WAVE_SYNC();
            first = stack[3 * sp]; last = stack[3 * sp + 1]; depth = stack[3 * sp + 2];
        }
    }
    // __final_insertion_sort == stable sort of the current arrangement.  Every segment the partitions left has <= 16
    // elements and no element of a later segment is less than one of an earlier segment, so an element's final position
    // is   lo + #{j in [lo, hi] : a[j] < a[i] or (a[j] == a[i] and j < i)}   for any window [lo, hi] that contains its segment:
    // 15 neighbours on either side.
    OCT_WAVE_SYNC();
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const uint64_t mine = a[min(i, n - 1)];
        const uint64_t v = mine >> 16;
        const int lo = max(i - 15, 0);
        int r = lo;
#pragma unroll
        for (int d0 = -15; d0 <= 15; d0 += 8) {
            uint64_t w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) w[u] = a[min(max(i + d0 + u, 0), n - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int d = d0 + u, j = i + d;
                if (d <= 15 && d != 0) {
                    const uint64_t x = w[u] >> 16;
                    r += (j >= 0 && j < n) && (d < 0 ? x <= v : x < v);
                }
            }
        }
        if (i < n) tmp[r] = mine;
    }
    OCT_WAVE_SYNC();
    for (int i = lane; i < n; i += 64) a[i] = tmp[i];
    OCT_WAVE_SYNC();
    SORT_TICK(14);
#undef SORT_TICK
    return true;
}

// debug kernel: sort (count, ulx) pairs with the wave-parallel replica (falls back like k_octree_par does)
__global__ __launch_bounds__(64) void k_debug_sort_par(const int32_t *count, const int32_t *ulx, int n, int32_t *perm, uint64_t *scratch,
                                                       uint64_t *tmp, uint16_t *ls, uint16_t *rs, int32_t *fellback) {
    __shared__ int stack[kOctStackInts];
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64)
        scratch[i] = ((uint64_t)(uint32_t)count[i] << 32) | ((uint64_t)(uint32_t)ulx[i] << 16) | (uint64_t)i;
    OCT_WAVE_SYNC();
    const bool ok = oct_par_sort(scratch, tmp, n, ls, rs, stack, lane);
    if (!ok) {
        OCT_WAVE_SYNC();
        for (int i = lane; i < n; i += 64)
            scratch[i] = ((uint64_t)(uint32_t)count[i] << 32) | ((uint64_t)(uint32_t)ulx[i] << 16) | (uint64_t)i;
        OCT_WAVE_SYNC();
        LdsArr arr{scratch};
        if (lane == 0) { oct_std_sort(arr, n, stack); *fellback = 1; }
        OCT_WAVE_SYNC();
    }
    for (int i = lane; i < n; i += 64) perm[i] = (int32_t)(scratch[i] & 0xffff);
}

// ---- compact_level (k_compact until round 3): vToDistributeKeys of every (frame, level) (:805-870 pushes the cells' keypoints in cell row-major order):
// exclusive scan of the per-cell counts + gather of the cell slots into one dense key array.  A separate, wide launch
// (256 threads per (frame, level)) because the single quad-tree wave cannot hide the latency of these dependent global
// loads.  Since round 3 the first tier of k_octree_par_t calls it for its own (frame, level).
// one (frame, level): returns the number of keys gathered (the same value in every thread); 256 threads, one barrier pair per 256 cells
__device__ __forceinline__ int compact_level(const LevelInfo &L, const int32_t *__restrict__ ccnt, const uint32_t *__restrict__ ent,
                                             uint32_t *__restrict__ dst) {
    __shared__ int wsum[4];
    const int ncell = L.nCols * L.nRows;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int run = 0;
    for (int c0 = 0; c0 < ncell; c0 += 256) {
        const int c = c0 + tid;
        const int n = (c < ncell) ? ccnt[c] : 0;
        const int incl = wave_incl_scan(n);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        const int s0 = wsum[0], s1 = wsum[1], s2 = wsum[2], s3 = wsum[3];
        __syncthreads();
        const int excl = run + incl - n + (w > 0 ? s0 : 0) + (w > 1 ? s1 : 0) + (w > 2 ? s2 : 0);
        const uint32_t *src = ent + (size_t)c * L.cell_cap;
        for (int k = 0; k < n; k += 4) {
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = (k + u < n) ? src[k + u] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) if (k + u < n) dst[excl + k + u] = v[u];
        }
        run += s0 + s1 + s2 + s3;
    }
    return run;
}
// ---- the kernel body ---------------------------------------------------------------------------------------------
// exclusive SEGMENTED prefix sum over the 256 threads of the workgroup: the value accumulated since the last segment head
// (fl = this thread contains a head; v = its accumulation after its last head, or over all its keys when it has none);
// DPP scan inside a wave (row_shr 1/2/4/8, row_bcast15, row_bcast31), LDS across the four waves
template <int CTRL, int ROWS>
__device__ __forceinline__ void oct_seg_step(uint32_t &lo, uint32_t &hi, int &f) {
    const uint32_t plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, ROWS, 0xf, false);
    const uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, ROWS, 0xf, false);
    const int pf = __builtin_amdgcn_update_dpp(0, f, CTRL, ROWS, 0xf, false);
    if (!f) { lo += plo; hi += phi; }  // the four 16-bit counters never carry into each other
    f |= pf;
}
__device__ __forceinline__ uint64_t oct_blk_seg_excl(uint64_t v, bool fl, uint64_t *wtot, int *wflag, int lane, int wave) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    int f = fl ? 1 : 0;
    oct_seg_step<0x111, 0xf>(lo, hi, f);
    oct_seg_step<0x112, 0xf>(lo, hi, f);
    oct_seg_step<0x114, 0xf>(lo, hi, f);
    oct_seg_step<0x118, 0xf>(lo, hi, f);
    oct_seg_step<0x142, 0xa>(lo, hi, f);
    oct_seg_step<0x143, 0xc>(lo, hi, f);
    if (lane == 63) { wtot[wave] = ((uint64_t)hi << 32) | lo; wflag[wave] = f; }
    __syncthreads();
    // exclusive value: the previous lane's inclusive one (wave_shr:1)
    const uint32_t elo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x138, 0xf, 0xf, false);
    const uint32_t ehi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x138, 0xf, 0xf, false);
    const int ef = __builtin_amdgcn_update_dpp(0, f, 0x138, 0xf, 0xf, false);
    const uint64_t ev = ((uint64_t)ehi << 32) | elo;
    uint64_t carry = 0;
    for (int w = 0; w < wave; w++) carry = wflag[w] ? wtot[w] : carry + wtot[w];
    const uint64_t r = ef ? ev : ev + carry;
    __syncthreads();
    return r;
}

// KEYS = capacity of the LDS key buffers of the 256-thread form (4096, or 2048 for k_octree_par_t's small tier); kE = key slots per thread
// WIDE (round 6; only with BLK == false): the global-memory form on all 256 threads of the workgroup.  The single-wave chunked form walks every key of a level in
// chunks of 64, twice per pass (child histogram, stable scatter), and its scatter hands a running offset from chunk to chunk -- one wave, 2 ms for a level
// of 22 900 candidates (the texture scene: 2.2 of 3.3 ms per step).  Here the chunks are independent: a table in LDS (the key area the LDS form would use)
// holds, per chunk, how many keys of the chunk's LAST node go to each child; its prefix sum P gives the number of a node's keys in earlier chunks as
// P[chunk] - P[chunk where the node starts] (a node's keys are contiguous, so every chunk in between belongs to it entirely), and a key's place is
// child offset + that + its rank inside the chunk.  No atomics on the offsets, no order between chunks: the four waves stride the chunks.  C <= 65535
// (16-bit fields); larger levels keep the single-wave form.
constexpr int kOctWideMaxKeys = 65535;
template <bool BLK, int KEYS = kOctParLdsKeys, bool WIDE = false>
__device__ __forceinline__ void octree_par_body(const LevelInfo &L, uint8_t *smem, int max_pool, const int C, uint32_t *gk0, uint32_t *gk1,
                                                uint16_t *gn0, uint16_t *gn1, uint32_t *__restrict__ out,
                                                int32_t *__restrict__ lvlcnt_out, int32_t *__restrict__ err, long long *dbg) {
    constexpr int kE = (KEYS >> 8) | 1;   // ceil(KEYS / 256) | 1 (odd: conflict-free LDS stride); 17 for 4096 keys
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // optional phase timing of one workgroup (ORBX_OCT_DBG): 100 MHz wall clock ticks per phase
    long long tmark = dbg ? (long long)wall_clock64() : 0;
    const long long tstart = tmark;
#define OCT_TICK(k) do { if (dbg) { const long long tn = (long long)wall_clock64(); if (tid == 0) dbg[k] += tn - tmark; tmark = tn; } } while (0)
    static_assert(!(BLK && WIDE), "WIDE is a form of the global-memory body");
    constexpr int NT = (BLK || WIDE) ? 256 : 64;
    constexpr int NW = NT / 64;
    const int pool = L.pool;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#define OCT_SYNC_ALL() do { if (BLK || WIDE) __syncthreads(); else OCT_WAVE_SYNC(); } while (0)
    // WIDE: per-chunk tables in the LDS key area: ctab[c] = four 16-bit counters (then their exclusive prefix sums), one per child
    uint64_t *const ctab = reinterpret_cast<uint64_t *>(smem + oct_par_pool_bytes(max_pool));
    const int nch = (C + 63) >> 6;
    // exclusive prefix sum of ctab[0 .. nch) in place, by wave 0 (the fields never carry into each other: they count disjoint keys, C <= 65535); total -> ctl[2], ctl[3]
    auto chunk_scan = [&](int *ctl_) {
        if (wave == 0) {
            uint32_t rlo = 0, rhi = 0;
            for (int c0 = 0; c0 < nch; c0 += 64) {
                const int c = c0 + lane;
                const uint64_t v = c < nch ? ctab[c] : 0ull;
                const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
                const uint32_t ilo = (uint32_t)wave_incl_scan((int)lo), ihi = (uint32_t)wave_incl_scan((int)hi);
                if (c < nch) ctab[c] = ((uint64_t)(rhi + ihi - hi) << 32) | (uint64_t)(rlo + ilo - lo);
                rlo += (uint32_t)__builtin_amdgcn_readlane((int)ilo, 63);
                rhi += (uint32_t)__builtin_amdgcn_readlane((int)ihi, 63);
            }
            if (lane == 0) { ctl_[2] = (int)rlo; ctl_[3] = (int)rhi; }
        }
    };

    // carve (8-byte arrays first)
    uint8_t *p = smem;
    int *stack = (int *)p; p += (size_t)kOctStackInts * 4;
    uint64_t *wtot = (uint64_t *)p; p += 32;
    int *wflag = (int *)p; p += 16;
    int *ctl = (int *)p; p += 48;
    uint64_t *sa = (uint64_t *)p; p += (size_t)pool * 8;
    uint64_t *sb = (uint64_t *)p; p += (size_t)pool * 8;
    // the two generations of the node list: base + generation * pool (plain offset arithmetic keeps the accesses ds_* instead of
    // flat_*, which an array of two pointers indexed at run time does not)
    OctBnd *const bndb = (OctBnd *)p; p += (size_t)pool * 16;
    OctSeg *const segb = (OctSeg *)p; p += (size_t)pool * 16;
    uint32_t *hist = (uint32_t *)p; p += (size_t)pool * 16;
    uint16_t *cpos = (uint16_t *)p; p += (size_t)pool * 8;
    uint16_t *npos = (uint16_t *)p; p += (size_t)pool * 2;   // new position of a node that is not divided; 0xffff = divided
    uint16_t *eof = (uint16_t *)p; p += (size_t)pool * 2;    // processing index of a divisible node
    uint16_t *nebe = (uint16_t *)p; p += (size_t)pool * 2;   // by processing index: #children | #children with >1 keys << 3
    uint16_t *cbe = (uint16_t *)p; p += (size_t)pool * 2;    // by processing index: children created before
    uint16_t *abe = (uint16_t *)p; p += (size_t)pool * 2;    // by processing index: divisible children created before
    uint16_t *ls = (uint16_t *)p; p += (size_t)pool * 2;
    uint16_t *rs = (uint16_t *)p; p += (size_t)pool * 2;

    uint32_t *kb[2];
    uint16_t *nof[2];
    if (BLK) {
        uint8_t *lk = smem + oct_par_pool_bytes(max_pool);
        // every thread holds its keys in registers between the read and the write sweep of a pass: one buffer, updated in place
        kb[0] = kb[1] = (uint32_t *)lk;
        nof[0] = nof[1] = (uint16_t *)(kb[0] + KEYS);
    } else {
        kb[0] = gk0; kb[1] = gk1; nof[0] = gn0; nof[1] = gn1;
    }
    const int E = ((C + 255) >> 8) | 1;  // BLK: keys per thread (<= kE)
    // per-key registers of the 256-thread form: key; node (11 bits: pool <= 2047) | q << 11 | head << 14 | last << 15 |
    // valid << 16 | rank inside the child << 17 (13 bits: <= KEYS)
    uint32_t kv[kE], r1[kE];
    (void)kv; (void)r1;

    // ---- 2./3. roots (:559-602): key -> root (int)(x / hX), stable; empty roots are dropped.  vToDistributeKeys is in gk1
    int cur = 0, size = 0, nA = 0;
    if (BLK) {
        uint64_t acc = 0;
#pragma unroll
        for (int e = 0; e < kE; e++) {
            const int i = tid * E + e;
            r1[e] = 0;
            if (e < E && i < C) {
                const uint32_t key = gk1[i];
                int root = (int)((float)key_x(key) / L.hX);
                root = min(root, L.nIni - 1);
                kv[e] = key;
                r1[e] = ((uint32_t)root << 11) | (1u << 16);
                acc += 1ull << (16 * root);
            }
        }
        const uint64_t cin = oct_blk_seg_excl(acc, false, wtot, wflag, lane, wave);
        if (tid == 255) *reinterpret_cast<uint64_t *>(ctl + 4) = cin + acc;  // keys per root
        __syncthreads();
        const uint64_t tot = *reinterpret_cast<const uint64_t *>(ctl + 4);
        int rstart[4], rpos[4], off = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int cnt = (int)((tot >> (16 * r)) & 0xffff);
            rstart[r] = off; rpos[r] = size;
            if (r < L.nIni && cnt > 0) {
                if (tid == 0) {
                    bndb[size] = OctBnd{(int16_t)(int)(L.hX * (float)r), 0, (int16_t)(int)(L.hX * (float)(r + 1)), (int16_t)(L.h - 2 * kBorder)};
                    segb[size] = OctSeg{off, cnt};
                }
                size++;
            }
            off += cnt;
        }
        acc = cin;
#pragma unroll
        for (int e = 0; e < kE; e++) {
            if (r1[e] & (1u << 16)) {
                const int q = (r1[e] >> 11) & 3;
                const int pos = (q == 0 ? rstart[0] : q == 1 ? rstart[1] : q == 2 ? rstart[2] : rstart[3]) + (int)((acc >> (16 * q)) & 0xffff);
                kb[0][pos] = kv[e];
                nof[0][pos] = (uint16_t)(q == 0 ? rpos[0] : q == 1 ? rpos[1] : q == 2 ? rpos[2] : rpos[3]);
                acc += 1ull << (16 * q);
            }
        }
        __syncthreads();
    } else if (WIDE) {
        int wr = 0;
        for (int r = 0; r < L.nIni; r++) {
            auto root_of = [&](uint32_t key) { return min((int)((float)key_x(key) / L.hX), L.nIni - 1); };
            for (int c = wave; c < nch; c += NW) {   // keys of root r per chunk
                const int i = c * 64 + lane;
                const bool mine = i < C && root_of(kb[1][i]) == r;
                const unsigned long long b = __ballot(mine);
                if (lane == 0) ctab[c] = (uint64_t)__popcll(b);
            }
            __syncthreads();
            chunk_scan(ctl);
            __syncthreads();
            const int cnt = ctl[2];
            for (int c = wave; c < nch; c += NW) {
                const int i = c * 64 + lane;
                uint32_t key = 0;
                bool mine = false;
                if (i < C) { key = kb[1][i]; mine = root_of(key) == r; }
                const unsigned long long b = __ballot(mine);
                if (mine) {
                    const int pos = wr + (int)(uint32_t)ctab[c] + __popcll(b & lt_mask);
                    kb[0][pos] = key;
                    nof[0][pos] = (uint16_t)size;
                }
            }
            if (cnt > 0) {
                if (tid == 0) {
                    bndb[size] = OctBnd{(int16_t)(int)(L.hX * (float)r), 0, (int16_t)(int)(L.hX * (float)(r + 1)), (int16_t)(L.h - 2 * kBorder)};
                    segb[size] = OctSeg{wr, cnt};
                }
                size++;
            }
            wr += cnt;
            __syncthreads();   // ctab is reused by the next root; the keys written here are read by the first pass
        }
    } else {
        int wr = 0;
        for (int r = 0; r < L.nIni; r++) {
            const int beg = wr;
            for (int i0 = 0; i0 < C; i0 += 64) {
                const int i = i0 + lane;
                uint32_t key = 0;
                bool mine = false;
                if (i < C) {
                    key = kb[1][i];
                    int root = (int)((float)key_x(key) / L.hX);
                    root = min(root, L.nIni - 1);
                    mine = (root == r);
                }
                const unsigned long long b = __ballot(mine);
                if (mine) {
                    const int pos = wr + __popcll(b & lt_mask);
                    kb[0][pos] = key;
                    nof[0][pos] = (uint16_t)size;
                }
                wr += __popcll(b);
            }
            const int cnt = wr - beg;
            if (cnt > 0) {
                if (lane == 0) {
                    bndb[size] = OctBnd{(int16_t)(int)(L.hX * (float)r), 0, (int16_t)(int)(L.hX * (float)(r + 1)), (int16_t)(L.h - 2 * kBorder)};
                    segb[size] = OctSeg{beg, cnt};
                }
                size++;
            }
        }
        OCT_WAVE_SYNC();
    }
    const int N = L.quota;
    OCT_TICK(1);

    // One pass over the list: every node with more than one key whose processing index is <= the break index is divided.
    //   sorted == false: breadth pass (:618-680), processing order = list order, no break
    //   sorted == true : size-ordered round (:690-750), processing order = eof[] (set from the sorted array), break at N
    auto pass = [&](bool sorted, int nB) {
        const int nxt = cur ^ 1;
        const OctBnd *B0 = bndb + cur * pool;
        const OctSeg *S0 = segb + cur * pool;
        OctBnd *B1 = bndb + nxt * pool;
        OctSeg *S1 = segb + nxt * pool;
        const uint32_t *K0 = kb[cur];
        const uint16_t *O0 = nof[cur];
        // A. child key counts of every divisible node (and, 256-thread form, the rank of every key inside its child)
        if (BLK) {
            uint64_t acc = 0;
            bool seen = false;
            // groups of six keys: load round (key + node), load round (node record), arithmetic -- the LDS latencies of a
            // group overlap and the temporaries stay within the register budget of three workgroups per CU
#pragma unroll
            for (int g = 0; g < kE; g += 6) {
                if (g < E) {
                    OctSeg sgs[6];
                    OctBnd bns[6];
#pragma unroll
                    for (int u = 0; u < 6; u++) {
                        const int e = g + u;
                        if (e < kE) { const int i = min(tid * E + e, C - 1); kv[e] = K0[i]; r1[e] = O0[i]; }
                    }
#pragma unroll
                    for (int u = 0; u < 6; u++) {
                        const int e = g + u;
                        if (e < kE) { sgs[u] = S0[r1[e]]; bns[u] = B0[r1[e]]; }
                    }
#pragma unroll
                    for (int u = 0; u < 6; u++) {
                        const int e = g + u;
                        if (e < kE) {
                            const int i = tid * E + e;
                            if (e < E && i < C) {
                                const uint32_t key = kv[e];
                                const int j = (int)r1[e];
                                const OctSeg sg = sgs[u];
                                int q = 4;
                                if (sg.cnt > 1) {
                                    const OctBnd b = bns[u];
                                    const int sx = b.x0 + ((b.x1 - b.x0 + 1) >> 1), sy = b.y0 + ((b.y1 - b.y0 + 1) >> 1);  // ceil(d / 2), :482-483
                                    q = (key_x(key) < sx ? 0 : 1) + (key_y(key) < sy ? 0 : 2);
                                }
                                const bool hd = (i == sg.beg), lst = (i == sg.beg + sg.cnt - 1);
                                if (hd) { acc = 0; seen = true; }
                                if (q < 4) acc += 1ull << (16 * q);
                                r1[e] = (uint32_t)j | ((uint32_t)q << 11) | ((uint32_t)hd << 14) | ((uint32_t)lst << 15) | (1u << 16);
                            } else {
                                r1[e] = 0;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 6; u++) if (g + u < kE) r1[g + u] = 0;
                }
            }
            acc = oct_blk_seg_excl(acc, seen, wtot, wflag, lane, wave);
#pragma unroll
            for (int e = 0; e < kE; e++) {
                if (r1[e] & (1u << 16)) {
                    const int q = (r1[e] >> 11) & 7;
                    if (r1[e] & (1u << 14)) acc = 0;
                    if (q < 4) {
                        r1[e] |= (uint32_t)((acc >> (16 * q)) & 0xffff) << 17;
                        acc += 1ull << (16 * q);
                        if (r1[e] & (1u << 15))
                            *reinterpret_cast<uint4 *>(&hist[4 * (r1[e] & 0x7ff)]) =
                                make_uint4((uint32_t)(acc & 0xffff), (uint32_t)((acc >> 16) & 0xffff), (uint32_t)((acc >> 32) & 0xffff), (uint32_t)(acc >> 48));
                    }
                }
            }
        } else if (WIDE) {
            for (int i = tid; i < size * 4; i += NT) hist[i] = 0;
            __syncthreads();
            for (int c = wave; c < nch; c += NW) {
                const int i0 = c * 64, i = i0 + lane;
                int j = -1, q = 4;
                if (i < C) {
                    const uint32_t key = K0[i];
                    j = O0[i];
                    if (S0[j].cnt > 1) {
                        const OctBnd b = B0[j];
                        const int sx = b.x0 + ((b.x1 - b.x0 + 1) >> 1), sy = b.y0 + ((b.y1 - b.y0 + 1) >> 1);
                        q = (key_x(key) < sx ? 0 : 1) + (key_y(key) < sy ? 0 : 2);
                        atomicAdd(&hist[4 * j + q], 1u);
                    }
                }
                // the chunk's contribution to the node it ends in: keys of that node per child (what a later chunk of the same node finds before it)
                const int jl = __shfl(j, min(63, C - 1 - i0));
                const bool same = i < C && j == jl;
                const unsigned long long t0 = __ballot(same && q == 0), t1 = __ballot(same && q == 1), t2 = __ballot(same && q == 2), t3 = __ballot(same && q == 3);
                if (lane == 0)
                    ctab[c] = (uint64_t)__popcll(t0) | ((uint64_t)__popcll(t1) << 16) | ((uint64_t)__popcll(t2) << 32) | ((uint64_t)__popcll(t3) << 48);
            }
            __syncthreads();
            chunk_scan(ctl);   // (the barrier that follows publishes it)
        } else {
            for (int i = lane; i < size * 4; i += 64) hist[i] = 0;
            OCT_WAVE_SYNC();
            for (int i0 = 0; i0 < C; i0 += 64) {
                const int i = i0 + lane;
                if (i < C) {
                    const uint32_t key = K0[i];
                    const int j = O0[i];
                    if (S0[j].cnt > 1) {
                        const OctBnd b = B0[j];
                        const int sx = b.x0 + ((b.x1 - b.x0 + 1) >> 1), sy = b.y0 + ((b.y1 - b.y0 + 1) >> 1);
                        const int q = (key_x(key) < sx ? 0 : 1) + (key_y(key) < sy ? 0 : 2);
                        atomicAdd(&hist[4 * j + q], 1u);
                    }
                }
            }
        }
        OCT_SYNC_ALL();
        OCT_TICK(2);
        if (wave == 0) {
            // B1. per node: number of children / divisible children, by processing index
            {
                int ebase = 0;
                for (int j0 = 0; j0 < size; j0 += 64) {
                    const int j = j0 + lane;
                    bool cand = false;
                    int ne = 0, nx = 0;
                    if (j < size && S0[j].cnt > 1) {
                        cand = true;
                        const uint4 h = *reinterpret_cast<const uint4 *>(&hist[4 * j]);
                        ne = (h.x > 0) + (h.y > 0) + (h.z > 0) + (h.w > 0);
                        nx = (h.x > 1) + (h.y > 1) + (h.z > 1) + (h.w > 1);
                    }
                    const unsigned long long bc = __ballot(cand);
                    if (cand) {
                        int e;
                        if (sorted) e = eof[j];
                        else { e = ebase + __popcll(bc & lt_mask); eof[j] = (uint16_t)e; }
                        nebe[e] = (uint16_t)(ne | (nx << 3));
                    }
                    ebase += __popcll(bc);
                }
                if (!sorted) nB = ebase;
            }
            OCT_WAVE_SYNC();
            // B2. over the processing order: children created before each node, and the break index m (:737)
            int m = nB - 1, T = 0, newA = 0;
            {
                int cb = 0, ab = 0;
                bool done = false;
                for (int e0 = 0; e0 < nB && !done; e0 += 64) {
                    const int e = e0 + lane;
                    int ne = 0, nx = 0;
                    if (e < nB) { const int v = nebe[e]; ne = v & 7; nx = v >> 3; }
                    const int ine = wave_incl_scan(ne), inx = wave_incl_scan(nx);
                    if (e < nB) { cbe[e] = (uint16_t)(cb + ine - ne); abe[e] = (uint16_t)(ab + inx - nx); }
                    int last = min(63, nB - 1 - e0);
                    if (sorted) {  // list size after dividing e: size + (children so far) - (nodes divided so far)
                        const bool reach = e < nB && size + (cb + ine) - (e + 1) >= N;
                        const unsigned long long br = __ballot(reach);
                        if (br) { last = __ffsll((long long)br) - 1; m = e0 + last; done = true; }
                    }
                    cb += __builtin_amdgcn_readlane(ine, last);
                    ab += __builtin_amdgcn_readlane(inx, last);
                }
                T = cb; newA = ab;
            }
            OCT_WAVE_SYNC();
            // B3. write the new list: children at T-1-creation index, the others after them in their old order
            int ubase = 0;
            for (int j0 = 0; j0 < size; j0 += 64) {
                const int j = j0 + lane;
                bool keepn = false, divd = false;
                OctSeg sg{0, 0};
                OctBnd b{0, 0, 0, 0};
                int e = 0;
                if (j < size) {
                    sg = S0[j]; b = B0[j];
                    if (sg.cnt > 1) { e = eof[j]; divd = e <= m; }
                    keepn = !divd;
                }
                const unsigned long long bk = __ballot(keepn);
                if (keepn) {
                    const int pos = T + ubase + __popcll(bk & lt_mask);
                    B1[pos] = b; S1[pos] = sg;
                    npos[j] = (uint16_t)pos;
                }
                ubase += __popcll(bk);
                if (divd) {
                    npos[j] = 0xffff;
                    const uint4 h = *reinterpret_cast<const uint4 *>(&hist[4 * j]);
                    const int cc[4] = {(int)h.x, (int)h.y, (int)h.z, (int)h.w};
                    const int sx = b.x0 + ((b.x1 - b.x0 + 1) >> 1), sy = b.y0 + ((b.y1 - b.y0 + 1) >> 1);
                    const int cx0[4] = {b.x0, sx, b.x0, sx}, cx1[4] = {sx, b.x1, sx, b.x1};
                    const int cy0[4] = {b.y0, b.y0, sy, sy}, cy1[4] = {sy, sy, b.y1, b.y1};
                    int ci = cbe[e], ai = abe[e], off = sg.beg;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        hist[4 * j + k] = (uint32_t)off;  // becomes the (running) scatter offset of child k
                        if (cc[k] > 0) {
                            const int pos = T - 1 - ci;
                            ci++;
                            B1[pos] = OctBnd{(int16_t)cx0[k], (int16_t)cy0[k], (int16_t)cx1[k], (int16_t)cy1[k]};
                            S1[pos] = OctSeg{off, cc[k]};
                            cpos[4 * j + k] = (uint16_t)pos;
                            if (cc[k] > 1) {
                                sa[ai] = ((uint64_t)(uint32_t)cc[k] << 32) | ((uint64_t)(uint16_t)cx0[k] << 16) | (uint64_t)pos;
                                ai++;
                            }
                        }
                        off += cc[k];
                    }
                }
            }
            if (lane == 0) { ctl[0] = T + ubase; ctl[1] = newA; }
        }
        OCT_SYNC_ALL();
        OCT_TICK(3);
        size = ctl[0];
        nA = ctl[1];
        // C. stable scatter of the keys of the divided nodes (the parent's key order is kept inside every child)
        uint32_t *K1 = kb[nxt];
        uint16_t *O1 = nof[nxt];
        if (BLK) {
#pragma unroll
            for (int e = 0; e < kE; e++) {
                if (r1[e] & (1u << 16)) {
                    const int j = r1[e] & 0x7ff, q = (r1[e] >> 11) & 7;
                    const int np = npos[j];
                    if (np == 0xffff) {
                        const int slot = 4 * j + q;
                        const int pos = (int)hist[slot] + (int)(r1[e] >> 17);
                        K1[pos] = kv[e];
                        O1[pos] = cpos[slot];
                    } else {
                        const int i = tid * E + e;
                        K1[i] = kv[e];
                        O1[i] = (uint16_t)np;
                    }
                }
            }
        } else if (WIDE) {
            for (int c = wave; c < nch; c += NW) {
                const int i0 = c * 64, i = i0 + lane;
                uint32_t key = 0;
                int j = 0, q = 4, np = 0, s0 = 0, beg = 0;
                if (i < C) {
                    key = K0[i];
                    j = O0[i];
                    np = npos[j];
                    if (np == 0xffff) {
                        const OctBnd b = B0[j];
                        const int sx = b.x0 + ((b.x1 - b.x0 + 1) >> 1), sy = b.y0 + ((b.y1 - b.y0 + 1) >> 1);
                        q = (key_x(key) < sx ? 0 : 1) + (key_y(key) < sy ? 0 : 2);
                        beg = S0[j].beg;
                        s0 = max(beg - i0, 0);  // first lane of this node's run inside the chunk
                    }
                }
                const unsigned long long b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
                if (i < C) {
                    if (q < 4) {
                        const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
                        const int rank = __popcll(bq & lt_mask & ~((1ull << s0) - 1ull));
                        int earlier = 0;   // keys of this node and child in earlier chunks: every chunk from the node's first one up to this one ends in the node
                        if (beg < i0) earlier = (int)(((ctab[c] >> (16 * q)) - (ctab[beg >> 6] >> (16 * q))) & 0xffff);
                        const int slot = 4 * j + q;
                        const int pos = (int)hist[slot] + earlier + rank;
                        K1[pos] = key;
                        O1[pos] = cpos[slot];
                    } else {
                        K1[i] = key;
                        O1[i] = (uint16_t)np;
                    }
                }
            }
        } else {
            for (int i0 = 0; i0 < C; i0 += 64) {
                const int i = i0 + lane;
                uint32_t key = 0;
                int j = 0, q = 4, np = 0, s0 = 0;
                if (i < C) {
                    key = K0[i];
                    j = O0[i];
                    np = npos[j];
                    if (np == 0xffff) {
                        const OctBnd b = B0[j];
                        const int sx = b.x0 + ((b.x1 - b.x0 + 1) >> 1), sy = b.y0 + ((b.y1 - b.y0 + 1) >> 1);
                        q = (key_x(key) < sx ? 0 : 1) + (key_y(key) < sy ? 0 : 2);
                        s0 = max(S0[j].beg - i0, 0);  // first lane of this node's run inside the chunk
                    }
                }
                const unsigned long long b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
                if (i < C) {
                    if (q < 4) {
                        const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
                        const int rank = __popcll(bq & lt_mask & ~((1ull << s0) - 1ull));
                        const int slot = 4 * j + q;
                        const int pos = (int)hist[slot] + rank;
                        K1[pos] = key;
                        O1[pos] = cpos[slot];
                        atomicAdd(&hist[slot], 1u);  // LDS/memory operations of a wave are performed in order: the next chunk sees it
                    } else {
                        K1[i] = key;
                        O1[i] = (uint16_t)np;
                    }
                }
            }
        }
        OCT_SYNC_ALL();
        OCT_TICK(4);
        cur = nxt;
        if (dbg && tid == 0) dbg[sorted ? 8 : 7] += 1;
    };

    // ---- 4. main loop (:604-755): breadth passes, then size-ordered rounds once size + 3 * nToExpand > N (:686) --------
    bool finish = (size == 0), sorted = false;
    while (!finish) {
        const int prevSize = size;
        const int nB = nA;
        if (sorted) {
            if (wave == 0) {
                // vPrevSizeAndPointerToNode = vSizeAndPointerToNode; sort (:694-697)
                for (int i = lane; i < nB; i += 64) sb[i] = sa[i];
                OCT_WAVE_SYNC();
                if (!oct_par_sort(sb, reinterpret_cast<uint64_t *>(hist), nB, ls, rs, stack, lane, dbg)) {
                    OCT_WAVE_SYNC();
                    for (int i = lane; i < nB; i += 64) sb[i] = sa[i];
                    OCT_WAVE_SYNC();
                    LdsArr arr{sb};
                    if (lane == 0) oct_std_sort(arr, nB, stack);
                    OCT_WAVE_SYNC();
                }
                for (int k = lane; k < nB; k += 64) eof[(int)(sb[k] & 0xffff)] = (uint16_t)(nB - 1 - k);  // divided from the back (:700)
            }
            OCT_SYNC_ALL();
            OCT_TICK(5);
            if (dbg && tid == 0) dbg[10] = max(dbg[10], (long long)nB);
        }
        pass(sorted, nB);
        if (size >= N || size == prevSize) finish = true;
        else if (!sorted && size + nA * 3 > N) sorted = true;
    }

    // ---- 5. best response per node, first wins ties (:757-776), in list order ---------------------------------------
    const int nn = size;
    uint32_t *best = hist;
    for (int i = tid; i < nn; i += NT) best[i] = 0;
    OCT_SYNC_ALL();
    for (int i = tid; i < C; i += NT) {
        const uint32_t key = kb[cur][i];
        atomicMax(&best[nof[cur][i]], ((uint32_t)key_s(key) << 20) | (0xfffffu - (uint32_t)i));
    }
    OCT_SYNC_ALL();
    for (int i = tid; i < nn && i < L.lvl_cap; i += NT) {
        const uint32_t key = kb[cur][0xfffffu - (best[i] & 0xfffffu)];
        // keypoints[i].pt += minBorder (:884-886): store level coordinates
        out[i] = pack_key(key_x(key) + kBorder, key_y(key) + kBorder, key_s(key));
    }
    if (tid == 0) {
        if (nn > L.lvl_cap) atomicExch(err, 3);
        *lvlcnt_out = min(nn, L.lvl_cap);
    }
    OCT_TICK(6);
    if (dbg && tid == 0) { dbg[0] += (long long)wall_clock64() - tstart; dbg[9] = C; dbg[11] = nn; }
#undef OCT_TICK
#undef OCT_SYNC_ALL
}

// The 256-thread form handles a (frame, level) when its keys fit the LDS buffers, it has at most four roots and node indices
// fit 11 bits; the single-wave chunked form (one wave, passes over global key buffers: k_octree_rest) handles the others.  Both kernels are
// launched over all (frame, level) pairs and each returns at once for the pairs that belong to the other.
__device__ __forceinline__ bool oct_blk_form(const LevelInfo &L, int C) { return C <= kOctParLdsKeys && L.nIni <= 4 && L.pool <= 2047; }

// k_octree_par_t: the 256-thread form in two tiers (the first tier gathers the keys itself: compact_level).  Sized for 4096 candidates on every level the form
// needs 45 KB of LDS and 161 VGPRs (three workgroups per CU), and a level with 1100 candidates still walks 17 key slots per thread in
// every unrolled sweep.  The small tier (LO < C <= 2048: every level of the EuRoC-shaped bench) has 9 slots, 118 VGPRs and 33 KB:
// four workgroups per CU (2048 workgroups = two dispatch rounds instead of three) and about 30 % fewer instructions; levels with
// 2048 < C <= 4096 and the rest take the second launch (k_octree_rest).  Round 4: the first tier at 1792 keys (7 slots per thread) and 96 VGPRs
// (__launch_bounds__(256, 5): five spilled registers) fits FIVE workgroups per CU (31.7 KB of LDS each): 89 -> 85 us, TUM-VI step -1.8 %.  Round 3, profiles/r03_a_ab_prepared_kernels.log: step 1.175 ->
// 1.142 ms against the single-tier kernel, which is gone.
// grid (B, nlevels), block 256, dynamic LDS = oct_par_pool_bytes(max pool) + KEYS * 6
template <int KEYS, int LO>
__global__ __launch_bounds__(256, (KEYS <= 2048 ? 5 : 3)) void k_octree_par_t(const LevelInfo *__restrict__ lv, size_t ent_frame_stride,
                                                                              uint32_t *__restrict__ keys1, uint32_t *__restrict__ lvlkp,
                                                                              size_t lvlkp_frame_stride, int32_t *__restrict__ lvlcnt, int nlevels,
                                                                              int32_t *__restrict__ cand_total, int32_t *__restrict__ err,
                                                                              int max_pool, const int32_t *__restrict__ cellcnt, int total_cells,
                                                                              const uint32_t *__restrict__ cellent) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int level = blockIdx.y, f = blockIdx.x;
    const LevelInfo L = lv[level];
    uint32_t *gk1 = keys1 + (size_t)f * ent_frame_stride + L.cand_off;
    int C;
    if (LO < 0 && cellcnt) {   // the first tier gathers vToDistributeKeys of EVERY (frame, level) itself (k_compact's work without its launch: the
                               // second launch reads the keys and the count it leaves behind)
        C = __builtin_amdgcn_readfirstlane(compact_level(L, cellcnt + (size_t)f * total_cells + L.cell_base,
                                                         cellent + (size_t)f * ent_frame_stride + L.cand_off, gk1));
        if (threadIdx.x == 0) cand_total[f * nlevels + level] = C;
        __syncthreads();   // the workgroup's own stores to gk1 are visible to its loads below
    } else C = __builtin_amdgcn_readfirstlane(cand_total[f * nlevels + level]);
    if (!(C > LO && C <= KEYS && L.nIni <= 4 && L.pool <= 2047)) return;
    octree_par_body<true, KEYS>(L, smem, max_pool, C, nullptr, gk1, nullptr, nullptr, lvlkp + (size_t)f * lvlkp_frame_stride + L.lvl_off,
                                lvlcnt + f * nlevels + level, err, nullptr);
}

// k_octree_rest (round 4): the second tier of the 256-thread form (2048 < C <= 4096) and the single-wave chunked form (levels the 256-thread form does not
// take: more than 4096 candidates, more than four roots, a node pool beyond 2047) as ONE launch.  Both are near-empty launches for the bench's frames
// (every level has at most 2048 candidates), 5 us each plus a launch boundary on the latency-bound stretch between the FAST strips and k_describe -- and
// on every single-frame call.  A workgroup reads its level's candidate count (left by the first tier) and takes the form that applies; in the
// single-wave form waves 1..3 leave at once.
// grid (B, nlevels), block 256, dynamic LDS = max(oct_par_lds_bytes(max pool), oct_par_pool_bytes(max pool))
__global__ __launch_bounds__(256, 3) void k_octree_rest(const LevelInfo *__restrict__ lv, size_t ent_frame_stride, uint32_t *__restrict__ keys0,
                                                        uint32_t *__restrict__ keys1, uint16_t *__restrict__ nof0, uint16_t *__restrict__ nof1,
                                                        uint32_t *__restrict__ lvlkp, size_t lvlkp_frame_stride, int32_t *__restrict__ lvlcnt,
                                                        int nlevels, const int32_t *__restrict__ cand_total, int32_t *__restrict__ err, int max_pool, int single_wave) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int level = blockIdx.y, f = blockIdx.x;
    const LevelInfo L = lv[level];
    int32_t *cnt_out = lvlcnt + f * nlevels + level;
    const int C = __builtin_amdgcn_readfirstlane(cand_total[f * nlevels + level]);
    uint32_t *gk1 = keys1 + (size_t)f * ent_frame_stride + L.cand_off;
    if (oct_blk_form(L, C)) {
        if (C <= kOctTier1Keys) return;   // the first tier's (k_octree_par_t<kOctTier1Keys, -1>)
        octree_par_body<true, kOctParLdsKeys>(L, smem, max_pool, C, nullptr, gk1, nullptr, nullptr, lvlkp + (size_t)f * lvlkp_frame_stride + L.lvl_off, cnt_out,
                                              err, nullptr);
        return;
    }
    uint32_t *gk0 = keys0 + (size_t)f * ent_frame_stride + L.cand_off;
    uint16_t *gn0 = nof0 + (size_t)f * ent_frame_stride + L.cand_off;
    uint16_t *gn1 = nof1 + (size_t)f * ent_frame_stride + L.cand_off;
    if (C <= kOctWideMaxKeys && !single_wave) {   // the global-memory form on the whole workgroup (round 6)
        octree_par_body<false, kOctParLdsKeys, true>(L, smem, max_pool, C, gk0, gk1, gn0, gn1, lvlkp + (size_t)f * lvlkp_frame_stride + L.lvl_off, cnt_out, err, nullptr);
        return;
    }
    if (threadIdx.x >= 64) return;   // the chunked single-wave form (levels beyond 65535 candidates; ORBX_OCTREE=w1)
    if (C >= 0xfffff) {  // the best-response pick packs the key position into 20 bits
        if (threadIdx.x == 0) { atomicExch(err, 2); *cnt_out = 0; }
        return;
    }
    octree_par_body<false>(L, smem, max_pool, C, gk0, gk1, gn0, gn1, lvlkp + (size_t)f * lvlkp_frame_stride + L.lvl_off, cnt_out, err, nullptr);
}

}  // namespace orbx
