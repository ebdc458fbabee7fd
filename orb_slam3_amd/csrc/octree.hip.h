// octree.hip.h -- k_octree: exact device emulation of ORBextractor::DistributeOctTree
// (/root/reference/src/ORBextractor.cc:555-779, DivideNode :480-536, compareNodes :538-553).
//
// The reference algorithm is sequential and order sensitive: nodes live in a std::list (children are
// push_front'ed, parents erased while iterating), the size-ordered expansion uses std::sort with a comparator
// that leaves ties, and the output order is the final list order.  To return bit-identical keypoint sets in
// identical order, one wave (64 lanes) per (frame, level) runs that control flow verbatim:
//   * lane 0 is the "scalar thread" for list surgery and for the libstdc++ introsort replica;
//   * all 64 lanes cooperate on the data-parallel parts: gathering the cell slots in reference order, the
//     stable 4-way key partition of DivideNode (ballot + prefix popcount), and the per-node best-response pick.
// Node state lives in LDS; keys (packed x|y<<12|score<<24, relative to the 16-px border) ping-pong between two
// global buffers that stay L2 resident.
#pragma once

#include "orbx_internal.h"

namespace orbx {

constexpr int kNil = 0xffff;

struct OctLds {  // carved from dynamic LDS, `pool` entries each
    int16_t *x0, *y0, *x1, *y1;  // UL.x, UL.y, UR.x, BR.y
    int32_t *beg, *cnt;          // key range in buffer `buf`
    uint8_t *buf, *leaf;         // ping-pong id, bNoMore
    uint16_t *next, *prev;       // list links
    uint16_t *freelist;          // stack of free node ids
    uint64_t *sa, *sb;           // vSizeAndPointerToNode / vPrevSizeAndPointerToNode: cnt<<32 | ulx<<16 | node
    uint16_t *order;             // final list order
};

constexpr int kOctLdsKeys = 3072;  // keys per ping-pong buffer kept in LDS (levels with more candidates use global memory)

__host__ __device__ inline size_t oct_pool_bytes(int pool) {
    // 64 bytes of scalars, then the per-node arrays (8-byte arrays first for alignment), rounded to 16
    return (64 + (size_t)pool * (8 + 8 + 4 + 4 + 2 * 4 + 2 * 3 + 2 + 1 + 1) + 64 + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t oct_lds_bytes(int pool) { return oct_pool_bytes(pool) + 2 * (size_t)kOctLdsKeys * 4; }

// ---- libstdc++ (GCC 11) std::sort replica on u64 entries compared by (entry >> 16) ----------------------
// compareNodes(e1,e2): e1.first < e2.first, or equal and e1.second->UL.x < e2.second->UL.x.  With the packing
// cnt<<32 | ulx<<16 | node this is (a >> 16) < (b >> 16).
__device__ __forceinline__ bool oct_less(uint64_t a, uint64_t b) { return (a >> 16) < (b >> 16); }

__device__ inline void oct_unguarded_linear_insert(uint64_t *v, int last) {
    const uint64_t val = v[last];
    int next = last - 1;
    while (oct_less(val, v[next])) {
        v[last] = v[next];
        last = next;
        --next;
    }
    v[last] = val;
}
__device__ inline void oct_insertion_sort(uint64_t *v, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (oct_less(v[i], v[first])) {
            const uint64_t val = v[i];
            for (int k = i; k > first; --k) v[k] = v[k - 1];  // move_backward
            v[first] = val;
        } else {
            oct_unguarded_linear_insert(v, i);
        }
    }
}
__device__ inline void oct_push_heap(uint64_t *v, int first, int hole, int top, uint64_t value) {
    int parent = (hole - 1) / 2;
    while (hole > top && oct_less(v[first + parent], value)) {
        v[first + hole] = v[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v[first + hole] = value;
}
__device__ inline void oct_adjust_heap(uint64_t *v, int first, int hole, int len, uint64_t value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (oct_less(v[first + child], v[first + child - 1])) child--;
        v[first + hole] = v[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v[first + hole] = v[first + child - 1];
        hole = child - 1;
    }
    oct_push_heap(v, first, hole, top, value);
}
__device__ inline void oct_heap_sort(uint64_t *v, int first, int last) {  // __partial_sort(first,last,last)
    const int len = last - first;
    if (len >= 2) {  // __make_heap
        int parent = (len - 2) / 2;
        while (true) {
            const uint64_t value = v[first + parent];
            oct_adjust_heap(v, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    int l = last;
    while (l - first > 1) {  // __sort_heap / __pop_heap
        --l;
        const uint64_t value = v[l];
        v[l] = v[first];
        oct_adjust_heap(v, first, 0, l - first, value);
    }
}
__device__ inline void oct_std_sort(uint64_t *v, int n) {
    if (n <= 0) return;
    // __introsort_loop with an explicit stack for the right-hand recursion
    int stack_first[64], stack_last[64], stack_depth[64];
    int sp = 0;
    int lg = 31 - __clz(n);
    stack_first[sp] = 0; stack_last[sp] = n; stack_depth[sp] = 2 * lg; sp++;
    while (sp > 0) {
        sp--;
        int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
        while (last - first > 16) {
            if (depth == 0) { oct_heap_sort(v, first, last); break; }
            --depth;
            // __unguarded_partition_pivot
            const int mid = first + (last - first) / 2;
            {   // __move_median_to_first(first, first+1, mid, last-1)
                const int a = first + 1, b = mid, c = last - 1;
                int m;
                if (oct_less(v[a], v[b])) {
                    if (oct_less(v[b], v[c])) m = b;
                    else if (oct_less(v[a], v[c])) m = c;
                    else m = a;
                } else if (oct_less(v[a], v[c])) m = a;
                else if (oct_less(v[b], v[c])) m = c;
                else m = b;
                const uint64_t t = v[first]; v[first] = v[m]; v[m] = t;
            }
            int lo = first + 1, hi = last;
            const uint64_t pivot = v[first];
            while (true) {  // __unguarded_partition
                while (oct_less(v[lo], pivot)) ++lo;
                --hi;
                while (oct_less(pivot, v[hi])) --hi;
                if (!(lo < hi)) break;
                const uint64_t t = v[lo]; v[lo] = v[hi]; v[hi] = t;
                ++lo;
            }
            const int cut = lo;
            // recurse on [cut, last) first (as libstdc++ does), then continue with [first, cut)
            // order of processing does not change the result: the two ranges are disjoint
            stack_first[sp] = cut; stack_last[sp] = last; stack_depth[sp] = depth; sp++;
            last = cut;
        }
    }
    // __final_insertion_sort
    if (n > 16) {
        oct_insertion_sort(v, 0, 16);
        for (int i = 16; i != n; ++i) oct_unguarded_linear_insert(v, i);
    } else {
        oct_insertion_sort(v, 0, n);
    }
}

// debug kernel: sort (count, ulx) pairs with the replica; perm out
__global__ void k_debug_sort(const int32_t *count, const int32_t *ulx, int n, int32_t *perm, uint64_t *scratch) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < n; i++) scratch[i] = ((uint64_t)(uint32_t)count[i] << 32) | ((uint64_t)(uint32_t)ulx[i] << 16) | (uint64_t)i;
        // NB: node ids here are 16 bit; n <= 65535
        oct_std_sort(scratch, n);
        for (int i = 0; i < n; i++) perm[i] = (int32_t)(scratch[i] & 0xffff);
    }
}

// ---- the quad-tree kernel ----------------------------------------------------------------------------------
// grid (nlevels, B), block 64, dynamic LDS = oct_lds_bytes(max pool): node pool + two LDS key buffers
__global__ __launch_bounds__(64) void k_octree(const LevelInfo *__restrict__ lv, const int32_t *__restrict__ cellcnt,
                                               int total_cells, const uint32_t *__restrict__ cellent,
                                               size_t ent_frame_stride, uint32_t *__restrict__ keys0,
                                               uint32_t *__restrict__ keys1, uint32_t *__restrict__ lvlkp,
                                               size_t lvlkp_frame_stride, int32_t *__restrict__ lvlcnt, int nlevels,
                                               int32_t *__restrict__ cand_total, int32_t *__restrict__ err, int max_pool) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int level = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    const LevelInfo L = lv[level];
    const int pool = L.pool;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // carve LDS
    OctLds S;
    // scalars live in the dynamic region too (keeps the carve 16-byte aligned, no static LDS in front)
    int *shv = (int *)smem;
#define sh_head shv[0]
#define sh_tail shv[1]
#define sh_size shv[2]
#define sh_nfree shv[3]
#define sh_nA shv[4]
#define sh_nToExpand shv[5]
#define sh_err shv[6]
    {
        uint8_t *p = smem + 64;
        S.sa = (uint64_t *)p; p += (size_t)pool * 8;
        S.sb = (uint64_t *)p; p += (size_t)pool * 8;
        S.beg = (int32_t *)p; p += (size_t)pool * 4;
        S.cnt = (int32_t *)p; p += (size_t)pool * 4;
        S.x0 = (int16_t *)p; p += (size_t)pool * 2;
        S.y0 = (int16_t *)p; p += (size_t)pool * 2;
        S.x1 = (int16_t *)p; p += (size_t)pool * 2;
        S.y1 = (int16_t *)p; p += (size_t)pool * 2;
        S.next = (uint16_t *)p; p += (size_t)pool * 2;
        S.prev = (uint16_t *)p; p += (size_t)pool * 2;
        S.freelist = (uint16_t *)p; p += (size_t)pool * 2;
        S.order = (uint16_t *)p; p += (size_t)pool * 2;
        S.buf = p; p += pool;
        S.leaf = p; p += pool;
    }
    const uint32_t *ent = cellent + (size_t)f * ent_frame_stride + L.cand_off;
    const int32_t *ccnt = cellcnt + (size_t)f * total_cells + L.cell_base;
    const int ncell = L.nCols * L.nRows;

    // ---- 1. gather the cell slots in reference order (cell row-major); C = total -----------------------------
    int C = 0;
    for (int c0 = 0; c0 < ncell; c0 += 64) {
        int n = (c0 + lane < ncell) ? ccnt[c0 + lane] : 0;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) n += __shfl_xor(n, s);
        C += n;
    }
    // key ping-pong buffers: LDS when the level's candidates fit, else the global scratch slabs (generic pointers)
    uint32_t *kb[2];
    if (C <= kOctLdsKeys) {
        uint32_t *lk = reinterpret_cast<uint32_t *>(smem + oct_pool_bytes(max_pool));
        kb[0] = lk; kb[1] = lk + kOctLdsKeys;
    } else {
        kb[0] = keys0 + (size_t)f * ent_frame_stride + L.cand_off;
        kb[1] = keys1 + (size_t)f * ent_frame_stride + L.cand_off;
    }
    {
        int run = 0;
        for (int c0 = 0; c0 < ncell; c0 += 64) {
            const int c = c0 + lane;
            const int n = (c < ncell) ? ccnt[c] : 0;
            int incl = n;  // inclusive scan over the wave
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const int t = __shfl_up(incl, s);
                if (lane >= s) incl += t;
            }
            const int excl = run + incl - n;
            for (int k = 0; k < n; k++) kb[1][excl + k] = ent[(size_t)c * L.cell_cap + k];
            run += __shfl(incl, 63);
        }
    }
    __syncthreads();
    if (lane == 0 && cand_total) cand_total[f * nlevels + level] = C;

    // ---- 2. roots (:559-587): nIni nodes of width hX; key -> root (int)(x / hX); stable split into kb[0] ----
    // ---- 3. list init (:589-602): drop empty roots, single-key roots are leaves
    if (lane == 0) {
        sh_head = kNil; sh_tail = kNil; sh_size = 0; sh_nA = 0; sh_nToExpand = 0; sh_err = 0;
        int nf = 0;
        for (int i = pool - 1; i >= 0; i--) S.freelist[nf++] = (uint16_t)i;  // pop gives 0,1,2,...
        sh_nfree = nf;
    }
    __syncthreads();
    {
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
                if (mine) kb[0][wr + __popcll(b & lt_mask)] = key;
                wr += __popcll(b);
            }
            const int cnt = wr - beg;
            if (lane == 0 && cnt > 0) {  // push_back
                const int id = S.freelist[--sh_nfree];
                S.x0[id] = (int16_t)(int)(L.hX * (float)r);
                S.x1[id] = (int16_t)(int)(L.hX * (float)(r + 1));
                S.y0[id] = 0;
                S.y1[id] = (int16_t)(L.h - 2 * kBorder);
                S.beg[id] = beg; S.cnt[id] = cnt; S.buf[id] = 0; S.leaf[id] = (cnt == 1);
                S.next[id] = kNil; S.prev[id] = (uint16_t)sh_tail;
                if (sh_tail != kNil) S.next[sh_tail] = (uint16_t)id; else sh_head = id;
                sh_tail = id;
                sh_size++;
            }
        }
    }
    __syncthreads();

    const int N = L.quota;

    // DivideNode + child bookkeeping for node `id` (all lanes call it with the same id).
    // Children are push_front'ed in the order n1..n4; those with >1 keys are appended to S.sa.
    auto divide = [&](int id, bool count_expand) {
        const int px0 = S.x0[id], py0 = S.y0[id], px1 = S.x1[id], py1 = S.y1[id];
        const int beg = S.beg[id], cnt = S.cnt[id], sb = S.buf[id];
        const int halfX = (int)ceilf((float)(px1 - px0) / 2), halfY = (int)ceilf((float)(py1 - py0) / 2);
        const int sx = px0 + halfX, sy = py0 + halfY;
        const uint32_t *src = kb[sb] + beg;
        uint32_t *dst = kb[sb ^ 1] + beg;
        int c[4] = {0, 0, 0, 0};
        if (cnt <= 64) {
            uint32_t key = 0;
            int q = 4;
            if (lane < cnt) {
                key = src[lane];
                q = (key_x(key) < sx ? 0 : 1) + (key_y(key) < sy ? 0 : 2);
            }
            const unsigned long long b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
            c[0] = __popcll(b0); c[1] = __popcll(b1); c[2] = __popcll(b2); c[3] = __popcll(b3);
            if (q < 4) {
                const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
                const int off = (q > 0 ? c[0] : 0) + (q > 1 ? c[1] : 0) + (q > 2 ? c[2] : 0);
                dst[off + __popcll(bq & lt_mask)] = key;
            }
        } else {
            for (int i0 = 0; i0 < cnt; i0 += 64) {
                const int i = i0 + lane;
                int q = 4;
                if (i < cnt) { const uint32_t key = src[i]; q = (key_x(key) < sx ? 0 : 1) + (key_y(key) < sy ? 0 : 2); }
                c[0] += __popcll(__ballot(q == 0)); c[1] += __popcll(__ballot(q == 1));
                c[2] += __popcll(__ballot(q == 2)); c[3] += __popcll(__ballot(q == 3));
            }
            int o0 = 0, o1 = c[0], o2 = c[0] + c[1], o3 = c[0] + c[1] + c[2];
            for (int i0 = 0; i0 < cnt; i0 += 64) {
                const int i = i0 + lane;
                uint32_t key = 0;
                int q = 4;
                if (i < cnt) { key = src[i]; q = (key_x(key) < sx ? 0 : 1) + (key_y(key) < sy ? 0 : 2); }
                const unsigned long long b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
                if (q < 4) {
                    const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
                    const int off = q == 0 ? o0 : q == 1 ? o1 : q == 2 ? o2 : o3;
                    dst[off + __popcll(bq & lt_mask)] = key;
                }
                o0 += __popcll(b0); o1 += __popcll(b1); o2 += __popcll(b2); o3 += __popcll(b3);
            }
        }
        if (lane == 0) {
            // child rectangles (:485-507)
            const int cx0[4] = {px0, sx, px0, sx}, cx1[4] = {sx, px1, sx, px1};
            const int cy0[4] = {py0, py0, sy, sy}, cy1[4] = {sy, sy, py1, py1};
            int off = beg;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (c[k] > 0) {
                    if (sh_nfree <= 0) { sh_err = 1; break; }
                    const int ch = S.freelist[--sh_nfree];
                    S.x0[ch] = (int16_t)cx0[k]; S.x1[ch] = (int16_t)cx1[k];
                    S.y0[ch] = (int16_t)cy0[k]; S.y1[ch] = (int16_t)cy1[k];
                    S.beg[ch] = off; S.cnt[ch] = c[k]; S.buf[ch] = (uint8_t)(sb ^ 1); S.leaf[ch] = (c[k] == 1);
                    // push_front
                    S.prev[ch] = kNil; S.next[ch] = (uint16_t)sh_head;
                    if (sh_head != kNil) S.prev[sh_head] = (uint16_t)ch; else sh_tail = ch;
                    sh_head = ch;
                    sh_size++;
                    if (c[k] > 1) {
                        if (count_expand) sh_nToExpand++;
                        S.sa[sh_nA++] = ((uint64_t)(uint32_t)c[k] << 32) | ((uint64_t)(uint16_t)cx0[k] << 16) | (uint64_t)ch;
                    }
                }
                off += c[k];
            }
            // erase the parent
            const int pn = S.next[id], pv = S.prev[id];
            if (pv != kNil) S.next[pv] = (uint16_t)pn; else sh_head = pn;
            if (pn != kNil) S.prev[pn] = (uint16_t)pv; else sh_tail = pv;
            sh_size--;
            S.freelist[sh_nfree++] = (uint16_t)id;
        }
        __syncthreads();
    };

    // ---- 4. main loop (:604-755) ----------------------------------------------------------------------------
    bool finish = (sh_size == 0);
    while (!finish) {
        const int prevSize = sh_size;
        if (lane == 0) { sh_nToExpand = 0; sh_nA = 0; }
        __syncthreads();
        int lit = sh_head;
        while (lit != kNil) {
            if (S.leaf[lit]) { lit = S.next[lit]; continue; }
            const int nxt = S.next[lit];
            divide(lit, true);
            if (sh_err) break;
            lit = nxt;
        }
        if (sh_err) break;
        if (sh_size >= N || sh_size == prevSize) {
            finish = true;
        } else if (sh_size + sh_nToExpand * 3 > N) {
            while (!finish) {
                const int prevSize2 = sh_size;
                const int nB = sh_nA;
                for (int i = lane; i < nB; i += 64) S.sb[i] = S.sa[i];
                __syncthreads();
                if (lane == 0) { sh_nA = 0; oct_std_sort(S.sb, nB); }
                __syncthreads();
                for (int j = nB - 1; j >= 0; j--) {
                    const int id = (int)(S.sb[j] & 0xffff);
                    divide(id, false);
                    if (sh_err) break;
                    if (sh_size >= N) break;
                }
                if (sh_err) break;
                if (sh_size >= N || sh_size == prevSize2) finish = true;
            }
        }
        if (sh_err) break;
    }
    if (sh_err) {
        if (lane == 0) { atomicExch(err, 2); lvlcnt[f * nlevels + level] = 0; }
        return;
    }

    // ---- 5. best response per node, first wins ties (:757-776), in list order -------------------------------
    const int nn = sh_size;
    if (lane == 0) {
        int k = 0;
        for (int it = sh_head; it != kNil; it = S.next[it]) S.order[k++] = (uint16_t)it;
    }
    __syncthreads();
    uint32_t *out = lvlkp + (size_t)f * lvlkp_frame_stride + L.lvl_off;
    for (int i0 = 0; i0 < nn; i0 += 64) {
        const int i = i0 + lane;
        if (i < nn && i < L.lvl_cap) {
            const int id = S.order[i];
            const uint32_t *src = kb[S.buf[id]] + S.beg[id];
            const int cnt = S.cnt[id];
            uint32_t best = src[0];
            for (int k = 1; k < cnt; k++) {
                const uint32_t key = src[k];
                if (key_s(key) > key_s(best)) best = key;
            }
            // keypoints[i].pt += minBorder (:884-886): store level coordinates
            out[i] = pack_key(key_x(best) + kBorder, key_y(best) + kBorder, key_s(best));
        }
    }
    if (lane == 0) {
        if (nn > L.lvl_cap) atomicExch(err, 3);
        lvlcnt[f * nlevels + level] = min(nn, L.lvl_cap);
    }
#undef sh_head
#undef sh_tail
#undef sh_size
#undef sh_nfree
#undef sh_nA
#undef sh_nToExpand
#undef sh_err
}

}  // namespace orbx
