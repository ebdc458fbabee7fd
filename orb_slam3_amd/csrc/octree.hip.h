// octree.hip.h -- k_octree: exact device emulation of ORBextractor::DistributeOctTree
// (/root/reference/src/ORBextractor.cc:555-779, DivideNode :480-536, compareNodes :538-553).
//
// The reference algorithm is sequential and order sensitive: nodes live in a std::list (children are
// push_front'ed, parents erased while iterating), the size-ordered expansion uses std::sort with a comparator
// that leaves ties, and the output order is the final list order.  To return bit-identical keypoint sets in
// identical order, one wave (64 lanes) per (frame, level) runs that control flow:
//   * the list/bookkeeping state (head, tail, size, free count, ...) is wave-uniform and lives in registers; every
//     lane executes the same scalar control flow, lane 0 alone writes the node arrays in LDS;
//   * a breadth pass of the reference (":618-680": walk the list, divide every node that is not bNoMore) visits
//     exactly the children with >1 keys created by the previous pass, in REVERSE creation order (children are
//     push_front'ed), i.e. vSizeAndPointerToNode backwards -- so no list walk is needed to find them;
//   * the libstdc++ introsort replica of the size-ordered expansion runs on lane 0 over an LDS array (a VGPR-resident
//     array addressed with v_readlane was measured 2x slower: ~40 instructions per access vs one LDS round trip);
//   * all 64 lanes cooperate on the data-parallel parts: gathering the cell slots in reference order, the stable
//     4-way key partition of DivideNode (ballot + prefix popcount), and the per-node best-response pick.
// Keys (packed x|y<<12|score<<24, relative to the 16-px border) ping-pong between two LDS buffers (global scratch
// when a level has more candidates than fit).
#pragma once

#include "orbx_internal.h"

namespace orbx {

constexpr int kNil = 0xffff;
constexpr int kOctLdsKeys = 3072;   // keys per ping-pong buffer kept in LDS
constexpr int kOctStackInts = 3 * 64;

struct OctLds {  // carved from dynamic LDS, `pool` entries each
    int16_t *x0, *y0, *x1, *y1;  // UL.x, UL.y, UR.x, BR.y
    int32_t *beg, *cnt;          // key range in buffer `buf`
    uint8_t *buf;                // ping-pong id
    uint16_t *next, *prev;       // list links
    uint16_t *freelist;          // stack of free node ids
    uint64_t *sa, *sb;           // LDS fallback of vSizeAndPointerToNode / vPrevSizeAndPointerToNode
    uint16_t *order;             // final list order
    int *stack;                  // introsort recursion stack
};

__host__ __device__ inline size_t oct_pool_bytes(int pool) {
    // sort stack, then the per-node arrays (8-byte arrays first for alignment), rounded to 16
    return ((size_t)kOctStackInts * 4 + (size_t)pool * (8 + 8 + 4 + 4 + 2 * 4 + 2 * 3 + 2 + 1) + 64 + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t oct_lds_bytes(int pool) { return oct_pool_bytes(pool) + 2 * (size_t)kOctLdsKeys * 4; }

// ---- sort entries: cnt << 32 | ulx << 16 | node.  compareNodes(e1,e2) (:538-553): e1.first < e2.first, or equal and
// e1.second->UL.x < e2.second->UL.x  ==  (a >> 16) < (b >> 16) ------------------------------------------------
__device__ __forceinline__ bool oct_less(uint64_t a, uint64_t b) { return (a >> 16) < (b >> 16); }

struct LdsArr {  // array in LDS (any size)
    uint64_t *p;
    __device__ __forceinline__ uint64_t get(int i) const { return p[i]; }
    __device__ __forceinline__ void set(int i, uint64_t v) { p[i] = v; }
};

// ---- libstdc++ (GCC 11) std::sort replica: introsort (median-of-3 pivot moved to first, unguarded partition,
// threshold 16, depth limit 2*lg(n) with heap-sort fallback) + final insertion sort --------------------------
template <class A> __device__ inline void oct_unguarded_linear_insert(A &v, int last) {
    const uint64_t val = v.get(last);
    int next = last - 1;
    uint64_t nv = v.get(next);
    while (oct_less(val, nv)) {
        v.set(last, nv);
        last = next;
        --next;
        nv = v.get(next);
    }
    v.set(last, val);
}
template <class A> __device__ inline void oct_insertion_sort(A &v, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        const uint64_t val = v.get(i);
        if (oct_less(val, v.get(first))) {
            for (int k = i; k > first; --k) v.set(k, v.get(k - 1));  // move_backward
            v.set(first, val);
        } else {
            oct_unguarded_linear_insert(v, i);
        }
    }
}
template <class A> __device__ inline void oct_push_heap(A &v, int first, int hole, int top, uint64_t value) {
    int parent = (hole - 1) / 2;
    while (hole > top && oct_less(v.get(first + parent), value)) {
        v.set(first + hole, v.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v.set(first + hole, value);
}
template <class A> __device__ inline void oct_adjust_heap(A &v, int first, int hole, int len, uint64_t value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (oct_less(v.get(first + child), v.get(first + child - 1))) child--;
        v.set(first + hole, v.get(first + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v.set(first + hole, v.get(first + child - 1));
        hole = child - 1;
    }
    oct_push_heap(v, first, hole, top, value);
}
template <class A> __device__ inline void oct_heap_sort(A &v, int first, int last) {  // __partial_sort(first,last,last)
    const int len = last - first;
    if (len >= 2) {  // __make_heap
        int parent = (len - 2) / 2;
        while (true) {
            const uint64_t value = v.get(first + parent);
            oct_adjust_heap(v, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    int l = last;
    while (l - first > 1) {  // __sort_heap / __pop_heap
        --l;
        const uint64_t value = v.get(l);
        v.set(l, v.get(first));
        oct_adjust_heap(v, first, 0, l - first, value);
    }
}
// `stack` = kOctStackInts ints of (LDS) scratch for the right-hand recursion
template <class A> __device__ inline void oct_std_sort(A &v, int n, int *stack) {
    if (n <= 0) return;
    int sp = 0;
    const int lg = 31 - __clz(n);
    stack[0] = 0; stack[1] = n; stack[2] = 2 * lg; sp = 1;
    while (sp > 0) {
        sp--;
        int first = stack[3 * sp], last = stack[3 * sp + 1], depth = stack[3 * sp + 2];
        while (last - first > 16) {
            if (depth == 0) { oct_heap_sort(v, first, last); break; }
            --depth;
            // __unguarded_partition_pivot: __move_median_to_first(first, first+1, mid, last-1)
            const int mid = first + (last - first) / 2;
            {
                const int ia = first + 1, ib = mid, ic = last - 1;
                const uint64_t a = v.get(ia), b = v.get(ib), c = v.get(ic);
                int m;
                if (oct_less(a, b)) {
                    if (oct_less(b, c)) m = ib;
                    else if (oct_less(a, c)) m = ic;
                    else m = ia;
                } else if (oct_less(a, c)) m = ia;
                else if (oct_less(b, c)) m = ic;
                else m = ib;
                const uint64_t t = v.get(first), mv = v.get(m);
                v.set(first, mv); v.set(m, t);
            }
            int lo = first + 1, hi = last;
            const uint64_t pivot = v.get(first);
            while (true) {  // __unguarded_partition
                uint64_t vl = v.get(lo);
                while (oct_less(vl, pivot)) { ++lo; vl = v.get(lo); }
                --hi;
                uint64_t vh = v.get(hi);
                while (oct_less(pivot, vh)) { --hi; vh = v.get(hi); }
                if (!(lo < hi)) break;
                v.set(lo, vh); v.set(hi, vl);
                ++lo;
            }
            const int cut = lo;
            // libstdc++ recurses on [cut, last) and loops on [first, cut); the two ranges are disjoint, so the
            // processing order does not influence the result
            stack[3 * sp] = cut; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth; sp++;
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

// debug kernel: sort (count, ulx) pairs with the replica (n <= 65535)
__global__ __launch_bounds__(64) void k_debug_sort(const int32_t *count, const int32_t *ulx, int n, int32_t *perm, uint64_t *scratch) {
    __shared__ int stack[kOctStackInts];
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64)
        scratch[i] = ((uint64_t)(uint32_t)count[i] << 32) | ((uint64_t)(uint32_t)ulx[i] << 16) | (uint64_t)i;
    __syncthreads();
    LdsArr a{scratch};
    if (lane == 0) oct_std_sort(a, n, stack);
    __syncthreads();
    for (int i = lane; i < n; i += 64) perm[i] = (int32_t)(scratch[i] & 0xffff);
}

// ---- the quad-tree kernel body ------------------------------------------------------------------------------
__device__ __forceinline__ void octree_body(const LevelInfo &L, uint8_t *smem, int max_pool, const int32_t *__restrict__ ccnt,
                                            const uint32_t *__restrict__ ent, uint32_t *gk0, uint32_t *gk1,
                                            uint32_t *__restrict__ out, int32_t *__restrict__ lvlcnt_out,
                                            int32_t *__restrict__ cand_total_out, int32_t *__restrict__ err) {
    const int lane = threadIdx.x;
    const int pool = L.pool;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    OctLds S;
    {
        uint8_t *p = smem;
        S.stack = (int *)p; p += (size_t)kOctStackInts * 4;
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
    }
    const int ncell = L.nCols * L.nRows;

    // ---- 1. gather the cell slots in reference order (cell row-major); C = total -----------------------------
    int C = 0;
    for (int c0 = 0; c0 < ncell; c0 += 64) {
        int n = (c0 + lane < ncell) ? ccnt[c0 + lane] : 0;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) n += __shfl_xor(n, s);
        C += n;
    }
    C = __builtin_amdgcn_readfirstlane(C);
    uint32_t *kb[2];
    if (C <= kOctLdsKeys) {
        uint32_t *lk = reinterpret_cast<uint32_t *>(smem + oct_pool_bytes(max_pool));
        kb[0] = lk; kb[1] = lk + kOctLdsKeys;
    } else {
        kb[0] = gk0; kb[1] = gk1;
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
    if (lane == 0 && cand_total_out) *cand_total_out = C;

    // ---- wave-uniform list state in registers -------------------------------------------------------------------
    int head = kNil, tail = kNil, size = 0, nfree = pool, nA = 0, nToExpand = 0, errflag = 0;
    for (int i = lane; i < pool; i += 64) S.freelist[i] = (uint16_t)(pool - 1 - i);  // pop order 0,1,2,...
    __syncthreads();

    LdsArr la{S.sa}, lb{S.sb};  // vSizeAndPointerToNode / vPrevSizeAndPointerToNode

    // ---- 2. roots (:559-587): nIni nodes of width hX; key -> root (int)(x / hX); stable split kb[1] -> kb[0];
    // ---- 3. list init (:589-602): empty roots are dropped, single-key roots are bNoMore, the others are the
    //         first pass's work list (forward order: roots were push_back'ed)
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
            if (cnt > 0) {  // push_back
                const int id = S.freelist[nfree - 1];
                nfree--;
                const int ux0 = (int)(L.hX * (float)r);
                if (lane == 0) {
                    S.x0[id] = (int16_t)ux0;
                    S.x1[id] = (int16_t)(int)(L.hX * (float)(r + 1));
                    S.y0[id] = 0;
                    S.y1[id] = (int16_t)(L.h - 2 * kBorder);
                    S.beg[id] = beg; S.cnt[id] = cnt; S.buf[id] = 0;
                    S.next[id] = kNil; S.prev[id] = (uint16_t)tail;
                    if (tail != kNil) S.next[tail] = (uint16_t)id;
                }
                if (tail == kNil) head = id;
                tail = id;
                size++;
                if (cnt > 1) {
                    const uint64_t e = ((uint64_t)(uint32_t)cnt << 32) | ((uint64_t)(uint16_t)ux0 << 16) | (uint64_t)id;
                    if (lane == 0) la.set(nA, e);
                    nA++;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    const int N = L.quota;

    // DivideNode (:480-536) + child bookkeeping for node `id` (uniform).  Children are push_front'ed in the order
    // n1..n4; those with >1 keys are appended to vSizeAndPointerToNode.
    auto divide = [&](int id, bool count_expand) {
        const int px0 = S.x0[id], py0 = S.y0[id], px1 = S.x1[id], py1 = S.y1[id];
        const int beg = S.beg[id], cnt = S.cnt[id], sb = S.buf[id];
        // up to four fresh node ids, fetched in one LDS round trip
        const int f0 = S.freelist[max(nfree - 1, 0)], f1 = S.freelist[max(nfree - 2, 0)], f2 = S.freelist[max(nfree - 3, 0)],
                  f3 = S.freelist[max(nfree - 4, 0)];
        const int halfX = (int)ceilf((float)(px1 - px0) / 2), halfY = (int)ceilf((float)(py1 - py0) / 2);
        const int sx = px0 + halfX, sy = py0 + halfY;
        const uint32_t *src = kb[sb] + beg;
        uint32_t *dst = kb[sb ^ 1] + beg;
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (cnt <= 64) {
            uint32_t key = 0;
            int q = 4;
            if (lane < cnt) {
                key = src[lane];
                q = (key_x(key) < sx ? 0 : 1) + (key_y(key) < sy ? 0 : 2);
            }
            const unsigned long long b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
            c0 = __popcll(b0); c1 = __popcll(b1); c2 = __popcll(b2); c3 = __popcll(b3);
            if (q < 4) {
                const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
                const int off = (q > 0 ? c0 : 0) + (q > 1 ? c1 : 0) + (q > 2 ? c2 : 0);
                dst[off + __popcll(bq & lt_mask)] = key;
            }
        } else {
            for (int i0 = 0; i0 < cnt; i0 += 64) {
                const int i = i0 + lane;
                int q = 4;
                if (i < cnt) { const uint32_t key = src[i]; q = (key_x(key) < sx ? 0 : 1) + (key_y(key) < sy ? 0 : 2); }
                c0 += __popcll(__ballot(q == 0)); c1 += __popcll(__ballot(q == 1));
                c2 += __popcll(__ballot(q == 2)); c3 += __popcll(__ballot(q == 3));
            }
            int o0 = 0, o1 = c0, o2 = c0 + c1, o3 = c0 + c1 + c2;
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
        // child rectangles (:485-507); all lanes track the list state, lane 0 writes the node arrays
        const int cc[4] = {c0, c1, c2, c3};
        const int fr[4] = {f0, f1, f2, f3};
        const int cx0[4] = {px0, sx, px0, sx}, cx1[4] = {sx, px1, sx, px1};
        const int cy0[4] = {py0, py0, sy, sy}, cy1[4] = {sy, sy, py1, py1};
        int off = beg, used = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (cc[k] > 0) {
                if (nfree <= 0) { errflag = 1; }
                else {
                    const int ch = used == 0 ? fr[0] : used == 1 ? fr[1] : used == 2 ? fr[2] : fr[3];
                    used++;
                    nfree--;
                    if (lane == 0) {
                        S.x0[ch] = (int16_t)cx0[k]; S.x1[ch] = (int16_t)cx1[k];
                        S.y0[ch] = (int16_t)cy0[k]; S.y1[ch] = (int16_t)cy1[k];
                        S.beg[ch] = off; S.cnt[ch] = cc[k]; S.buf[ch] = (uint8_t)(sb ^ 1);
                        S.prev[ch] = kNil; S.next[ch] = (uint16_t)head;  // push_front
                        if (head != kNil) S.prev[head] = (uint16_t)ch;
                    }
                    if (head == kNil) tail = ch;
                    head = ch;
                    size++;
                    if (cc[k] > 1) {
                        if (count_expand) nToExpand++;
                        const uint64_t e = ((uint64_t)(uint32_t)cc[k] << 32) | ((uint64_t)(uint16_t)cx0[k] << 16) | (uint64_t)ch;
                        if (lane == 0) la.set(nA, e);
                        nA++;
                    }
                }
            }
            off += cc[k];
        }
        __builtin_amdgcn_wave_barrier();
        // erase the parent (after the pushes: prev[id] has just been rewritten if id was the head)
        const int pn = S.next[id], pv = S.prev[id];
        if (lane == 0) {
            if (pv != kNil) S.next[pv] = (uint16_t)pn;
            if (pn != kNil) S.prev[pn] = (uint16_t)pv;
            S.freelist[nfree] = (uint16_t)id;
        }
        if (pv == kNil) head = pn;
        if (pn == kNil) tail = pv;
        size--;
        nfree++;
        __builtin_amdgcn_wave_barrier();
    };

    // ---- 4. main loop (:604-755) ----------------------------------------------------------------------------
    bool finish = (size == 0);
    bool first_pass = true;
    while (!finish && !errflag) {
        const int prevSize = size;
        nToExpand = 0;
        // breadth pass: the nodes that are not bNoMore, in list order
        const int nB = nA;
        for (int i = lane; i < nB; i += 64) S.sb[i] = S.sa[i];
        __syncthreads();
        nA = 0;
        for (int jj = 0; jj < nB && !errflag; jj++) {
            const int j = first_pass ? jj : nB - 1 - jj;  // roots: creation order; later passes: reverse creation order
            const int id = (int)(lb.get(j) & 0xffff);
            divide(id, true);
        }
        first_pass = false;
        if (errflag) break;
        if (size >= N || size == prevSize) {
            finish = true;
        } else if (size + nToExpand * 3 > N) {
            while (!finish && !errflag) {
                const int prevSize2 = size;
                const int nB2 = nA;
                for (int i = lane; i < nB2; i += 64) S.sb[i] = S.sa[i];
                __syncthreads();
                nA = 0;
                if (lane == 0) oct_std_sort(lb, nB2, S.stack);
                __syncthreads();
                for (int j = nB2 - 1; j >= 0; j--) {
                    const int id = (int)(lb.get(j) & 0xffff);
                    divide(id, false);
                    if (errflag) break;
                    if (size >= N) break;
                }
                if (size >= N || size == prevSize2) finish = true;
            }
        }
    }
    if (errflag) {
        if (lane == 0) { atomicExch(err, 2); *lvlcnt_out = 0; }
        return;
    }

    // ---- 5. best response per node, first wins ties (:757-776), in list order -------------------------------
    const int nn = size;
    if (lane == 0) {
        int k = 0;
        for (int it = head; it != kNil; it = S.next[it]) S.order[k++] = (uint16_t)it;
    }
    __syncthreads();
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
        *lvlcnt_out = min(nn, L.lvl_cap);
    }
}

// grid (nlevels, B), block 64, dynamic LDS = oct_lds_bytes(max pool): sort stack + node pool + two LDS key buffers
__global__ __launch_bounds__(64) void k_octree(const LevelInfo *__restrict__ lv, const int32_t *__restrict__ cellcnt,
                                               int total_cells, const uint32_t *__restrict__ cellent,
                                               size_t ent_frame_stride, uint32_t *__restrict__ keys0,
                                               uint32_t *__restrict__ keys1, uint32_t *__restrict__ lvlkp,
                                               size_t lvlkp_frame_stride, int32_t *__restrict__ lvlcnt, int nlevels,
                                               int32_t *__restrict__ cand_total, int32_t *__restrict__ err, int max_pool) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int level = blockIdx.x, f = blockIdx.y;
    const LevelInfo L = lv[level];
    const int32_t *ccnt = cellcnt + (size_t)f * total_cells + L.cell_base;
    const uint32_t *ent = cellent + (size_t)f * ent_frame_stride + L.cand_off;
    uint32_t *gk0 = keys0 + (size_t)f * ent_frame_stride + L.cand_off;
    uint32_t *gk1 = keys1 + (size_t)f * ent_frame_stride + L.cand_off;
    uint32_t *out = lvlkp + (size_t)f * lvlkp_frame_stride + L.lvl_off;
    int32_t *cnt_out = lvlcnt + f * nlevels + level;
    int32_t *ct_out = cand_total ? cand_total + f * nlevels + level : nullptr;
    octree_body(L, smem, max_pool, ccnt, ent, gk0, gk1, out, cnt_out, ct_out, err);
}

}  // namespace orbx
