// extractor_kernels.hip.h -- hand-written HIP kernels (gfx950, wave64) of the ORB extractor hot path.
//
// Stage -> reference function (all line numbers in /root/reference/src/ORBextractor.cc):
//   k_pyr_base        copyMakeBorder of the input into level 0                 :1188-1191
//   k_pyr_resize2     cv::resize(INTER_LINEAR) level l-1 -> l + REFLECT_101    :1183-1186
//   k_fast_strip      per-cell cv::FAST(ini) + NMS, a strip of cells per workgroup  :805-842   (fast_strip.hip.h)
//   k_fast_wave_list  cells the first pass left empty: cv::FAST(ini) / fallback cv::FAST(min)   :843-870
//                     (k_fast_cells = generic workgroup-per-cell form for cells wider than 57 px)
//   k_octree_par_t (octree_par.hip.h; its first tier gathers the keys: compact_level)  DistributeOctTree / DivideNode / compareNodes   :480-779
//                     (k_octree in octree.hip.h = sequential emulation for node pools beyond the LDS budget)
//   k_finalize        level concatenation + lapping split slots                :1117-1162
//   k_blur_stream     GaussianBlur 7x7 sigma 2 (fixed point)                   :1132-1133
//   k_describe        IC_Angle + computeOrbDescriptor + keypoint record        :76-146, 1143-1162
//   k_describe_fused  the same with the GaussianBlur computed on demand around each keypoint (no k_blur_stream launch; chosen per geometry)
//
// Integer pixel / bit work: no MFMA.  Built with -ffp-contract=off; the only fused float ops are the explicit
// __fmaf_rn / __fma_rn calls (and __builtin_elementwise_fma on float pairs) that reproduce the reference binary (see describe_tail).
#pragma once

#include "orbx_internal.h"

namespace orbx {

// Workgroups are handed to the 8 XCDs round-robin by linear workgroup id, and each XCD has its own L2.  Kernels whose workgroups
// share data inside a frame (FAST cells share tile aprons, descriptor patches overlap) are launched on a grid (8, blocks per frame,
// ceil(frames / 8)): x is the fastest dimension of the linear id, so blockIdx.x IS the XCD, and frame = 8 * blockIdx.z + blockIdx.x
// keeps ALL workgroups of a frame on one XCD (frames f = 8k + x belong to XCD x).  Measured on the FAST stage: 1.25 GB -> 0.33 GB
// of HBM reads per launch (request-size counters), the aprons and partial lines now hit the XCD's L2.
// Batches of fewer than 8 frames (single-frame calls) use a grid (1, blocks per frame, frames) instead: a frame's workgroups then spread
// over all XCDs -- one frame alone would otherwise run on an eighth of the machine.  The kernels read the multiplier from gridDim.x.
__device__ __forceinline__ bool xcd_frame_map(int n_frames, int *bx, int *f) {
    *f = (int)(blockIdx.z * gridDim.x + blockIdx.x);
    *bx = (int)blockIdx.y;
    return *f < n_frames;
}
inline dim3 xcd_grid(int blocks_per_frame, int n_frames, bool local = true) {
    if (n_frames < 8 || !local) return dim3(1, (unsigned)blocks_per_frame, (unsigned)n_frames);
    return dim3(8, (unsigned)blocks_per_frame, (unsigned)((n_frames + 7) / 8));
}

// Level 0 in place: the batch's frames as the caller holds them (no padded copy in the pyramid slab).  img == nullptr: level 0 lives in the slab.
struct Level0Src {
    const uint8_t *img;
    size_t row_stride, frame_stride;
};

__device__ __forceinline__ int reflect101(int p, int len) {
    // [OCV] borderInterpolate(BORDER_REFLECT_101); |p| excursions here are < len
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - 2 - p;
    return p;
}

// ---------------------------------------------------------------------------------------------------------
// Level 0: copy the input image into the padded level-0 slab, ring = REFLECT_101 of the image.
// One thread writes 16 consecutive bytes of a padded row (aligned 16-byte store); threads are mapped flat over
// (row, chunk) so every lane is busy.
// grid xcd_grid(ceil(words_per_frame/256), B)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_base(const LevelInfo L, const uint8_t *__restrict__ img,
                                                  size_t row_stride, size_t frame_stride, uint8_t *__restrict__ pyr,
                                                  size_t pyr_frame_stride, int32_t *__restrict__ zero_word, uint32_t cpr_rcp, int n_frames) {
    int bx, f;
    if (!xcd_frame_map(n_frames, &bx, &f)) return;   // a frame's rows stay on one XCD: lines shared by neighbouring workgroups hit its L2
    if (zero_word && bx == 0 && f == 0 && threadIdx.x == 0) *zero_word = 0;  // k_fast_wave's overflow counter of this batch
    const int cpr = L.pitch >> 4;  // 16-byte chunks per padded row (the pitch is a multiple of 64)
    const int idx = bx * 256 + threadIdx.x;
    const int py = (int)__umulhi((uint32_t)idx, cpr_rcp), ci = idx - py * cpr;   // idx / cpr, cpr_rcp = ceil(2^32 / cpr)
    if (py >= L.h + 2 * kEdge) return;
    const uint8_t *src = img + (size_t)f * frame_stride;
    const int sy = reflect101(py - kEdge, L.h);
    const uint8_t *srow = src + (size_t)sy * row_stride;
    const int x0 = ci * 16 - kRoiX;  // ROI x of the first byte (kRoiX is a multiple of 16)
    uint4 out;
    auto reversed16 = [&](int first, int x_lo) {
        // out[k] = srow[first + 15 - k] for the bytes whose ROI x = x_lo + k lies inside the 19-px ring, 0 elsewhere
        // (REFLECT_101 maps a ring segment onto a reversed run of source bytes)
        uint4 v;
        __builtin_memcpy(&v, srow + first, 16);
        uint32_t o[4] = {__builtin_amdgcn_perm(0u, v.w, 0x00010203u), __builtin_amdgcn_perm(0u, v.z, 0x00010203u),
                         __builtin_amdgcn_perm(0u, v.y, 0x00010203u), __builtin_amdgcn_perm(0u, v.x, 0x00010203u)};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t m = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = x_lo + 4 * d + k;
                m |= (x >= -kEdge && x < L.w + kEdge) ? (0xffu << (8 * k)) : 0u;
            }
            o[d] &= m;
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    };
    if (x0 >= 0 && x0 + 15 < L.w) {
        __builtin_memcpy(&out, srow + x0, 16);          // interior: 16-byte copy (any source alignment)
    } else if (x0 + 15 < 0) {
        if (x0 + 15 < -kEdge) out = make_uint4(0, 0, 0, 0);
        else out = reversed16(-x0 - 15, x0);            // left ring: x -> -x
    } else if (x0 >= L.w && L.w >= 48) {
        if (x0 >= L.w + kEdge) out = make_uint4(0, 0, 0, 0);
        else out = reversed16(2 * L.w - 2 - x0 - 15, x0);  // right ring: x -> 2w-2-x
    } else {
        uint32_t o[4];  // chunk straddling the right end of the ROI
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = x0 + 4 * d + k;
                if (x >= -kEdge && x < L.w + kEdge) v |= (uint32_t)srow[reflect101(x, L.w)] << (8 * k);
            }
            o[d] = v;
        }
        out = make_uint4(o[0], o[1], o[2], o[3]);
    }
    uint8_t *drow = pyr + (size_t)f * pyr_frame_stride + L.off + (size_t)py * L.pitch;
    *reinterpret_cast<uint4 *>(drow + ci * 16) = out;
}

// ---------------------------------------------------------------------------------------------------------
// Level l from level l-1: [OCV] resize INTER_LINEAR 8U (Q11 taps, (b*(H>>4))>>16 vertical form) evaluated at
// the REFLECT_101-mapped coordinate, so ROI and ring are written in one pass.  One thread = 4 consecutive bytes of
// a padded row (aligned dword store), threads mapped flat over (row, dword).  (An LDS-staged tile variant was
// measured 1.6x slower: the footprint set-up adds a third dependent memory round trip per small workgroup.)
// grid xcd_grid(ceil(words_per_frame/256), B)
// ---------------------------------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 as_pk(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

constexpr int kResizeRows = 2;  // output rows per thread (measured: 1 -> 0.032, 2 -> 0.024, 4 -> 0.033 ms per level launch)

// The vertical pass runs without the two arithmetic shifts and the clamp per pixel.  With Q11 taps b0 + b1 = 2048 and
// horizontal sums h <= 255 * 2048, both products b * (h >> 4) are non-negative and below 2^27, so "(p0 >> 16) + (p1 >> 16) + 2" is
// ONE v_add_u32_sdwa of the two high words once the rounding constant rides in p0 (p0 = b0 * (h0 >> 4) + (2 << 16), a v_mad_u32_u24),
// the result is at most 1022 (no clamp to 255 after the >> 2), and two pixels shift + pack in one v_ashr_pk_u8_i32 (gfx950).
// k_pyr_resize2: one thread = TWO adjacent dword columns (8 pixels) x 2 rows.  The resize chain is bound by the
// waves' lifetime (table loads -> source rows -> store: two dependent memory round trips, 8 waves per SIMD at most), not by issue or
// bandwidth (2 TB/s), so a wave that keeps twice the bytes in flight for the same two round trips doubles the rate until VALU issue
// binds; the two columns share their row taps, source rows (adjacent 8-byte loads) and one 8-byte store per row.
// grid xcd_grid(ceil((pitch / 8) * ceil(rows / 2) / 256), B)
// one work item of the two-column form: row pair pg, column pair wg of level L of the frame whose pyramid slab starts at `frame`
__device__ __forceinline__ void resize2_item(const LevelInfo &L, const LevelInfo &P, const ResizeTap *__restrict__ xtab,
                                             const ResizeTap *__restrict__ ytab, const ResizeGroup *__restrict__ xg, uint8_t *frame,
                                             const int pg, const int wg) {
    const int rows = L.h + 2 * kEdge;
    const int py0 = pg * kResizeRows;
    if (py0 >= rows) return;
    const uint8_t *proi = frame + P.off + (size_t)kEdge * P.pitch + kRoiX;  // previous level ROI origin
    const uint4 *gp = reinterpret_cast<const uint4 *>(&xg[L.xg_off + 2 * wg]);   // two ResizeGroup entries = 64 contiguous bytes
    uint4 gh[2], gc[2];
    gh[0] = gp[0]; gc[0] = gp[1]; gh[1] = gp[2]; gc[1] = gp[3];
    ResizeTap ty[kResizeRows];
#pragma unroll
    for (int r = 0; r < kResizeRows; r++) ty[r] = ytab[L.ytab_off + reflect101(min(py0 + r, rows - 1) - kEdge, L.h)];
    uint32_t so0[kResizeRows], so1[kResizeRows];   // byte offsets of the two source rows of each output row
#pragma unroll
    for (int r = 0; r < kResizeRows; r++) {
        const int sy0 = min(max(ty[r].ofs, 0), P.h - 1), sy1 = min(max(ty[r].ofs + 1, 0), P.h - 1);
        so0[r] = __umul24((uint32_t)sy0, (uint32_t)P.pitch);
        so1[r] = __umul24((uint32_t)sy1, (uint32_t)P.pitch);
    }
    // every source load of the thread before any arithmetic: one memory round trip
    uint2 r0[2][kResizeRows], r1[2][kResizeRows];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int r = 0; r < kResizeRows; r++) {
            // unconditional (a column without taps reads the row's first bytes and ignores them): loads inside divergent blocks make the
            // compiler wait for ALL outstanding loads at every block boundary -- four round trips instead of one
            const uint32_t gx = gh[c].z == 1 ? gh[c].x : 0u;
            __builtin_memcpy(&r0[c][r], proi + (so0[r] + gx), 8);
            __builtin_memcpy(&r1[c][r], proi + (so1[r] + gx), 8);
        }
    uint32_t out[2][kResizeRows];
    constexpr uint32_t kPair[4] = {0x0c040c00u, 0x0c050c01u, 0x0c060c02u, 0x0c070c03u};  // (left tap k, right tap k) as two u16
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const uint32_t selr = gh[c].y + 0x01010101u;
        const uint32_t cc[4] = {gc[c].x, gc[c].y, gc[c].z, gc[c].w};
#pragma unroll
        for (int r = 0; r < kResizeRows; r++) {
            uint32_t o = 0;   // gh.z == 2: pitch padding outside the ring, zeros
            if (gh[c].z == 1) {
                const uint32_t l0 = __builtin_amdgcn_perm(r0[c][r].y, r0[c][r].x, gh[c].y), q0 = __builtin_amdgcn_perm(r0[c][r].y, r0[c][r].x, selr);
                const uint32_t l1 = __builtin_amdgcn_perm(r1[c][r].y, r1[c][r].x, gh[c].y), q1 = __builtin_amdgcn_perm(r1[c][r].y, r1[c][r].x, selr);
                const int b0 = ty[r].c0, b1 = ty[r].c1;   // bilinear tap pair: 0 <= b0, b1 <= 2048
                int t[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t h0 = __builtin_amdgcn_udot2(as_pk(__builtin_amdgcn_perm(q0, l0, kPair[k])), as_pk(cc[k]), 0u, false);
                    const uint32_t h1 = __builtin_amdgcn_udot2(as_pk(__builtin_amdgcn_perm(q1, l1, kPair[k])), as_pk(cc[k]), 0u, false);
                    const uint32_t p0 = __umul24(h0 >> 4, (uint32_t)b0) + 0x20000u, p1 = __umul24(h1 >> 4, (uint32_t)b1);
                    t[k] = (int)((p0 >> 16) + (p1 >> 16));
                }
                const uint32_t lo = (uint16_t)__builtin_amdgcn_ashr_pk_u8_i32(t[0], t[1], 2);
                const uint32_t hi = (uint16_t)__builtin_amdgcn_ashr_pk_u8_i32(t[2], t[3], 2);
                o = lo | (hi << 16);
            } else if (gh[c].z == 0) {   // taps further apart than 8 bytes (scale factor > 2): table form, as in k_pyr_resize
                const uint8_t *S0 = proi + so0[r], *S1 = proi + so1[r];
                const int b0 = ty[r].c0, b1 = ty[r].c1;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int x = (2 * wg + c) * 4 + k - kRoiX;
                    if (x >= -kEdge && x < L.w + kEdge) {
                        const ResizeTap tx = xtab[L.xtab_off + reflect101(x, L.w)];
                        const int h0 = S0[tx.ofs] * tx.c0 + S0[tx.ofs + 1] * tx.c1;
                        const int h1 = S1[tx.ofs] * tx.c0 + S1[tx.ofs + 1] * tx.c1;
                        int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                        v = min(max(v, 0), 255);
                        o |= (uint32_t)v << (8 * k);
                    }
                }
            }
            out[c][r] = o;
        }
    }
    uint8_t *drow = frame + L.off + (size_t)py0 * L.pitch + wg * 8;
#pragma unroll
    for (int r = 0; r < kResizeRows; r++)
        if (py0 + r < rows) *reinterpret_cast<uint2 *>(drow + (uint32_t)(r * L.pitch)) = make_uint2(out[0][r], out[1][r]);
}

__global__ __launch_bounds__(256) void k_pyr_resize2(const LevelInfo L, const LevelInfo P, const ResizeTap *__restrict__ xtab,
                                                     const ResizeTap *__restrict__ ytab, const ResizeGroup *__restrict__ xg,
                                                     uint8_t *__restrict__ pyr, size_t pyr_frame_stride, uint32_t wpc_rcp, int n_frames) {
    int bx, f;
    if (!xcd_frame_map(n_frames, &bx, &f)) return;
    const int wpc = L.pitch >> 3;   // column pairs per padded row (the pitch is a multiple of 64)
    const int idx = bx * 256 + threadIdx.x;
    const int pg = (int)__umulhi((uint32_t)idx, wpc_rcp), wg = idx - (int)__umul24((uint32_t)pg, (uint32_t)wpc);
    resize2_item(L, P, xtab, ytab, xg, pyr + (size_t)f * pyr_frame_stride, pg, wg);
}

// ---------------------------------------------------------------------------------------------------------
// k_pyr_resize_march (round 3): the same bilinear arithmetic, register-marching.  k_pyr_resize2 spends a table round trip (64 B of
// ResizeGroup + row taps per 16 output bytes), then a source round trip, then the store, per thread, and moves 2.2 TB/s whatever is
// done to its instruction count.  Here a lane owns one dword column (4 output pixels) of a block of RB output rows
// and walks down it: the column's ResizeGroup is loaded ONCE, the row taps are wave-uniform scalar loads, the source rows of the block
// are streamed in order (every source row is read once per block and its horizontal pass computed once -- an output row pair shares
// it: 1.2 instead of 2 source rows per output row at scale 1.2), CH rows in flight while the previous CH are filtered, and a wave
// stores 256 contiguous bytes per row.  The 19 ring rows above / below the ROI are REFLECT_101 copies of ROI rows 1..19 / h-2..h-20
// (copyMakeBorder of the level itself, ORBextractor.cc:1185-1186): the wave that produces such a row stores it twice.
// Needs every tap pair of a dword column within 8 source bytes (scale factor <= 2); k_pyr_resize2 stays for the other case.
// grid xcd_grid(ceil(nstrips * ceil(h / RB) / 4), B), block 256 (four independent waves)
// ---------------------------------------------------------------------------------------------------------
// one block of the marching resize: `item` = (row block, column strip) of level L of the frame whose pyramid slab starts at `frame`; one wave
template <int CH>
__device__ __forceinline__ void resize_march_block(const LevelInfo &L, const LevelInfo &P, const ResizeTap *__restrict__ ytab,
                                                   const ResizeGroup *__restrict__ xg, uint8_t *frame, int rb_rows, int nstrips,
                                                   uint32_t nstrips_rcp, int item, int lane) {
    const int rb = nstrips == 1 ? item : __builtin_amdgcn_readfirstlane((int)__umulhi((uint32_t)item, nstrips_rcp)), cs = item - rb * nstrips;   // ceil(2^32 / 1) does not fit
    const int ncol = L.pitch >> 2;
    const int col = cs * 64 + lane;
    const bool live = col < ncol;
    const uint4 *gp = reinterpret_cast<const uint4 *>(&xg[L.xg_off + min(col, ncol - 1)]);
    const uint4 gh = gp[0], gc = gp[1];   // base, sel, valid, pad | cc[4]
    const uint8_t *proi = frame + P.off + (size_t)kEdge * P.pitch + kRoiX + (gh.z == 1 ? gh.x : 0u);   // this lane's first source byte of row 0
    uint8_t *dcol = frame + L.off + 4 * (uint32_t)col;
    const uint32_t sel = gh.y, selr = gh.y + 0x01010101u;
    const uint32_t cc[4] = {gc.x, gc.y, gc.z, gc.w};
    constexpr uint32_t kPair[4] = {0x0c040c00u, 0x0c050c01u, 0x0c060c02u, 0x0c070c03u};  // (left tap k, right tap k) as two u16
    const int y0 = rb * rb_rows, y1 = min(y0 + rb_rows, L.h);
    // the row taps of the whole block in ONE vector load (lane i holds row y0 + i; rb_rows <= 64), read back with v_readlane: a scalar load
    // per output row sat in the middle of the emit loop's dependence chain (16 - 32 exposed round trips per block)
    const uint2 ytl = *reinterpret_cast<const uint2 *>(ytab + L.ytab_off + min(y0 + lane, L.h - 1));
#define RM_TAP_OFS(i) __builtin_amdgcn_readlane((int)ytl.x, (i))
#define RM_TAP_CC(i) ((uint32_t)__builtin_amdgcn_readlane((int)ytl.y, (i)))
    int r = y0;
    int ty_ofs = RM_TAP_OFS(0);
    uint32_t ty_cc = RM_TAP_CC(0);
    int s = ty_ofs;                                   // next source row to fetch
    const int s_last = RM_TAP_OFS(y1 - 1 - y0) + 1;   // last source row the block needs
    const int hmax = P.h - 1;
    uint2 cur[CH], nxt[CH];
    uint32_t Hp[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < CH; k++) __builtin_memcpy(&cur[k], proi + (size_t)(uint32_t)(min(s + k, hmax) * P.pitch), 8);
    while (r < y1) {
        if (s + CH <= s_last) {   // wave-uniform: the next CH rows are requested before these are filtered
#pragma unroll
            for (int k = 0; k < CH; k++) __builtin_memcpy(&nxt[k], proi + (size_t)(uint32_t)(min(s + CH + k, hmax) * P.pitch), 8);
        }
        uint32_t H[CH][4];   // horizontal pass of the chunk's rows, already >> 4 ([OCV] the vertical pass multiplies (sum >> 4))
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const uint32_t l = __builtin_amdgcn_perm(cur[k].y, cur[k].x, sel), q = __builtin_amdgcn_perm(cur[k].y, cur[k].x, selr);
#pragma unroll
            for (int j = 0; j < 4; j++) H[k][j] = __builtin_amdgcn_udot2(as_pk(__builtin_amdgcn_perm(q, l, kPair[j])), as_pk(cc[j]), 0u, false) >> 4;
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            // the output row whose source rows are (s + k - 1, s + k), if there is one (at most one: the scale factor is >= 1)
            if (r < y1 && ty_ofs + 1 == s + k) {   // wave-uniform
                const uint32_t *A = k == 0 ? Hp : H[k == 0 ? 0 : k - 1], *B = H[k];
                const uint32_t b0 = ty_cc & 0xffffu, b1 = ty_cc >> 16;   // c0 | c1 << 16: bilinear tap pair, 0 <= b <= 2048
                int t[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t p0 = __umul24(A[j], b0) + 0x20000u, p1 = __umul24(B[j], b1);
                    t[j] = (int)((p0 >> 16) + (p1 >> 16));
                }
                uint32_t o = ((uint32_t)(uint16_t)__builtin_amdgcn_ashr_pk_u8_i32(t[0], t[1], 2)) | ((uint32_t)(uint16_t)__builtin_amdgcn_ashr_pk_u8_i32(t[2], t[3], 2) << 16);
                if (gh.z != 1) o = 0u;   // pitch padding outside the ring
                if (live) {
                    *reinterpret_cast<uint32_t *>(dcol + (size_t)(uint32_t)((kEdge + r) * L.pitch)) = o;
                    if (r >= 1 && r <= kEdge) *reinterpret_cast<uint32_t *>(dcol + (size_t)(uint32_t)((kEdge - r) * L.pitch)) = o;                              // ring above
                    if (r <= L.h - 2 && r >= L.h - 1 - kEdge) *reinterpret_cast<uint32_t *>(dcol + (size_t)(uint32_t)((kEdge + 2 * (L.h - 1) - r) * L.pitch)) = o;   // ring below
                }
                r++;
                ty_ofs = RM_TAP_OFS(min(r, y1 - 1) - y0);
                ty_cc = RM_TAP_CC(min(r, y1 - 1) - y0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) Hp[j] = H[CH - 1][j];
#pragma unroll
        for (int k = 0; k < CH; k++) cur[k] = nxt[k];
        s += CH;
        if (s > s_last + CH) break;   // cannot happen with monotone row taps; never spin on a bad table
    }
#undef RM_TAP_OFS
#undef RM_TAP_CC
}

template <int CH>
__global__ __launch_bounds__(256) void k_pyr_resize_march(const LevelInfo L, const LevelInfo P, const ResizeTap *__restrict__ ytab,
                                                          const ResizeGroup *__restrict__ xg, uint8_t *__restrict__ pyr, size_t pyr_frame_stride,
                                                          int rb_rows, int nstrips, uint32_t nstrips_rcp, int n_items, int n_frames) {
    int bx, f;
    if (!xcd_frame_map(n_frames, &bx, &f)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = bx * 4 + wave;
    if (item >= n_items) return;
    resize_march_block<CH>(L, P, ytab, xg, pyr + (size_t)f * pyr_frame_stride, rb_rows, nstrips, nstrips_rcp, item, lane);
}

// ---------------------------------------------------------------------------------------------------------
// FAST-9/16 corner score of one pixel.  [OCV] cornerScore<16>: with d[k] = v - p[k] on the Bresenham circle,
//   score = max( max_arcs min(d over 9 contiguous), max_arcs min(-d over 9 contiguous) ) - 1
// and "p is a corner at threshold t"  <=>  score >= t  (SURVEY.md 8c-R2).
// ---------------------------------------------------------------------------------------------------------
// three-input min / max as ONE instruction.  Written as min(a, min(b, c)) the compiler reassociates the overlapping windows of
// fast_score16 to share pair minima and ends up with two-input ops only (110 instead of 80 per score).
__device__ __forceinline__ int min3i(int a, int b, int c) {
    int r;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int max3i(int a, int b, int c) {
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ int fast_score16(const uint8_t *__restrict__ c, int pp) {
    // [OCV] cornerScore<16>: with d[k] = v - p[k] the score is max(max_arcs min(d), -min_arcs max(d)) - 1 over the 16 arcs of 9.
    // min over an arc of (v - p) = v - max over the arc of p, so the arcs are evaluated on the raw circle pixels and v enters
    // twice at the end (sixteen subtractions fewer):   score = max(v - min_arcs max9(p), max_arcs min9(p) - v) - 1
    const int v = c[0];
    int p[16];
    p[0] = c[3 * pp];
    p[1] = c[3 * pp + 1];
    p[2] = c[2 * pp + 2];
    p[3] = c[pp + 3];
    p[4] = c[3];
    p[5] = c[-pp + 3];
    p[6] = c[-2 * pp + 2];
    p[7] = c[-3 * pp + 1];
    p[8] = c[-3 * pp];
    p[9] = c[-3 * pp - 1];
    p[10] = c[-2 * pp - 2];
    p[11] = c[-pp - 3];
    p[12] = c[-3];
    p[13] = c[pp - 3];
    p[14] = c[2 * pp - 2];
    p[15] = c[3 * pp - 1];
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = min3i(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
        hi3[k] = max3i(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
    }
    int maxmin = 0, minmax = 255;  // max over the arcs of min9(p), min over the arcs of max9(p)
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const int a0 = min3i(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);          // min over p[k..k+8]
        const int a1 = min3i(lo3[k + 1], lo3[(k + 4) & 15], lo3[(k + 7) & 15]);
        const int b0 = max3i(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);          // max over p[k..k+8]
        const int b1 = max3i(hi3[k + 1], hi3[(k + 4) & 15], hi3[(k + 7) & 15]);
        maxmin = max3i(maxmin, a0, a1);
        minmax = min3i(minmax, b0, b1);
    }
    return max(v - minmax, maxmin - v) - 1;
}

// cornerScore of TWO pixels per lane: the 16 circle pixels of both as packed u16 pairs (low half = pixel A, high half = pixel B), the arc
// minima / maxima by v_pk_minimum3_f16 / v_pk_maximum3_f16 (gfx950) -- three-input, two pixels per instruction, 80 instructions for both
// pixels where fast_score16 needs 80 for one.  The operands are INTEGERS 0 .. 255 in 16-bit lanes: as f16 bit patterns they are
// non-negative (sub)normal numbers, whose order is the order of their bit patterns, and minimum / maximum return one of their inputs
// unchanged (the kernels run with f16 denormals preserved, the AMDGPU default) -- so the result is the integer min / max.
__device__ __forceinline__ uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void fast_score16_x2(const uint8_t *__restrict__ ca, const uint8_t *__restrict__ cb, int pp, int *sa, int *sb) {
    uint32_t p[16];
#define FS_P2(k, off) p[k] = (uint32_t)ca[off] | ((uint32_t)cb[off] << 16)
    FS_P2(0, 3 * pp); FS_P2(1, 3 * pp + 1); FS_P2(2, 2 * pp + 2); FS_P2(3, pp + 3); FS_P2(4, 3); FS_P2(5, -pp + 3); FS_P2(6, -2 * pp + 2);
    FS_P2(7, -3 * pp + 1); FS_P2(8, -3 * pp); FS_P2(9, -3 * pp - 1); FS_P2(10, -2 * pp - 2); FS_P2(11, -pp - 3); FS_P2(12, -3);
    FS_P2(13, pp - 3); FS_P2(14, 2 * pp - 2); FS_P2(15, 3 * pp - 1);
#undef FS_P2
    uint32_t lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = pk_min3(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
        hi3[k] = pk_max3(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
    }
    uint32_t maxmin = 0u, minmax = 0x00ff00ffu;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const uint32_t a0 = pk_min3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
        const uint32_t a1 = pk_min3(lo3[k + 1], lo3[(k + 4) & 15], lo3[(k + 7) & 15]);
        const uint32_t b0 = pk_max3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
        const uint32_t b1 = pk_max3(hi3[k + 1], hi3[(k + 4) & 15], hi3[(k + 7) & 15]);
        maxmin = pk_max3(maxmin, a0, a1);
        minmax = pk_min3(minmax, b0, b1);
    }
    const int va = ca[0], vb = cb[0];
    const int mma = (int)(maxmin & 0xffffu), mmb = (int)(maxmin >> 16), mna = (int)(minmax & 0xffffu), mnb = (int)(minmax >> 16);
    *sa = max(va - mna, mma - va) - 1;
    *sb = max(vb - mnb, mmb - vb) - 1;
}

// ---------------------------------------------------------------------------------------------------------
// One workgroup per FAST cell (ORBextractor.cc:805-870).  The cell sub-image [iniX, maxX) x [iniY, maxY) is
// staged in LDS; scores are computed for its interior (3 px inside), NMS sees only same-cell neighbours (the
// reference runs cv::FAST on the sub-image, so outside-interior neighbours score 0), then the cell emits
//   S20 = {survivors with score >= iniTh}  if non-empty, else  S7 = {survivors with score >= minTh}
// in row-major order into its slot of the frame's candidate slab.
//
// Phases: (0) dword-coalesced tile load; (1) cheap NECESSARY test on the 4 antipodal circle pairs at minTh (a
// 9-arc contains one pixel of every antipodal pair) -> passing pixels are queued in LDS with a wave-aggregated
// atomic; (2) exact score only for queued pixels, all lanes busy; (3) NMS; (4) ordered ballot compaction.
// grid (total_cells, B), block 256, dynamic LDS: pixel tile + score tile + queue
// ---------------------------------------------------------------------------------------------------------
template <int TPB>
__device__ __forceinline__ void fast_cell_body(const TileRef t, const int f, uint8_t *smem, const LevelInfo *__restrict__ lv,
                                               const uint8_t *__restrict__ pyr, size_t pyr_frame_stride, int32_t *__restrict__ cellcnt,
                                               int total_cells, uint32_t *__restrict__ cellent, size_t ent_frame_stride, int iniTh, int minTh) {
    const LevelInfo L = lv[t.level];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int cell = t.ti * L.nCols + t.tj;
    int32_t *cnt_out = cellcnt + (size_t)f * total_cells + L.cell_base + cell;

    const int maxBX = L.w - kBorder, maxBY = L.h - kBorder;
    const int iniX = kBorder + t.tj * L.wCell, iniY = kBorder + t.ti * L.hCell;
    const int maxX = min(iniX + L.wCell + 6, maxBX), maxY = min(iniY + L.hCell + 6, maxBY);
    const int cols = maxX - iniX, rows = maxY - iniY;
    const int iw = cols - 6, ih = rows - 6;
    // :813 / :821 skip rules, plus sub-images too small to have an interior
    if (iniY >= maxBY - 3 || iniX >= maxBX - 6 || iw <= 0 || ih <= 0) {
        if (tid == 0) *cnt_out = 0;
        return;
    }
    const int ax = iniX & 3;                   // the tile is loaded from the enclosing aligned dwords
    const int pp = (cols + ax + 3) & ~3;       // LDS pitch of the pixel tile (bytes, multiple of 4)
    const int sp = iw + 2;                     // pitch of the score tile (1-px zero apron)
    const int n = iw * ih;
    // carve (all offsets multiples of 16)
    int *ctrl = reinterpret_cast<int *>(smem);                 // [4] queue length, [5] survivor count
    uint8_t *pix = smem + 32;                                  // rows * pp   (later reused as the survivor map)
    uint8_t *sco = pix + ((rows * pp + 15) & ~15);             // (ih+2) * sp
    uint16_t *queue = reinterpret_cast<uint16_t *>(sco + (((ih + 2) * sp + 15) & ~15));  // n entries
    const uint32_t rcp = ((1u << 20) + (uint32_t)iw - 1u) / (uint32_t)iw;  // exact i / iw for i < 2^20 / iw

    {   // phase 0: dword loads (the ROI row starts 64-B aligned; iniX - ax is 4-B aligned)
        const uint8_t *src = pyr + (size_t)f * pyr_frame_stride + L.off + (size_t)(kEdge + iniY) * L.pitch + kRoiX + (iniX - ax);
        const int nd = pp >> 2;
        const uint32_t rcpd = ((1u << 20) + (uint32_t)nd - 1u) / (uint32_t)nd;
        for (int i = tid; i < rows * nd; i += TPB) {
            const int r = (int)(((uint32_t)i * rcpd) >> 20), c = i - r * nd;
            reinterpret_cast<uint32_t *>(pix)[i] = *reinterpret_cast<const uint32_t *>(src + (size_t)r * L.pitch + 4 * c);
        }
        for (int i = tid; i < (((ih + 2) * sp + 3) >> 2); i += TPB) reinterpret_cast<uint32_t *>(sco)[i] = 0;
        if (tid == 0) ctrl[4] = 0;
    }
    __syncthreads();

    // phase 1: necessary condition at minTh on the antipodal pairs (0,8) (4,12) (2,10) (6,14); branch-free
    for (int i0 = 0; i0 < n; i0 += TPB) {
        const int i = i0 + tid;
        int pass = 0;
        if (i < n) {
            const int y = (int)(((uint32_t)i * rcp) >> 20), x = i - y * iw;
            const uint8_t *c = pix + (y + 3) * pp + x + 3 + ax;
            const int v = c[0], hi = v + minTh, lo = v - minTh;
            const int p0 = c[3 * pp], p8 = c[-3 * pp], p4 = c[3], p12 = c[-3];
            const int p2 = c[2 * pp + 2], p10 = c[-2 * pp - 2], p6 = c[-2 * pp + 2], p14 = c[2 * pp - 2];
            const int br = ((p0 > hi) | (p8 > hi)) & ((p4 > hi) | (p12 > hi)) & ((p2 > hi) | (p10 > hi)) & ((p6 > hi) | (p14 > hi));
            const int dk = ((p0 < lo) | (p8 < lo)) & ((p4 < lo) | (p12 < lo)) & ((p2 < lo) | (p10 < lo)) & ((p6 < lo) | (p14 < lo));
            pass = br | dk;
        }
        const unsigned long long b = __ballot(pass != 0);
        if (b) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&ctrl[4], __popcll(b));
            base = __shfl(base, 0);
            if (pass) queue[base + __popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)i;
        }
    }
    __syncthreads();

    // phase 2: exact score of the queued pixels
    const int qn = ctrl[4];
    for (int e = tid; e < qn; e += TPB) {
        const int i = queue[e];
        const int y = (int)(((uint32_t)i * rcp) >> 20), x = i - y * iw;
        int s = fast_score16(pix + (y + 3) * pp + x + 3 + ax, pp);
        s = (s >= minTh) ? s : 0;
        sco[(y + 1) * sp + x + 1] = (uint8_t)s;
    }
    if (tid == 0) ctrl[5] = 0;
    __syncthreads();

    // phase 3: NMS over the queued pixels only, strict '>' against all 8 neighbours ([OCV] FAST_t nonmax stage);
    // survivors (few per cell) go to an unordered LDS list as (linear index << 8 | score)
    uint32_t *surv = reinterpret_cast<uint32_t *>(pix);  // the pixel tile is dead after phase 2
    int any_ini = 0;
    for (int e0 = 0; e0 < qn; e0 += TPB) {
        const int e = e0 + tid;
        int keep = 0, s = 0, i = 0;
        if (e < qn) {
            i = queue[e];
            const int y = (int)(((uint32_t)i * rcp) >> 20), x = i - y * iw;
            const uint8_t *p = sco + (y + 1) * sp + x + 1;
            s = p[0];
            keep = (s > 0) & (s > p[-1]) & (s > p[1]) & (s > p[-sp - 1]) & (s > p[-sp]) & (s > p[-sp + 1]) &
                   (s > p[sp - 1]) & (s > p[sp]) & (s > p[sp + 1]);
        }
        any_ini |= (keep & (s >= iniTh));
        const unsigned long long b = __ballot(keep != 0);
        __syncthreads();  // (first iteration) every wave is done reading the pixel tile before surv overwrites it
        if (b) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&ctrl[5], __popcll(b));
            base = __shfl(base, 0);
            if (keep) surv[base + __popcll(b & ((1ull << lane) - 1ull))] = ((uint32_t)i << 8) | (uint32_t)s;
        }
    }
    any_ini = __syncthreads_or(any_ini);
    const int thr = any_ini ? iniTh : minTh;
    const int ns = min(ctrl[5], L.cell_cap);

    // phase 4: ordered (row-major) emission: rank of a selected survivor = number of selected survivors before it
    uint32_t *slot = cellent + (size_t)f * ent_frame_stride + L.cand_off + (size_t)cell * L.cell_cap;
    int total = 0;
    for (int e0 = 0; e0 < ns; e0 += TPB) {
        const int e = e0 + tid;
        if (e < ns) {
            const uint32_t me = surv[e];
            const int s = (int)(me & 0xff);
            if (s >= thr) {
                int rank = 0;
                for (int k = 0; k < ns; k++) {
                    const uint32_t o = surv[k];
                    rank += ((int)(o & 0xff) >= thr) & (o < me);  // linear index in the high bits orders the keys
                }
                const int i = (int)(me >> 8);
                const int y = (int)(((uint32_t)i * rcp) >> 20), x = i - y * iw;
                slot[rank] = pack_key(x + 3 + t.tj * L.wCell, y + 3 + t.ti * L.hCell, s);
            }
        }
    }
    if (tid < 64) {  // first wave counts the selected survivors
        for (int k = tid; k < ns; k += 64) total += ((int)(surv[k] & 0xff) >= thr);
        for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o);
        if (tid == 0) *cnt_out = total;
    }
}

template <int TPB>
__global__ __launch_bounds__(TPB) void k_fast_cells(const LevelInfo *__restrict__ lv, const TileRef *__restrict__ tiles,
                                                    const uint8_t *__restrict__ pyr, size_t pyr_frame_stride,
                                                    int32_t *__restrict__ cellcnt, int total_cells,
                                                    uint32_t *__restrict__ cellent, size_t ent_frame_stride, int iniTh,
                                                    int minTh) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    fast_cell_body<TPB>(tiles[blockIdx.x], blockIdx.y, smem, lv, pyr, pyr_frame_stride, cellcnt, total_cells, cellent, ent_frame_stride, iniTh, minTh);
}


// ---------------------------------------------------------------------------------------------------------
// k_fast_wave: the same per-cell FAST + NMS as k_fast_cells with ONE WAVE per cell (no barriers, no atomics) and the
// necessary-condition test evaluated for four horizontally adjacent pixels per lane with packed 16-bit min/max
// (v_pk_max_u16 / v_pk_min_u16): the kernel is VALU-issue bound, so instructions per pixel is what counts.
// Used when every level's cell sub-image fits a 64-byte LDS pitch (wCell + 7 <= 64); k_fast_cells otherwise.
//   LDS: pixel tile rows x 64 (sub-image column s at byte s + 1, interior pixel x at byte x + 4, so 4-pixel groups are
//   dword aligned), score tile (ih + 2) x 64 with a zero apron, queue of (y << 8 | x) u16 in row-major order.
// grid (total_cells, B), block 64
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u16x2 pk_even(uint32_t v) { return as_pk(__builtin_amdgcn_perm(0u, v, 0x0c020c00u)); }  // bytes 0, 2
__device__ __forceinline__ u16x2 pk_odd(uint32_t v) { return as_pk(__builtin_amdgcn_perm(0u, v, 0x0c030c01u)); }   // bytes 1, 3
__device__ __forceinline__ u16x2 pk_max(u16x2 a, u16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ u16x2 pk_min(u16x2 a, u16x2 b) { return __builtin_elementwise_min(a, b); }

// nonzero 16-bit half <=> that pixel passes the antipodal-pair test at threshold t (both polarities)
__device__ __forceinline__ uint32_t quick_pairs(u16x2 c, u16x2 p0, u16x2 p8, u16x2 p4, u16x2 p12, u16x2 p2, u16x2 p10, u16x2 p6,
                                                u16x2 p14, u16x2 t2) {
    // the minimum / maximum over the four pairs as ONE three-input op + one two-input op (v_pk_minimum3_f16 / v_pk_maximum3_f16 on integers 0 .. 255
    // in 16-bit lanes, as in fast_score16_x2) instead of three two-input ops
    const u16x2 mb = as_pk(pk_min3(as_u32(pk_max(p0, p8)), as_u32(pk_max(p4, p12)), as_u32(pk_min(pk_max(p2, p10), pk_max(p6, p14)))));
    const u16x2 md = as_pk(pk_max3(as_u32(pk_min(p0, p8)), as_u32(pk_min(p4, p12)), as_u32(pk_max(pk_min(p2, p10), pk_min(p6, p14)))));
    const u16x2 hi = c + t2;
    const u16x2 lo = __builtin_elementwise_sub_sat(c, t2);
    return as_u32(__builtin_elementwise_sub_sat(mb, hi)) | as_u32(__builtin_elementwise_sub_sat(lo, md));
}

template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_add(int acc, int v) {
    return acc + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, false);
}
// inclusive prefix sum over the 64 lanes of a wave (row_shr 1/2/4/8, then row_bcast15 / row_bcast31)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v = dpp_add<0x111, 0xf>(v, v);
    v = dpp_add<0x112, 0xf>(v, v);
    v = dpp_add<0x114, 0xf>(v, v);
    v = dpp_add<0x118, 0xf>(v, v);
    v = dpp_add<0x142, 0xa>(v, v);
    v = dpp_add<0x143, 0xc>(v, v);
    return v;
}

// phase 0 of the one-wave-per-cell FAST kernels: the cell's sub-image (+1 byte left, so that interior groups are dword aligned in LDS)
// into the LDS tile.  16 lanes per row, 4 rows per step; the global loads of up to 12 steps (48 rows) are issued back to back and
// only then stored -- ONE memory round trip per wave instead of one per step (the wave's lifetime, not VALU issue, bounds these
// kernels once the per-pixel work is cheap: 175 waves per SIMD take turns, 8 at a time).
template <int P>
__device__ __forceinline__ void fast_tile_load(const uint8_t *__restrict__ src, int pitch, int rows, int cols, uint8_t *pix, int lane) {
    const int c = lane & 15, nd = (cols + 4) >> 2, r0 = lane >> 4;
    if (c >= nd) return;
    for (int base = 0; base < rows; base += 48) {
        uint32_t v[12];
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int r = base + r0 + 4 * k;
            v[k] = 0;
            if (r < rows) __builtin_memcpy(&v[k], src + (size_t)r * pitch + 4 * c, 4);
        }
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int r = base + r0 + 4 * k;
            if (r < rows) *reinterpret_cast<uint32_t *>(pix + r * P + 4 * c) = v[k];
        }
    }
}

// LDS of one wave (= one cell): pixel tile rows x P | queue[qcap] u16 | score per queue entry [qcap] u8.  After the scores are
// known the pixel tile is dead and its memory becomes the zero-aproned score tile of the NMS; the survivors of the NMS
// overwrite the head of the queue in place.  ~4.7 KB for EuRoC (P = 48, qcap = 768) -> 32 waves per CU: the kernel's time
// falls with occupancy (measured 0.50 / 0.42 / 0.36 / 0.32 ms at 7 / 9 / 11 / 14 waves per CU).
__host__ __device__ inline size_t fast_wave_lds_bytes(int P, int max_rows, int qcap) {
    return (((size_t)max_rows * P + 16 + 15) & ~(size_t)15) + (size_t)qcap * 3 + 16;
}

template <int P>
__device__ __forceinline__ void fast_wave_cell(const TileRef t, const int f, const uint32_t tile_index, uint8_t *smem,
                                               const LevelInfo *__restrict__ lv, const uint8_t *__restrict__ pyr, size_t pyr_frame_stride,
                                               int32_t *__restrict__ cellcnt, int total_cells, uint32_t *__restrict__ cellent,
                                               size_t ent_frame_stride, int iniTh, int minTh, int max_rows, int qcap,
                                               uint32_t *__restrict__ ovf_list, int32_t *__restrict__ ovf_count, const Level0Src src0) {
    const LevelInfo L = lv[t.level];
    const int lane = threadIdx.x;
    const int cell = t.ti * L.nCols + t.tj;
    int32_t *cnt_out = cellcnt + (size_t)f * total_cells + L.cell_base + cell;

    const int maxBX = L.w - kBorder, maxBY = L.h - kBorder;
    const int iniX = kBorder + t.tj * L.wCell, iniY = kBorder + t.ti * L.hCell;
    const int maxX = min(iniX + L.wCell + 6, maxBX), maxY = min(iniY + L.hCell + 6, maxBY);
    const int cols = maxX - iniX, rows = maxY - iniY;
    const int iw = cols - 6, ih = rows - 6;
    if (iniY >= maxBY - 3 || iniX >= maxBX - 6 || iw <= 0 || ih <= 0) {  // :813 / :821 skip rules
        if (lane == 0) *cnt_out = 0;
        return;
    }
    uint8_t *pix = smem;                                                                  // rows * P (+16 slack)
    uint16_t *queue = reinterpret_cast<uint16_t *>(smem + (((size_t)max_rows * P + 16 + 15) & ~(size_t)15));  // qcap entries
    uint8_t *scq = reinterpret_cast<uint8_t *>(queue + qcap);                             // qcap scores

    // phase 0: (unaligned) dword loads starting one byte left of the sub-image
    const bool in_place = src0.img != nullptr && t.level == 0;   // level 0 in place: the caller's frame
    fast_tile_load<P>(in_place ? src0.img + (size_t)f * src0.frame_stride + (size_t)iniY * src0.row_stride + iniX - 1
                               : pyr + (size_t)f * pyr_frame_stride + L.off + (size_t)(kEdge + iniY) * L.pitch + kRoiX + iniX - 1,
                      in_place ? (int)src0.row_stride : L.pitch, rows, cols, pix, lane);
    __syncthreads();  // single-wave workgroup: a compiler/LDS ordering point, no s_barrier

    // phase 1: antipodal-pair test at minTh, four pixels per lane; passing pixels are queued in row-major order
    int qn = 0;
    {
        // lane -> (row inside the iteration, 4-pixel group): 64 / G whole rows per iteration, so nothing is divided in the loop
        const int G = (iw + 3) >> 2, RPI = 64 / G;
        const uint32_t rcpG = ((1u << 20) + (uint32_t)G - 1u) / (uint32_t)G;
        const int lrow = (int)(((uint32_t)lane * rcpG) >> 20), lg = lane - lrow * G;
        const uint32_t vmask = lrow < RPI ? (0xfu >> max(4 * lg + 3 - (iw - 1), 0)) : 0u;  // pixels of the last group beyond the interior
        const uint8_t *Abase = pix + lrow * P + 4 * lg + 4;
        const uint32_t ebase = ((uint32_t)lrow << 8) | (uint32_t)(4 * lg);
        const u16x2 t2 = as_pk((uint32_t)minTh * 0x00010001u);
        constexpr int D = P / 4;  // dwords per tile row
        for (int y0 = 0; y0 < ih; y0 += RPI) {
            const bool act = y0 + lrow < ih;
            const uint32_t *A = reinterpret_cast<const uint32_t *>(Abase + y0 * P);  // row y-3 of the centre row y+3
            const uint32_t r8 = A[0], r0 = A[6 * D];
            const uint32_t aL = A[1 * D - 1], aC = A[1 * D], aR = A[1 * D + 1];    // centre row - 2
            const uint32_t cL = A[3 * D - 1], cC = A[3 * D], cR = A[3 * D + 1];    // centre row
            const uint32_t bL = A[5 * D - 1], bC = A[5 * D], bR = A[5 * D + 1];    // centre row + 2
            // even / odd pixels of a neighbour dword that starts s bytes into the pair (hi, lo): ONE v_perm_b32 each
            // (byte s, 0, byte s + 2, 0) / (byte s + 1, 0, byte s + 3, 0) instead of v_alignbyte + two unpacks
#define FW_EVEN(hi, lo, s) as_pk(__builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)(s) | ((uint32_t)((s) + 2) << 16)))
#define FW_ODD(hi, lo, s) as_pk(__builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)((s) + 1) | ((uint32_t)((s) + 3) << 16)))
            const uint32_t fe = quick_pairs(pk_even(cC), pk_even(r0), pk_even(r8), FW_EVEN(cR, cC, 3), FW_EVEN(cC, cL, 1), FW_EVEN(bR, bC, 2),
                                            FW_EVEN(aC, aL, 2), FW_EVEN(aR, aC, 2), FW_EVEN(bC, bL, 2), t2);   // pixels 0 (low half) and 2
            const uint32_t fo = quick_pairs(pk_odd(cC), pk_odd(r0), pk_odd(r8), FW_ODD(cR, cC, 3), FW_ODD(cC, cL, 1), FW_ODD(bR, bC, 2),
                                            FW_ODD(aC, aL, 2), FW_ODD(aR, aC, 2), FW_ODD(bC, bL, 2), t2);      // pixels 1 and 3
#undef FW_EVEN
#undef FW_ODD
            // flags -> 4-bit mask: min(f, 1) per half (inline asm: the compiler expands the packed min into compares + selects)
            uint32_t ze, zo;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(ze) : "v"(fe), "v"(0x00010001u));
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(zo) : "v"(fo), "v"(0x00010001u));
            const uint32_t z = ze | (zo << 1);  // bits 0, 1, 16, 17
            uint32_t m4 = (z | (z >> 14)) & 0xfu;
            m4 &= act ? vmask : 0u;
            const int c = __popc(m4);
            const int incl = wave_incl_scan(c);
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            if (qn + tot > qcap) {  // more candidates than the LDS queue holds: the generic kernel takes this cell (k_fast_overflow)
                if (lane == 0 && ovf_list) ovf_list[atomicAdd(ovf_count, 1)] = ((uint32_t)f << 16) | tile_index;
                return;
            }
            int pos = qn + incl - c;
            const uint32_t e0 = ebase + ((uint32_t)y0 << 8);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (m4 & (1u << k)) { queue[pos] = (uint16_t)(e0 + k); pos++; }
            }
            qn += tot;
        }
    }
    __syncthreads();

    // phase 2: exact score of the queued pixels, kept per queue entry while the pixel tile is still being read
    for (int e0 = 0; e0 < qn; e0 += 128) {   // two queue entries per lane (fast_score16_x2)
        const int ea = e0 + lane, eb = ea + 64;
        const int qa = queue[min(ea, qn - 1)], qb = queue[min(eb, qn - 1)];
        int sa, sb;
        fast_score16_x2(pix + ((qa >> 8) + 3) * P + (qa & 0xff) + 4, pix + ((qb >> 8) + 3) * P + (qb & 0xff) + 4, P, &sa, &sb);
        if (ea < qn) scq[ea] = (uint8_t)((sa >= minTh) ? sa : 0);
        if (eb < qn) scq[eb] = (uint8_t)((sb >= minTh) ? sb : 0);
    }
    __syncthreads();
    // the pixel tile is dead: its memory becomes the score tile (pitch P, 1-px zero apron)
    uint8_t *sco = pix;
    for (int i = lane; i < (ih + 2) * (P / 4); i += 64) reinterpret_cast<uint32_t *>(sco)[i] = 0;
    __syncthreads();
    for (int e = lane; e < qn; e += 64) {
        const int s = scq[e];
        if (s) { const int q = queue[e]; sco[((q >> 8) + 1) * P + (q & 0xff) + 1] = (uint8_t)s; }
    }
    __syncthreads();

    // phase 3: NMS over the queued pixels, strict '>' against all 8 neighbours ([OCV] FAST_t nonmax stage); the survivors
    // (row-major order kept) overwrite the head of the queue
    int ns = 0;
    bool any_ini = false;
    for (int e0 = 0; e0 < qn; e0 += 64) {
        const int e = e0 + lane;
        int keep = 0, s = 0, q = 0;
        if (e < qn) {
            q = queue[e];
            s = scq[e];
            const uint8_t *p = sco + ((q >> 8) + 1) * P + (q & 0xff) + 1;
            keep = (s > 0) & (s > p[-1]) & (s > p[1]) & (s > p[-P - 1]) & (s > p[-P]) & (s > p[-P + 1]) & (s > p[P - 1]) & (s > p[P]) &
                   (s > p[P + 1]);
        }
        const unsigned long long b = __ballot(keep != 0);
        any_ini |= __ballot(keep && s >= iniTh) != 0ull;
        if (keep) queue[ns + __popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)q;  // ns + rank <= e: never ahead of an unread entry
        ns += __popcll(b);
    }
    __syncthreads();

    // phase 4: emission in row-major order of the survivors at the cell's threshold
    const int thr = any_ini ? iniTh : minTh;
    uint32_t *slot = cellent + (size_t)f * ent_frame_stride + L.cand_off + (size_t)cell * L.cell_cap;
    int total = 0;
    for (int e0 = 0; e0 < ns; e0 += 64) {
        const int e = e0 + lane;
        int q = 0, s = 0;
        if (e < ns) { q = queue[e]; s = sco[((q >> 8) + 1) * P + (q & 0xff) + 1]; }
        const bool sel = e < ns && s >= thr;
        const unsigned long long b = __ballot(sel);
        if (sel) slot[total + __popcll(b & ((1ull << lane) - 1ull))] = pack_key((q & 0xff) + 3 + t.tj * L.wCell, (q >> 8) + 3 + t.ti * L.hCell, s);
        total += __popcll(b);
    }
    if (lane == 0) *cnt_out = total;
}

// A wave hands data to itself through LDS: its LDS instructions execute in program order, so a wave-private hand-over needs no s_barrier --
// only the compiler must not move LDS accesses across the point
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// k_fast_wave_list: the cells k_fast_strip put on its list -- no corner at iniTh (6 % of the cells of the EuRoC-like bench: the reference's
// second pass, :843-846) or a strip whose queues overflowed -- one wave per cell with the complete ini / min logic of fast_wave_cell and a
// queue that holds a whole cell.  grid (any), block 64, LDS for qcap = max interior pixels
template <int P>
__global__ __launch_bounds__(64) void k_fast_wave_list(const LevelInfo *__restrict__ lv, const TileRef *__restrict__ tiles,
                                                       const uint8_t *__restrict__ pyr, size_t pyr_frame_stride,
                                                       int32_t *__restrict__ cellcnt, int total_cells, uint32_t *__restrict__ cellent,
                                                       size_t ent_frame_stride, int iniTh, int minTh, int max_rows, int qcap,
                                                       const uint32_t *__restrict__ list, const int32_t *__restrict__ list_count, const Level0Src src0) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int count = *list_count;
    for (int k = blockIdx.x; k < count; k += gridDim.x) {
        const uint32_t e = list[k];
        fast_wave_cell<P>(tiles[e & 0xffffu], (int)(e >> 16), e & 0xffffu, smem, lv, pyr, pyr_frame_stride, cellcnt, total_cells, cellent,
                          ent_frame_stride, iniTh, minTh, max_rows, qcap, nullptr, nullptr, src0);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Quad-tree cull: one wave per (frame, level) emulating DistributeOctTree exactly (sequential list semantics,
// libstdc++ std::sort replica for the size-ordered expansion).  See octree.hip.h.
// ---------------------------------------------------------------------------------------------------------
}  // namespace orbx

#include "pyr_stream.hip.h"
#include "fast_strip.hip.h"
#include "octree.hip.h"
#include "octree_par.hip.h"

namespace orbx {

// ---------------------------------------------------------------------------------------------------------
// Output slots (ORBextractor.cc:1117-1162): keypoints of all levels in level order; a keypoint whose SCALED x
// lies in [lap0, lap1] is written from the back (stereoIndex--), the others from the front (monoIndex++).
// One workgroup per frame.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(const LevelInfo *__restrict__ lv, int nlevels,
                                                  const uint32_t *__restrict__ lvlkp, size_t lvlkp_frame_stride,
                                                  const int32_t *__restrict__ lvlcnt, WorkItem *__restrict__ work,
                                                  int cap, int32_t *__restrict__ count, int32_t *__restrict__ mono,
                                                  int lap0, int lap1, int32_t *__restrict__ err) {
    __shared__ int wsum[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int N = 0;
    for (int l = 0; l < nlevels; l++) N += lvlcnt[f * nlevels + l];
    if (N > cap) {
        if (tid == 0) { atomicExch(err, 1); count[f] = 0; mono[f] = 0; }
        return;
    }
    int g0 = 0, monoIdx = 0, stereoIdx = N - 1;
    const float flap0 = (float)lap0, flap1 = (float)lap1;
    for (int l = 0; l < nlevels; l++) {
        const LevelInfo L = lv[l];
        const int nl = lvlcnt[f * nlevels + l];
        const uint32_t *src = lvlkp + (size_t)f * lvlkp_frame_stride + L.lvl_off;
        for (int i0 = 0; i0 < nl; i0 += 256) {
            const int i = i0 + tid;
            uint32_t key = 0;
            bool valid = i < nl, lap = false;
            if (valid) {
                key = src[i];
                float x = (float)key_x(key);
                if (l != 0) x = x * L.scale;  // keypoint->pt *= scale (:1150)
                lap = (x >= flap0 && x <= flap1);
            }
            const unsigned long long bl = __ballot(valid && lap), bm = __ballot(valid && !lap);
            if (lane == 0) wsum[wv] = (__popcll(bl) << 16) | __popcll(bm);
            __syncthreads();
            int offl = 0, offm = 0, totl = 0, totm = 0;
            for (int k = 0; k < 4; k++) {
                const int v = wsum[k];
                if (k < wv) { offl += v >> 16; offm += v & 0xffff; }
                totl += v >> 16; totm += v & 0xffff;
            }
            if (valid) {
                const unsigned long long lt = (1ull << lane) - 1ull;
                int pos;
                if (lap) pos = stereoIdx - (offl + __popcll(bl & lt));
                else pos = monoIdx + offm + __popcll(bm & lt);
                WorkItem w;
                w.key = key; w.level = l; w.pos = pos;
                w.pitches = (uint32_t)L.pitch | ((uint32_t)L.bpitch << 16); w.off = (uint32_t)L.off; w.boff = (uint32_t)L.boff;
                w.scale = L.scale; w.size = L.size;
                work[(size_t)f * cap + g0 + i] = w;
            }
            monoIdx += totm;
            stereoIdx -= totl;
            __syncthreads();
        }
        g0 += nl;
    }
    if (tid == 0) { count[f] = N; mono[f] = monoIdx; }
}

// ---------------------------------------------------------------------------------------------------------
// [OCV] GaussianBlur 7x7 sigma 2 on 8U, fixed-point (Q8 taps, exact 2-D sum, one rounding, saturate).
// The reference blurs a ring-less clone with BORDER_REFLECT_101 (:1132-1133); the padded pyramid level already
// carries exactly that reflection in its 19-px ring, so the kernel reads the ring and needs no border logic.
//
// k_blur_stream (round 4; replaces k_blur_pk, whose arithmetic it keeps): a register-marching separable filter, no LDS, no barriers.  A lane owns a
// 4-pixel-wide column and walks down the rows of a strip (256 pixels x 42 rows):
//   * horizontal pass without byte alignment: the window of pixel x0 + j is three (j = 1, 2) or two (j = 0, 3) v_dot4_u32_u8 of the ALIGNED dwords
//     (left neighbour's, own, right neighbour's) against tap dwords shifted instead of the data;
//   * vertical pass on PAIRS of consecutive rows of horizontal sums packed 2 x u16 (a sum is at most 255 * 257): per output pixel three
//     v_dot2_u32_u16 + one v_mad_u32_u24 with the rounding constant as the addend; one v_lshl_or per pixel and row builds the pair (rows r-1, r), a
//     ring of six pair slots (rotation resolved at compile time: the row loop is unrolled by six) serves rows r-5, r-3, r-1;
//   * the result byte is bits 16..23 of the sum: with taps summing to 256 it cannot exceed 255 (SAT = false: no clamp); taps summing to 257 (the
//     OpenCV <= 4.5.0 table) clamp the sum first (SAT = true).
// What is new against k_blur_pk (one 256-thread workgroup per tile, one row requested ahead, three dwords per lane and row):
//   * a fixed number of single-wave workgroups, each walking through MANY strips back to back as one continuous row stream: SIX source rows in
//     flight per wave, the first rows of the next strip requested while the last rows of the current one are filtered (no bubble between strips).
//     The loads are inline assembly and the waits counted by hand (see load_row): left to the compiler the stream was drained at every loop header;
//   * ONE dword per lane and row (+ one for the two dwords beyond the wave's ends); the neighbours' dwords come from the neighbouring lanes
//     (DPP wave_shr / wave_shl): a third of the bytes through the texture path;
//   * item = strip (BlurItem, precomputed per geometry): up to 48 source rows = 8 groups of 6; group 0 only fills the vertical window, the others emit;
//   * wave = blockIdx.x: group x = blockIdx.x % nx takes the frames f = x (mod nx) (nx = 8: a frame's strips stay on the XCD whose L2 the FAST strips
//     of the same frame fill -- speed only, nothing depends on it), wave k of the group takes the items k, k + K, ... of the group's frame-major
//     item sequence: all waves of a group work on the same one or two frames at a time.
// The idea was a blur THROTTLED to the four wave slots per CU that k_fast_strip (7 workgroups = 28 of 32 slots, VALU bound) leaves free, so that the
// two overlap instead of fighting for slots.  Measured (profiles/r04_c_*): the strips take 404 us beside either form of the blur (272 us alone) --
// what stretches them is the memory system under the blur's traffic, not the wave slots -- and 1024 waves (4 per CU) are too few for the blur itself
// (300 us alone; 512: 454, 2048: 207, 4096: 175 = k_blur_pk).  What it does buy, at 2048 waves: EuRoC step 1.12 -> 1.09 ms, KITTI 1.51 -> 1.47,
// TUM-VI 1.61 -> 1.62.
// No inter-workgroup dependency.  grid (waves), block 64
// ---------------------------------------------------------------------------------------------------------
constexpr int kBlurTW = 256;            // pixels per strip row
constexpr int kBlurRows = 42;           // output rows per strip (6 rows that fill the window + 6 x 7 rows that emit)

struct BlurRaw {
    uint32_t m, c, p;  // pixels x0-4 .. x0-1, x0 .. x0+3, x0+4 .. x0+7
};

struct BlurItem {         // four dwords, fetched with one scalar load (sub-dword fields would become vector loads the row stream has to wait for)
    uint32_t src_off;    // byte offset inside a frame's pyramid slab of (level row y0 - 3, ROI column x0w): a ring row for y0 = 0
    uint32_t dst_off;    // byte offset inside a frame's blur slab of (row y0, column x0w)
    uint32_t pitches;    // pitch | bpitch << 16
    uint32_t misc;       // rmax | rows_out << 16 | nlanes << 24: rmax = last source row that exists, counted from level row y0 - 3 (rows past it are
                         // clamped, their results never stored); rows_out = min(42, h - y0); nlanes = lanes with pixels, ceil(min(256, w - x0w) / 4)
    __host__ __device__ uint32_t pitch() const { return pitches & 0xffffu; }
    __host__ __device__ uint32_t bpitch() const { return pitches >> 16; }
    __host__ __device__ uint32_t rmax() const { return misc & 0xffffu; }
    __host__ __device__ int rows_out() const { return (int)((misc >> 16) & 0xffu); }
    __host__ __device__ int nlanes() const { return (int)(misc >> 24); }
};
static_assert(sizeof(BlurItem) == 16, "BlurItem layout");

template <bool SAT>
__global__ __launch_bounds__(64) void k_blur_stream(const BlurItem *__restrict__ items, int nitems, const uint8_t *__restrict__ pyr, size_t pyr_frame_stride,
                                                    uint8_t *__restrict__ blur, size_t blur_frame_stride, int g0, int g1, int g2, int g3, int n_frames,
                                                    int nx) {
    const int lane = threadIdx.x;
    const int x = (int)(blockIdx.x % (unsigned)nx), k = (int)(blockIdx.x / (unsigned)nx), K = (int)(gridDim.x / (unsigned)nx);
    const int nfx = (n_frames - x + nx - 1) / nx;            // frames of this group: f = x + nx * fi
    const int total = nfx * nitems;
    if (k >= total) return;                                   // wave-uniform
    const int nmine = (total - k + K - 1) / K;
    auto tapd = [&](int d) -> uint32_t {
        d = d < 0 ? -d : d;
        return d == 0 ? (uint32_t)g3 : d == 1 ? (uint32_t)g2 : d == 2 ? (uint32_t)g1 : d == 3 ? (uint32_t)g0 : 0u;
    };
    uint32_t ht[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < 3; q++) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) v |= tapd(4 * q + b - (j + 4)) << (8 * b);
            ht[j][q] = v;
        }
    const u16x2 vp0 = as_pk((uint32_t)g0 | ((uint32_t)g1 << 16));   // rows r-6, r-5
    const u16x2 vp1 = as_pk((uint32_t)g2 | ((uint32_t)g3 << 16));   // rows r-4, r-3
    const u16x2 vp2 = as_pk((uint32_t)g2 | ((uint32_t)g1 << 16));   // rows r-2, r-1
    auto hsum = [&](const BlurRaw &v, uint32_t h[4]) {
        h[0] = __builtin_amdgcn_udot4(v.c, ht[0][1], __builtin_amdgcn_udot4(v.m, ht[0][0], 0u, false), false);
        h[1] = __builtin_amdgcn_udot4(v.p, ht[1][2], __builtin_amdgcn_udot4(v.c, ht[1][1], __builtin_amdgcn_udot4(v.m, ht[1][0], 0u, false), false), false);
        h[2] = __builtin_amdgcn_udot4(v.p, ht[2][2], __builtin_amdgcn_udot4(v.c, ht[2][1], __builtin_amdgcn_udot4(v.m, ht[2][0], 0u, false), false), false);
        h[3] = __builtin_amdgcn_udot4(v.p, ht[3][2], __builtin_amdgcn_udot4(v.c, ht[3][1], 0u, false), false);
    };
    // cursor over this wave's items: (fi, it) of item number j is k + j * K in the group's frame-major sequence
    auto advance = [&](int &fi, int &it) {
        it += K;
        while (it >= nitems) { it -= nitems; fi++; }
    };
    // r-th source row of the strip: ONE dword per lane (its own four pixels) + one for the two dwords beyond the wave's ends (lane 0: the dword to
    // its left, lane 63: the dword to its right; the lanes between repeat their own: an L1 hit).  The neighbours' dwords of the 12-pixel window come
    // from the neighbouring lanes (DPP wave_shr / wave_shl) when the row is filtered: a third of the bytes of three dwords per lane through the texture path, and the
    // rows in flight are single registers the register allocator can keep in place across the loop's back edge (as dwordx3 tuples they were copied
    // there, behind a wait for every outstanding load).
    // The loads are inline assembly and the waits are counted by hand (s_waitcnt vmcnt): left to the compiler, the row stream was drained at every loop
    // header (it cannot bound the number of outstanding operations across the conditional stores).  A row = 2 loads; when row slot s of a group is
    // consumed, the operations issued after its own loads are: 2 loads (+ 1 store) for each of the 5 other slots and the 2 loads just issued for its
    // own slot = 12 loads (+ up to 6 stores): vmcnt(12) is exact without stores and waits for at most two rows more than necessary with them.
    struct RowRegs { uint32_t c, e; };
    auto load_row = [&](const BlurItem &D, const uint8_t *src, int r, uint32_t lofs, uint32_t eofs) -> RowRegs {
        const uint32_t rel = min((uint32_t)r, D.rmax());
        const uint8_t *row = src + (size_t)(rel * D.pitch()) - 4;   // wave-uniform (SGPR base); the lane offsets below are >= 0
        RowRegs v;
        asm volatile("global_load_dword %0, %1, %2" : "=v"(v.c) : "v"(lofs), "s"(row) : "memory");
        asm volatile("global_load_dword %0, %1, %2" : "=v"(v.e) : "v"(eofs), "s"(row) : "memory");
        return v;
    };
    auto wait_row = [&](RowRegs &v) { asm volatile("s_waitcnt vmcnt(12)" : "+v"(v.c), "+v"(v.e)); };
    auto window = [&](const RowRegs &v) -> BlurRaw {
        BlurRaw w;
        w.c = v.c;
        w.m = (uint32_t)__builtin_amdgcn_update_dpp((int)v.e, (int)v.c, 0x138, 0xf, 0xf, false);   // wave_shr:1: lane i <- lane i - 1, lane 0 keeps e
        w.p = (uint32_t)__builtin_amdgcn_update_dpp((int)v.e, (int)v.c, 0x130, 0xf, 0xf, false);   // wave_shl:1: lane i <- lane i + 1, lane 63 keeps e
        return w;
    };
    // a lane past the level's width loads the dword right of the last lane with pixels (ring bytes its neighbour's window needs), then repeats it
    auto lane_ofs = [&](const BlurItem &D, uint32_t &lofs, uint32_t &eofs) {
        lofs = (uint32_t)min(lane, D.nlanes()) * 4u + 4u;   // + 4: the row base points one dword left of the strip
        // lane 63's right neighbour exists (and is needed) only when the strip is full: a part-filled strip of a narrow level may end within four
        // bytes of the row's -- for the last ring row of the last level of the last frame: of the slab's -- end
        eofs = lane == 0 ? 0u : (lane == 63 && D.nlanes() == 64) ? 260u : lofs;
    };
    int fi = 0, it = k;
    while (it >= nitems) { it -= nitems; fi++; }
    BlurItem Dc = items[__builtin_amdgcn_readfirstlane(it)];   // wave-uniform index, dword fields: one scalar load
    const uint8_t *srcc = pyr + (size_t)(x + nx * fi) * pyr_frame_stride + Dc.src_off;
    uint8_t *dstc = blur + (size_t)(x + nx * fi) * blur_frame_stride + Dc.dst_off;
    uint32_t lofc, eofc;
    lane_ofs(Dc, lofc, eofc);
    // Two sets of six row slots, used alternately (group g reads set g & 1 and requests the rows of group g + 1 into the other set): with ONE set
    // the loop-carried copies at the back edge made the compiler wait for every outstanding load once per group -- no rows in flight across groups.
    RowRegs rawA[6], rawB[6];
#pragma unroll
    for (int s = 0; s < 6; s++) rawA[s] = load_row(Dc, srcc, s, lofc, eofc);
    uint32_t pr[6][4];   // pr[q % 6] = (sums of row q-1) | (sums of row q) << 16
    uint32_t hprev[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int s = 0; s < 6; s++)
#pragma unroll
        for (int j = 0; j < 4; j++) pr[s][j] = 0u;
    for (int jm = 0; jm < nmine; jm++) {
        // the item after this one (its descriptor is needed when group 7 prefetches): fetched now, a whole strip ahead of its use
        int fin = fi, itn = it;
        const bool have_next = jm + 1 < nmine;
        if (have_next) advance(fin, itn);
        const BlurItem Dn = items[__builtin_amdgcn_readfirstlane(itn)];
        const uint8_t *srcn = pyr + (size_t)(x + nx * fin) * pyr_frame_stride + Dn.src_off;
        uint32_t lofn, eofn;
        lane_ofs(Dn, lofn, eofn);
        const bool live = lane < Dc.nlanes();
        // groups of this strip: one that fills the window + ceil(rows_out / 6) that emit, rounded up to an even number (the two sets of row slots
        // alternate, and the next strip starts with set A again): 8 for a full strip, fewer for the short last strip of a level
        const int ngroups = (1 + (Dc.rows_out() + 5) / 6 + 1) & ~1;
        // one group of six rows: filter the rows in `cons`, request the rows of the next group into `prod`
        auto group = [&](RowRegs (&cons)[6], RowRegs (&prod)[6], const int ph) {
            // rows of the NEXT group: group ph + 1 of this strip, or group 0 of the next strip (at the very end of the stream: six more rows of
            // this strip, clamped to its last row and never used)
            const bool wrap = ph == ngroups - 1 && have_next;
            const BlurItem &Dl = wrap ? Dn : Dc;
            const uint8_t *srcl = wrap ? srcn : srcc;
            const uint32_t lofl = wrap ? lofn : lofc, eofl = wrap ? eofn : eofc;
            const int rl = wrap ? 0 : 6 * (ph + 1);
#pragma unroll
            for (int s = 0; s < 6; s++) {
                prod[s] = load_row(Dl, srcl, rl + s, lofl, eofl);
                wait_row(cons[s]);
                uint32_t h[4];
                hsum(window(cons[s]), h);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    pr[s][j] = hprev[j] | (h[j] << 16);
                    hprev[j] = h[j];
                }
                if (ph > 0) {   // wave-uniform: group 0 only fills the window
                    const int yo = 6 * ph + s - 6;
                    uint32_t sum[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t a = __umul24(h[j], (uint32_t)g0) + 32768u;                        // row r
                        a = __builtin_amdgcn_udot2(as_pk(pr[(s + 5) % 6][j]), vp2, a, false);      // rows r-2, r-1
                        a = __builtin_amdgcn_udot2(as_pk(pr[(s + 3) % 6][j]), vp1, a, false);      // rows r-4, r-3
                        a = __builtin_amdgcn_udot2(as_pk(pr[(s + 1) % 6][j]), vp0, a, false);      // rows r-6, r-5
                        sum[j] = SAT ? min(a, 0x00ffffffu) : a;
                    }
                    if (live && yo < Dc.rows_out()) {
                        const uint32_t lo = __builtin_amdgcn_perm(sum[1], sum[0], 0x0c0c0602u);    // byte 2 of sum[0], byte 2 of sum[1]
                        const uint32_t hi = __builtin_amdgcn_perm(sum[3], sum[2], 0x06020c0cu);
                        *reinterpret_cast<uint32_t *>(dstc + (size_t)((uint32_t)yo * Dc.bpitch()) + 4u * (uint32_t)lane) = lo | hi;
                    }
                }
            }
        };
#pragma unroll 1
        for (int ph = 0; ph < ngroups; ph += 2) {
            group(rawA, rawB, ph);
            group(rawB, rawA, ph + 1);
        }
        // next strip
        fi = fin; it = itn;
        Dc = Dn; srcc = srcn; lofc = lofn; eofc = eofn;
        dstc = blur + (size_t)(x + nx * fi) * blur_frame_stride + Dc.dst_off;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the six rows requested beyond the end of the stream
}

// ---------------------------------------------------------------------------------------------------------
// glibc 2.35 sinf / cosf ("fma" ifunc variant: every multiply-add of the double polynomial fused), restated
// for |x| < 120; bit-identical to the x86-64 libm the reference links against (validated exhaustively on the
// CPU oracle, which uses the same formulation).  Tables: __sincosf_table.
// ---------------------------------------------------------------------------------------------------------
// Branch-free form (the two halves of a k_describe wave work on different keypoints, a data-dependent branch would run both sides):
// glibc's operations in glibc's order, the variant picked by selects (checked against the oracle's branching restatement on the CPU:
// tests/simt, every float angle in [0, 360) degrees at 1e-4 steps and 10^7 integer moment pairs).
__device__ __forceinline__ void glibc_sincosf(float y, float *sn, float *cs) {
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ff;
    const bool small = top < 0x3f4, tiny = top < 0x398;
    const double x0 = (double)y;
    const double r = __dmul_rn(x0, 0x1.45F306DC9C883p+23);
    const int n = small ? 0 : (((int32_t)r + 0x800000) >> 24);
    const double xr = small ? x0 : __fma_rn(-(double)n, 0x1.921FB54442D18p0, x0);
    const double x2 = __dmul_rn(xr, xr);
    const bool neg = (n & 2) != 0;                           // table T1: c0, c1, c2, c3, c4 negated; s1 .. s3 unchanged
    const double sg = neg ? -1.0 : 1.0;
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5, c3 = sg * -0x1.6c087e89a359dp-10,
                 c4 = sg * 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    const double xs = small ? xr : __dmul_rn(xr, sgn);        // |y| < pi/4 uses x itself (no multiplication by 1.0: same value)
    // sin polynomial of xs, cos polynomial of x2 (sc_sin_poly / sc_cos_poly, operation for operation)
    const double x3 = __dmul_rn(xs, x2);
    const double sp1 = __fma_rn(x2, s3, s2);
    const double x7 = __dmul_rn(x3, x2);
    const double sp = __fma_rn(x3, s1, xs);
    const float fs = (float)__fma_rn(x7, sp1, sp);
    const double x4 = __dmul_rn(x2, x2);
    const double cq2 = __fma_rn(x2, c4, c3);
    const double cq1 = __fma_rn(x2, c1, c0);
    const double x6 = __dmul_rn(x4, x2);
    const double cq = __fma_rn(x4, c2, cq1);
    const float fc = (float)__fma_rn(x6, cq2, cq);
    const bool swap = (n & 1) != 0;                          // odd quadrant: sin <- cos polynomial, cos <- sin polynomial
    *sn = tiny ? y : (swap ? fc : fs);
    *cs = tiny ? 1.0f : (swap ? fs : fc);
}
__device__ __forceinline__ float fast_atan2_deg(float y, float x, const bool fma_poly) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale,
                p7 = -0.04432655554792128f * scale;
    const float eps = 2.2204460492503131e-16f;  // (float)DBL_EPSILON
    const float ax = fabsf(x), ay = fabsf(y);
    const bool xbig = ax >= ay;
    const float c = __fdiv_rn(xbig ? ay : ax, __fadd_rn(xbig ? ax : ay, eps));
    const float c2 = __fmul_rn(c, c);
    float a;
    if (fma_poly) {   // ORBX_FLAG_ATAN_FMA: the polynomial as a compiler that contracts a * b + c emits it (three fused steps; 90 - q * c is one fnmadd)
        const float q = __fmaf_rn(__fmaf_rn(__fmaf_rn(p7, c2, p5), c2, p3), c2, p1);
        a = xbig ? __fmul_rn(q, c) : __fmaf_rn(-q, c, 90.f);
    } else {
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
        a = xbig ? a : __fsub_rn(90.f, a);
    }
    a = x < 0 ? __fsub_rn(180.f, a) : a;
    a = y < 0 ? __fsub_rn(360.f, a) : a;
    return a;
}

// a pointer the program knows to be wave-uniform, made provably so for the compiler (buffer descriptors must live in SGPRs)
__device__ __forceinline__ const uint8_t *uniform_ptr(const uint8_t *p) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (const uint8_t *)(((uint64_t)hi << 32) | lo);
}

// fp_mode bits of the descriptor kernels (their `fp_mode` argument): which of the float forms the reference BINARY executes
constexpr int kFpDescStrict = 1;   // ORBX_FLAG_DESC_STRICT: x*b + y*a with separately rounded products (-ffp-contract=off build of ORBextractor.cc)
constexpr int kFpAtanFma = 2;      // ORBX_FLAG_ATAN_FMA: cv::fastAtan2's polynomial contracted to FMAs (an OpenCV whose baseline has FMA3 / NEON)

struct DescConst {
    int8_t vmax_of_u[16];              // orientation disc: largest |v| with umax[|v|] >= |u|
    int8_t pat[1024];                  // bit_pattern_31_ (x0,y0,x1,y1) x 256
    float patf[256][4];                // the same pattern as floats (x0, y0, x1, y1): k_describe_fused takes it converted (four v_cvt per pair less)
    uint32_t ic_mask[32][4][10];       // orientation disc, row form: row hl = v + 15 (row 31: empty), alignment axB = (kx - 18) & 3: byte t of the row is
                                       // 0xff where |t - 18 - axB| <= umax[|v|] (k_describe_fused, ic_moments_rows)
};

constexpr int kDescAP = 40, kDescAR = 31;   // orientation patch in LDS: 31 rows x 40 B (9 aligned dwords used)
constexpr int kDescBP = 44, kDescBR = 37;   // BRIEF patch in LDS: 37 rows x 44 B (10 aligned dwords used)
constexpr int kDescWaveLds = kDescAP * kDescAR + kDescBP * kDescBR + 12;  // 2880 B, multiple of 16

// orbx_extract (one frame per call, results wanted on the host at once): k_describe stores the frame's keypoints and descriptors a second time,
// into a pinned host block, with the error word / count / monocular index in front -- the call then ends with ONE stream synchronisation
// instead of three copy-and-wait round trips (85 of 256 us per call, tools/latency_timeline.sh).  hdr == nullptr: no mirror (every batched path).
struct HostMirror {
    int32_t *hdr;            // [0] device error word, [1] count, [2] monoIndex
    orbx_keypoint *kps;      // [cap]
    uint8_t *desc;           // [cap][32]
    const int32_t *err, *mono;
};

// IC_Angle moments of the half's keypoint, COLUMN form (:76-103): lane = disc column, m_10 = u * sum I, m_01 = sum v * I; c0 = centre of the
// orientation patch (pitch AP) + this lane's disc column du
template <int AP>
__device__ __forceinline__ void ic_moments_columns(const uint8_t *c0, const int dvmax, const int du, const int hw, int *M10, int *M01) {
    int sumI = 0, m01 = 0;
#pragma unroll
    for (int v = -kHalfPatch; v <= kHalfPatch; v++) {
        if ((v < 0 ? -v : v) <= dvmax) {
            const int I = c0[v * AP];
            sumI += I;
            m01 += v * I;
        }
    }
    const int s10 = wave_incl_scan(du * sumI), s01 = wave_incl_scan(m01);
    const int a10 = __builtin_amdgcn_readlane(s10, 31), b10 = __builtin_amdgcn_readlane(s10, 63);
    const int a01 = __builtin_amdgcn_readlane(s01, 31), b01 = __builtin_amdgcn_readlane(s01, 63);
    *M10 = hw ? b10 - a10 : a10;
    *M01 = hw ? b01 - a01 : a01;
}

// the same moments, ROW form (k_describe_fused, which is bound by VALU issue): lane = disc row v = hl - 15; the row's 40 bytes (ten aligned dwords of
// the orientation patch, byte t = pixel x0 + t) are masked to the disc (DescConst::ic_mask: 0xff where |t - ctr| <= umax[|v|], ctr = byte of the
// keypoint's column) and reduced with v_dot4: S = sum I, T = sum t * I (weights t = 0 .. 39 as literals), m_10 row = T - ctr * S, m_01 row = v * S.
// 32 instead of 78 vector instructions per wave, three loads of the mask row instead of 31 dependent LDS byte reads.
__device__ __forceinline__ void ic_moments_rows(const uint8_t *row, const uint32_t *__restrict__ mask_row, const int ctr, const int hw, const int hl,
                                                int *M10, int *M01) {
    uint32_t px[10], mk[10];
    const uint4 *r4 = reinterpret_cast<const uint4 *>(row);   // LDS rows are 16-byte aligned (pitch 48, patch base a multiple of 16)
    const uint4 pa = r4[0], pb = r4[1];
    const uint2 pc = *reinterpret_cast<const uint2 *>(row + 32);
    uint4 ma, mb;   // a mask row is 40 bytes: 8-byte aligned only
    uint2 mc;
    __builtin_memcpy(&ma, mask_row, 16);
    __builtin_memcpy(&mb, mask_row + 4, 16);
    __builtin_memcpy(&mc, mask_row + 8, 8);
    px[0] = pa.x; px[1] = pa.y; px[2] = pa.z; px[3] = pa.w; px[4] = pb.x; px[5] = pb.y; px[6] = pb.z; px[7] = pb.w; px[8] = pc.x; px[9] = pc.y;
    mk[0] = ma.x; mk[1] = ma.y; mk[2] = ma.z; mk[3] = ma.w; mk[4] = mb.x; mk[5] = mb.y; mk[6] = mb.z; mk[7] = mb.w; mk[8] = mc.x; mk[9] = mc.y;
    uint32_t S = 0u, T = 0u;
#pragma unroll
    for (int d = 0; d < 10; d++) {
        const uint32_t I = px[d] & mk[d];
        S = __builtin_amdgcn_udot4(I, 0x01010101u, S, false);
        T = __builtin_amdgcn_udot4(I, 0x03020100u + 0x04040404u * (uint32_t)d, T, false);
    }
    const int m10 = (int)T - ctr * (int)S, m01 = (hl - kHalfPatch) * (int)S;   // lane 31 of a half: an all-zero mask row
    const int s10 = wave_incl_scan(m10), s01 = wave_incl_scan(m01);
    const int a10 = __builtin_amdgcn_readlane(s10, 31), b10 = __builtin_amdgcn_readlane(s10, 63);
    const int a01 = __builtin_amdgcn_readlane(s01, 31), b01 = __builtin_amdgcn_readlane(s01, 63);
    *M10 = hw ? b10 - a10 : a10;
    *M01 = hw ? b01 - a01 : a01;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The tail of both descriptor kernels in three pieces (k_describe runs them back to back in every wave; k_describe_fused runs the middle one -- the same
// for all 32 lanes of a half -- ONCE per workgroup, on eight lanes of wave 0, between two barriers):
//   describe_rotation  the keypoint's angle (:76-103 fastAtan2) and the rotation (cos, sin) of the steered pattern (:112-113)
//   describe_record    the keypoint record of slot w.pos
//   describe_brief     the 256 comparisons on the blurred patch (bc = its centre, pitch BP); lanes 0..7 of a half end up with descriptor dword hl
__device__ __forceinline__ void describe_rotation(const int M10, const int M01, const int fp_mode, float *angle, float *a, float *b) {
    *angle = fast_atan2_deg((float)M01, (float)M10, (fp_mode & kFpAtanFma) != 0);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    glibc_sincosf(__fmul_rn(*angle, factorPI), b, a);  // a = cos, b = sin
}

__device__ __forceinline__ void describe_record(const uint32_t key, const int level, const int pos, const float scale, const float size, const float angle,
                                                const int f, const int cap, orbx_keypoint *__restrict__ kps, const HostMirror &hm) {
    orbx_keypoint kp;
    float x = (float)key_x(key), y = (float)key_y(key);
    if (level != 0) { x = __fmul_rn(x, scale); y = __fmul_rn(y, scale); }
    kp.x = x; kp.y = y; kp.size = size; kp.angle = angle; kp.response = (float)key_s(key);
    kp.octave = level; kp.class_id = -1;
    kps[(size_t)f * cap + pos] = kp;
    if (hm.hdr) hm.kps[pos] = kp;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// patf[it] = (x0, y0, x1, y1) of pattern pair it * 32 + hl.  Returns the eight ballots' halves: out[it] = dword `it` of the descriptor of this half's keypoint
template <int BP>
__device__ __forceinline__ void describe_brief(const float a, const float b, const uint8_t *bc, const int hw, const f32x4 (&patf)[8],
                                               const int fp_mode, uint32_t (&out)[8]) {
    constexpr float kMagic = 12582912.f;           // cvRound by magic add, see k_describe
    constexpr uint32_t kMagicBits = 0x4B400000u;
    const uint32_t cbm = (uint32_t)(uintptr_t)bc - (0x400000u * (uint32_t)BP + kMagicBits);
    // the two points of a pattern pair as ONE packed operation each (v_pk_mul / v_pk_fma / v_pk_add_f32 on (point 0, point 1)): the same multiplications,
    // fused multiply-adds and additions per element as the scalar form of the reference binary
    const f32x2 aa = {a, a}, bb = {b, b}, mg = {kMagic, kMagic};
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const f32x2 X = {patf[it].x, patf[it].z}, Y = {patf[it].y, patf[it].w};
        f32x2 R, Q;
        if (fp_mode & kFpDescStrict) {
            R = X * bb + Y * aa;      // -ffp-contract=off: products and sums round separately
            Q = X * aa - Y * bb;
        } else {  // GCC -O3 -march=native: fma(x, b, y*a), fma(x, a, -(y*b))
            R = __builtin_elementwise_fma(X, bb, Y * aa);
            Q = __builtin_elementwise_fma(X, aa, -(Y * bb));
        }
        const f32x2 Rm = R + mg, Qm = Q + mg;
        const uint32_t a0 = __umul24(__float_as_uint(Rm.x), (uint32_t)BP) + __float_as_uint(Qm.x) + cbm;
        const uint32_t a1 = __umul24(__float_as_uint(Rm.y), (uint32_t)BP) + __float_as_uint(Qm.y) + cbm;
        const int t0 = *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>(a0);
        const int t1 = *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>(a1);
        const unsigned long long bal = __ballot(t0 < t1);
        out[it] = hw ? (uint32_t)(bal >> 32) : (uint32_t)bal;
    }
}

template <int BP>
__device__ __forceinline__ void describe_tail(const int M10, const int M01, const uint8_t *bc, const int hw, const int hl,
                                              const uint32_t (&pat8)[8], const int fp_mode, const bool live, const WorkItem &w, const int kx,
                                              const int ky, const int f, const int cap, orbx_keypoint *__restrict__ kps, uint8_t *__restrict__ desc,
                                              const HostMirror &hm) {
    float angle, a, b;
    describe_rotation(M10, M01, fp_mode, &angle, &a, &b);
    f32x4 patf[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const char4 pt = __builtin_bit_cast(char4, pat8[it]);
        patf[it] = f32x4{(float)pt.x, (float)pt.y, (float)pt.z, (float)pt.w};
    }
    uint32_t d8[8];
    describe_brief<BP>(a, b, bc, hw, patf, fp_mode, d8);
    uint32_t mine = 0;   // lanes 0..7 of a half: descriptor dword hl
#pragma unroll
    for (int it = 0; it < 8; it++) mine = hl == it ? d8[it] : mine;
    if (!live) return;
    const size_t slot = (size_t)f * cap + w.pos;
    if (hl < 8) reinterpret_cast<uint32_t *>(desc + slot * 32)[hl] = mine;
    if (hm.hdr && hl < 8) reinterpret_cast<uint32_t *>(hm.desc + (size_t)w.pos * 32)[hl] = mine;
    if (hl == 0) describe_record(w.key, w.level, w.pos, w.scale, w.size, angle, f, cap, kps, hm);
}

// ---------------------------------------------------------------------------------------------------------
// k_describe: IC_Angle on the UNBLURRED level (:76-103), steered 256-pair BRIEF on the BLURRED level (:107-146), keypoint record +
// descriptor written to the final slot.  TWO keypoints per wave, one per half (32 lanes): about a third of the one-keypoint-per-wave
// form's instructions were wave-uniform work the vector unit executed for 64 identical lanes (work-item unpacking, fastAtan2, the FP64
// sincos); with one keypoint per half that work serves two keypoints at once (round 3, profiles/r03_a_ab_prepared_kernels.log:
// step 1.175 -> 1.160 ms; the one-keypoint form is gone):
//   disc  : lane = column u of the orientation disc, 31 row steps; per-half totals from one wave prefix sum (lanes 31 and 63)
//   brief : lane i of a half evaluates pattern pairs i, i + 32, ..., i + 224; each of the 8 ballots holds one dword of BOTH descriptors
//   the two branches of fastAtan2 / sincosf become selects (glibc_sincosf, fast_atan2_deg: bit-identical, checked on the CPU)
// grid xcd_grid(ceil(cap / 8), B), block 256
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_describe(const DescConst *__restrict__ dc, const WorkItem *__restrict__ work,
                                                   const int32_t *__restrict__ count, int cap, const uint8_t *__restrict__ pyr,
                                                   size_t pyr_frame_stride, const uint8_t *__restrict__ blur, size_t blur_frame_stride,
                                                   orbx_keypoint *__restrict__ kps, uint8_t *__restrict__ desc, int fp_mode, int n_frames,
                                                   const HostMirror hm) {
    __shared__ __attribute__((aligned(16))) uint8_t patches[4 * 2 * kDescWaveLds];
    int bx, f;
    if (!xcd_frame_map(n_frames, &bx, &f)) return;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, hw = lane >> 5, hl = lane & 31;
    const int cnt = count[f];
    if (hm.hdr && bx == 0 && f == 0 && threadIdx.x == 0) { hm.hdr[0] = *hm.err; hm.hdr[1] = cnt; hm.hdr[2] = hm.mono[0]; }
    const int g0 = (bx * 4 + wv) * 2;   // first keypoint of the wave
    if (g0 >= cnt) return;              // wave-uniform
    const bool live = g0 + hw < cnt;    // an odd count leaves the last wave's second half idle: it repeats the last keypoint and stores nothing
    const WorkItem w = work[(size_t)f * cap + min(g0 + hw, cnt - 1)];
    uint8_t *A = patches + (wv * 2 + hw) * kDescWaveLds;
    uint8_t *Bp = A + kDescAP * kDescAR;
    const int pitch = (int)(w.pitches & 0xffffu), bpitch = (int)(w.pitches >> 16);
    const int kx = key_x(w.key), ky = key_y(w.key);
    const int du = hl - kHalfPatch;
    const int dvmax = hl <= 2 * kHalfPatch ? dc->vmax_of_u[du < 0 ? -du : du] : -1;
    uint32_t pat8[8];
#pragma unroll
    for (int it = 0; it < 8; it++) pat8[it] = reinterpret_cast<const uint32_t *>(dc->pat)[it * 32 + hl];

    // ---- both patches of both keypoints: 16 lanes per row, 2 rows per step and half, every load issued before the first LDS store ----
    const int axA = (kx - kHalfPatch) & 3, axB = (kx - 18) & 3;
    {
        // through one buffer descriptor per slab of the FRAME (wave-uniform; the two halves address different levels through their 32-bit
        // offsets): a lane that has nothing to fetch -- columns past the patch width, the odd half's row past the last one -- gets an offset
        // outside the descriptor's range (no memory access, no per-load predicate, no zero-filled registers)
        const auto srdA = __builtin_amdgcn_make_buffer_rsrc((void *)uniform_ptr(pyr + (size_t)f * pyr_frame_stride), 0, (int)min(pyr_frame_stride, (size_t)0x7fffffff), 0x00020000);
        const auto srdB = __builtin_amdgcn_make_buffer_rsrc((void *)uniform_ptr(blur + (size_t)f * blur_frame_stride), 0, (int)min(blur_frame_stride, (size_t)0x7fffffff), 0x00020000);
        const int c = hl & 15, r0 = hl >> 4;
        constexpr uint32_t kOut = 0x80000000u;
        const uint32_t offA = (c < 9 ? 0u : kOut) + w.off + (uint32_t)((kEdge + ky - kHalfPatch + r0) * pitch + kRoiX + (kx - kHalfPatch - axA) + 4 * c);
        const uint32_t offB = (c < 10 ? 0u : kOut) + w.boff + (uint32_t)((ky - 18 + r0) * bpitch + (kx - 18 - axB) + 4 * c);
        const uint32_t last = r0 == 0 ? 0u : kOut;   // rows 30 (A) and 36 (B) exist for the even half only
        uint32_t va[16], vb[19];
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = __builtin_amdgcn_raw_buffer_load_b32(srdA, (int)((offA + (uint32_t)(2 * k * pitch)) | (k == 15 ? last : 0u)), 0, 0);
#pragma unroll
        for (int k = 0; k < 19; k++) vb[k] = __builtin_amdgcn_raw_buffer_load_b32(srdB, (int)((offB + (uint32_t)(2 * k * bpitch)) | (k == 18 ? last : 0u)), 0, 0);
        if (c < 9) {
            uint8_t *d = A + r0 * kDescAP + 4 * c;
#pragma unroll
            for (int k = 0; k < 15; k++) *reinterpret_cast<uint32_t *>(d + 2 * k * kDescAP) = va[k];
            if (r0 == 0) *reinterpret_cast<uint32_t *>(d + 30 * kDescAP) = va[15];
        }
        if (c < 10) {
            uint8_t *d = Bp + r0 * kDescBP + 4 * c;
#pragma unroll
            for (int k = 0; k < 18; k++) *reinterpret_cast<uint32_t *>(d + 2 * k * kDescBP) = vb[k];
            if (r0 == 0) *reinterpret_cast<uint32_t *>(d + 36 * kDescBP) = vb[18];
        }
    }
    wave_lds_sync();

    int M10, M01;
    ic_moments_columns<kDescAP>(A + kHalfPatch * kDescAP + kHalfPatch + axA + du, dvmax, du, hw, &M10, &M01);
    describe_tail<kDescBP>(M10, M01, Bp + 18 * kDescBP + 18 + axB, hw, hl, pat8, fp_mode, live, w, kx, ky, f, cap, kps, desc, hm);
}

// ---------------------------------------------------------------------------------------------------------
// k_describe_fused (round 4): k_describe with the GaussianBlur of :1132-1133 computed ON DEMAND -- the 37 x 37 blurred pixels the steered
// pattern can reach (|rotated coordinate| <= 18), from the 43 x 43 raw pixels around the keypoint -- instead of read back from a blurred copy
// of the whole pyramid.  The blur is linear up to its final rounding (k_blur_stream: sum of tap products + 32768, byte 2, saturated for the
// <= 4.5.0 taps), and the ring of a level in the pyramid slab IS its BORDER_REFLECT_101 extension, so the values are the ones k_blur_stream
// writes.  What it trades: no k_blur_stream launch (2 P bytes per frame read + written) and one 43-row patch read per keypoint instead of a
// 31-row and a 37-row one, against 19 row filters per lane.
//   lanes of a half (one keypoint): three segments of ten dword columns, lane = (segment, column c); columns 4 c .. 4 c + 3 from the aligned
//     dword that holds pixel kx - 18; segment s filters raw rows R0 .. R0 + 18 (R0 = 0, 13, 25 counted from row ky - 21) into blurred rows
//     R0 .. R0 + 12 (counted from ky - 18): 13 + 12 (+ 1 again) + 12 (+ 1 never read) rows, the same code in every segment
//   a row = ONE 12-byte load per lane: the dword left of its column, its own, the one right of it (the kernel is bound by VALU issue: with the
//     neighbours' dwords over DPP as in k_blur_stream, a row costs two v_mov_dpp + two v_cndmask at the segment ends on top of its 15 filter
//     instructions; the overlapping loads hit the L1)
//   the raw rows go to LDS as they are (orientation patch: rows ky - 21 .. ky + 22, IC_Angle reads ky - 15 .. ky + 15), the blurred ones beside them;
//     lanes 30 / 31 of a half run along as columns 10 / 11 of segment 2 (offsets outside the descriptor: zeros) and write the two pad dwords of a row
// grid xcd_grid(ceil(cap / 8), B), block 256
// ---------------------------------------------------------------------------------------------------------
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
constexpr int kDfP = 48;                  // pitch of both patches in LDS: 12 dwords per row, 10 used
constexpr int kDfAR = 44, kDfBR = 38;     // raw rows ky - 21 .. ky + 22, blurred rows ky - 18 .. ky + 19
constexpr int kDfRows = 19;               // raw rows per segment
constexpr int kDfWaveLds = kDfP * (kDfAR + kDfBR);   // 3936 B per keypoint
static_assert(kDfWaveLds % 16 == 0, "LDS carve");

template <bool SAT>
__global__ __launch_bounds__(256) void k_describe_fused(const DescConst *__restrict__ dc, const WorkItem *__restrict__ work,
                                                         const int32_t *__restrict__ count, int cap, const uint8_t *__restrict__ pyr,
                                                         size_t pyr_frame_stride, int g0, int g1, int g2, int g3, orbx_keypoint *__restrict__ kps,
                                                         uint8_t *__restrict__ desc, int fp_mode, int n_frames, const HostMirror hm,
                                                         const Level0Src src0, int W0, int H0, uint8_t *__restrict__ dbg_patch, int dbg_frame) {
    __shared__ __attribute__((aligned(16))) uint8_t patches[4 * 2 * kDfWaveLds];
    int bx, f;
    if (!xcd_frame_map(n_frames, &bx, &f)) return;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, hw = lane >> 5, hl = lane & 31;
    const int cnt = count[f];
    if (hm.hdr && bx == 0 && f == 0 && threadIdx.x == 0) { hm.hdr[0] = *hm.err; hm.hdr[1] = cnt; hm.hdr[2] = hm.mono[0]; }
    const int g0i = (bx * 4 + wv) * 2;   // first keypoint of the wave
    if (bx * 8 >= cnt) return;           // workgroup-uniform (the barriers below are for every wave of a workgroup that has a keypoint)
    const bool live = g0i + hw < cnt;    // halves past the frame's last keypoint (the last workgroup only) repeat it and store nothing
    const WorkItem w = work[(size_t)f * cap + min(g0i + hw, cnt - 1)];
    uint8_t *A = patches + (wv * 2 + hw) * kDfWaveLds;
    uint8_t *Bp = A + kDfP * kDfAR;
    const int pitch = (int)(w.pitches & 0xffffu);
    const int kx = key_x(w.key), ky = key_y(w.key);

    const int seg = hl < 10 ? 0 : hl < 20 ? 1 : 2, c = hl - 10 * seg, R0 = seg == 0 ? 0 : seg == 1 ? 13 : 25;
    const int axB = (kx - 18) & 3;
    u32x3 vw[kDfRows];   // a row's 12-pixel window of this lane: the dword left of its own, its own, the dword right of it
    // Level 0 in place (src0.img != nullptr): a level-0 keypoint reads the caller's frame, which has no REFLECT_101 ring.  Its window (columns
    // kx - 25 .. kx + 25, rows ky - 21 .. ky + 22 at most) lies inside the image unless the keypoint sits within 25 px of a side / 22 px of the top
    // or bottom (keypoints start 16 px inside): those few take the staged form below, which applies the reflection itself.
    const bool ip = src0.img != nullptr && w.level == 0;
    const bool border = ip && (kx < 25 || kx > W0 - 27 || ky < 21 || ky > H0 - 23);
    constexpr uint32_t kOut = 0x80000000u;   // an offset outside the descriptor: no memory access, the load returns 0
    const unsigned long long bip = __ballot(ip);
    if (__ballot(border) == 0ull && (bip == 0ull || bip == ~0ull)) {
        // both keypoints of the wave in the slab, or both on the in-place level 0: one descriptor, one 12-byte load per row (wave-uniform choice)
        const size_t img_bytes = (size_t)(H0 - 1) * src0.row_stride + (size_t)W0;
        const uint8_t *base = bip ? src0.img + (size_t)f * src0.frame_stride : pyr + (size_t)f * pyr_frame_stride;
        const auto srd = __builtin_amdgcn_make_buffer_rsrc((void *)uniform_ptr(base), 0, (int)min(bip ? img_bytes : pyr_frame_stride, (size_t)0x7fffffff), 0x00020000);
        const int lp = bip ? (int)src0.row_stride : pitch;
        const uint32_t own = bip ? (uint32_t)((ky - 21 + R0) * lp + (kx - 18 - axB) + 4 * c)
                                 : w.off + (uint32_t)((kEdge + ky - 21 + R0) * lp + kRoiX + (kx - 18 - axB) + 4 * c);
        const uint32_t offW = c < 10 ? own - 4u : kOut;
#pragma unroll
        for (int k = 0; k < kDfRows; k++) vw[k] = __builtin_amdgcn_raw_buffer_load_b96(srd, (int)(offW + (uint32_t)(k * lp)), 0, 0);
    } else {
        // staged form: the 44 x 48-byte window (columns X0 - 4 .. X0 + 43, X0 = kx - 18 - axB a multiple of 4) of each half's keypoint goes through the raw
        // patch's LDS rows (byte 0 = column X0 - 4; the filter loop below rewrites them in their final layout), 528 dwords over 32 lanes.  A dword of the
        // window is one dword of the image, possibly byte-reversed: columns x < 0 come from -x, columns x >= W from 2 W - 2 - x ([OCV] BORDER_REFLECT_101),
        // rows likewise.  A slab level in the other half takes the same route with the identity mapping (its ring is in the slab), and so does the one
        // wave of a frame whose halves straddle the in-place level 0 and level 1 (two base addresses: no common descriptor).
        const uint8_t *roi = ip ? src0.img + (size_t)f * src0.frame_stride : pyr + (size_t)f * pyr_frame_stride + w.off + (size_t)(kEdge * pitch + kRoiX);
        const int sp = ip ? (int)src0.row_stride : pitch;
        const int X0m4 = kx - 18 - axB - 4;
        uint32_t stage[17];
#pragma unroll
        for (int t = 0; t < 17; t++) {
            const int i = min(hl + 32 * t, 527);
            const int r = (i * 43691) >> 19, j = i - 12 * r;   // i / 12
            int y = ky - 21 + r, x0 = X0m4 + 4 * j;
            uint32_t sel = 0x03020100u;
            if (ip) {
                y = reflect101(y, H0);
                const int d = W0 - x0;               // pixels of the dword inside the image
                if (x0 < 0) { x0 = -x0 - 3; sel = 0x00010203u; }
                else if (d <= 0) { x0 = 2 * W0 - 5 - x0; sel = 0x00010203u; }
                else if (d < 4) { x0 = W0 - 4; sel = d == 1 ? 0x00010203u : d == 2 ? 0x01020302u : 0x02030201u; }   // the dword straddles the right edge
            }
            uint32_t v;
            __builtin_memcpy(&v, roi + (ptrdiff_t)y * sp + x0, 4);
            stage[t] = __builtin_amdgcn_perm(0u, v, sel);
        }
#pragma unroll
        for (int t = 0; t < 17; t++) {
            const int i = hl + 32 * t;
            const int r = (i * 43691) >> 19, j = i - 12 * r;
            if (i < 528) *reinterpret_cast<uint32_t *>(A + r * kDfP + 4 * j) = stage[t];
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < kDfRows; k++) {
            u32x3 v = {0u, 0u, 0u};
            if (c < 10) __builtin_memcpy(&v, A + (R0 + k) * kDfP + 4 * c, 12);
            vw[k] = v;
        }
        wave_lds_sync();
    }
    {
        // horizontal taps of the four pixels of a dword over its 12-pixel window (m | c | p), as byte vectors for v_dot4 (k_blur_stream's)
        auto tapd = [&](int d) -> uint32_t {
            d = d < 0 ? -d : d;
            return d == 0 ? (uint32_t)g3 : d == 1 ? (uint32_t)g2 : d == 2 ? (uint32_t)g1 : d == 3 ? (uint32_t)g0 : 0u;
        };
        uint32_t ht[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < 3; q++) {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) v |= tapd(4 * q + b - (j + 4)) << (8 * b);
                ht[j][q] = v;
            }
        // Vertical taps over PAIRS of rows (v_dot2).  Only the pairs that start at an even row are formed (one v_lshl_or per pixel and row pair): an
        // output row whose window starts at an odd row takes that row from the high half of the pair before it (taps 0, g0) and its last two rows as a pair.
        const u16x2 te0 = as_pk((uint32_t)g0 | ((uint32_t)g1 << 16)), te1 = as_pk((uint32_t)g2 | ((uint32_t)g3 << 16)), te2 = as_pk((uint32_t)g2 | ((uint32_t)g1 << 16));
        const u16x2 to0 = as_pk((uint32_t)g0 << 16), to1 = as_pk((uint32_t)g1 | ((uint32_t)g2 << 16)), to2 = as_pk((uint32_t)g3 | ((uint32_t)g2 << 16)),
                    to3 = as_pk((uint32_t)g1 | ((uint32_t)g0 << 16));
        uint8_t *ad = A + R0 * kDfP + 4 * c, *bd = Bp + R0 * kDfP + 4 * c;
        uint32_t pe[4][4];   // pe[(e / 2) % 4] = (sums of row e) | (sums of row e + 1) << 16 for the even rows e
        uint32_t hev[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < kDfRows; k++) {
            const uint32_t m = vw[k].x, cc = vw[k].y, p = vw[k].z;
            *reinterpret_cast<uint32_t *>(ad + k * kDfP) = cc;
            uint32_t h[4];
            h[0] = __builtin_amdgcn_udot4(cc, ht[0][1], __builtin_amdgcn_udot4(m, ht[0][0], 0u, false), false);
            h[1] = __builtin_amdgcn_udot4(p, ht[1][2], __builtin_amdgcn_udot4(cc, ht[1][1], __builtin_amdgcn_udot4(m, ht[1][0], 0u, false), false), false);
            h[2] = __builtin_amdgcn_udot4(p, ht[2][2], __builtin_amdgcn_udot4(cc, ht[2][1], __builtin_amdgcn_udot4(m, ht[2][0], 0u, false), false), false);
            h[3] = __builtin_amdgcn_udot4(p, ht[3][2], __builtin_amdgcn_udot4(cc, ht[3][1], 0u, false), false);
            const int q = k >> 1;
            if (k & 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) pe[q & 3][j] = hev[j] | (h[j] << 16);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) hev[j] = h[j];
            }
            if (k >= 6) {
                uint32_t sum[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t a;
                    if (k & 1) {
                        a = __builtin_amdgcn_udot2(as_pk(pe[(q - 3) & 3][j]), to0, 32768u, false);   // row r-6
                        a = __builtin_amdgcn_udot2(as_pk(pe[(q - 2) & 3][j]), to1, a, false);        // rows r-5, r-4
                        a = __builtin_amdgcn_udot2(as_pk(pe[(q - 1) & 3][j]), to2, a, false);        // rows r-3, r-2
                        a = __builtin_amdgcn_udot2(as_pk(pe[q & 3][j]), to3, a, false);              // rows r-1, r
                    } else {
                        a = __umul24(h[j], (uint32_t)g0) + 32768u;                                   // row r
                        a = __builtin_amdgcn_udot2(as_pk(pe[(q - 1) & 3][j]), te2, a, false);        // rows r-2, r-1
                        a = __builtin_amdgcn_udot2(as_pk(pe[(q - 2) & 3][j]), te1, a, false);        // rows r-4, r-3
                        a = __builtin_amdgcn_udot2(as_pk(pe[(q - 3) & 3][j]), te0, a, false);        // rows r-6, r-5
                    }
                    sum[j] = SAT ? min(a, 0x00ffffffu) : a;
                }
                const uint32_t lo = __builtin_amdgcn_perm(sum[1], sum[0], 0x0c0c0602u);    // byte 2 of sum[0], byte 2 of sum[1]
                const uint32_t hi = __builtin_amdgcn_perm(sum[3], sum[2], 0x06020c0cu);
                *reinterpret_cast<uint32_t *>(bd + (k - 6) * kDfP) = lo | hi;
            }
        }
    }
    wave_lds_sync();
    if (dbg_patch != nullptr && f == dbg_frame && live) {   // debug readout (orbx_debug_fused_patches): the 37 x 37 blurred pixels around the keypoint, as they lie in LDS
        uint8_t *o = dbg_patch + (size_t)(g0i + hw) * (37 * 37);
        for (int r = hl; r < 37; r += 32)
            for (int j = 0; j < 37; j++) o[r * 37 + j] = Bp[r * kDfP + j + axB];
    }
    // the lane's eight pattern pairs, as floats: issued here (the row windows' registers are free), in flight across the moments and the barriers below
    f32x4 patf[8];
#pragma unroll
    for (int it = 0; it < 8; it++) patf[it] = reinterpret_cast<const f32x4 *>(dc->patf)[it * 32 + hl];
    int M10, M01;
    ic_moments_rows(A + (6 + min(hl, 30)) * kDfP, dc->ic_mask[min(hl, 31)][axB], 18 + axB, hw, hl, &M10, &M01);
    // fastAtan2, sincosf and the keypoint record are the same ~120 vector instructions in all 32 lanes of a half: they run ONCE per workgroup instead, its
    // eight keypoints on lanes 0..7 of wave 0, while waves 1..3 wait (the kernel is bound by VALU issue: 844 -> ~760 instructions per wave)
    __shared__ uint32_t kp_in[8][8];   // per keypoint: m_10, m_01, key, level, output slot, scale, size
    __shared__ float rot[8][2];        // cos, sin of its angle
    const int kq = wv * 2 + hw;
    if (hl == 0) {
        kp_in[kq][0] = (uint32_t)M10; kp_in[kq][1] = (uint32_t)M01; kp_in[kq][2] = w.key; kp_in[kq][3] = (uint32_t)w.level;
        kp_in[kq][4] = (uint32_t)w.pos; kp_in[kq][5] = __float_as_uint(w.scale); kp_in[kq][6] = __float_as_uint(w.size);
    }
    __syncthreads();
    if (wv == 0) {
        const int k = lane & 7;
        float angle, ca, sb;
        describe_rotation((int)kp_in[k][0], (int)kp_in[k][1], fp_mode, &angle, &ca, &sb);
        if (lane < 8) {
            rot[k][0] = ca; rot[k][1] = sb;
            if (bx * 8 + k < cnt)
                describe_record(kp_in[k][2], (int)kp_in[k][3], (int)kp_in[k][4], __uint_as_float(kp_in[k][5]), __uint_as_float(kp_in[k][6]), angle, f, cap, kps, hm);
        }
    }
    __syncthreads();
    uint32_t d8[8];
    describe_brief<kDfP>(rot[kq][0], rot[kq][1], Bp + 18 * kDfP + 18 + axB, hw, patf, fp_mode, d8);
    if (live && hl == 0) {   // the half's 32 descriptor bytes from one lane
        uint4 *o = reinterpret_cast<uint4 *>(desc + ((size_t)f * cap + w.pos) * 32);
        o[0] = make_uint4(d8[0], d8[1], d8[2], d8[3]);
        o[1] = make_uint4(d8[4], d8[5], d8[6], d8[7]);
        if (hm.hdr) {
            uint4 *oh = reinterpret_cast<uint4 *>(hm.desc + (size_t)w.pos * 32);
            oh[0] = make_uint4(d8[0], d8[1], d8[2], d8[3]);
            oh[1] = make_uint4(d8[4], d8[5], d8[6], d8[7]);
        }
    }
}

}  // namespace orbx
