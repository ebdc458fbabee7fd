// fast_strip.hip.h -- k_fast_strip: the first pass of the reference's per-cell detection, cv::FAST(cell, iniThFAST, nonmax) for every
// cell of a frame (/root/reference/src/ORBextractor.cc:805-842), one workgroup per STRIP of cells instead of one wave per cell.
// Included by extractor_kernels.hip.h (uses fast_score16, quick_pairs, the DPP scan and the packed-u16 helpers defined there).
//
// A strip = up to floor(256 / wCell) horizontally adjacent cells of one cell row of one level.  The score of a pixel does not depend on
// the cell it lies in ("corner at threshold t" <=> cornerScore >= t, and the cell interiors tile the detection window
// exactly), so the detection stages run over the strip's interior as ONE image, 4 pixels per lane across its whole width:
//   phase 0  the strip's sub-image rows (interior + 3-px apron, one extra byte left so that interior 4-pixel groups are dword aligned
//            in LDS) -> LDS, rows interleaved over the 4 waves, every load of a wave in flight at once (buffer descriptor: no
//            per-row predicates)                                                                                   | barrier
//   stage A  per 4-pixel group: v_sad_u8 rejection on the vertical and horizontal antipodal pairs (if both pixels of a pair are
//            within t of all four centres no 9-arc exists); lanes = groups of 64 / G whole rows (G = groups per row), every wave
//            on its own band of rows; surviving groups queued row-major (ballot + mbcnt)
//   stage B  the four-antipodal-pair test on the queued groups only (packed u16, both polarities); passing pixels queued row-major
//   scores   exact cornerScore of the queued pixels, TWO per lane on packed u16 halves (v_pk_minimum3 / maximum3_f16, fast_score16_x2);
//            corners (score >= t) compacted in place                                                             | barrier
//   2nd pass cells of the strip WITHOUT a corner at iniThFAST in any wave's band (known now, while the pixel tile is still alive): the reference
//            runs cv::FAST(cell, minThFAST) for them (:843-846).  The four-pair test and the exact scores at minTh over those cells' columns
//            only (one wave per such cell, all its rows), appended to that wave's queue; the NMS below separates cells by a zero column, so the two thresholds never meet.
//            (Rounds 3-4 sent these cells -- 6 % of the bench's -- to k_fast_wave_list, which re-read their sub-images from HBM: 92 MB per
//            256 frames and a launch.)                                                                             | barrier (only if any)
//   NMS      the pixel tile is dead: each wave zeroes its band of the score tile and scatters its corners into it, with one ZERO
//            COLUMN between adjacent cells -- cv::FAST runs per cell sub-image, so a neighbour in another cell scores 0 (the
//            strip's rows belong to one cell row, the apron rows are zero)                                         | barrier
//            strict '>' against the 8 neighbours; survivors compacted in place, counted per (wave, cell)           | barrier
//   emit     a cell's survivors in row-major order = wave 0's, then wave 1's, ... (bands are row ranges): slot position = the
//            earlier waves' counts of that cell + the rank inside the wave.  What is left for the list pass (k_fast_wave_list): cells whose
//            corners at iniTh ALL lost the NMS (ties only: the reference then runs FAST(minTh) on them) and strips whose queues overflowed.
// What it removes against one wave per cell (k_fast_ini, rounds 1-2): the (wCell + 6)(hCell + 6) / (wCell hCell) apron re-read per
// cell (now only the 6 apron rows per strip and 8 columns per 256), the per-cell set-up and the part-filled last iterations of every
// stage (a strip's queues are 5-7 cells long), the lanes a 36-px cell row leaves idle (9 groups x 6-7 rows = 54-63 of 64).
// grid xcd_grid(n_strips, B), block 256, dynamic LDS = fast_strip_lds_bytes(...)
#pragma once

namespace orbx {

// LDS bytes per tile row: 4 + interior + 4, rounded up to 16.  Three tile shapes share one launch (and one LDS budget per workgroup): strips of
// tall cells (the top pyramid levels: a level with two or three cell rows has cells of up to 63 px) are cut narrower, so that rows x pitch
// stays inside the budget that the ordinary 44 - 46-row strips of 7 cells set -- a taller tile would cost EVERY workgroup an occupancy step
constexpr int kStripPitch = 272;   // interior <= 256
constexpr int kStripPitchMid = 208, kStripPitchLow = 144;   // interior <= 192 / <= 128
__host__ __device__ inline int strip_max_interior(int pitch) { return pitch == kStripPitch ? 256 : pitch == kStripPitchMid ? 192 : 128; }
constexpr int kStripMaxCells = 8;  // floor(256 / 35) = 7 cells at most

struct StripTile {   // one strip, precomputed per geometry (48 bytes, scalar loads)
    uint32_t src_off;      // byte offset inside a frame's pyramid slab of tile byte (row 0, col 0) = level pixel (X0 - 4, Y0 - 3)
    int32_t pitch;         // bytes per padded row of the level
    int16_t iw, ih;        // interior size: iw <= 256 pixels, ih <= 63 rows
    int16_t ox, oy;        // candidate coordinates of interior pixel (0, 0): 3 + j0 * wCell, 3 + i * hCell
    uint32_t cell0;        // index of the strip's first cell in a frame's cell-count array (= its TileRef index for the list pass)
    uint32_t slot0;        // entry offset of that cell's slot inside a frame's candidate slab
    uint32_t cell_cap;     // entries per cell slot
    uint32_t rcp_wcell;    // ceil(2^16 / wCell): cell of interior column x = (x * rcp_wcell) >> 16  (exact for x < 256, wCell < 256)
    uint32_t rcp_groups;   // ceil(2^20 / G), G = (iw + 3) / 4 dword groups per interior row: lane -> (row, group) without a division
    uint16_t rows_per_iter, ncell;   // 64 / G rows per stage-A iteration; cells in the strip
    uint32_t lds_pitch;    // kStripPitch / kStripPitchMid / kStripPitchLow
    uint32_t wcell;        // cell width (interior columns [c * wcell, min((c + 1) * wcell, iw)) belong to cell c of the strip)
};
static_assert(sizeof(StripTile) == 48, "StripTile layout");

// per wave: group queue u16[gcap] (dead after stage B: the scores u8[qcap] of the pixel queue reuse its bytes, qcap <= 2 gcap) | pixel queue u16[qcap]
__host__ __device__ inline size_t fast_strip_wave_bytes(int gcap, int qcap) { return ((size_t)gcap * 2 + (size_t)qcap * 2 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t fast_strip_lds_bytes(int waves, int pix_bytes, int gcap, int qcap) {
    return (size_t)pix_bytes + waves * fast_strip_wave_bytes(gcap, qcap) + (waves * kStripMaxCells + 4) * sizeof(int32_t);
}

template <int W, int P>   // W waves per workgroup = row bands per strip; P = LDS pitch of the tile
__device__ __forceinline__ void fast_strip_body(const StripTile &T, const int f, uint8_t *smem, const uint8_t *__restrict__ tile_src, const int spitch,
                                                int32_t *__restrict__ cellcnt, int total_cells, uint32_t *__restrict__ cellent, size_t ent_frame_stride,
                                                int iniTh, int minTh, int pix_bytes, int gcap, int qcap, uint32_t *__restrict__ list,
                                                int32_t *__restrict__ list_count, int second_pass) {
    constexpr int D = P / 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int iw = T.iw, ih = T.ih, rows = ih + 6;
    uint8_t *pix = smem;
    uint8_t *wbase = smem + (size_t)pix_bytes + (size_t)wave * fast_strip_wave_bytes(gcap, qcap);
    uint16_t *gq = reinterpret_cast<uint16_t *>(wbase);          // group queue
    uint16_t *pq = gq + gcap;                                     // pixel queue -> corners -> survivors (compacted in place)
    uint8_t *ps = reinterpret_cast<uint8_t *>(gq);                // score per pixel-queue entry, written when the group queue is dead
    int32_t *cnt = reinterpret_cast<int32_t *>(smem + (size_t)pix_bytes + W * fast_strip_wave_bytes(gcap, qcap));   // [W][8] survivors per (wave, cell)
    int32_t *ovf = cnt + W * kStripMaxCells;   // [0] a queue overflowed in the first pass, [1] cells with a corner at iniTh (bit per cell), [2] cells whose second pass overflowed (bit per cell)
    if (threadIdx.x == 0) { ovf[0] = 0; ovf[1] = 0; ovf[2] = 0; }

    // ---- phase 0: rows wave, wave + 4, ... ; lane = dword of the row (G + 2 <= 66 dwords: the two beyond lane 63 in a second sweep) ----
    {
        const int nd = ((iw + 3) >> 2) + 2;
        const uint64_t a = (uint64_t)tile_src;   // tile byte (row 0, col 0) = level pixel (X0 - 4, Y0 - 3)
        const uint32_t alo = __builtin_amdgcn_readfirstlane((uint32_t)a), ahi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        const int nbytes = (rows - 1) * spitch + 4 * nd;
        const auto srd = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)ahi << 32) | alo), 0, nbytes, 0x00020000);
        const int oob = 0x40000000;
        const int voff = (lane < nd ? 0 : oob) + 4 * lane;
        for (int rb = wave; rb < rows; rb += 16 * W) {   // 16 rows of this wave in flight at once (rows past the tile: out of range, no access)
            uint32_t v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b32(srd, voff + (rb + W * k) * spitch, 0, 0);
            if (lane < nd) {
                uint8_t *d = pix + rb * P + 4 * lane;
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (rb + W * k < rows) *reinterpret_cast<uint32_t *>(d + W * k * P) = v[k];
            }
        }
        if (nd > 64)   // dwords 64, 65 of rows wave + W * (lane >> 1) [+ 32 W, ...]
            for (int xb = wave; xb < rows; xb += 32 * W) {
                const int xr = xb + W * (lane >> 1), xd = 64 + (lane & 1);
                const bool xlive = xd < nd && xr < rows;
                const uint32_t xv = __builtin_amdgcn_raw_buffer_load_b32(srd, (xlive ? 0 : oob) + xr * spitch + 4 * xd, 0, 0);
                if (xlive) *reinterpret_cast<uint32_t *>(pix + xr * P + 4 * xd) = xv;
            }
    }
    __syncthreads();

    const int BH = (ih + W - 1) / W;                       // rows per wave band
    const int yb0 = wave * BH, yb1 = min(ih, yb0 + BH);
    const uint32_t ut = (uint32_t)iniTh;

    // ---- stage A: groups whose vertical AND horizontal antipodal pairs cannot both be rejected by the SAD bound ----
    // counters are wave-uniform (ballot popcounts: scalar unit); an entry past the queue's end is dropped and the count runs on,
    // so that an overflow is seen after the loop (no data-dependent exit inside it)
    int gn = 0;
    {
        const int G = (iw + 3) >> 2, RPI = (int)T.rows_per_iter;
        const int lrow0 = (int)(((uint32_t)lane * T.rcp_groups) >> 20);
        const bool lane_live = lrow0 < RPI;                 // the lanes past the last whole row of an iteration idle
        const int lrow = lane_live ? lrow0 : 0, lg = lane_live ? lane - lrow0 * G : 0;
        const uint8_t *Ap = pix + (yb0 + lrow) * P + 4 * lg + 4;
        uint32_t ent = ((uint32_t)(yb0 + lrow) << 8) | (uint32_t)(4 * lg);
        int left = lane_live ? yb1 - yb0 - lrow : 0;        // rows of the band at and below this lane's row
        // (no unroll request: the trip count is data dependent and the body holds a ballot -- hipcc refuses it with a warning, rounds 3-4)
        for (int y0 = yb0; y0 < yb1; y0 += RPI) {
            const uint32_t *A = reinterpret_cast<const uint32_t *>(Ap);   // tile row y = level row of the centre - 3
            const uint32_t r8 = A[0], r0 = A[6 * D];
            const uint32_t cL = A[3 * D - 1], cC = A[3 * D], cR = A[3 * D + 1];
            const uint32_t sv = max(__builtin_amdgcn_sad_u8(cC, r0, 0u), __builtin_amdgcn_sad_u8(cC, r8, 0u));
            const uint32_t p4 = __builtin_amdgcn_alignbyte(cR, cC, 3), p12 = __builtin_amdgcn_alignbyte(cC, cL, 1);
            const uint32_t sh = max(__builtin_amdgcn_sad_u8(cC, p4, 0u), __builtin_amdgcn_sad_u8(cC, p12, 0u));
            const bool keep = left > 0 && min(sv, sh) > ut;
            const unsigned long long b = __ballot(keep);
            const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, (uint32_t)gn));
            if (keep && pos < gcap) gq[pos] = (uint16_t)ent;
            gn += __popcll(b);
            Ap += RPI * P; ent += (uint32_t)RPI << 8; left -= RPI;
        }
    }
    bool over = gn > gcap;
    wave_lds_sync();

    // ---- stage B: the four-pair test on the queued groups, four pixels per lane, row-major queue order ----
    int qn = 0;
    if (!over) {
        const u16x2 t2 = as_pk(ut * 0x00010001u);
        for (int g0 = 0; g0 < gn; g0 += 64) {
            const int gi = g0 + lane;
            const bool act = gi < gn;
            const uint32_t e0 = act ? (uint32_t)gq[gi] : 0u;
            const int y = (int)(e0 >> 8), x4 = (int)(e0 & 0xff);
            const uint32_t vmask = act ? 0xfu >> max(x4 + 3 - (iw - 1), 0) : 0u;   // pixels of the last group beyond the interior
            const uint32_t *A = reinterpret_cast<const uint32_t *>(pix + y * P + x4 + 4);
            const uint32_t r8 = A[0], r0 = A[6 * D];
            const uint32_t aL = A[1 * D - 1], aC = A[1 * D], aR = A[1 * D + 1];
            const uint32_t cL = A[3 * D - 1], cC = A[3 * D], cR = A[3 * D + 1];
            const uint32_t bL = A[5 * D - 1], bC = A[5 * D], bR = A[5 * D + 1];
#define FS_EVEN(hi, lo, s) as_pk(__builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)(s) | ((uint32_t)((s) + 2) << 16)))
#define FS_ODD(hi, lo, s) as_pk(__builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)((s) + 1) | ((uint32_t)((s) + 3) << 16)))
            const uint32_t fe = quick_pairs(pk_even(cC), pk_even(r0), pk_even(r8), FS_EVEN(cR, cC, 3), FS_EVEN(cC, cL, 1), FS_EVEN(bR, bC, 2),
                                            FS_EVEN(aC, aL, 2), FS_EVEN(aR, aC, 2), FS_EVEN(bC, bL, 2), t2);
            const uint32_t fo = quick_pairs(pk_odd(cC), pk_odd(r0), pk_odd(r8), FS_ODD(cR, cC, 3), FS_ODD(cC, cL, 1), FS_ODD(bR, bC, 2),
                                            FS_ODD(aC, aL, 2), FS_ODD(aR, aC, 2), FS_ODD(bC, bL, 2), t2);
            uint32_t ze, zo;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(ze) : "v"(fe), "v"(0x00010001u));
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(zo) : "v"(fo), "v"(0x00010001u));
            const uint32_t z = ze | (zo << 1);
            const uint32_t m4 = (z | (z >> 14)) & vmask;
            const int c = __popc(m4);
            const int incl = wave_incl_scan(c);
            int pos = qn + incl - c;
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            if (qn + tot <= qcap) {   // wave-uniform: a chunk that would run past the queue's end is dropped whole (the strip overflows: `over` below)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (m4 & (1u << k)) pq[pos] = (uint16_t)(e0 + k);
                    pos += (int)(m4 >> k & 1u);
                }
            }
            qn += tot;
#undef FS_EVEN
#undef FS_ODD
        }
        over = qn > qcap;
    }
    wave_lds_sync();

    // ---- exact scores of the queued pixels; corners (score >= th) compacted in place, row-major order kept ----
    // entries [begin, end) of the pixel queue -> corners appended at `out` (out <= begin: the compaction never overtakes an unread entry)
    auto score_compact = [&](const int begin, const int end, const int th, int out) -> int {
        for (int e0 = begin; e0 < end; e0 += 128) {   // two queue entries per lane: e0 + lane and e0 + 64 + lane of 128 per iteration
            const int ea = e0 + lane, eb = e0 + 64 + lane;
            const int qa = pq[min(ea, end - 1)], qb = pq[min(eb, end - 1)];
            int sa, sb;
            fast_score16_x2(pix + ((qa >> 8) + 3) * P + (qa & 0xff) + 4, pix + ((qb >> 8) + 3) * P + (qb & 0xff) + 4, P, &sa, &sb);
            const bool ca = sa >= th && ea < end, cb = sb >= th && eb < end;
            const unsigned long long ba = __ballot(ca), bb = __ballot(cb);
            __builtin_amdgcn_wave_barrier();   // all 128 entries read before any is overwritten
            const int na = __popcll(ba);
            if (ca) {
                const int o = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ba >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ba, (uint32_t)out));
                pq[o] = (uint16_t)qa;
                ps[o] = (uint8_t)sa;
            }
            if (cb) {
                const int o = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bb, (uint32_t)(out + na)));
                pq[o] = (uint16_t)qb;
                ps[o] = (uint8_t)sb;
            }
            out += na + __popcll(bb);
        }
        return out;
    };
    int nc = 0;
    if (!over) nc = score_compact(0, qn, iniTh, 0);
    if (second_pass == 1 && !over) {   // which cells have a corner at iniTh in this wave's band
        wave_lds_sync();
        uint32_t m = 0u;
        for (int e = lane; e < nc; e += 64) m |= 1u << (((uint32_t)(pq[e] & 0xff) * T.rcp_wcell) >> 16);
        // OR over the wave by the DPP steps of the prefix scan (six v_or; until round 6 a ballot per cell; an LDS atomic per LANE was measured at + 22 us per launch:
        // 64 lanes on one address serialize), then one LDS atomic per wave
#define FS_OR_DPP(ctrl, rows) m |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, ctrl, rows, 0xf, false)
        FS_OR_DPP(0x111, 0xf); FS_OR_DPP(0x112, 0xf); FS_OR_DPP(0x114, 0xf); FS_OR_DPP(0x118, 0xf); FS_OR_DPP(0x142, 0xa); FS_OR_DPP(0x143, 0xc);
#undef FS_OR_DPP
        if (lane == 63 && m) atomicOr(reinterpret_cast<uint32_t *>(ovf + 1), m);
    }
    if (over && lane == 0) ovf[0] = 1;
    __syncthreads();   // without a second pass the pixel tile is dead from here on
    bool ovf_any = ovf[0] != 0;
    // ---- second pass (:843-846): cells without any corner at iniTh, at minTh, while their pixels are still in LDS ----
    const uint32_t needy = (second_pass == 1 && !ovf_any) ? ~(uint32_t)ovf[1] & ((1u << T.ncell) - 1u) : 0u;   // the same value in every wave
    if (needy) {
        const u16x2 t2 = as_pk((uint32_t)minTh * 0x00010001u);
        const int wcell = (int)T.wcell;
        int qn2 = nc;   // the candidates of the second pass queue up behind the corners of the first
        // ONE wave takes a cell, all its rows (needy cell number i goes to wave i mod W): 64 / Gc rows per iteration fill the lanes (a wave's own band
        // of ~10 rows would leave a fifth of them idle and cost every wave the set-up); the other waves wait at the barrier below
        int turn = 0;
        for (uint32_t rest = needy; rest; rest &= rest - 1u, turn++) {
            if ((turn & (W - 1)) != wave) continue;                               // wave-uniform
            const int c = __builtin_ctz(rest);
            const int xa = c * wcell, xb = min(xa + wcell, iw);                   // the cell's interior columns [xa, xb)
            const int g0 = xa >> 2, Gc = ((xb - 1) >> 2) - g0 + 1;                // its 4-pixel groups (the first and last may straddle a neighbour)
            const uint32_t rcpG = (65536u + (uint32_t)Gc - 1u) / (uint32_t)Gc;    // lane / Gc for lane < 64
            const int RPI = (int)((64u * rcpG) >> 16);                            // whole rows per iteration
            const int lrow = (int)(((uint32_t)lane * rcpG) >> 16), lg = lane - lrow * Gc;
            const int x4 = 4 * (g0 + lg);
            // pixels of the group inside the cell (and so inside the interior: xb <= iw)
            const uint32_t cmask = lrow < RPI ? ((0xfu << max(xa - x4, 0)) & 0xfu) & (0xfu >> max(x4 + 3 - (xb - 1), 0)) : 0u;
            const int qstart = qn2;
            for (int y0 = 0; y0 < ih; y0 += RPI) {
                const int y = y0 + lrow;
                const bool act = cmask != 0u && y < ih;
                const uint32_t *A = reinterpret_cast<const uint32_t *>(pix + (act ? y : 0) * P + (act ? x4 : 4 * g0) + 4);
                const uint32_t r8 = A[0], r0 = A[6 * D];
                const uint32_t aL = A[1 * D - 1], aC = A[1 * D], aR = A[1 * D + 1];
                const uint32_t cL = A[3 * D - 1], cC = A[3 * D], cR = A[3 * D + 1];
                const uint32_t bL = A[5 * D - 1], bC = A[5 * D], bR = A[5 * D + 1];
#define FS_EVEN(hi, lo, s) as_pk(__builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)(s) | ((uint32_t)((s) + 2) << 16)))
#define FS_ODD(hi, lo, s) as_pk(__builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)((s) + 1) | ((uint32_t)((s) + 3) << 16)))
                const uint32_t fe = quick_pairs(pk_even(cC), pk_even(r0), pk_even(r8), FS_EVEN(cR, cC, 3), FS_EVEN(cC, cL, 1), FS_EVEN(bR, bC, 2),
                                                FS_EVEN(aC, aL, 2), FS_EVEN(aR, aC, 2), FS_EVEN(bC, bL, 2), t2);
                const uint32_t fo = quick_pairs(pk_odd(cC), pk_odd(r0), pk_odd(r8), FS_ODD(cR, cC, 3), FS_ODD(cC, cL, 1), FS_ODD(bR, bC, 2),
                                                FS_ODD(aC, aL, 2), FS_ODD(aR, aC, 2), FS_ODD(bC, bL, 2), t2);
#undef FS_EVEN
#undef FS_ODD
                uint32_t ze, zo;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(ze) : "v"(fe), "v"(0x00010001u));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(zo) : "v"(fo), "v"(0x00010001u));
                const uint32_t z = ze | (zo << 1);
                const uint32_t m4 = act ? (z | (z >> 14)) & cmask : 0u;
                if (__ballot(m4 != 0u) == 0ull) continue;   // nothing in these rows passes (the usual case in a cell without a corner at iniTh): no scan, no queue writes
                const int cn = __popc(m4);
                const int incl = wave_incl_scan(cn);
                int pos = qn2 + incl - cn;
                const uint32_t e0 = ((uint32_t)y << 8) | (uint32_t)x4;
                const int tot = __builtin_amdgcn_readlane(incl, 63);
                if (qn2 + tot <= qcap) {   // wave-uniform (a chunk past the queue's end is dropped whole: the cell takes the list pass, below)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (m4 & (1u << k)) pq[pos] = (uint16_t)(e0 + k);
                        pos += (int)(m4 >> k & 1u);
                    }
                }
                qn2 += tot;
            }
            if (qn2 > qcap) {   // this cell's candidates do not fit behind the wave's corners: the cell alone takes the list pass, its entries are dropped
                qn2 = qstart;
                if (lane == 0) atomicOr(reinterpret_cast<uint32_t *>(ovf + 2), 1u << c);
            }
        }
        wave_lds_sync();
        nc = score_compact(nc, qn2, minTh, nc);
        __syncthreads();   // the pixel tile is dead from here on
    }
    const uint32_t listed = needy ? (uint32_t)ovf[2] : 0u;   // cells of the second pass handed to the list pass (bit per cell)
    if (ovf_any) {     // a queue of some wave overflowed: the whole strip takes the second-pass kernel, cell by cell
        if (threadIdx.x < T.ncell) list[atomicAdd(list_count, 1)] = ((uint32_t)f << 16) | (T.cell0 + threadIdx.x);
        return;
    }

    // ---- score tile (aliases the pixel tile): row y + 1, column 1 + x + cell(x); this wave's band + the apron row(s) next to it ----
    uint8_t *sco = pix;
    {
        const int r0 = wave == 0 ? 0 : yb0 + 1, r1 = (wave == W - 1 || yb1 == ih) ? max(ih + 2, yb1 + 1) : yb1 + 1;   // rows [r0, r1)
        uint4 *z = reinterpret_cast<uint4 *>(sco + r0 * P);
        const int n16 = (r1 - r0) * (P / 16);
        for (int i = lane; i < n16; i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (needy) __syncthreads();   // second-pass corners lie in any band: every band zeroed before any wave scatters (workgroup-uniform condition)
    else wave_lds_sync();         // first-pass corners lie in the wave's own band
    for (int e = lane; e < nc; e += 64) {
        const int q = pq[e], x = q & 0xff;
        sco[((q >> 8) + 1) * P + 1 + x + (int)(((uint32_t)x * T.rcp_wcell) >> 16)] = ps[e];
    }
    __syncthreads();

    // ---- NMS: strict '>' against all 8 neighbours ([OCV] FAST_t nonmax stage); survivors compacted in place ----
    int ns = 0;
    for (int e0 = 0; e0 < nc; e0 += 64) {
        const int e = e0 + lane;
        int keep = 0, s = 0, q = 0;
        if (e < nc) {
            q = pq[e];
            s = ps[e];
            const int x = q & 0xff;
            const uint8_t *p = sco + ((q >> 8) + 1) * P + 1 + x + (int)(((uint32_t)x * T.rcp_wcell) >> 16);
            keep = s > max3i(max3i(p[-1], p[1], p[-P - 1]), max3i(p[-P], p[-P + 1], p[P - 1]), max((int)p[P], (int)p[P + 1]));   // strictly above all 8 neighbours
        }
        const unsigned long long b = __ballot(keep != 0);
        __builtin_amdgcn_wave_barrier();
        if (keep) {
            const int o = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, (uint32_t)ns));
            pq[o] = (uint16_t)q;
            ps[o] = (uint8_t)s;
        }
        ns += __popcll(b);
    }
    wave_lds_sync();
    // per-cell counts of this wave's survivors and every survivor's rank inside its cell from ONE pair of wave scans: a survivor adds 1 to the byte field of
    // its cell (cells 0-3 in one dword, 4-7 in a second; a chunk has 64 lanes, so a field never carries), the inclusive scan minus the lane's own field is its
    // rank, lane 63's value holds all counts.  (Rounds 3-5: a ballot per cell in the counting loop and again, with a v_readlane and two mbcnt, in the emission:
    // ~11 vector instructions per cell and survivor chunk, 7 cells.)  Lane c keeps the running count of cell c.
    const int ncell = (int)T.ncell;
    auto cell_fields = [&](const int cellx, uint32_t &lo, uint32_t &hi) {
        const uint32_t one = cellx >= 0 ? 1u << ((cellx & 3) * 8) : 0u;
        lo = cellx < 4 ? one : 0u; hi = cellx >= 4 ? one : 0u;
    };
    auto lane_field = [&](const int il, const int ih) -> int {   // lane c: the chunk's count of cell c
        const uint32_t tl = (uint32_t)__builtin_amdgcn_readlane(il, 63), th = (uint32_t)__builtin_amdgcn_readlane(ih, 63);
        return (int)(((lane < 4 ? tl : th) >> ((lane & 3) * 8)) & 0xffu);
    };
    int mycnt = 0;
    int il0 = 0, ih0 = 0;   // the first chunk's scans, used again by the emission (a wave seldom has more than 64 survivors)
    for (int e0 = 0; e0 < ns; e0 += 64) {
        const int e = e0 + lane;
        const int cellx = e < ns ? (int)(((uint32_t)(pq[e] & 0xff) * T.rcp_wcell) >> 16) : -1;
        uint32_t lo, hi;
        cell_fields(cellx, lo, hi);
        const int il = wave_incl_scan((int)lo), ih = wave_incl_scan((int)hi);
        if (e0 == 0) { il0 = il; ih0 = ih; }
        mycnt += lane_field(il, ih);
    }
    if (lane < kStripMaxCells) cnt[wave * kStripMaxCells + lane] = lane < ncell ? mycnt : 0;
    __syncthreads();

    // ---- emission: slot position = survivors of the cell in earlier waves (earlier rows) + rank inside this wave ----
    uint32_t *slots = cellent + (size_t)f * ent_frame_stride + T.slot0;
    int base = 0;   // lane c: survivors of cell c in the earlier waves, then also this wave's earlier chunks
    if (lane < ncell)
        for (int w = 0; w < wave; w++) base += cnt[w * kStripMaxCells + lane];
    for (int e0 = 0; e0 < ns; e0 += 64) {
        const int e = e0 + lane;
        const int q = e < ns ? (int)pq[e] : 0;
        const int cellx = e < ns ? (int)(((uint32_t)(q & 0xff) * T.rcp_wcell) >> 16) : -1;
        uint32_t lo, hi;
        cell_fields(cellx, lo, hi);
        int il = il0, ih = ih0;
        if (e0 != 0) { il = wave_incl_scan((int)lo); ih = wave_incl_scan((int)hi); }   // wave-uniform
        const uint32_t excl = cellx < 4 ? (uint32_t)il - lo : (uint32_t)ih - hi;
        const int pos = __shfl(base, cellx & 7) + (int)((excl >> ((cellx & 3) * 8)) & 0xffu);
        if (e < ns) slots[(size_t)cellx * T.cell_cap + pos] = pack_key((q & 0xff) + T.ox, (q >> 8) + T.oy, ps[e]);
        if (e0 + 64 < ns) base += lane_field(il, ih);
    }
    if (wave == 0 && lane < ncell) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < W; w++) total += cnt[w * kStripMaxCells + lane];
        // an empty cell is final when FAST(minTh) has been run on it above (or cannot find more: !second_pass); a cell whose corners at iniTh all
        // lost the NMS (equal scores side by side) is empty for the reference as well, which then runs FAST(minTh) on it: the list pass (rare)
        if (!((listed >> lane) & 1u) && (total > 0 || !second_pass || ((needy >> lane) & 1u))) cellcnt[(size_t)f * total_cells + T.cell0 + lane] = total;
        else list[atomicAdd(list_count, 1)] = ((uint32_t)f << 16) | (T.cell0 + lane);
    }
}

template <int W>
__global__ __launch_bounds__(64 * W) void k_fast_strip(const StripTile *__restrict__ tiles, const uint8_t *__restrict__ pyr,
                                                       size_t pyr_frame_stride, int32_t *__restrict__ cellcnt, int total_cells,
                                                       uint32_t *__restrict__ cellent, size_t ent_frame_stride, int iniTh, int minTh,
                                                       int pix_bytes, int gcap, int qcap, uint32_t *__restrict__ list,
                                                       int32_t *__restrict__ list_count, int second_pass, int n_frames, const Level0Src src0, int n_strips0) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int tile, f;
    if (!xcd_frame_map(n_frames, &tile, &f)) return;   // a frame's strips stay on one XCD (apron rows hit its L2)
    const StripTile T = tiles[tile];
    // level 0 in place (src0.img != nullptr): its strips -- the first n_strips0 of a frame -- read the caller's frame; FAST never looks at the ring
    // (the detection window starts 16 px inside the image, ORBextractor.cc:789-792)
    const bool in_place = src0.img != nullptr && tile < n_strips0;   // workgroup-uniform
    const uint8_t *tile_src = in_place ? src0.img + (size_t)f * src0.frame_stride + (size_t)(kBorder + T.oy - 3) * src0.row_stride + (kBorder + T.ox - 4)
                                       : pyr + (size_t)f * pyr_frame_stride + T.src_off;
    const int spitch = in_place ? (int)src0.row_stride : T.pitch;
#define ORBX_STRIP_BODY(PITCH) fast_strip_body<W, PITCH>(T, f, smem, tile_src, spitch, cellcnt, total_cells, cellent, ent_frame_stride, iniTh, minTh, pix_bytes, gcap, qcap, list, list_count, second_pass)
    if (T.lds_pitch == (uint32_t)kStripPitch) ORBX_STRIP_BODY(kStripPitch);          // wave-uniform
    else if (T.lds_pitch == (uint32_t)kStripPitchMid) ORBX_STRIP_BODY(kStripPitchMid);
    else ORBX_STRIP_BODY(kStripPitchLow);
#undef ORBX_STRIP_BODY
}

}  // namespace orbx
