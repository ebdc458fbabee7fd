// pyr_stream.hip.h -- k_pyr_stream: levels 1 .. n-1 of the image pyramid (ORBextractor::ComputePyramid, /root/reference/src/ORBextractor.cc:1170-1195:
// cv::resize(INTER_LINEAR) of the level before + copyMakeBorder(REFLECT_101)) of a batch of frames in ONE launch, no hand-off between workgroups.
//
// Rounds 2-4 ran one launch per level (k_pyr_resize_march x 7): every level was written to HBM and read back by the next launch, each launch
// lasted as long as a wave's two dependent memory round trips, and seven drains sat on the main stream's critical path (252 us of a 944-us
// step for 657 MB: 0.33 of the HBM roofline).  Here a workgroup owns a BAND of a frame (the whole frame, or a half / quarter of its rows) and
// streams down it through all levels at once:
//   * the frame's rows enter an LDS ring ("level 0") through a ninth wave that does nothing else (sixteen bytes per lane and chunk);
//   * level l keeps its most recent rows in an LDS ring; in step s it produces the rows whose two source rows of level l - 1 were complete
//     after step s - 1 (all levels advance in the same step: ONE barrier per step, no barrier between levels);
//   * a row is computed once, stored once to the padded slab in HBM (plus its REFLECT_101 copies in the 19-row ring above / below) and once to
//     the LDS ring the next level reads; nothing is read back from HBM.
// The row schedule is a per-geometry table built on the host (build_pyr_stream, orbx_extractor.hip): the kernel holds no index arithmetic beyond
// "task -> LDS offsets".  A task = one wave x 64 dword columns x one or two output rows of one level (two rows share the horizontal pass of
// their common source row).  Bands of one frame overlap by the few rows the cascade needs (9 of 480 rows for two bands of a 752x480 frame);
// rows in the overlap are computed by both bands and stored by the one that owns them.
// Arithmetic = k_pyr_resize_march's ([OCV] resize INTER_LINEAR 8U: Q11 taps, horizontal sums >> 4, (b * H) >> 16 per source row, + 2 >> 2), bit for bit.
// grid xcd_grid(bands per frame, B), block 64 x (worker waves + 1: the loader wave), dynamic LDS = PyrStreamGeom::lds_bytes
#pragma once

namespace orbx {

// horizontal pass of one source row for the four pixels of a dword column: sums >> 4 ([OCV] the vertical pass multiplies (sum >> 4)).
// The column's 8 source bytes start at any byte of the row: three ALIGNED dwords + two v_alignbyte (measured: an 8-byte LDS read at an odd address
// costs the kernel 66 of 228 us -- the LDS serves it in pieces).
__device__ __forceinline__ void pyr_hpass(const uint8_t *row4, const uint32_t osh, const uint32_t sel, const uint32_t selr, const uint32_t (&cc)[4], uint32_t (&H)[4]) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(row4);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    const uint32_t vx = __builtin_amdgcn_alignbyte(d1, d0, osh), vy = __builtin_amdgcn_alignbyte(d2, d1, osh);
    constexpr uint32_t kPair[4] = {0x0c040c00u, 0x0c050c01u, 0x0c060c02u, 0x0c070c03u};  // (left tap k, right tap k) as two u16
    const uint32_t l = __builtin_amdgcn_perm(vy, vx, sel), q = __builtin_amdgcn_perm(vy, vx, selr);
#pragma unroll
    for (int j = 0; j < 4; j++) H[j] = __builtin_amdgcn_udot2(as_pk(__builtin_amdgcn_perm(q, l, kPair[j])), as_pk(cc[j]), 0u, false) >> 4;
}
// vertical pass: (b0 * A >> 16) + (b1 * B >> 16) + 2 >> 2 per pixel, four pixels packed (see k_pyr_resize_march)
__device__ __forceinline__ uint32_t pyr_vpass(const uint32_t (&A)[4], const uint32_t (&B)[4], const uint32_t bb) {
    const uint32_t b0 = bb & 0xffffu, b1 = bb >> 16;
    int t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t p0 = __umul24(A[j], b0) + 0x20000u, p1 = __umul24(B[j], b1);
        t[j] = (int)((p0 >> 16) + (p1 >> 16));
    }
    return ((uint32_t)(uint16_t)__builtin_amdgcn_ashr_pk_u8_i32(t[0], t[1], 2)) | ((uint32_t)(uint16_t)__builtin_amdgcn_ashr_pk_u8_i32(t[2], t[3], 2) << 16);
}

__global__ __launch_bounds__(1024) void k_pyr_stream(const PyrStreamGeom G, const uint4 *__restrict__ xg24, const PyrStep *__restrict__ steps,
                                                                   const PyrTask *__restrict__ tasks, const uint32_t *__restrict__ band_task0,
                                                                   const uint8_t *__restrict__ img, size_t row_stride, size_t frame_stride,
                                                                   uint8_t *__restrict__ pyr, size_t pyr_frame_stride, int32_t *__restrict__ zero_word, int n_frames) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int band, f;
    if (!xcd_frame_map(n_frames, &band, &f)) return;   // the bands of a frame stay on one XCD (their overlap rows hit its L2)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = (int)G.workers;   // worker waves (the block has one wave more: the loader)
    if (zero_word && band == 0 && f == 0 && tid == 0) *zero_word = 0;   // the FAST stage's overflow counter of this batch (k_pyr_base's side job)
    for (uint32_t i = (uint32_t)tid; i < G.xg_bytes / 16u; i += 64u * (G.workers + 1u)) reinterpret_cast<uint4 *>(smem)[i] = xg24[i];
    const PyrStep *st = steps + (size_t)band * G.steps_per_band;
    __syncthreads();
    if (wave == NW) {
        // ---- the loader wave: the frame rows of every step into the LDS ring, 16 bytes per lane and chunk.  A wave of its own because a wave's
        // memory operations retire in order: a worker that had requested frame rows would have to wait for every store it issued after them
        // (s_waitcnt vmcnt counts both) before it could hand the rows to LDS -- the workers below never wait for memory at all.
        const uint8_t *frame = img + (size_t)f * frame_stride;
        // chunk i of a step = (row i / cpr0, 16-byte chunk i % cpr0); the last chunk of a row is pulled back to end at the row's last pixel
        for (uint32_t s = 0; s < G.steps_per_band; s++) {
            const PyrStep d = st[s];
            const uint32_t y0 = d.y0_rows & 0xffffu, nrows = d.y0_rows >> 16;
            uint4 stg[kPyrStreamStage];
#pragma unroll
            for (int j = 0; j < kPyrStreamStage; j++) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (64u * (uint32_t)j < nrows * G.cpr0) {   // wave-uniform
                    const uint32_t i = (uint32_t)lane + 64u * (uint32_t)j, row = __umulhi(i, G.cpr0_rcp);
                    const uint32_t x = min((i - row * G.cpr0) * 16u, (uint32_t)G.w0 - 16u);
                    __builtin_memcpy(&v, frame + (size_t)(y0 + min(row, nrows - 1u)) * row_stride + x, 16);
                }
                stg[j] = v;
            }
#pragma unroll
            for (int j = 0; j < kPyrStreamStage; j++) {
                const uint32_t i = (uint32_t)lane + 64u * (uint32_t)j, row = __umulhi(i, G.cpr0_rcp);
                if (64u * (uint32_t)j < nrows * G.cpr0 && row < nrows) {
                    const uint32_t x = min((i - row * G.cpr0) * 16u, (uint32_t)G.w0 - 16u);
                    uint32_t slot = d.slot0 + row;
                    if (slot >= G.ring0_rows) slot -= G.ring0_rows;
                    const uint4 v = stg[j];
                    __builtin_memcpy(smem + G.ring0_off + slot * G.ring0_pitch + x, &v, 16);
                }
            }
            __syncthreads();
        }
        return;
    }
    // ---- the worker waves: the step's tasks, wave w takes tasks w, w + NW, ... ----
    uint8_t *slab = pyr + (size_t)f * pyr_frame_stride;
    const PyrTask *tk = tasks + band_task0[band];
    PyrStep d = st[0];
    const uint32_t last_task = band_task0[band + 1] - band_task0[band] - 1u;   // a step without tasks points one past its predecessor's: never fetch beyond the band's list
    PyrTask T = tk[min(min(d.task_begin + (uint32_t)wave, d.task_end - (d.task_end > d.task_begin ? 1u : 0u)), last_task)];
    for (uint32_t s = 0; s < G.steps_per_band; s++) {
        // the next step's descriptor and this wave's first task of it are requested now and waited for after the barrier: a step does not start
        // with two dependent scalar-memory round trips
        const PyrStep dn = st[min(s + 1u, G.steps_per_band - 1u)];
        for (uint32_t t = d.task_begin + (uint32_t)wave; t < d.task_end; t += NW) {
            const PyrTask C = T;
            // the next task's descriptor is on its way while this one is computed (after the step's last one: the first of the next step)
            T = tk[t + NW < d.task_end ? t + NW : min(min(dn.task_begin + (uint32_t)wave, dn.task_end - (dn.task_end > dn.task_begin ? 1u : 0u)), last_task)];
            const uint32_t two = C.hdr & 1u, nsrc = (C.hdr >> 1) & 7u, nlive = (C.hdr >> 4) & 127u, roi_lo = (C.hdr >> 11) & 127u, roi_n = (C.hdr >> 18) & 127u;
            const bool live = (uint32_t)lane < nlive;
            const uint8_t *e = smem + C.xg + (uint32_t)min(lane, (int)nlive - 1) * 24u;
            const uint2 e0 = *reinterpret_cast<const uint2 *>(e);
            uint32_t cc[4];
            { uint2 a, b; a = *reinterpret_cast<const uint2 *>(e + 8); b = *reinterpret_cast<const uint2 *>(e + 16); cc[0] = a.x; cc[1] = a.y; cc[2] = b.x; cc[3] = b.y; }
            const uint32_t base = e0.x & 0xfffcu, osh = e0.x & 3u, valid = e0.x >> 30, sel = e0.y, selr = e0.y + 0x01010101u;
            uint32_t H0[4], H1[4], H2[4], H3[4];
            pyr_hpass(smem + (C.src01 & 0xffffu) * 16u + base, osh, sel, selr, cc, H0);
            pyr_hpass(smem + (C.src01 >> 16) * 16u + base, osh, sel, selr, cc, H1);
            uint32_t o0 = pyr_vpass(H0, H1, C.b[0]), o1 = 0u;
            if (two) {   // wave-uniform: three source rows (the middle one shared) or four (two independent pairs)
                pyr_hpass(smem + (C.src23 & 0xffffu) * 16u + base, osh, sel, selr, cc, H2);
                if (nsrc == 4u) {
                    pyr_hpass(smem + (C.src23 >> 16) * 16u + base, osh, sel, selr, cc, H3);
                    o1 = pyr_vpass(H2, H3, C.b[1]);
                } else {
                    o1 = pyr_vpass(H1, H2, C.b[1]);
                }
            }
            if (valid != 1u) { o0 = 0u; o1 = 0u; }   // a dword past the ring's last pixel
            uint8_t *gl = slab + 4u * (uint32_t)lane;
            const bool in_roi = (uint32_t)lane - roi_lo < roi_n;   // this lane's dword belongs to the ROI row the next level reads
            uint8_t *ll = smem + 4u * ((uint32_t)lane - roi_lo);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (k == 1 && !two) break;
                const uint32_t o = k ? o1 : o0;
                if (live) {
                    if (C.goff[k] != 0xffffffffu) *reinterpret_cast<uint32_t *>(gl + C.goff[k]) = o;   // wave-uniform conditions
                    if (C.moff[k] != 0xffffffffu) *reinterpret_cast<uint32_t *>(gl + C.moff[k]) = o;
                }
                const uint32_t dl = k ? C.dlds >> 16 : C.dlds & 0xffffu;
                if (in_roi && dl != 0xffffu) *reinterpret_cast<uint32_t *>(ll + dl * 4u) = o;
            }
        }
        if (d.task_begin + (uint32_t)wave >= d.task_end)   // no task of this step for this wave: nothing has fetched the next step's first one
            T = tk[min(min(dn.task_begin + (uint32_t)wave, dn.task_end - (dn.task_end > dn.task_begin ? 1u : 0u)), last_task)];
        d = dn;
        __syncthreads();
    }
}

}  // namespace orbx
