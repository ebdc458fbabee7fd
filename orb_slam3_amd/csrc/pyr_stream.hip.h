// pyr_stream.hip.h -- k_pyr_stream: levels 1 .. n-1 of the image pyramid (ORBextractor::ComputePyramid, /root/reference/src/ORBextractor.cc:1170-1195:
// cv::resize(INTER_LINEAR) of the level before + copyMakeBorder(REFLECT_101)) of a batch of frames in ONE launch, no hand-off between workgroups.
//
// Rounds 2-4 ran one launch per level (k_pyr_resize_march x 7): every level was written to HBM and read back by the next launch, each launch
// lasted as long as a wave's two dependent memory round trips, and seven drains sat on the main stream's critical path (252 us of a 944-us
// step for 657 MB: 0.33 of the HBM roofline).  Here a workgroup owns a BAND of a frame (the whole frame, or a half / quarter of its rows) and
// streams down it through all levels at once:
//   * the frame's rows enter an LDS ring ("level 0") sixteen bytes per thread, requested at the start of a step and stored at its end;
//   * level l keeps its most recent rows in an LDS ring; in step s it produces the rows whose two source rows of level l - 1 were complete
//     after step s - 1 (all levels advance in the same step: ONE barrier per step, no barrier between levels);
//   * a row is computed once, stored once to the padded slab in HBM (plus its REFLECT_101 copies in the 19-row ring above / below) and once to
//     the LDS ring the next level reads; nothing is read back from HBM.
// The row schedule is a per-geometry table built on the host (build_pyr_stream, orbx_extractor.hip): the kernel holds no index arithmetic beyond
// "task -> LDS offsets".  A task = one wave x 64 dword columns x one or two output rows of one level (two rows share the horizontal pass of
// their common source row).  Bands of one frame overlap by the few rows the cascade needs (9 of 480 rows for two bands of a 752x480 frame);
// rows in the overlap are computed by both bands and stored by the one that owns them.
// Arithmetic = k_pyr_resize_march's ([OCV] resize INTER_LINEAR 8U: Q11 taps, horizontal sums >> 4, (b * H) >> 16 per source row, + 2 >> 2), bit for bit.
// grid xcd_grid(bands per frame, B), block 512, dynamic LDS = PyrStreamGeom::lds_bytes
#pragma once

namespace orbx {

// horizontal pass of one source row for the four pixels of a dword column: sums >> 4 ([OCV] the vertical pass multiplies (sum >> 4))
__device__ __forceinline__ void pyr_hpass(const uint8_t *row, const uint32_t sel, const uint32_t selr, const uint32_t (&cc)[4], uint32_t (&H)[4]) {
    uint2 v;
    __builtin_memcpy(&v, row, 8);   // unaligned 8-byte LDS read (ds_read_b64, unaligned access mode)
    constexpr uint32_t kPair[4] = {0x0c040c00u, 0x0c050c01u, 0x0c060c02u, 0x0c070c03u};  // (left tap k, right tap k) as two u16
    const uint32_t l = __builtin_amdgcn_perm(v.y, v.x, sel), q = __builtin_amdgcn_perm(v.y, v.x, selr);
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

__global__ __launch_bounds__(kPyrStreamThreads) void k_pyr_stream(const PyrStreamGeom G, const PyrStreamLevel *__restrict__ levels, const uint4 *__restrict__ xg24, const PyrStep *__restrict__ steps,
                                                                   const PyrTask *__restrict__ tasks, const uint32_t *__restrict__ band_task0,
                                                                   const uint8_t *__restrict__ img, size_t row_stride, size_t frame_stride,
                                                                   uint8_t *__restrict__ pyr, size_t pyr_frame_stride, int32_t *__restrict__ zero_word, int n_frames) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int band, f;
    if (!xcd_frame_map(n_frames, &band, &f)) return;   // the bands of a frame stay on one XCD (their overlap rows hit its L2)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = kPyrStreamThreads / 64;
    if (zero_word && band == 0 && f == 0 && tid == 0) *zero_word = 0;   // the FAST stage's overflow counter of this batch (k_pyr_base's side job)
    for (uint32_t i = (uint32_t)tid; i < G.xg_bytes / 16u; i += kPyrStreamThreads) reinterpret_cast<uint4 *>(smem)[i] = xg24[i];
    const uint8_t *frame = img + (size_t)f * frame_stride;
    uint8_t *slab = pyr + (size_t)f * pyr_frame_stride;
    const PyrStep *st = steps + (size_t)band * G.steps_per_band;
    const PyrTask *tk = tasks + band_task0[band];
    // frame row chunk of this thread: chunk i of a step = (row i / cpr0, 16-byte chunk i % cpr0); the last chunk of a row is pulled back to end at
    // the row's last pixel (no read past the frame's last row)
    uint32_t srow[kPyrStreamStage], sx[kPyrStreamStage];
#pragma unroll
    for (int j = 0; j < kPyrStreamStage; j++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)j * kPyrStreamThreads;
        srow[j] = __umulhi(i, G.cpr0_rcp);
        sx[j] = min((i - srow[j] * G.cpr0) * 16u, (uint32_t)G.w0 - 16u);
    }
    __syncthreads();
    for (uint32_t s = 0; s < G.steps_per_band; s++) {
        const PyrStep d = st[s];
        const uint32_t y0 = d.y0_rows & 0xffffu, nrows = d.y0_rows >> 16;
        uint4 stg[kPyrStreamStage];
#pragma unroll
        for (int j = 0; j < kPyrStreamStage; j++) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if ((uint32_t)j * kPyrStreamThreads < nrows * G.cpr0) {   // wave-uniform
                const uint32_t r = min(srow[j], nrows - 1u);
                __builtin_memcpy(&v, frame + (size_t)(y0 + r) * row_stride + sx[j], 16);
            }
            stg[j] = v;
        }
        PyrTask T = tk[min(d.task_begin + (uint32_t)wave, d.task_end - (d.task_end > d.task_begin ? 1u : 0u))];
        for (uint32_t t = d.task_begin + (uint32_t)wave; t < d.task_end; t += NW) {
            const PyrTask C = T;
            T = tk[min(t + NW, d.task_end - 1u)];   // the next task's descriptor is on its way while this one is computed
            const uint32_t lvl = C.hdr & 15u, chunk = (C.hdr >> 4) & 15u, two = (C.hdr >> 8) & 1u, nsrc = (C.hdr >> 9) & 7u;
            const PyrStreamLevel L = levels[lvl];   // wave-uniform index: scalar loads (indexing the kernel argument itself would copy it to scratch)
            const uint32_t col = chunk * 64u + (uint32_t)lane;
            const bool live = col < L.ncol;
            const uint8_t *e = smem + L.xg_lds + min(col, L.ncol - 1u) * 24u;
            const uint2 e0 = *reinterpret_cast<const uint2 *>(e);
            uint32_t cc[4];
            { uint2 a, b; a = *reinterpret_cast<const uint2 *>(e + 8); b = *reinterpret_cast<const uint2 *>(e + 16); cc[0] = a.x; cc[1] = a.y; cc[2] = b.x; cc[3] = b.y; }
            const uint32_t base = e0.x & 0xffffu, valid = e0.x >> 30, sel = e0.y, selr = e0.y + 0x01010101u;
            uint32_t H0[4], H1[4], H2[4], H3[4];
            pyr_hpass(smem + (uint32_t)C.src[0] * 16u + base, sel, selr, cc, H0);
            pyr_hpass(smem + (uint32_t)C.src[1] * 16u + base, sel, selr, cc, H1);
            uint32_t o0 = pyr_vpass(H0, H1, C.b[0]), o1 = 0u;
            if (two) {   // wave-uniform: three source rows (the middle one shared) or four (two independent pairs)
                pyr_hpass(smem + (uint32_t)C.src[2] * 16u + base, sel, selr, cc, H2);
                if (nsrc == 4u) {
                    pyr_hpass(smem + (uint32_t)C.src[3] * 16u + base, sel, selr, cc, H3);
                    o1 = pyr_vpass(H2, H3, C.b[1]);
                } else {
                    o1 = pyr_vpass(H1, H2, C.b[1]);
                }
            }
            if (valid != 1u) { o0 = 0u; o1 = 0u; }   // a dword past the ring's last pixel
            const uint32_t dcol = L.col0 + col;                  // dword column inside the padded row
            const uint32_t roi = dcol - (uint32_t)(kRoiX / 4);   // ... inside the ROI row (wraps for ring columns)
            uint8_t *gcol = slab + (((uint64_t)L.off_hi << 32) | L.off_lo) + 4u * dcol;
            const int h = L.h;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (k == 1 && !two) break;
                const uint32_t o = k ? o1 : o0;
                const int r = (int)C.row + k;
                if (live && ((C.hdr >> (12 + k)) & 1u)) {
                    *reinterpret_cast<uint32_t *>(gcol + (size_t)(uint32_t)((kEdge + r) * (int)L.pitch)) = o;
                    if (r >= 1 && r <= kEdge) *reinterpret_cast<uint32_t *>(gcol + (size_t)(uint32_t)((kEdge - r) * (int)L.pitch)) = o;                            // ring above
                    if (r <= h - 2 && r >= h - 1 - kEdge) *reinterpret_cast<uint32_t *>(gcol + (size_t)(uint32_t)((kEdge + 2 * (h - 1) - r) * (int)L.pitch)) = o;   // ring below
                }
                if (live && C.dst[k] != 0xffffu && roi < L.roi_dw) *reinterpret_cast<uint32_t *>(smem + (uint32_t)C.dst[k] * 16u + 4u * roi) = o;
            }
        }
        // the frame rows requested at the top of the step
#pragma unroll
        for (int j = 0; j < kPyrStreamStage; j++)
            if ((uint32_t)j * kPyrStreamThreads < nrows * G.cpr0 && srow[j] < nrows) {
                uint32_t slot = d.slot0 + srow[j];
                if (slot >= G.ring0_rows) slot -= G.ring0_rows;
                const uint4 v = stg[j];
                __builtin_memcpy(smem + G.ring0_off + slot * G.ring0_pitch + sx[j], &v, 16);
            }
        __syncthreads();
    }
}

}  // namespace orbx
