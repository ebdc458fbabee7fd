// orbx_internal.h -- shared host/device definitions of liborbx (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orbx.h"

namespace orbx {

constexpr int kEdge = 19;       // EDGE_THRESHOLD   (ORBextractor.cc:73)
constexpr int kHalfPatch = 15;  // HALF_PATCH_SIZE  (ORBextractor.cc:72)
constexpr int kPatch = 31;      // PATCH_SIZE       (ORBextractor.cc:71)
constexpr int kBorder = 16;     // minBorder = EDGE_THRESHOLD-3 (ORBextractor.cc:789)
constexpr int kRoiX = 64;       // byte offset of ROI column 0 inside a padded row (64-B aligned ROI rows)
constexpr int kRingX = kRoiX - kEdge;  // byte offset of padded column 0
constexpr int kMaxLevels = 16;
constexpr int kMaxDim = 4095;   // x,y packed in 12 bits each

// Packed candidate / keypoint key: x | y << 12 | score << 24
__host__ __device__ inline uint32_t pack_key(int x, int y, int s) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)s << 24); }
__host__ __device__ inline int key_x(uint32_t k) { return (int)(k & 0xfffu); }
__host__ __device__ inline int key_y(uint32_t k) { return (int)((k >> 12) & 0xfffu); }
__host__ __device__ inline int key_s(uint32_t k) { return (int)(k >> 24); }

// Per-level constants for one (width, height) configuration.
struct LevelInfo {
    int32_t w, h;            // ROI size of the level
    int32_t pitch;           // bytes per padded row
    int32_t bpitch;          // bytes per blurred row
    uint64_t off;            // byte offset of the level (row 0 of the padded image) inside a frame's pyramid slab
    uint64_t boff;           // byte offset inside a frame's blur slab
    int32_t nCols, nRows, wCell, hCell;  // FAST grid (ORBextractor.cc:797-803)
    int32_t cell_base;       // first cell index of the level inside a frame's cell arrays
    int32_t cell_cap;        // capacity (entries) of one cell slot
    uint32_t cand_off;       // entry offset of the level inside a frame's candidate slabs
    uint32_t cand_cap;       // nCols*nRows*cell_cap
    int32_t quota;           // mnFeaturesPerLevel[level]
    int32_t nIni;            // number of quad-tree roots (ORBextractor.cc:559)
    float hX;                // root width (ORBextractor.cc:560)
    int32_t lvl_cap;         // capacity of the per-level selected keypoint list
    uint32_t lvl_off;        // entry offset of the level in the per-frame selected-keypoint slab
    float scale;             // mvScaleFactor[level]
    float size;              // (float)(int)(31*scale)
    uint32_t xtab_off, ytab_off;  // offsets into the resize tables
    uint32_t xg_off;         // offset into the ResizeGroup table (pitch/4 entries per level)
    int32_t pool;            // quad-tree node pool size
};

// Bilinear resize table entry (cv::resize INTER_LINEAR 8U): source offset and Q11 coefficients
struct ResizeTap {
    int32_t ofs;
    int16_t c0, c1;
};

// Four horizontally adjacent output pixels (one dword of a padded row) of the bilinear resize: all eight source bytes of a
// row lie inside [base, base+8), sel holds the byte offset of each pixel's left tap (v_perm_b32 selector), cc the Q11
// coefficient pairs (c0 | c1 << 16; 0 for pixels outside the ring).  valid = 0: offsets do not fit (scale factor > 2);
// valid = 2: the whole dword lies outside the ring (pitch padding), the kernel stores zeros.
struct ResizeGroup {
    int32_t base;
    uint32_t sel;
    uint32_t valid;
    uint32_t pad;
    uint32_t cc[4];
};

// ---- k_pyr_stream (pyr_stream.hip.h): per-geometry tables built by build_pyr_stream (orbx_extractor.hip) ----
constexpr int kPyrStreamStage = 8;      // 16-byte chunks of frame rows a lane of the loader wave may have in flight per step

struct PyrColumn {        // one dword column of a padded row (ResizeGroup without its padding: 24 bytes, LDS resident)
    uint32_t base_valid;  // first source byte of the column's taps (ROI x of the level before) | valid << 30 (1: taps, otherwise zeros)
    uint32_t sel;         // byte offset of each pixel's left tap inside the 8 source bytes (v_perm_b32 selector)
    uint32_t cc[4];       // Q11 tap pairs c0 | c1 << 16
};
static_assert(sizeof(PyrColumn) == 24, "PyrColumn layout");

struct PyrStreamLevel {
    uint32_t xg_lds;      // byte offset of the level's PyrColumn table inside LDS
    uint32_t col0, ncol;  // first dword column of a padded row that holds ring / ROI pixels, number of such columns
    uint32_t pitch;       // bytes per padded row in the slab
    uint32_t off_lo, off_hi;   // byte offset of the level inside a frame's pyramid slab
    int32_t h;            // ROI rows
    uint32_t roi_dw;      // dwords of an ROI row kept in the LDS ring: ceil(w / 4)
};

struct PyrStreamGeom {    // kernel argument (scalar loads)
    uint32_t xg_bytes;        // bytes of the PyrColumn tables (multiple of 16)
    uint32_t steps_per_band;
    uint32_t workers;         // waves of a workgroup that compute (one more stages the frame rows)
    uint32_t cpr0, cpr0_rcp;  // 16-byte chunks per frame row, ceil(2^32 / cpr0)
    uint32_t ring0_off, ring0_pitch, ring0_rows;   // LDS ring of frame rows
    int32_t w0;
};

struct PyrStep {          // 16 bytes, one scalar load
    uint32_t task_begin, task_end;   // indices into the band's task list
    uint32_t y0_rows;         // first frame row to stage | number of rows << 16
    uint32_t slot0;           // ring slot of that row (the following rows take the following slots, wrapping at ring0_rows)
};

struct PyrTask {          // 48 bytes, scalar loads; everything a wave needs to know about its 64 dword columns of one or two output rows
    uint32_t hdr;         // (rows - 1) | source rows << 1 | live lanes (1 .. 64) << 4 | first lane with an ROI dword << 11 | lanes with an ROI dword << 18
    uint32_t src01, src23;   // LDS byte offsets / 16 of the source rows (ROI byte 0), consecutive rows of the level before: row 0 | row 1 << 16, row 2 | row 3 << 16
                          // (every field a whole dword: a 16-bit field would be fetched with a VECTOR load, whose wait also waits for the stores before it)
    uint32_t b[2];        // vertical tap pair c0 | c1 << 16 of each output row
    uint32_t goff[2];     // byte offset inside a frame's pyramid slab of lane 0's dword of each output row; 0xffffffff: another band stores this row
    uint32_t moff[2];     // ... of its REFLECT_101 copy in the 19-row ring above / below the level; 0xffffffff: none
    uint32_t dlds;        // LDS byte offset / 4 of the first ROI dword of each output row in this level's ring (row 0 | row 1 << 16); 0xffff: the last level keeps none
    uint32_t xg;          // LDS byte offset of lane 0's PyrColumn
    uint32_t pad;
};
static_assert(sizeof(PyrTask) == 48 && sizeof(PyrStep) == 16, "task table layout");

struct TileRef {  // blockIdx.x -> (level, tile) mapping for multi-level launches
    int16_t level;
    int16_t ti, tj;
    int16_t pad;
};


struct WorkItem {  // one selected keypoint of a frame, in level order, with the constants of its level (k_describe needs no LevelInfo load)
    uint32_t key;     // level coords x,y + score
    int32_t level;
    int32_t pos;      // output slot (lapping split, ORBextractor.cc:1153-1162)
    uint32_t pitches; // pitch | bpitch << 16 (bytes per padded / blurred row)
    uint32_t off;     // byte offset of the level inside a frame's pyramid slab
    uint32_t boff;    // ... inside a frame's blur slab
    float scale;      // mvScaleFactor[level]
    float size;       // keypoint size of the level
};
static_assert(sizeof(WorkItem) == 32, "WorkItem layout");

}  // namespace orbx
