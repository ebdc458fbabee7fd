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
