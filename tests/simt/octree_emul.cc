// octree_emul.cc -- liborbx's quad-tree kernels (orb_slam3_amd/csrc/octree.hip.h, octree_par.hip.h: the DEVICE code, unmodified)
// compiled for the CPU SIMT emulator and exported through a C ABI for tests/test_simt_octree.py.  Test infrastructure only.
#include "hip/hip_runtime.h"

namespace orbx {

// wave_incl_scan of extractor_kernels.hip.h (k_compact and the quad-tree body use it): inclusive prefix sum over the 64 lanes
template <int CTRL, int ROWS>
inline int dpp_add(int acc, int v) { return acc + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, false); }
inline int wave_incl_scan(int v) {
    v = dpp_add<0x111, 0xf>(v, v);
    v = dpp_add<0x112, 0xf>(v, v);
    v = dpp_add<0x114, 0xf>(v, v);
    v = dpp_add<0x118, 0xf>(v, v);
    v = dpp_add<0x142, 0xa>(v, v);
    v = dpp_add<0x143, 0xc>(v, v);
    return v;
}
}  // namespace orbx

#include "octree_par.hip.h"   // the build step's copy (tests/simt/build/)

using namespace orbx;

static LevelInfo make_level(int w, int h, int quota) {
    LevelInfo L;
    memset(&L, 0, sizeof(L));
    L.w = w; L.h = h; L.quota = quota;
    L.nIni = (int)std::round((float)(w - 2 * kBorder) / (float)(h - 2 * kBorder));   // configure(), orbx_extractor.hip
    L.hX = (float)(w - 2 * kBorder) / L.nIni;
    L.lvl_cap = std::max(L.quota + 4, 4 * L.nIni) + 4;
    L.pool = std::max(L.quota, 4 * L.nIni) + 16;
    return L;
}

extern "C" {

// form: 0 = k_octree_par body (256 threads, keys in LDS), 1 = the single-wave chunked form (k_octree_par1), 2 = k_octree (sequential
// emulation).  keys: C packed candidates (x | y << 12 | score << 24, window coordinates) in vToDistributeKeys order.
// out: selected keys in list order (capacity lvl_cap), returns their number, -1 on a device-side error code.
int simt_octree(int form, int w, int h, int quota, const uint32_t *keys, int C, uint32_t *out, int out_cap, int *err_out) {
    LevelInfo L = make_level(w, h, quota);
    if (L.lvl_cap > out_cap) return -2;
    std::vector<uint32_t> k0(std::max(C, 1) + 16384), k1(std::max(C, 1) + 16384), res(L.lvl_cap + 4096, 0);
    std::vector<uint16_t> n0(std::max(C, 1) + 16384), n1(std::max(C, 1) + 16384);
    memcpy(k1.data(), keys, 4 * (size_t)C);
    int32_t cnt = -1, err = 0;
    const int max_pool = L.pool;
    simt::Dim3 grid, bidx;
    if (form == 0) {
        if (oct_par_lds_bytes(max_pool) > 200 * 1024) return -3;
        simt::run_block(grid, bidx, 256, [&] {
            octree_par_body<true>(L, simt::dyn_lds(), max_pool, C, nullptr, k1.data(), nullptr, nullptr, res.data(), &cnt, &err, nullptr);
        });
    } else if (form == 1) {
        simt::run_block(grid, bidx, 64, [&] {
            octree_par_body<false>(L, simt::dyn_lds(), max_pool, C, k0.data(), k1.data(), n0.data(), n1.data(), res.data(), &cnt, &err, nullptr);
        });
    } else {
        return -4;
    }
    if (err_out) *err_out = err;
    if (cnt < 0 || cnt > L.lvl_cap) return -1;
    memcpy(out, res.data(), 4 * (size_t)cnt);
    return cnt;
}

}  // extern "C"
