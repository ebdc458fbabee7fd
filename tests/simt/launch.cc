// launch.cc -- grid loop + dynamic LDS of the SIMT emulator (one definition per emulated library)
#include "hip/hip_runtime.h"
namespace simt {
static uint8_t *g_lds = nullptr;
static uint8_t g_anchor[16];
uintptr_t bss_anchor() { return (uintptr_t)g_anchor; }   // static LDS arrays of the kernels live in this library's .bss too
uint8_t *dyn_lds() {
    if (!g_lds) g_lds = (uint8_t *)aligned_alloc(4096, 256 * 1024);
    return g_lds;
}
void launch(const char *name, dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body) {
    static const bool trace = getenv("SIMT_TRACE") != nullptr;
    if (trace) fprintf(stderr, "simt: %s grid (%u, %u, %u) block %u lds %zu\n", name, grid.x, grid.y, grid.z, block.x, lds_bytes);
    if (lds_bytes > 200 * 1024) { fprintf(stderr, "simt: %zu bytes of dynamic LDS\n", lds_bytes); abort(); }
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads > 1024) { fprintf(stderr, "simt: %d threads per block\n", nthreads); abort(); }
    Dim3 g; g.x = grid.x; g.y = grid.y; g.z = grid.z;
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                Dim3 b; b.x = x; b.y = y; b.z = z;
                // LDS is not zero on entry: a fixed pattern, or (SIMT_LDS_RANDOM=<seed>) different garbage for every block -- a result
                // that changes with it depends on LDS the kernel never wrote
                static const char *rnd = getenv("SIMT_LDS_RANDOM");
                if (rnd) {
                    static unsigned long long st = strtoull(rnd, nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
                    uint64_t *p = (uint64_t *)dyn_lds();
                    for (size_t i = 0; i < (lds_bytes + 7) / 8; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; p[i] = st; }
                } else memset(dyn_lds(), 0xcd, lds_bytes);
                run_block(g, b, nthreads, body);
            }
}
}  // namespace simt
