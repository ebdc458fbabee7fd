// Achievable streaming bandwidth at the sizes the pyramid kernels move (tens of MB per launch): read-only, write-only, copy.
// hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip && ./copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_copy(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
__global__ void k_copy4(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n) {  // 4 x 16 B per thread, strided by the grid
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (i + k * s < n) ? a[i + k * s] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + k * s < n) b[i + k * s] = v[k];
}
__global__ void k_read(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint4 v = a[i]; if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) b[0] = v; }
}
__global__ void k_write(uint4 *__restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = make_uint4((unsigned)i, 1, 2, 3);
}
int main() {
    for (size_t mb : {50, 100, 200, 1000}) {
        size_t bytes = mb << 20, n = bytes / 16;
        uint4 *a, *b;
        hipMalloc(&a, bytes); hipMalloc(&b, bytes);
        hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
        // a second pair so that consecutive launches do not hit the 256 MB infinity cache
        uint4 *c, *d; hipMalloc(&c, 1 << 30); hipMalloc(&d, 1 << 30);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](const char *name, int kind, double moved) {
            float best = 1e9f;
            for (int it = 0; it < 6; it++) {
                k_write<<<(1 << 30) / 16 / 256, 256>>>(c, (1 << 30) / 16);  // flush caches
                hipEventRecord(e0);
                unsigned g = (unsigned)((n + 255) / 256);
                if (kind == 0) k_copy<<<g, 256>>>(a, b, n);
                if (kind == 1) k_copy4<<<(g + 3) / 4, 256>>>(a, b, n);
                if (kind == 2) k_read<<<g, 256>>>(a, b, n);
                if (kind == 3) k_write<<<g, 256>>>(b, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%5zu MB %-8s %8.1f us  %7.1f GB/s\n", mb, name, best * 1e3, moved / (best * 1e-3) / 1e9);
        };
        run("copy", 0, 2.0 * bytes); run("copy4", 1, 2.0 * bytes); run("read", 2, 1.0 * bytes); run("write", 3, 1.0 * bytes);
        hipFree(a); hipFree(b); hipFree(c); hipFree(d);
    }
    return 0;
}
