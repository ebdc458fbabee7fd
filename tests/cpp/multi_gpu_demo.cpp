// SURVEY.md 8(e) as a C++ SLAM process would run it: ONE process, one host worker thread per camera sequence, sequence s on GPU s mod G, each
// worker with its own extractor (= its own HIP streams) and its own pinned staging ring; no collective, results return to the worker's host
// buffers.  (bench.py --gpus N runs the same partitioning with one PROCESS per GPU; `bench.py --threads` is this program's shape in Python.)
//   usage: multi_gpu_demo frames.bin width height frames_per_sequence n_threads steps out.bin
//     frames.bin: n_threads sequences of frames_per_sequence raw 8-bit frames; out.bin: per thread int32 cap, then for every frame of its LAST
//     step: int32 n, int32 monoIndex, n keypoints (28 B), n descriptors (32 B), n int32 (its match vector against the frame before)
// Every step of a worker must reproduce its first step bit for bit (the input does not change); the Python test compares the first step with
// the CPU oracle.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

#include "../../include/orbx.h"

struct Worker {
    int index = 0, device = 0, w = 0, h = 0, nf = 0, steps = 0;
    const uint8_t *seq = nullptr;            // the sequence's frames (pageable: the camera driver's memory)
    std::vector<orbx_keypoint> kps;          // results of the last step
    std::vector<uint8_t> desc;
    std::vector<int32_t> cnt, mono, match, nmatch;
    int cap = 0, mismatching_steps = 0, status = 0;
    double seconds = 0;
    long features = 0;
    std::string error;
};

#define CHECK(expr)                                                                            \
    do {                                                                                       \
        int s_ = (expr);                                                                       \
        if (s_ != ORBX_OK) { W->status = s_; W->error = std::string(#expr) + ": " + orbx_last_error(); return; } \
    } while (0)

static void run_worker(Worker *W) {
    if (hipSetDevice(W->device) != hipSuccess) { W->status = -1; W->error = "hipSetDevice"; return; }
    orbx_params prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.nfeatures = 1000; prm.scale_factor = 1.2f; prm.nlevels = 8; prm.ini_th_fast = 20; prm.min_th_fast = 7;
    orbx_extractor *ex = nullptr;
    CHECK(orbx_create(&prm, W->device, W->w, W->h, W->nf, &ex));
    const size_t fbytes = (size_t)W->w * W->h, bytes = fbytes * W->nf;
    bool pin_failed = false;
    auto pin = [&](size_t b) -> void * { void *p = nullptr; if (hipHostMalloc(&p, b, hipHostMallocDefault) != hipSuccess) { pin_failed = true; p = nullptr; } return p; };
    uint8_t *ring[2] = {(uint8_t *)pin(bytes), (uint8_t *)pin(bytes)};   // the worker's pinned staging ring: frames are copied here, the library uploads from here
    W->cap = orbx_output_capacity(ex, W->w, W->h);
    const size_t n = (size_t)W->nf * W->cap;
    orbx_keypoint *h_kps = (orbx_keypoint *)pin(n * sizeof(orbx_keypoint));
    uint8_t *h_desc = (uint8_t *)pin(n * 32);
    int32_t *h_cnt = (int32_t *)pin(W->nf * 4), *h_mono = (int32_t *)pin(W->nf * 4), *h_match = (int32_t *)pin(n * 4), *h_nm = (int32_t *)pin(W->nf * 4);
    if (pin_failed) { W->status = -1; W->error = "hipHostMalloc"; return; }
    std::vector<orbx_keypoint> first_kps;
    std::vector<uint8_t> first_desc;
    std::vector<int32_t> first_cnt, first_match;
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < W->steps; s++) {
        uint8_t *stage = ring[s & 1];
        std::memcpy(stage, W->seq, bytes);   // the camera thread hands over the batch
        CHECK(orbx_extract_batch_host(ex, stage, W->nf, W->w, W->h, (size_t)W->w, fbytes, 0, 1000));
        CHECK(orbx_match_consecutive_device(ex, 15.0f, -2.0f, -1.0f, 1, nullptr, nullptr));   // SearchByProjection(frame t, frame t-1), results kept in the library's buffers
        CHECK(orbx_batch_download_async(ex, h_kps, h_desc, h_cnt, h_mono, h_match, h_nm));
        CHECK(orbx_download_wait(ex));
        for (int f = 0; f < W->nf; f++) W->features += h_cnt[f];
        if (s == 0) {
            first_kps.assign(h_kps, h_kps + n); first_desc.assign(h_desc, h_desc + n * 32);
            first_cnt.assign(h_cnt, h_cnt + W->nf); first_match.assign(h_match, h_match + n);
        } else {
            bool same = std::memcmp(first_cnt.data(), h_cnt, (size_t)W->nf * 4u) == 0;
            for (int f = 0; f < W->nf && same; f++) {
                const size_t o = (size_t)f * W->cap, c = (size_t)h_cnt[f];
                same = std::memcmp(&first_kps[o], h_kps + o, c * sizeof(orbx_keypoint)) == 0 && std::memcmp(&first_desc[o * 32], h_desc + o * 32, c * 32) == 0 &&
                       (f == 0 || std::memcmp(&first_match[o], h_match + o, c * 4) == 0);
            }
            if (!same) W->mismatching_steps++;
        }
    }
    W->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    W->kps.assign(h_kps, h_kps + n); W->desc.assign(h_desc, h_desc + n * 32);
    W->cnt.assign(h_cnt, h_cnt + W->nf); W->mono.assign(h_mono, h_mono + W->nf);
    W->match.assign(h_match, h_match + n); W->nmatch.assign(h_nm, h_nm + W->nf);
    orbx_destroy(ex);
    for (void *p : {(void *)ring[0], (void *)ring[1], (void *)h_kps, (void *)h_desc, (void *)h_cnt, (void *)h_mono, (void *)h_match, (void *)h_nm}) (void)hipHostFree(p);
}

int main(int argc, char **argv) {
    if (argc < 8) { std::fprintf(stderr, "usage: multi_gpu_demo frames.bin width height frames_per_sequence n_threads steps out.bin\n"); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]), nt = atoi(argv[5]), steps = atoi(argv[6]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { std::fprintf(stderr, "no HIP device\n"); return 3; }
    std::vector<uint8_t> buf((size_t)w * h * nf * nt);
    std::ifstream f(argv[1], std::ios::binary);
    f.read((char *)buf.data(), (std::streamsize)buf.size());
    if (!f) { std::fprintf(stderr, "short input\n"); return 2; }
    std::vector<Worker> W(nt);
    std::vector<std::thread> th;
    for (int i = 0; i < nt; i++) {
        W[i].index = i; W[i].device = i % ndev; W[i].w = w; W[i].h = h; W[i].nf = nf; W[i].steps = steps;
        W[i].seq = buf.data() + (size_t)i * w * h * nf;
        th.emplace_back(run_worker, &W[i]);
    }
    for (auto &t : th) t.join();
    std::ofstream o(argv[7], std::ios::binary);
    int bad = 0;
    double slowest = 0;
    long feats = 0;
    for (const Worker &x : W) {
        if (x.status != ORBX_OK) { std::fprintf(stderr, "worker %d: %s\n", x.index, x.error.c_str()); return 4; }
        bad += x.mismatching_steps;
        slowest = std::max(slowest, x.seconds);
        feats += x.features;
        std::printf("worker %d on GPU %d: %d steps x %d frames in %.3f s, %ld features, %d steps differ from the first\n", x.index, x.device, x.steps, x.nf, x.seconds, x.features,
                    x.mismatching_steps);
        const int32_t cap = x.cap;
        o.write((const char *)&cap, 4);
        for (int fr = 0; fr < x.nf; fr++) {
            const int32_t n = x.cnt[fr], mono = x.mono[fr];
            o.write((const char *)&n, 4);
            o.write((const char *)&mono, 4);
            o.write((const char *)&x.kps[(size_t)fr * x.cap], (std::streamsize)(n * sizeof(orbx_keypoint)));
            o.write((const char *)&x.desc[(size_t)fr * x.cap * 32], (std::streamsize)(n * 32));
            o.write((const char *)&x.match[(size_t)fr * x.cap], (std::streamsize)(n * 4));
        }
    }
    std::printf("threads %d devices %d mismatches %d aggregate %.1f kfeatures/s\n", nt, ndev, bad, feats / slowest / 1e3);
    return bad == 0 ? 0 : 1;
}
