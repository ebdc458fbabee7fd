/*
 * orb_oracle.cc -- CPU ORACLE (extractor half).  TEST INFRASTRUCTURE ONLY -- see orb_oracle.h.
 *
 * Every function cites the reference lines (/root/reference/...) or the OpenCV / glibc routine it
 * restates.  "[OCV]" marks behaviour of OpenCV 4.x restated from its published algorithm; OpenCV is an
 * un-vendored dependency of the reference (CMakeLists.txt:33-36, "find_package(OpenCV 4.4)") and is
 * not available in this image, so those parts are **parity unpinned**; everything that is the reference's own
 * code in this file is pinned against the compiled reference source (see header, `make ref`).
 *
 * Build: g++ -O3 -march=x86-64-v3 -ffp-contract=off (see oracle/Makefile).  Contraction is OFF so that
 * every float expression rounds exactly as written; the one place where the reference's own build
 * (-O3 -march=native, CMakeLists.txt:10-13) fuses a multiply-add (ORBextractor.cc:117-119) is
 * written with an explicit fmaf under ORBO_FLAG_DESC_FMA.
 */
#include "orb_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

namespace {

/* ------------------------------------------------------------------------------------------------
 * [OCV] cvRound: round-half-to-even (SSE cvtss2si / cvtsd2si in the default rounding mode).
 * ---------------------------------------------------------------------------------------------- */
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

const int kPatch = 31;       /* PATCH_SIZE       ORBextractor.cc:71 */
const int kHalfPatch = 15;   /* HALF_PATCH_SIZE  ORBextractor.cc:72 */
const int kEdge = 19;        /* EDGE_THRESHOLD   ORBextractor.cc:73 */

const int8_t kPattern[1024] = {
#include "orb_pattern_data.inc"
};

/* ------------------------------------------------------------------------------------------------
 * [OCV] borderInterpolate(p, len, BORDER_REFLECT_101):  ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ...
 * ---------------------------------------------------------------------------------------------- */
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* ------------------------------------------------------------------------------------------------
 * [OCV] cv::resize, 8UC1, INTER_LINEAR, generic (non-IPP, non-area) path:
 * resizeGeneric_<HResizeLinear<uchar,int,short,2048>, VResizeLinear<uchar,int,short,FixedPtCast<22>>>.
 * Called by ORBextractor.cc:1183 from the previous level's ROI to this level's ROI.
 * ---------------------------------------------------------------------------------------------- */
struct LinearTab {
    std::vector<int> ofs;
    std::vector<short> c0, c1;
};

LinearTab linear_tab(int ssize, int dsize, bool horizontal) {
    LinearTab t;
    t.ofs.resize(dsize);
    t.c0.resize(dsize);
    t.c1.resize(dsize);
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (horizontal) {
            /* x: the fraction is reset at both borders */
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        t.ofs[d] = s;
        /* saturate_cast<short>(float) == cvRound, then clamp to short */
        int a0 = cv_round((1.f - f) * 2048.f), a1 = cv_round(f * 2048.f);
        t.c0[d] = (short)std::min(std::max(a0, -32768), 32767);
        t.c1[d] = (short)std::min(std::max(a1, -32768), 32767);
    }
    return t;
}

void resize_linear_u8(const uint8_t *src, int sw, int sh, size_t sstride, uint8_t *dst, int dw, int dh,
                      size_t dstride) {
    LinearTab tx = linear_tab(sw, dw, true), ty = linear_tab(sh, dh, false);
    std::vector<int> row0(dw), row1(dw);
    auto hpass = [&](int sy, std::vector<int> &out) {
        const uint8_t *S = src + (size_t)sy * sstride;
        for (int x = 0; x < dw; x++) {
            int sx = tx.ofs[x];
            if (sx + 1 < sw) out[x] = S[sx] * tx.c0[x] + S[sx + 1] * tx.c1[x];
            else out[x] = S[sx] * 2048; /* tail: D[dx] = S[xofs[dx]]*ONE */
        }
    };
    for (int y = 0; y < dh; y++) {
        /* vertical: rows sy, sy+1 index-clamped to [0, sh-1]; the fraction is NOT reset */
        int sy0 = std::min(std::max(ty.ofs[y], 0), sh - 1);
        int sy1 = std::min(std::max(ty.ofs[y] + 1, 0), sh - 1);
        hpass(sy0, row0);
        hpass(sy1, row1);
        const int b0 = ty.c0[y], b1 = ty.c1[y];
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++) {
            int v = (((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2;
            D[x] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
}

/* [OCV] copyMakeBorder(..., BORDER_REFLECT_101 [+BORDER_ISOLATED]) in place: the ROI sits at (border,border)
 * of the padded buffer (ORBextractor.cc:1185-1191). */
void border_reflect101(uint8_t *padded, int w, int h, size_t stride, int border) {
    const int W = w + 2 * border, H = h + 2 * border;
    for (int y = 0; y < h; y++) {
        uint8_t *row = padded + (size_t)(y + border) * stride;
        for (int x = 0; x < border; x++) row[x] = row[border + reflect101(x - border, w)];
        for (int x = w + border; x < W; x++) row[x] = row[border + reflect101(x - border, w)];
    }
    for (int y = 0; y < H; y++) {
        if (y >= border && y < h + border) continue;
        int sy = reflect101(y - border, h) + border;
        memcpy(padded + (size_t)y * stride, padded + (size_t)sy * stride, W);
    }
}

/* ------------------------------------------------------------------------------------------------
 * [OCV] cv::FAST, TYPE_9_16, nonmaxSuppression=true  (FAST_t<16> + cornerScore<16>), as called per cell
 * at ORBextractor.cc:826-827 and :845-846.
 * ---------------------------------------------------------------------------------------------- */
const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* cornerScore<16>: largest threshold for which the pixel is still a 9-of-16 corner */
int corner_score16(const uint8_t *ptr, const ptrdiff_t pixel[25], int threshold) {
    const int v = ptr[0];
    short d[25];
    for (int k = 0; k < 25; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) a = std::min(a, (int)d[k + j]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        for (int j = 3; j <= 5; j++) b = std::max(b, (int)d[k + j]);
        if (b >= b0) continue;
        for (int j = 6; j <= 8; j++) b = std::max(b, (int)d[k + j]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

void make_offsets(ptrdiff_t pixel[25], size_t stride) {
    for (int k = 0; k < 25; k++) pixel[k] = kCircle[k % 16][0] + (ptrdiff_t)kCircle[k % 16][1] * (ptrdiff_t)stride;
}

/* is the pixel a corner at `threshold`: more than 8 contiguous circle pixels darker, or brighter */
bool is_corner16(const uint8_t *ptr, const ptrdiff_t pixel[25], int threshold) {
    const int v = ptr[0];
    int count = 0;
    for (int k = 0; k < 25; k++) {
        if (ptr[pixel[k]] < v - threshold) { if (++count > 8) return true; }
        else count = 0;
    }
    count = 0;
    for (int k = 0; k < 25; k++) {
        if (ptr[pixel[k]] > v + threshold) { if (++count > 8) return true; }
        else count = 0;
    }
    return false;
}

void fast9_16(const uint8_t *img, int cols, int rows, size_t stride, int threshold, std::vector<orbo_keypoint> &out) {
    out.clear();
    if (cols < 7 || rows < 7) return;
    ptrdiff_t pixel[25];
    make_offsets(pixel, stride);
    threshold = std::min(std::max(threshold, 0), 255);
    /* three rolling score rows, zero-initialised: out-of-interior and non-corner neighbours read as 0 */
    std::vector<uint8_t> buf((size_t)cols * 3, 0);
    std::vector<int> cpos((size_t)(cols + 1) * 3, 0);
    for (int i = 3; i < rows - 2; i++) {
        const uint8_t *ptr = img + (size_t)i * stride + 3;
        uint8_t *curr = &buf[(size_t)((i - 3) % 3) * cols];
        int *cornerpos = &cpos[(size_t)((i - 3) % 3) * (cols + 1)] + 1;
        memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; j++, ptr++) {
                if (is_corner16(ptr, pixel, threshold)) {
                    cornerpos[ncorners++] = j;
                    curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t *prev = &buf[(size_t)((i - 4 + 3) % 3) * cols];
        const uint8_t *pprev = &buf[(size_t)((i - 5 + 3) % 3) * cols];
        cornerpos = &cpos[(size_t)((i - 4 + 3) % 3) * (cols + 1)] + 1;
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
                orbo_keypoint kp = {(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1};
                out.push_back(kp);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * [OCV] GaussianBlur(7x7, sigma 2) on 8U, fixed-point path (ufixedpoint16 taps, Q8.8):
 * exact 2-D sum of products, single rounding ((sum + 2^15) >> 16), saturate to u8; REFLECT_101.
 * ORBextractor.cc:1132-1133 (the clone drops the ring, so the border is the level itself).
 * ---------------------------------------------------------------------------------------------- */
const int kGaussNew[7] = {18, 34, 48, 56, 48, 34, 18}; /* OpenCV >= 4.5.1: error-diffused, sums to 256 */
const int kGaussOld[7] = {18, 34, 49, 55, 49, 34, 18}; /* OpenCV <= 4.5.0: each tap rounded, sums to 257 */

void gauss7_u8(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride, bool ocv440) {
    const int *g = ocv440 ? kGaussOld : kGaussNew;
    std::vector<uint32_t> hbuf((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = 0; k < 7; k++) s += (uint32_t)g[k] * S[reflect101(x + k - 3, w)];
            hbuf[(size_t)y * w + x] = s; /* <= 255*257 = 65535: never saturates ufixedpoint16 */
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int k = 0; k < 7; k++) s += (uint32_t)g[k] * hbuf[(size_t)reflect101(y + k - 3, h) * w + x];
            uint32_t v = (s + 32768u) >> 16;
            D[x] = (uint8_t)std::min(v, 255u);
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * [OCV] cv::fastAtan2(y, x) in degrees -- scalar atan_f32 polynomial, plain float ops (baseline build,
 * no FMA).  Called at ORBextractor.cc:102.
 * ---------------------------------------------------------------------------------------------- */
const float kAtanP1 = 0.9997878412794807f * (float)(180 / M_PI);
const float kAtanP3 = -0.3258083974640975f * (float)(180 / M_PI);
const float kAtanP5 = 0.1555786518463281f * (float)(180 / M_PI);
const float kAtanP7 = -0.04432655554792128f * (float)(180 / M_PI);

float fast_atan2(float y, float x, bool fma_poly = false) {
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (fma_poly) {
        /* ORBO_FLAG_ATAN_FMA: the same expression as a compiler that contracts a*b + c emits it (GCC -ffp-contract=fast with FMA in
         * the baseline ISA: aarch64, or x86 with CPU_BASELINE >= FMA3): three fused Horner steps, and `90 - q*c` as one fnmadd.
         * Written with explicit fmaf so that it does not depend on how THIS file is compiled; tests/test_oracle_known_answers.py
         * checks it against the source expression compiled with -mfma -ffp-contract=fast. */
        const bool xbig = ax >= ay;
        c = xbig ? ay / (ax + (float)DBL_EPSILON) : ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        const float q = std::fmaf(std::fmaf(std::fmaf(kAtanP7, c2, kAtanP5), c2, kAtanP3), c2, kAtanP1);
        a = xbig ? q * c : std::fmaf(-q, c, 90.f);
    } else if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((kAtanP7 * c2 + kAtanP5) * c2 + kAtanP3) * c2 + kAtanP1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((kAtanP7 * c2 + kAtanP5) * c2 + kAtanP3) * c2 + kAtanP1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ------------------------------------------------------------------------------------------------
 * glibc 2.35 sinf/cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c), the
 * x86_64 "fma" ifunc variant this image's libm selects on FMA hardware: every multiply-add of the
 * double polynomial is fused.  Constants read from this image's libm.so.6 (__sincosf_table).
 * Valid for |x| < 120 (the descriptor angle is in [0, 2*pi]); larger arguments are not needed.
 * orbo_check_sincos_vs_libm verifies bit-equality with the host libm exhaustively.
 * ---------------------------------------------------------------------------------------------- */
struct SinCosTab {
    double sign[4], hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3;
};
const SinCosTab kSC[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2,
     0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2,
     -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};

inline uint32_t abstop12(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return (u >> 20) & 0x7ff;
}
inline float sin_poly(double x, double x2, const SinCosTab &p) {
    double x3 = x * x2;
    double s1 = std::fma(x2, p.s3, p.s2);
    double x7 = x3 * x2;
    double s = std::fma(x3, p.s1, x);
    return (float)std::fma(x7, s1, s);
}
inline float cos_poly(double x2, const SinCosTab &p) {
    double x4 = x2 * x2;
    double c2 = std::fma(x2, p.c4, p.c3);
    double c1 = std::fma(x2, p.c1, p.c0);
    double x6 = x4 * x2;
    double c = std::fma(x4, p.c2, c1);
    return (float)std::fma(x6, c2, c);
}
inline double reduce_fast(double x, int *np) {
    double r = x * kSC[0].hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return std::fma(-(double)n, kSC[0].hpi, x);
}
float ref_sinf(float y) {
    double x = y;
    if (abstop12(y) < 0x3f4) { /* |y| < pi/4 */
        if (abstop12(y) < 0x398) return y; /* |y| < 2^-12 */
        return sin_poly(x, x * x, kSC[0]);
    }
    if (abstop12(y) < 0x42f) { /* |y| < 120 */
        int n;
        x = reduce_fast(x, &n);
        const SinCosTab &p = kSC[(n & 2) ? 1 : 0];
        if (n & 1) return cos_poly(x * x, p);
        return sin_poly(x * kSC[0].sign[n & 3], x * x, p);
    }
    return sinf(y); /* outside the domain the descriptor path can produce */
}
float ref_cosf(float y) {
    double x = y;
    if (abstop12(y) < 0x3f4) {
        if (abstop12(y) < 0x398) return 1.0f;
        return cos_poly(x * x, kSC[0]);
    }
    if (abstop12(y) < 0x42f) {
        int n;
        x = reduce_fast(x, &n);
        const SinCosTab &p = kSC[(n & 2) ? 1 : 0];
        if (n & 1) return sin_poly(x * kSC[0].sign[n & 3], x * x, p);
        return cos_poly(x * x, p);
    }
    return cosf(y);
}

/* ------------------------------------------------------------------------------------------------
 * IC_Angle: ORBextractor.cc:76-103
 * ---------------------------------------------------------------------------------------------- */
struct Tables {
    int nfeatures, nlevels, ini_th, min_th, flags;
    double scale_factor; /* the member is a double initialised from a float (ORBextractor.h:94) */
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> quota;
    int umax[16];
};

float ic_angle(const uint8_t *center, size_t stride, const int umax[16], bool atan_fma = false) {
    int m_01 = 0, m_10 = 0;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m_10 += u * center[u];
    const ptrdiff_t step = (ptrdiff_t)stride;
    for (int v = 1; v <= kHalfPatch; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10, atan_fma);
}

/* ------------------------------------------------------------------------------------------------
 * computeOrbDescriptor: ORBextractor.cc:105-146.
 * fma_mode: GCC 11 -O3 -march=native on an FMA CPU turns  x*b + y*a  into fma(x, b, y*a)  and
 * x*a - y*b  into fma(x, a, -(y*b))  (checked by disassembly).
 * ---------------------------------------------------------------------------------------------- */
void orb_descriptor(const uint8_t *center, size_t stride, float angle_deg, bool fma_mode, bool libm, uint8_t *desc) {
    const float factorPI = (float)(M_PI / 180.f);
    const float angle = angle_deg * factorPI;
    const float a = libm ? cosf(angle) : ref_cosf(angle);
    const float b = libm ? sinf(angle) : ref_sinf(angle);
    const ptrdiff_t step = (ptrdiff_t)stride;
    auto sample = [&](int idx) -> int {
        const float px = (float)kPattern[2 * idx], py = (float)kPattern[2 * idx + 1];
        float fr, fc;
        if (fma_mode) {
            fr = fmaf(px, b, py * a);
            fc = fmaf(px, a, -(py * b));
        } else {
            fr = px * b + py * a;
            fc = px * a - py * b;
        }
        return center[cv_round(fr) * step + cv_round(fc)];
    };
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            int t0 = sample(16 * i + 2 * k), t1 = sample(16 * i + 2 * k + 1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------------------------------------
 * ExtractorNode / DivideNode / compareNodes / DistributeOctTree: ORBextractor.cc:480-779
 * ---------------------------------------------------------------------------------------------- */
struct QNode {
    std::vector<orbo_keypoint> keys;
    int ulx = 0, uly = 0, urx = 0, ury = 0, blx = 0, bly = 0, brx = 0, bry = 0;
    std::list<QNode>::iterator self;
    bool leaf = false; /* bNoMore */
};

void divide_node(const QNode &p, QNode c[4]) { /* :480-536 */
    const int halfX = (int)std::ceil(static_cast<float>(p.urx - p.ulx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(p.bry - p.uly) / 2);
    c[0].ulx = p.ulx;          c[0].uly = p.uly;
    c[0].urx = p.ulx + halfX;  c[0].ury = p.uly;
    c[0].blx = p.ulx;          c[0].bly = p.uly + halfY;
    c[0].brx = p.ulx + halfX;  c[0].bry = p.uly + halfY;
    c[1].ulx = c[0].urx;       c[1].uly = c[0].ury;
    c[1].urx = p.urx;          c[1].ury = p.ury;
    c[1].blx = c[0].brx;       c[1].bly = c[0].bry;
    c[1].brx = p.urx;          c[1].bry = p.uly + halfY;
    c[2].ulx = c[0].blx;       c[2].uly = c[0].bly;
    c[2].urx = c[0].brx;       c[2].ury = c[0].bry;
    c[2].blx = p.blx;          c[2].bly = p.bly;
    c[2].brx = c[0].brx;       c[2].bry = p.bly;
    c[3].ulx = c[2].urx;       c[3].uly = c[2].ury;
    c[3].urx = c[1].brx;       c[3].ury = c[1].bry;
    c[3].blx = c[2].brx;       c[3].bly = c[2].bry;
    c[3].brx = p.brx;          c[3].bry = p.bry;
    for (int k = 0; k < 4; k++) c[k].keys.reserve(p.keys.size());
    for (const orbo_keypoint &kp : p.keys) {
        if (kp.x < c[0].urx) {
            if (kp.y < c[0].bry) c[0].keys.push_back(kp);
            else c[2].keys.push_back(kp);
        } else if (kp.y < c[0].bry) c[1].keys.push_back(kp);
        else c[3].keys.push_back(kp);
    }
    for (int k = 0; k < 4; k++)
        if (c[k].keys.size() == 1) c[k].leaf = true;
}

typedef std::pair<int, QNode *> SizedNode;
bool compare_nodes(SizedNode &e1, SizedNode &e2) { /* :538-553; not a total order on ties */
    if (e1.first < e2.first) return true;
    if (e1.first > e2.first) return false;
    return e1.second->ulx < e2.second->ulx;
}

std::vector<orbo_keypoint> distribute_octree(const std::vector<orbo_keypoint> &in, int minX, int maxX, int minY,
                                             int maxY, int N) {
    std::vector<orbo_keypoint> result;
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY)); /* :559 */
    if (nIni <= 0) return result; /* the reference divides by zero here; we return nothing */
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<QNode> nodes;
    std::vector<QNode *> roots(nIni);
    for (int i = 0; i < nIni; i++) { /* :567-580 */
        QNode ni;
        ni.ulx = (int)(hX * static_cast<float>(i));      ni.uly = 0;
        ni.urx = (int)(hX * static_cast<float>(i + 1));  ni.ury = 0;
        ni.blx = ni.ulx;                                 ni.bly = maxY - minY;
        ni.brx = ni.urx;                                 ni.bry = maxY - minY;
        ni.keys.reserve(in.size());
        nodes.push_back(ni);
        roots[i] = &nodes.back();
    }
    for (const orbo_keypoint &kp : in) { /* :583-587 */
        size_t r = (size_t)(kp.x / hX);
        if (r >= roots.size()) r = roots.size() - 1; /* out of bounds in the reference; cannot happen for x < W */
        roots[r]->keys.push_back(kp);
    }
    for (auto it = nodes.begin(); it != nodes.end();) { /* :589-602 */
        if (it->keys.size() == 1) { it->leaf = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }
    bool finish = false;
    std::vector<SizedNode> expandable;
    expandable.reserve(nodes.size() * 4);
    auto add_children = [&](QNode c[4], int *n_to_expand) { /* :634-677 / :713-748 */
        for (int k = 0; k < 4; k++) {
            if (c[k].keys.size() > 0) {
                nodes.push_front(c[k]);
                if (c[k].keys.size() > 1) {
                    if (n_to_expand) ++*n_to_expand;
                    expandable.push_back(std::make_pair((int)c[k].keys.size(), &nodes.front()));
                    nodes.front().self = nodes.begin();
                }
            }
        }
    };
    while (!finish) { /* :611-755 */
        int prev_size = (int)nodes.size();
        auto it = nodes.begin();
        int n_to_expand = 0;
        expandable.clear();
        while (it != nodes.end()) {
            if (it->leaf) { ++it; continue; }
            QNode c[4];
            divide_node(*it, c);
            add_children(c, &n_to_expand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prev_size) {
            finish = true;
        } else if (((int)nodes.size() + n_to_expand * 3) > N) {
            while (!finish) {
                prev_size = (int)nodes.size();
                std::vector<SizedNode> prev = expandable;
                expandable.clear();
                std::sort(prev.begin(), prev.end(), compare_nodes);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    QNode c[4];
                    divide_node(*prev[j].second, c);
                    add_children(c, nullptr);
                    nodes.erase(prev[j].second->self);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prev_size) finish = true;
            }
        }
    }
    result.reserve(nodes.size());
    for (const QNode &n : nodes) { /* :757-776: best response per node, first wins ties */
        const orbo_keypoint *best = &n.keys[0];
        float max_response = best->response;
        for (size_t k = 1; k < n.keys.size(); k++)
            if (n.keys[k].response > max_response) { best = &n.keys[k]; max_response = n.keys[k].response; }
        result.push_back(*best);
    }
    return result;
}

}  // namespace

/* ================================================================================================
 * The extractor object: ORBextractor ctor (ORBextractor.cc:409-469), ComputePyramid (:1170-1195),
 * ComputeKeyPointsOctTree (:781-896), operator() (:1086-1168)
 * ============================================================================================== */
struct orbo_extractor {
    Tables t;
    struct Level {
        int w = 0, h = 0;
        size_t stride = 0;
        std::vector<uint8_t> padded;   /* (w+38) x (h+38) */
        std::vector<uint8_t> blurred;  /* w x h, empty if no keypoints */
        std::vector<orbo_keypoint> candidates, keypoints;
        uint8_t *roi() { return padded.data() + kEdge * stride + kEdge; }
        const uint8_t *roi() const { return padded.data() + kEdge * stride + kEdge; }
    };
    std::vector<Level> levels;
};

extern "C" {

orbo_extractor *orbo_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int flags) {
    if (nlevels < 1 || nlevels > 32 || nfeatures < 1 || !(scale_factor > 1.0f)) return nullptr;
    orbo_extractor *ex = new orbo_extractor();
    Tables &t = ex->t;
    t.nfeatures = nfeatures; t.nlevels = nlevels; t.ini_th = ini_th; t.min_th = min_th; t.flags = flags;
    t.scale_factor = scale_factor;
    t.scale.resize(nlevels); t.sigma2.resize(nlevels); t.inv_scale.resize(nlevels); t.inv_sigma2.resize(nlevels);
    t.scale[0] = 1.0f; t.sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) { /* :418-422: float * double -> float */
        t.scale[i] = (float)(t.scale[i - 1] * t.scale_factor);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    for (int i = 0; i < nlevels; i++) { /* :426-430 */
        t.inv_scale[i] = 1.0f / t.scale[i];
        t.inv_sigma2[i] = 1.0f / t.sigma2[i];
    }
    t.quota.resize(nlevels);
    float factor = (float)(1.0f / t.scale_factor); /* :435 */
    float desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels)); /* :436 */
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) { /* :439-444 */
        t.quota[level] = cv_round(desired);
        sum += t.quota[level];
        desired *= factor;
    }
    t.quota[nlevels - 1] = std::max(nfeatures - sum, 0);
    /* :453-468 umax */
    int v, v0, vmax = cv_floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v < 16; v++) t.umax[v] = 0;
    for (v = 0; v <= vmax; ++v) t.umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
    ex->levels.resize(nlevels);
    return ex;
}

void orbo_destroy(orbo_extractor *ex) { delete ex; }

int orbo_get_tables(const orbo_extractor *ex, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                    int *quota, int *umax16) {
    const Tables &t = ex->t;
    for (int i = 0; i < t.nlevels; i++) {
        if (scale) scale[i] = t.scale[i];
        if (inv_scale) inv_scale[i] = t.inv_scale[i];
        if (sigma2) sigma2[i] = t.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = t.inv_sigma2[i];
        if (quota) quota[i] = t.quota[i];
    }
    if (umax16) memcpy(umax16, t.umax, sizeof(t.umax));
    return t.nlevels;
}

int orbo_extract(orbo_extractor *ex, const uint8_t *img, int w, int h, size_t stride, int lap0, int lap1,
                 orbo_keypoint *kps, uint8_t *desc, int cap, int *n_out) {
    if (n_out) *n_out = 0;
    if (!img || w <= 0 || h <= 0) return -1; /* :1090 */
    const Tables &t = ex->t;
    const bool fma_mode = (t.flags & ORBO_FLAG_DESC_FMA) != 0, ocv440 = (t.flags & ORBO_FLAG_BLUR_OCV440) != 0,
               libm = (t.flags & ORBO_FLAG_LIBM_SINCOS) != 0;

    /* ---- ComputePyramid :1170-1195 ---- */
    for (int level = 0; level < t.nlevels; ++level) {
        orbo_extractor::Level &L = ex->levels[level];
        const float scale = t.inv_scale[level];
        L.w = cv_round((float)w * scale);
        L.h = cv_round((float)h * scale);
        if (L.w - 2 * (kEdge - 3) < 35 || L.h - 2 * (kEdge - 3) < 35) return -2; /* reference divides by zero (:800-803) */
        L.stride = (size_t)L.w + 2 * kEdge;
        L.padded.assign(L.stride * (L.h + 2 * kEdge), 0);
        if (level != 0) {
            const orbo_extractor::Level &P = ex->levels[level - 1];
            resize_linear_u8(P.roi(), P.w, P.h, P.stride, L.roi(), L.w, L.h, L.stride);
        } else {
            for (int y = 0; y < h; y++) memcpy(L.roi() + (size_t)y * L.stride, img + (size_t)y * stride, w);
        }
        border_reflect101(L.padded.data(), L.w, L.h, L.stride, kEdge);
        L.blurred.clear(); L.candidates.clear(); L.keypoints.clear();
    }

    /* ---- ComputeKeyPointsOctTree :781-896 ---- */
    const float W = 35;
    for (int level = 0; level < t.nlevels; ++level) {
        orbo_extractor::Level &L = ex->levels[level];
        const int minBorderX = kEdge - 3, minBorderY = minBorderX;
        const int maxBorderX = L.w - kEdge + 3, maxBorderY = L.h - kEdge + 3;
        std::vector<orbo_keypoint> &cand = L.candidates;
        cand.reserve(t.nfeatures * 10);
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        std::vector<orbo_keypoint> cell;
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                const uint8_t *sub = L.roi() + (size_t)(int)iniY * L.stride + (int)iniX;
                const int cols = (int)maxX - (int)iniX, rows = (int)maxY - (int)iniY;
                fast9_16(sub, cols, rows, L.stride, t.ini_th, cell);
                if (cell.empty()) fast9_16(sub, cols, rows, L.stride, t.min_th, cell);
                for (orbo_keypoint &kp : cell) {
                    kp.x += j * wCell;
                    kp.y += i * hCell;
                    cand.push_back(kp);
                }
            }
        }
        L.keypoints = distribute_octree(cand, minBorderX, maxBorderX, minBorderY, maxBorderY, t.quota[level]);
        const int scaledPatchSize = (int)(kPatch * t.scale[level]); /* :880 */
        for (orbo_keypoint &kp : L.keypoints) {
            kp.x += minBorderX;
            kp.y += minBorderY;
            kp.octave = level;
            kp.size = (float)scaledPatchSize;
        }
    }
    for (int level = 0; level < t.nlevels; ++level) { /* :894-895 orientation on the UNBLURRED level */
        orbo_extractor::Level &L = ex->levels[level];
        for (orbo_keypoint &kp : L.keypoints)
            kp.angle = ic_angle(L.roi() + (size_t)cv_round(kp.y) * L.stride + cv_round(kp.x), L.stride, t.umax, (t.flags & ORBO_FLAG_ATAN_FMA) != 0);
    }

    /* ---- operator() :1104-1167 ---- */
    int nkeypoints = 0;
    for (int level = 0; level < t.nlevels; ++level) nkeypoints += (int)ex->levels[level].keypoints.size();
    if (n_out) *n_out = nkeypoints;
    if (nkeypoints > cap) return -3;
    int monoIndex = 0, stereoIndex = nkeypoints - 1;
    uint8_t d[32];
    for (int level = 0; level < t.nlevels; ++level) {
        orbo_extractor::Level &L = ex->levels[level];
        if (L.keypoints.empty()) continue;
        L.blurred.resize((size_t)L.w * L.h);
        gauss7_u8(L.roi(), L.w, L.h, L.stride, L.blurred.data(), L.w, ocv440);
        const float scale = t.scale[level];
        for (const orbo_keypoint &lkp : L.keypoints) {
            orbo_keypoint kp = lkp;
            orb_descriptor(L.blurred.data() + (size_t)cv_round(kp.y) * L.w + cv_round(kp.x), L.w, kp.angle, fma_mode,
                           libm, d);
            if (level != 0) { kp.x *= scale; kp.y *= scale; }
            int pos;
            if (kp.x >= lap0 && kp.x <= lap1) pos = stereoIndex--;
            else pos = monoIndex++;
            kps[pos] = kp;
            memcpy(desc + (size_t)pos * 32, d, 32);
        }
    }
    return monoIndex;
}

int orbo_level_size(const orbo_extractor *ex, int level, int *w, int *h) {
    if (level < 0 || level >= ex->t.nlevels) return -1;
    *w = ex->levels[level].w; *h = ex->levels[level].h;
    return 0;
}
const uint8_t *orbo_level_padded(const orbo_extractor *ex, int level, size_t *stride) {
    *stride = ex->levels[level].stride;
    return ex->levels[level].padded.data();
}
const uint8_t *orbo_level_blurred(const orbo_extractor *ex, int level, size_t *stride) {
    *stride = ex->levels[level].w;
    return ex->levels[level].blurred.empty() ? nullptr : ex->levels[level].blurred.data();
}
static int copy_out(const std::vector<orbo_keypoint> &v, orbo_keypoint *out, int cap) {
    int n = (int)std::min<size_t>(v.size(), (size_t)std::max(cap, 0));
    if (out && n) memcpy(out, v.data(), (size_t)n * sizeof(orbo_keypoint));
    return (int)v.size();
}
int orbo_level_candidates(const orbo_extractor *ex, int level, orbo_keypoint *out, int cap) {
    return copy_out(ex->levels[level].candidates, out, cap);
}
int orbo_level_keypoints(const orbo_extractor *ex, int level, orbo_keypoint *out, int cap) {
    return copy_out(ex->levels[level].keypoints, out, cap);
}

/* ---- primitives ---- */
int orbo_cv_round_f(float v) { return cv_round(v); }
void orbo_resize_linear_u8(const uint8_t *src, int sw, int sh, size_t sstride, uint8_t *dst, int dw, int dh,
                           size_t dstride) {
    resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orbo_border_reflect101(uint8_t *padded, int w, int h, size_t stride, int border) {
    border_reflect101(padded, w, h, stride, border);
}
int orbo_fast9_16(const uint8_t *img, int cols, int rows, size_t stride, int threshold, orbo_keypoint *out, int cap) {
    std::vector<orbo_keypoint> v;
    fast9_16(img, cols, rows, stride, threshold, v);
    return copy_out(v, out, cap);
}
int orbo_fast_score(const uint8_t *center, size_t stride) {
    ptrdiff_t pixel[25];
    make_offsets(pixel, stride);
    /* with threshold 0 the formula returns max(0, A, B) - 1 where A/B are the best dark/bright arc minima */
    return corner_score16(center, pixel, 0);
}
void orbo_gauss7_u8(const uint8_t *src, int w, int h, size_t sstride, uint8_t *dst, size_t dstride, int ocv440) {
    gauss7_u8(src, w, h, sstride, dst, dstride, ocv440 != 0);
}
float orbo_fast_atan2(float y, float x) { return fast_atan2(y, x); }
float orbo_fast_atan2_fma(float y, float x) { return fast_atan2(y, x, true); }
void orbo_fast_atan2_n(const float *y, const float *x, float *out, int n, int fma_poly) {
    for (int i = 0; i < n; i++) out[i] = fast_atan2(y[i], x[i], fma_poly != 0);
}
float orbo_ic_angle(const uint8_t *center, size_t stride) {
    static const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    return ic_angle(center, stride, umax);
}
/* The same two functions as the x86_64 "sse2" ifunc variant computes them (a libm on a CPU without FMA, or built without the multiarch
 * variants): the identical source (sincosf.h) with every multiply-add rounded twice.  Used only to COUNT the arguments on which the two
 * variants differ (orbo_count_sincos_fma_vs_nofma): the descriptor path's dependence on which libm variant the reference process runs. */
static inline float sin_poly_nofma(double x, double x2, const SinCosTab &p) {
    double x3 = x * x2;
    double s1 = p.s2 + x2 * p.s3;
    double x7 = x3 * x2;
    double s = x + x3 * p.s1;
    return (float)(s + x7 * s1);
}
static inline float cos_poly_nofma(double x2, const SinCosTab &p) {
    double x4 = x2 * x2;
    double c2 = p.c3 + x2 * p.c4;
    double c1 = p.c0 + x2 * p.c1;
    double x6 = x4 * x2;
    double c = c1 + x4 * p.c2;
    return (float)(c + x6 * c2);
}
static inline void sincos_nofma(float y, float *sn, float *cs) {
    double x = y;
    if (abstop12(y) < 0x3f4) {
        if (abstop12(y) < 0x398) { *sn = y; *cs = 1.0f; return; }
        *sn = sin_poly_nofma(x, x * x, kSC[0]); *cs = cos_poly_nofma(x * x, kSC[0]);
        return;
    }
    double r = x * kSC[0].hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * kSC[0].hpi;
    const SinCosTab &p = kSC[(n & 2) ? 1 : 0];
    const float ps = sin_poly_nofma(x * kSC[0].sign[n & 3], x * x, p), pc = cos_poly_nofma(x * x, p);
    *sn = (n & 1) ? pc : ps;
    *cs = (n & 1) ? ps : pc;
}
/* number of float arguments with bit patterns in [lo, hi) (0 <= x < 120) for which sinf or cosf of the fma and the non-fma variant differ */
uint64_t orbo_count_sincos_fma_vs_nofma(uint32_t lo, uint32_t hi, uint32_t *first_bad) {
    uint64_t bad = 0;
    for (uint64_t b = lo; b < hi; b++) {
        uint32_t u = (uint32_t)b;
        float x, s1, c1;
        memcpy(&x, &u, 4);
        const float s0 = ref_sinf(x), c0 = ref_cosf(x);
        sincos_nofma(x, &s1, &c1);
        if (memcmp(&s0, &s1, 4) || memcmp(&c0, &c1, 4)) {
            if (!bad && first_bad) *first_bad = u;
            bad++;
        }
    }
    return bad;
}
float orbo_sinf(float x) { return ref_sinf(x); }
float orbo_cosf(float x) { return ref_cosf(x); }
uint64_t orbo_check_sincos_vs_libm(uint32_t lo, uint32_t hi, uint32_t *first_bad) {
    uint64_t bad = 0;
    for (uint64_t b = lo; b < hi; b++) {
        uint32_t u = (uint32_t)b;
        float x;
        memcpy(&x, &u, 4);
        float s0 = sinf(x), s1 = ref_sinf(x), c0 = cosf(x), c1 = ref_cosf(x);
        if (memcmp(&s0, &s1, 4) || memcmp(&c0, &c1, 4)) {
            if (!bad && first_bad) *first_bad = u;
            bad++;
        }
    }
    return bad;
}
void orbo_orb_descriptor(const uint8_t *center, size_t stride, float angle_deg, int fma_mode, int libm, uint8_t *desc) {
    orb_descriptor(center, stride, angle_deg, fma_mode != 0, libm != 0, desc);
}
int orbo_distribute_octree(const orbo_keypoint *in, int n_in, int minX, int maxX, int minY, int maxY, int N,
                           orbo_keypoint *out, int cap) {
    std::vector<orbo_keypoint> v(in, in + n_in);
    std::vector<orbo_keypoint> r = distribute_octree(v, minX, maxX, minY, maxY, N);
    return copy_out(r, out, cap);
}
void orbo_sort_nodes(const int *count, const int *ulx, int n, int *perm) {
    std::vector<QNode> dummy(n);
    std::vector<SizedNode> v(n);
    for (int i = 0; i < n; i++) { dummy[i].ulx = ulx[i]; v[i] = std::make_pair(count[i], &dummy[i]); }
    std::sort(v.begin(), v.end(), compare_nodes);
    for (int i = 0; i < n; i++) perm[i] = (int)(v[i].second - dummy.data());
}

}  // extern "C"
