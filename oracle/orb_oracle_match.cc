/*
 * orb_oracle_match.cc -- CPU ORACLE (matcher half).  TEST INFRASTRUCTURE ONLY -- see orb_oracle.h.
 *
 * Restates, over flattened arrays instead of the Frame/KeyFrame/MapPoint pointer graph:
 *   ORBmatcher::DescriptorDistance          /root/reference/src/ORBmatcher.cc:2058-2074
 *   ORBmatcher::ComputeThreeMaxima          ORBmatcher.cc:2012-2053
 *   Frame::AssignFeaturesToGrid / PosInGrid Frame.cc:385-416, 725-735
 *   Frame::GetFeaturesInArea                Frame.cc:657-723
 *   ORBmatcher::SearchByProjection (M1)     ORBmatcher.cc:43-213   (mono form, Nleft == -1)
 *   ORBmatcher::SearchByProjection (M2)     ORBmatcher.cc:1676-1887 (after the 3-D projection)
 *   Frame::ComputeStereoMatches (M8)        Frame.cc:811-981
 *   cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) as used by Frame.cc:1144 (M9) [OCV]
 */
#include "orb_oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

namespace {

const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30; /* ORBmatcher.cc:35-37 */
const int GRID_COLS = 64, GRID_ROWS = 48;                /* Frame.h:44-45 */

int descriptor_distance(const uint8_t *a, const uint8_t *b) { /* ORBmatcher.cc:2058-2074, SWAR popcount */
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

void three_maxima(const int *sizes, int L, int &ind1, int &ind2, int &ind3) { /* ORBmatcher.cc:2012-2053 */
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

}  // namespace

struct orbo_grid {
    float minx, maxx, miny, maxy, inv_w, inv_h;
    const orbo_keypoint *kps;
    int n;
    std::vector<int32_t> cell[GRID_COLS][GRID_ROWS];
};

namespace {

int grid_query(const orbo_grid *g, float x, float y, float r, int minLevel, int maxLevel, std::vector<int32_t> &out) {
    out.clear();
    const float factorX = r, factorY = r; /* Frame.cc:662-663 */
    const int nMinCellX = std::max(0, (int)std::floor((x - g->minx - factorX) * g->inv_w));
    if (nMinCellX >= GRID_COLS) return 0;
    const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x - g->minx + factorX) * g->inv_w));
    if (nMaxCellX < 0) return 0;
    const int nMinCellY = std::max(0, (int)std::floor((y - g->miny - factorY) * g->inv_h));
    if (nMinCellY >= GRID_ROWS) return 0;
    const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y - g->miny + factorY) * g->inv_h));
    if (nMaxCellY < 0) return 0;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<int32_t> &vCell = g->cell[ix][iy];
            for (int32_t idx : vCell) {
                const orbo_keypoint &kp = g->kps[idx];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                const float distx = kp.x - x, disty = kp.y - y;
                if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) out.push_back(idx);
            }
        }
    return (int)out.size();
}

}  // namespace

extern "C" {

int orbo_descriptor_distance(const uint8_t *a, const uint8_t *b) { return descriptor_distance(a, b); }

void orbo_three_maxima(const int *sizes, int L, int *ind1, int *ind2, int *ind3) {
    int a = -1, b = -1, c = -1;
    three_maxima(sizes, L, a, b, c);
    *ind1 = a; *ind2 = b; *ind3 = c;
}

orbo_grid *orbo_grid_create(const orbo_keypoint *kps, int n, float minx, float maxx, float miny, float maxy) {
    orbo_grid *g = new orbo_grid();
    g->minx = minx; g->maxx = maxx; g->miny = miny; g->maxy = maxy;
    g->inv_w = static_cast<float>(GRID_COLS) / (maxx - minx); /* Frame.cc:342-343 */
    g->inv_h = static_cast<float>(GRID_ROWS) / (maxy - miny);
    g->kps = kps; g->n = n;
    for (int i = 0; i < n; i++) { /* Frame.cc:403-415 + PosInGrid :725-735 */
        int px = (int)std::round((kps[i].x - minx) * g->inv_w);
        int py = (int)std::round((kps[i].y - miny) * g->inv_h);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
        g->cell[px][py].push_back(i);
    }
    return g;
}
void orbo_grid_destroy(orbo_grid *g) { delete g; }

int orbo_grid_query(const orbo_grid *g, float x, float y, float r, int min_level, int max_level, int32_t *out, int cap) {
    std::vector<int32_t> v;
    grid_query(g, x, y, r, min_level, max_level, v);
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

/* M1, ORBmatcher.cc:43-142 (left / mono branch; Nleft == -1) */
int orbo_search_by_projection_mappoints(const orbo_grid *grid, const orbo_keypoint *kps, const uint8_t *fdesc, int nF,
                                        const float *scale_factors, const float *u_right, const uint8_t *occupied,
                                        int n_mp, const float *proj_x, const float *proj_y, const float *proj_xr,
                                        const int32_t *pred_level, const float *view_cos, const uint8_t *mp_desc,
                                        const uint8_t *mp_in_view, const uint8_t *mp_has_obs, float th, float nnratio,
                                        int32_t *frame_match) {
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<uint8_t> occ(occupied ? std::vector<uint8_t>(occupied, occupied + nF) : std::vector<uint8_t>(nF, 0));
    for (int i = 0; i < nF; i++) frame_match[i] = -1;
    std::vector<int32_t> vIndices;
    for (int iMP = 0; iMP < n_mp; iMP++) {
        if (!mp_in_view[iMP]) continue; /* mbTrackInView; isBad / far points are folded into this flag by the caller */
        const int nPredictedLevel = pred_level[iMP];
        float r = (view_cos[iMP] > 0.998) ? 2.5f : 4.0f; /* RadiusByViewingCos :215-221 */
        if (bFactor) r *= th;
        grid_query(grid, proj_x[iMP], proj_y[iMP], r * scale_factors[nPredictedLevel], nPredictedLevel - 1,
                   nPredictedLevel, vIndices);
        if (vIndices.empty()) continue;
        const uint8_t *MPdescriptor = mp_desc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int32_t idx : vIndices) {
            if (occ[idx]) continue; /* F.mvpMapPoints[idx] && Observations()>0 : read-after-write inside the loop */
            if (u_right && u_right[idx] > 0) {
                const float er = std::fabs(proj_xr[iMP] - u_right[idx]);
                if (er > r * scale_factors[nPredictedLevel]) continue;
            }
            const int dist = descriptor_distance(MPdescriptor, fdesc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = kps[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = kps[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                frame_match[bestIdx] = iMP;
                occ[bestIdx] = mp_has_obs ? mp_has_obs[iMP] : 1;
                nmatches++;
            }
        }
    }
    return nmatches;
}

/* M1, the fisheye-stereo form (F.Nleft != -1), ORBmatcher.cc:43-213 whole: per map point the left search (:60-142) and then its
 * right-camera twin (:144-210) in the same loop iteration.  Features are indexed as the reference indexes them: [0, n_left) left,
 * [n_left, n_left + n_right) right; kps_left = F.mvKeys, kps_right = F.mvKeysRight, desc / occupied / frame_match cover all of them.
 * l2r / r2l = F.mvLeftToRightMatch / F.mvRightToLeftMatch (-1 = no stereo partner): an accepted match is written to the partner
 * slot as well and counts twice (:131-135, :197-201).  Differences from the left search that the twin really has: its radius is
 * RadiusByViewingCos(mTrackViewCosR) WITHOUT the th factor (:147), it needs mnTrackScaleLevelR != -1 (:146), and the stereo
 * coordinate gate (:92-97) exists only in the monocular form. */
int orbo_search_by_projection_mappoints_fisheye(const orbo_grid *grid_left, const orbo_grid *grid_right, const orbo_keypoint *kps_left,
                                                int n_left, const orbo_keypoint *kps_right, int n_right, const uint8_t *desc,
                                                const float *scale_factors, const int32_t *l2r, const int32_t *r2l,
                                                const uint8_t *occupied, int n_mp, const uint8_t *in_view, const float *proj_x,
                                                const float *proj_y, const int32_t *level, const float *view_cos,
                                                const uint8_t *in_view_r, const float *proj_xr, const float *proj_yr,
                                                const int32_t *level_r, const float *view_cos_r, const uint8_t *mp_desc,
                                                const uint8_t *mp_has_obs, float th, float nnratio, int32_t *frame_match) {
    const int N = n_left + n_right;
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<uint8_t> occ(occupied ? std::vector<uint8_t>(occupied, occupied + N) : std::vector<uint8_t>(N, 0));
    for (int i = 0; i < N; i++) frame_match[i] = -1;
    std::vector<int32_t> vIndices;
    auto assign = [&](int slot, int iMP) { frame_match[slot] = iMP; occ[slot] = mp_has_obs ? mp_has_obs[iMP] : 1; };
    for (int iMP = 0; iMP < n_mp; iMP++) {
        if (!in_view[iMP] && !in_view_r[iMP]) continue; /* :52-53 */
        const uint8_t *MPdescriptor = mp_desc + (size_t)iMP * 32;
        if (in_view[iMP]) {
            const int nPredictedLevel = level[iMP];
            float r = (view_cos[iMP] > 0.998) ? 2.5f : 4.0f;
            if (bFactor) r *= th;
            grid_query(grid_left, proj_x[iMP], proj_y[iMP], r * scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
            if (!vIndices.empty()) {
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (int32_t idx : vIndices) {
                    if (occ[idx]) continue;
                    const int dist = descriptor_distance(MPdescriptor, desc + (size_t)idx * 32);
                    if (dist < bestDist) {
                        bestDist2 = bestDist; bestDist = dist;
                        bestLevel2 = bestLevel; bestLevel = kps_left[idx].octave; /* idx < Nleft: F.mvKeys[idx] */
                        bestIdx = idx;
                    } else if (dist < bestDist2) {
                        bestLevel2 = kps_left[idx].octave;
                        bestDist2 = dist;
                    }
                }
                if (bestDist <= TH_HIGH) {
                    const bool rejected = bestLevel == bestLevel2 && bestDist > nnratio * bestDist2; /* :125-126: `continue` skips the twin too */
                    if (rejected) continue;
                    if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                        assign(bestIdx, iMP);
                        if (l2r[bestIdx] != -1) { assign(l2r[bestIdx] + n_left, iMP); nmatches++; }
                        nmatches++;
                    }
                }
            }
        }
        if (in_view_r[iMP]) {
            const int nPredictedLevel = level_r[iMP];
            if (nPredictedLevel != -1) {
                const float r = (view_cos_r[iMP] > 0.998) ? 2.5f : 4.0f;
                grid_query(grid_right, proj_xr[iMP], proj_yr[iMP], r * scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
                if (vIndices.empty()) continue;
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (int32_t idx : vIndices) {
                    if (occ[idx + n_left]) continue;
                    const int dist = descriptor_distance(MPdescriptor, desc + (size_t)(idx + n_left) * 32);
                    if (dist < bestDist) {
                        bestDist2 = bestDist; bestDist = dist;
                        bestLevel2 = bestLevel; bestLevel = kps_right[idx].octave;
                        bestIdx = idx;
                    } else if (dist < bestDist2) {
                        bestLevel2 = kps_right[idx].octave;
                        bestDist2 = dist;
                    }
                }
                if (bestDist <= TH_HIGH) {
                    if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                    if (r2l[bestIdx] != -1) { assign(r2l[bestIdx], iMP); nmatches++; }
                    assign(bestIdx + n_left, iMP);
                    nmatches++;
                }
            }
        }
    }
    return nmatches;
}

/* M2, ORBmatcher.cc:1676-1887 (mono form), queries = last-frame map points already projected */
int orbo_search_by_projection_frame(const orbo_grid *grid, const orbo_keypoint *ckps, const uint8_t *cdesc, int nC,
                                    const float *scale_factors, const float *cur_u_right, const uint8_t *cur_occupied,
                                    int n_q, const float *q_u, const float *q_v, const float *q_ur,
                                    const int32_t *q_octave, const float *q_angle, const uint8_t *q_desc,
                                    const uint8_t *q_has_obs, float th, int mode, int check_orientation,
                                    int32_t *cur_match) {
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH; /* :1684 -- NB: 1/30, so only bins 0..12 are reachable */
    std::vector<uint8_t> occ(cur_occupied ? std::vector<uint8_t>(cur_occupied, cur_occupied + nC)
                                          : std::vector<uint8_t>(nC, 0));
    for (int i = 0; i < nC; i++) cur_match[i] = -1;
    std::vector<int32_t> vIndices2;
    for (int i = 0; i < n_q; i++) {
        const int nLastOctave = q_octave[i];
        const float radius = th * scale_factors[nLastOctave];
        if (mode == 1) grid_query(grid, q_u[i], q_v[i], radius, nLastOctave, -1, vIndices2);
        else if (mode == 2) grid_query(grid, q_u[i], q_v[i], radius, 0, nLastOctave, vIndices2);
        else grid_query(grid, q_u[i], q_v[i], radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = q_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int32_t i2 : vIndices2) {
            if (occ[i2]) continue;
            if (cur_u_right && cur_u_right[i2] > 0) {
                const float er = std::fabs(q_ur[i] - cur_u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = descriptor_distance(dMP, cdesc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur_match[bestIdx2] = i;
            occ[bestIdx2] = q_has_obs ? q_has_obs[i] : 1;
            nmatches++;
            if (check_orientation) {
                float rot = q_angle[i] - ckps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) { /* :1865-1884 */
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { cur_match[idx] = -1; nmatches--; }
    }
    return nmatches;
}

/* M2 in its fisheye-stereo form (CurrentFrame.Nleft != -1), ORBmatcher.cc:1676-1885 whole: per last-frame map point the left
 * search and then the right-camera twin (:1794-1863) at the point's projection into the right camera (q_xr, q_yr), same radius
 * and level window, occupancy and result slots offset by n_left, rotation entries pushed for both.  An EMPTY left window skips
 * the twin as well (`continue`, :1738-1739); the stereo-coordinate gate (:1755-1761) exists only in the monocular form. */
int orbo_search_by_projection_frame_fisheye(const orbo_grid *grid_left, const orbo_grid *grid_right, const orbo_keypoint *kps_left,
                                            int n_left, const orbo_keypoint *kps_right, int n_right, const uint8_t *cdesc,
                                            const float *scale_factors, const uint8_t *cur_occupied, int n_q, const float *q_u,
                                            const float *q_v, const float *q_xr, const float *q_yr, const int32_t *q_octave,
                                            const float *q_angle, const uint8_t *q_desc, const uint8_t *q_has_obs, float th, int mode,
                                            int check_orientation, int32_t *cur_match) {
    const int nC = n_left + n_right;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<uint8_t> occ(cur_occupied ? std::vector<uint8_t>(cur_occupied, cur_occupied + nC) : std::vector<uint8_t>(nC, 0));
    for (int i = 0; i < nC; i++) cur_match[i] = -1;
    std::vector<int32_t> vIndices2;
    auto window = [&](const orbo_grid *g, float x, float y, float radius, int oct) {
        if (mode == 1) grid_query(g, x, y, radius, oct, -1, vIndices2);
        else if (mode == 2) grid_query(g, x, y, radius, 0, oct, vIndices2);
        else grid_query(g, x, y, radius, oct - 1, oct + 1, vIndices2);
    };
    auto push = [&](float a_last, float a_cur, int slot) {
        float rot = a_last - a_cur;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(slot);
    };
    for (int i = 0; i < n_q; i++) {
        const int nLastOctave = q_octave[i];
        const float radius = th * scale_factors[nLastOctave];
        const uint8_t *dMP = q_desc + (size_t)i * 32;
        const uint8_t obs = q_has_obs ? q_has_obs[i] : 1;
        window(grid_left, q_u[i], q_v[i], radius, nLastOctave);
        if (vIndices2.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int32_t i2 : vIndices2) {
            if (occ[i2]) continue;
            const int dist = descriptor_distance(dMP, cdesc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur_match[bestIdx2] = i; occ[bestIdx2] = obs;
            nmatches++;
            if (check_orientation) push(q_angle[i], kps_left[bestIdx2].angle, bestIdx2);
        }
        window(grid_right, q_xr[i], q_yr[i], radius, nLastOctave);
        bestDist = 256; bestIdx2 = -1;
        for (int32_t i2 : vIndices2) {
            if (occ[i2 + n_left]) continue;
            const int dist = descriptor_distance(dMP, cdesc + (size_t)(i2 + n_left) * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur_match[bestIdx2 + n_left] = i; occ[bestIdx2 + n_left] = obs;
            nmatches++;
            if (check_orientation) push(q_angle[i], kps_right[bestIdx2].angle, bestIdx2 + n_left);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { cur_match[idx] = -1; nmatches--; }
    }
    return nmatches;
}

/* M3 (ORBmatcher.cc:1889-2010) / M4 (ORBmatcher.cc:427-646) common form */
int orbo_search_by_projection_window(const orbo_grid *grid, const orbo_keypoint *kps, const uint8_t *desc, int n,
                                     const uint8_t *occupied, int n_q, const float *q_x, const float *q_y,
                                     const float *q_r, const int32_t *q_min, const int32_t *q_max, const float *q_angle,
                                     const uint8_t *q_desc, const uint8_t *q_has_obs, float max_dist,
                                     int check_orientation, int level_gate_in_loop, int32_t *match) {
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<uint8_t> occ(occupied ? std::vector<uint8_t>(occupied, occupied + n) : std::vector<uint8_t>(n, 0));
    for (int i = 0; i < n; i++) match[i] = -1;
    std::vector<int32_t> vIndices;
    for (int i = 0; i < n_q; i++) {
        if (level_gate_in_loop) grid_query(grid, q_x[i], q_y[i], q_r[i], -1, -1, vIndices);  /* KeyFrame::GetFeaturesInArea */
        else grid_query(grid, q_x[i], q_y[i], q_r[i], q_min[i], q_max[i], vIndices);
        if (vIndices.empty()) continue;
        const uint8_t *dMP = q_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (int32_t idx : vIndices) {
            if (occ[idx]) continue;
            if (level_gate_in_loop) {
                const int kpLevel = kps[idx].octave;
                if (kpLevel < q_min[i] || kpLevel > q_max[i]) continue;
            }
            const int dist = descriptor_distance(dMP, desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if ((float)bestDist <= max_dist) {
            match[bestIdx] = i;
            occ[bestIdx] = q_has_obs ? q_has_obs[i] : 1;
            nmatches++;
            if (check_orientation) {
                float rot = q_angle[i] - kps[bestIdx].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx);
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { match[idx] = -1; nmatches--; }
    }
    return nmatches;
}

/* M6, ORBmatcher.cc:648-763 */
int orbo_search_for_initialization(const orbo_keypoint *kps1, const uint8_t *desc1, int n1, const orbo_grid *grid2,
                                   const orbo_keypoint *kps2, const uint8_t *desc2, int n2, float *prev, int windowSize,
                                   float nnratio, int check_orientation, int32_t *vnMatches12) {
    int nmatches = 0;
    for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(n2, INT_MAX), vnMatches21(n2, -1);
    std::vector<int32_t> vIndices2;
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = kps1[i1].octave;
        if (level1 > 0) continue;
        grid_query(grid2, prev[2 * i1], prev[2 * i1 + 1], (float)windowSize, level1, level1, vIndices2);
        if (vIndices2.empty()) continue;
        const uint8_t *d1 = desc1 + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int32_t i2 : vIndices2) {
            const int dist = descriptor_distance(d1, desc2 + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    float rot = kps1[i1].angle - kps2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++) /* :757-760 update prev matched */
        if (vnMatches12[i1] >= 0) { prev[2 * i1] = kps2[vnMatches12[i1]].x; prev[2 * i1 + 1] = kps2[vnMatches12[i1]].y; }
    return nmatches;
}

}  // extern "C"

namespace {
/* the while loop over two DBoW2::FeatureVector maps (e.g. ORBmatcher.cc:239-381): visit equal node ids in ascending order */
template <class F> void join_nodes(const orbo_featvec *a, const orbo_featvec *b, F visit) {
    int ia = 0, ib = 0;
    while (ia < a->n_nodes && ib < b->n_nodes) {
        if (a->node_id[ia] == b->node_id[ib]) { visit(ia, ib); ia++; ib++; }
        else if (a->node_id[ia] < b->node_id[ib]) { while (ia < a->n_nodes && a->node_id[ia] < b->node_id[ib]) ia++; } /* lower_bound */
        else { while (ib < b->n_nodes && b->node_id[ib] < a->node_id[ia]) ib++; }
    }
}
inline int rot_bin(float a1, float a2) {
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}
}  // namespace

extern "C" {

/* M5, ORBmatcher.cc:223-425 (F.Nleft == -1) */
int orbo_search_by_bow_frame(const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid, int n_kf,
                             const orbo_featvec *kf_fv, const uint8_t *f_desc, const float *f_angle, int n_f,
                             const orbo_featvec *f_fv, float nnratio, int check_orientation, int32_t *f_match) {
    int nmatches = 0;
    for (int i = 0; i < n_f; i++) f_match[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    join_nodes(kf_fv, f_fv, [&](int ik, int jf) {
        for (int a = kf_fv->node_ptr[ik]; a < kf_fv->node_ptr[ik + 1]; a++) {
            const int realIdxKF = kf_fv->index[a];
            if (!kf_valid[realIdxKF]) continue; /* !pMP || pMP->isBad() */
            const uint8_t *dKF = kf_desc + (size_t)realIdxKF * 32;
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
            for (int b = f_fv->node_ptr[jf]; b < f_fv->node_ptr[jf + 1]; b++) {
                const int realIdxF = f_fv->index[b];
                if (f_match[realIdxF] >= 0) continue; /* vpMapPointMatches[realIdxF] */
                const int dist = descriptor_distance(dKF, f_desc + (size_t)realIdxF * 32);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 <= TH_LOW) {
                if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    f_match[bestIdxF] = realIdxKF;
                    if (check_orientation) rotHist[rot_bin(kf_angle[realIdxKF], f_angle[bestIdxF])].push_back(bestIdxF);
                    nmatches++;
                }
            }
        }
    });
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { f_match[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

/* M5 (KeyFrame -> Frame) in its fisheye-stereo form (F.Nleft != -1), ORBmatcher.cc:283-392: the frame's features of a node are
 * split by index into left (< n_f_left) and right; best / second-best are kept per camera; the right match is considered only
 * INSIDE the branch of a left best <= TH_LOW, and its ratio test is disabled by `|| true` (:359). */
int orbo_search_by_bow_frame_fisheye(const uint8_t *kf_desc, const float *kf_angle, const uint8_t *kf_valid, int n_kf,
                                     const orbo_featvec *kf_fv, const uint8_t *f_desc, const float *f_angle, int n_f, int n_f_left,
                                     const orbo_featvec *f_fv, float nnratio, int check_orientation, int32_t *f_match) {
    int nmatches = 0;
    for (int i = 0; i < n_f; i++) f_match[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    join_nodes(kf_fv, f_fv, [&](int ik, int jf) {
        for (int a = kf_fv->node_ptr[ik]; a < kf_fv->node_ptr[ik + 1]; a++) {
            const int realIdxKF = kf_fv->index[a];
            if (!kf_valid[realIdxKF]) continue;
            const uint8_t *dKF = kf_desc + (size_t)realIdxKF * 32;
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
            int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
            for (int b = f_fv->node_ptr[jf]; b < f_fv->node_ptr[jf + 1]; b++) {
                const int realIdxF = f_fv->index[b];
                if (f_match[realIdxF] >= 0) continue;
                const int dist = descriptor_distance(dKF, f_desc + (size_t)realIdxF * 32);
                if (realIdxF < n_f_left && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                else if (realIdxF < n_f_left && dist < bestDist2) bestDist2 = dist;
                if (realIdxF >= n_f_left && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
                else if (realIdxF >= n_f_left && dist < bestDist2R) bestDist2R = dist;
            }
            if (bestDist1 <= TH_LOW) {
                if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    f_match[bestIdxF] = realIdxKF;
                    if (check_orientation) rotHist[rot_bin(kf_angle[realIdxKF], f_angle[bestIdxF])].push_back(bestIdxF);
                    nmatches++;
                }
                if (bestDist1R <= TH_LOW) { /* ratio test `|| true` */
                    f_match[bestIdxFR] = realIdxKF;
                    if (check_orientation) rotHist[rot_bin(kf_angle[realIdxKF], f_angle[bestIdxFR])].push_back(bestIdxFR);
                    nmatches++;
                }
            }
        }
    });
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { f_match[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

/* M5, ORBmatcher.cc:765-905 */
int orbo_search_by_bow_keyframes(const uint8_t *desc1, const float *angle1, const uint8_t *valid1, int n1,
                                 const orbo_featvec *fv1, const uint8_t *desc2, const float *angle2,
                                 const uint8_t *valid2, int n2, const orbo_featvec *fv2, float nnratio,
                                 int check_orientation, int32_t *match12) {
    int nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<uint8_t> vbMatched2(n2, 0);
    std::vector<int> rotHist[HISTO_LENGTH];
    join_nodes(fv1, fv2, [&](int i1n, int i2n) {
        for (int a = fv1->node_ptr[i1n]; a < fv1->node_ptr[i1n + 1]; a++) {
            const int idx1 = fv1->index[a];
            if (!valid1[idx1]) continue;
            const uint8_t *d1 = desc1 + (size_t)idx1 * 32;
            int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
            for (int b = fv2->node_ptr[i2n]; b < fv2->node_ptr[i2n + 1]; b++) {
                const int idx2 = fv2->index[b];
                if (vbMatched2[idx2] || !valid2[idx2]) continue;
                const int dist = descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 < TH_LOW) {
                if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    match12[idx1] = bestIdx2;
                    vbMatched2[bestIdx2] = 1;
                    if (check_orientation) rotHist[rot_bin(angle1[idx1], angle2[bestIdx2])].push_back(idx1);
                    nmatches++;
                }
            }
        }
    });
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { match12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

/* M7, ORBmatcher.cc:907-1146.  NB vbMatched2 is never set in v1.0, so queries do not interact. */
int orbo_search_for_triangulation(const uint8_t *desc1, const float *angle1, const uint8_t *skip1, int n1,
                                  const orbo_featvec *fv1, const uint8_t *desc2, const float *angle2,
                                  const uint8_t *skip2, int n2, const orbo_featvec *fv2, int check_orientation,
                                  orbo_pair_predicate pair_ok, void *user, int32_t *vMatches12) {
    int nmatches = 0;
    for (int i = 0; i < n1; i++) vMatches12[i] = -1;
    std::vector<uint8_t> vbMatched2(n2, 0);
    std::vector<int> rotHist[HISTO_LENGTH];
    join_nodes(fv1, fv2, [&](int i1n, int i2n) {
        for (int a = fv1->node_ptr[i1n]; a < fv1->node_ptr[i1n + 1]; a++) {
            const int idx1 = fv1->index[a];
            if (skip1[idx1]) continue;
            const uint8_t *d1 = desc1 + (size_t)idx1 * 32;
            int bestDist = TH_LOW, bestIdx2 = -1;
            for (int b = fv2->node_ptr[i2n]; b < fv2->node_ptr[i2n + 1]; b++) {
                const int idx2 = fv2->index[b];
                if (vbMatched2[idx2] || skip2[idx2]) continue;
                const int dist = descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                if (dist > TH_LOW || dist > bestDist) continue;
                if (!pair_ok || pair_ok(user, idx1, idx2)) { bestIdx2 = idx2; bestDist = dist; }
            }
            if (bestIdx2 >= 0) {
                vMatches12[idx1] = bestIdx2;
                nmatches++;
                if (check_orientation) rotHist[rot_bin(angle1[idx1], angle2[bestIdx2])].push_back(idx1);
            }
        }
    });
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { vMatches12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

/* The two geometric gates of M7 for pinhole cameras, as the reference binary evaluates them.
 * (1) epipole distance, ORBmatcher.cc:1026-1034 (only when neither feature is stereo): distex*distex + distey*distey <
 *     100 * mvScaleFactors[kp2.octave] rejects.
 * (2) Pinhole::epipolarConstrain, CameraModels/Pinhole.cpp:107-129, with F12 = K1^-T [t12]x R12 K2^-1 supplied by the caller:
 *     a = x1 F00 + y1 F10 + F20 (b, c alike), num = a x2 + b y2 + c, den = a a + b b, dsqr = num num / den, pass iff den != 0 and
 *     (double)dsqr < 3.84 * (double)unc, unc = pKF2->mvLevelSigma2[kp2.octave].
 * fma_mode = 0: every operation rounds separately (the source as written).  fma_mode = 1: the contraction GCC 11 -O3 with FMA
 * (-ffp-contract=fast, the reference's build flags) applies to THIS text -- read off the compiled reference
 * (oracle/_ref/libframe_ref.so) and pinned against it on near-threshold pairs (tests/test_oracle_frame_vs_reference.py).  Which
 * product of a sum gets fused is the compiler's choice, not the source's, and it is not uniform:
 *   a = fma(x1, F00, y1*F10) + F20      b = fma(x1, F01, y1*F11) + F21      c = fma(y1, F12, x1*F02) + F22
 *   num = fma(b, y2, a*x2) + c          den = fma(a, a, b*b)
 * A build of the reference against real Eigen may fuse differently; callers that need the unfused semantics pass 0. */
int orbo_epipolar_pinhole(const float *F, float x1, float y1, float x2, float y2, float unc, int fma_mode) {
    float a, b, c, num, den;
    if (fma_mode) {
        a = std::fma(x1, F[0], y1 * F[3]) + F[6];
        b = std::fma(x1, F[1], y1 * F[4]) + F[7];
        c = std::fma(y1, F[5], x1 * F[2]) + F[8];
        num = std::fma(b, y2, a * x2) + c;
        den = std::fma(a, a, b * b);
    } else {
        a = (x1 * F[0] + y1 * F[3]) + F[6];
        b = (x1 * F[1] + y1 * F[4]) + F[7];
        c = (x1 * F[2] + y1 * F[5]) + F[8];
        num = (a * x2 + b * y2) + c;
        den = a * a + b * b;
    }
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)unc;
}

namespace {
struct PinholeGate {
    const orbo_keypoint *k1, *k2;
    const float *ur1, *ur2;           /* mvuRight or NULL (all < 0) */
    const float *scale2, *sigma2_2;   /* pKF2->mvScaleFactors, pKF2->mvLevelSigma2 */
    const float *F;
    float ex, ey;
    int fma_mode, coarse;
};
int pinhole_gate(void *user, int idx1, int idx2) {
    const PinholeGate &g = *(const PinholeGate *)user;
    const bool bStereo1 = g.ur1 && g.ur1[idx1] >= 0, bStereo2 = g.ur2 && g.ur2[idx2] >= 0;
    const orbo_keypoint &kp1 = g.k1[idx1], &kp2 = g.k2[idx2];
    if (!bStereo1 && !bStereo2) {
        const float distex = g.ex - kp2.x, distey = g.ey - kp2.y;
        const float d2 = g.fma_mode ? std::fma(distex, distex, distey * distey) : distex * distex + distey * distey;
        if (d2 < 100 * g.scale2[kp2.octave]) return 0;
    }
    if (g.coarse) return 1;
    return orbo_epipolar_pinhole(g.F, kp1.x, kp1.y, kp2.x, kp2.y, g.sigma2_2[kp2.octave], g.fma_mode);
}
}  // namespace

/* M7 with both gates evaluated here (pinhole key frames): kps*: mvKeysUn, u_right*: mvuRight (NULL = monocular), ep = epipole of
 * camera 1 in image 2 (:921), F12 row-major, coarse = bCoarse.  Everything else as orbo_search_for_triangulation. */
int orbo_search_for_triangulation_pinhole(const orbo_keypoint *kps1, const uint8_t *desc1, const uint8_t *skip1, const float *u_right1,
                                          int n1, const orbo_featvec *fv1, const orbo_keypoint *kps2, const uint8_t *desc2,
                                          const uint8_t *skip2, const float *u_right2, int n2, const orbo_featvec *fv2,
                                          const float *scale_factors2, const float *level_sigma2_2, const float *F12, float ep_x,
                                          float ep_y, int coarse, int check_orientation, int fma_mode, int32_t *matches12) {
    std::vector<float> a1(n1), a2(n2);
    for (int i = 0; i < n1; i++) a1[i] = kps1[i].angle;
    for (int i = 0; i < n2; i++) a2[i] = kps2[i].angle;
    PinholeGate g{kps1, kps2, u_right1, u_right2, scale_factors2, level_sigma2_2, F12, ep_x, ep_y, fma_mode, coarse};
    return orbo_search_for_triangulation(desc1, a1.data(), skip1, n1, fv1, desc2, a2.data(), skip2, n2, fv2, check_orientation,
                                         pinhole_gate, &g, matches12);
}

/* M7 between key frames of a fisheye rig (ORBmatcher.cc:1036-1072): the pair's cameras select the KannalaBrandt8 parameters and the relative pose
 * (R12 / t12 [2 * right1 + right2] = ll, lr, rl, rr of :934-944); KannalaBrandt8::epipolarConstrain evaluated lazily where the reference evaluates it.
 * cam1 / cam2: [2][8] parameters of (mpCamera, mpCamera2) of KF1 / KF2. */
namespace {
struct Kb8GateO {
    const orbo_keypoint *k1, *k2;
    int n_left1, n_left2;
    const float *sigma2_1, *sigma2_2, *cam1, *cam2, *R12, *t12;
    int coarse;
};
int kb8_gate(void *user, int i1, int i2) {
    const Kb8GateO &g = *(const Kb8GateO *)user;
    if (g.coarse) return 1;
    const int r1 = i1 >= g.n_left1 ? 1 : 0, r2 = i2 >= g.n_left2 ? 1 : 0, sel = 2 * r1 + r2;
    return orbo_kb8_triangulate_matches(g.cam1 + 8 * r1, g.cam2 + 8 * r2, g.k1[i1].x, g.k1[i1].y, g.k2[i2].x, g.k2[i2].y, g.R12 + 9 * sel, g.t12 + 3 * sel,
                                        g.sigma2_1[g.k1[i1].octave], g.sigma2_2[g.k2[i2].octave]) > 0.0001f;
}
}  // namespace
int orbo_search_for_triangulation_kb8(const orbo_keypoint *kps1, int n_left1, const uint8_t *desc1, const uint8_t *skip1, int n1, const orbo_featvec *fv1,
                                      const orbo_keypoint *kps2, int n_left2, const uint8_t *desc2, const uint8_t *skip2, int n2, const orbo_featvec *fv2,
                                      const float *level_sigma2_1, const float *level_sigma2_2, const float *cam1, const float *cam2, const float *R12,
                                      const float *t12, int coarse, int check_orientation, int32_t *matches12) {
    std::vector<float> a1(n1), a2(n2);
    for (int i = 0; i < n1; i++) a1[i] = kps1[i].angle;
    for (int i = 0; i < n2; i++) a2[i] = kps2[i].angle;
    Kb8GateO g{kps1, kps2, n_left1, n_left2, level_sigma2_1, level_sigma2_2, cam1, cam2, R12, t12, coarse};
    return orbo_search_for_triangulation(desc1, a1.data(), skip1, n1, fv1, desc2, a2.data(), skip2, n2, fv2, check_orientation, kb8_gate, &g, matches12);
}

/* M8, Frame.cc:811-981 */
int orbo_compute_stereo_matches(const orbo_keypoint *kl, const uint8_t *dl, int N, const orbo_keypoint *kr,
                                const uint8_t *dr, int Nr, const float *scale_factors, const float *inv_scale_factors,
                                int nlevels, const uint8_t *const *pyr_left, const uint8_t *const *pyr_right,
                                const int *pyr_w, const int *pyr_h, const size_t *pyr_stride, float bf, float b,
                                float *u_right, float *depth, int32_t *best_idx_r, int32_t *best_dist) {
    for (int i = 0; i < N; i++) { u_right[i] = -1.0f; depth[i] = -1.0f; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = pyr_h[0];
    std::vector<std::vector<int>> vRowIndices(nRows);
    for (int iR = 0; iR < Nr; iR++) { /* :824-838 */
        const float kpY = kr[iR].y;
        const float r = 2.0f * scale_factors[kr[iR].octave];
        const int maxr = (int)std::ceil(kpY + r), minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR); /* unguarded in the reference (Appendix A.8) */
    }
    const float minZ = b, minD = 0, maxD = bf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    vDistIdx.reserve(N);
    int nmatched = 0;
    for (int iL = 0; iL < N; iL++) {
        if (best_idx_r) best_idx_r[iL] = -1;
        if (best_dist) best_dist[iL] = TH_HIGH;
        const orbo_keypoint &kpL = kl[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const std::vector<int> &vCandidates = vRowIndices[(int)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        bool any = false;
        const uint8_t *dL = dl + (size_t)iL * 32;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const int iR = vCandidates[iC];
            const orbo_keypoint &kpR = kr[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = descriptor_distance(dL, dr + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; any = true; }
            }
        }
        if (any) {
            if (best_idx_r) best_idx_r[iL] = (int)bestIdxR;
            if (best_dist) best_dist[iL] = bestDist;
        }
        if (bestDist < thOrbDist) { /* :897-964 sub-pixel SAD refinement */
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = inv_scale_factors[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const int lvl = kpL.octave;
            int bestDistS = INT_MAX, bestincR = 0;
            std::vector<float> vDists(2 * L + 1);
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= pyr_w[lvl]) continue;
            const uint8_t *IL = pyr_left[lvl], *IRb = pyr_right[lvl];
            const size_t sl = pyr_stride[lvl];
            const int y0 = (int)(scaledvL - w), xl0 = (int)(scaleduL - w);
            for (int incR = -L; incR <= +L; incR++) {
                const int xr0 = (int)(scaleduR0 + incR - w);
                int sad = 0; /* cv::norm(IL, IR, NORM_L1) on 8U: exact integer sum */
                for (int yy = 0; yy < 2 * w + 1; yy++)
                    for (int xx = 0; xx < 2 * w + 1; xx++)
                        sad += std::abs((int)IL[(size_t)(y0 + yy) * sl + xl0 + xx] -
                                        (int)IRb[(size_t)(y0 + yy) * sl + xr0 + xx]);
                float dist = (float)sad;
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = scale_factors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01f; bestuR = uL - 0.01f; }
                depth[iL] = bf / disparity;
                u_right[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
                nmatched++;
            }
        }
    }
    if (vDistIdx.empty()) return 0; /* the reference indexes an empty vector here */
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        u_right[vDistIdx[i].second] = -1;
        depth[vDistIdx[i].second] = -1;
        nmatched--;
    }
    return nmatched;
}

/* TemplatedVocabulary.h:1206-1250 */
void orbo_bow_transform(const int32_t *child_ptr, const int32_t *child_idx, const uint8_t *node_desc, const int32_t *word_id,
                        int L, int levelsup, const uint8_t *desc, int n, int32_t *word_out, int32_t *node_out) {
    const int nid_level = L - levelsup;
    for (int i = 0; i < n; i++) {
        const uint8_t *feature = desc + (size_t)i * 32;
        if (nid_level <= 0) node_out[i] = 0; /* root */
        int final_id = 0, current_level = 0;
        do {
            ++current_level;
            const int b = child_ptr[final_id], e = child_ptr[final_id + 1];
            final_id = child_idx[b];
            double best_d = descriptor_distance(feature, node_desc + (size_t)final_id * 32);
            for (int c = b + 1; c < e; c++) {
                const int id = child_idx[c];
                const double d = descriptor_distance(feature, node_desc + (size_t)id * 32);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) node_out[i] = final_id;
        } while (child_ptr[final_id + 1] > child_ptr[final_id]); /* !isLeaf() */
        word_out[i] = word_id[final_id];
    }
}

/* ORBmatcher.cc:1246-1306 and :1405-1433 */
void orbo_fuse_search(const orbo_grid *grid, const orbo_keypoint *kps, const uint8_t *desc, int n, const float *u_right,
                      const float *inv_sigma2, int n_q, const float *q_u, const float *q_v, const float *q_ur,
                      const float *q_r, const int32_t *q_level, const uint8_t *q_desc, int fma_mode, int32_t *best_idx,
                      int32_t *best_dist) {
    std::vector<int32_t> vIndices;
    for (int i = 0; i < n_q; i++) {
        best_idx[i] = -1; best_dist[i] = 256;
        const int nPredictedLevel = q_level[i];
        grid_query(grid, q_u[i], q_v[i], q_r[i], -1, -1, vIndices); /* KeyFrame::GetFeaturesInArea: no level test */
        if (vIndices.empty()) continue;
        const uint8_t *dMP = q_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (int32_t idx : vIndices) {
            const orbo_keypoint &kp = kps[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (inv_sigma2) {
                if (u_right && u_right[idx] >= 0) { /* stereo reprojection error :1266-1278 */
                    const float ex = q_u[i] - kp.x, ey = q_v[i] - kp.y, er = q_ur[i] - u_right[idx];
                    const float e2 = fma_mode ? fmaf(er, er, fmaf(ex, ex, ey * ey)) : ex * ex + ey * ey + er * er;
                    if (e2 * inv_sigma2[kpLevel] > 7.8) continue;
                } else { /* :1281-1289 */
                    const float ex = q_u[i] - kp.x, ey = q_v[i] - kp.y;
                    const float e2 = fma_mode ? fmaf(ex, ex, ey * ey) : ex * ex + ey * ey;
                    if (e2 * inv_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = descriptor_distance(dMP, desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist;
    }
}

/* MapPoint.cc:369-397 */
void orbo_distinctive_descriptors(const uint8_t *desc, const int32_t *set_ptr, int n_sets, int32_t *best_idx) {
    for (int s = 0; s < n_sets; s++) {
        const int N = set_ptr[s + 1] - set_ptr[s];
        const uint8_t *D = desc + (size_t)set_ptr[s] * 32;
        if (N <= 0) { best_idx[s] = -1; continue; }
        std::vector<float> Distances((size_t)N * N);
        for (int i = 0; i < N; i++) {
            Distances[(size_t)i * N + i] = 0;
            for (int j = i + 1; j < N; j++) {
                int distij = descriptor_distance(D + (size_t)i * 32, D + (size_t)j * 32);
                Distances[(size_t)i * N + j] = (float)distij;
                Distances[(size_t)j * N + i] = (float)distij;
            }
        }
        int BestMedian = INT_MAX, BestIdx = 0;
        for (int i = 0; i < N; i++) {
            std::vector<int> vDists(Distances.begin() + (size_t)i * N, Distances.begin() + (size_t)(i + 1) * N);
            std::sort(vDists.begin(), vDists.end());
            int median = vDists[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = i; }
        }
        best_idx[s] = BestIdx;
    }
}

/* M9 [OCV]: BFMatcher(NORM_HAMMING).knnMatch(k=2): ascending train scan, strict '<' insertion => the
 * lower train index wins ties; Frame.cc:1144 */
void orbo_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist) {
    for (int i = 0; i < nq; i++) {
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            int d = descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
            else if (d < d1) { d1 = d; i1 = j; }
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        dist[2 * i] = i0 < 0 ? -1 : d0; dist[2 * i + 1] = i1 < 0 ? -1 : d1;
    }
}

/* Frame::ComputeStereoFishEyeMatches, Frame.cc:1126-1166 */
int orbo_stereo_fisheye_matches(const orbo_keypoint *kp_left, const uint8_t *desc_left, int n_left, int mono_left,
                                const orbo_keypoint *kp_right, const uint8_t *desc_right, int n_right, int mono_right,
                                const float *level_sigma2, orbo_triangulate_fn triangulate, void *ctx, int32_t *l2r, int32_t *r2l,
                                float *depth, float *u_right, float *p3d, int *desc_matches) {
    for (int i = 0; i < n_left; i++) { l2r[i] = -1; depth[i] = -1.0f; u_right[i] = -1.0f; p3d[3 * i] = p3d[3 * i + 1] = p3d[3 * i + 2] = 0.f; }  /* :1134-1138 */
    for (int i = 0; i < n_right; i++) r2l[i] = -1;
    const int nq = n_left - mono_left, nt = n_right - mono_right;   /* :1128-1132: the lapping-area tails */
    std::vector<int32_t> idx((size_t)2 * (nq > 0 ? nq : 0) + 2), dist(idx.size());
    if (nq > 0) orbo_knn2(desc_left + (size_t)mono_left * 32, nq, desc_right + (size_t)mono_right * 32, nt > 0 ? nt : 0, idx.data(), dist.data());  /* :1144 */
    int n_matches = 0, n_desc = 0;
    for (int q = 0; q < nq; q++) {
        if (idx[2 * q + 1] < 0) continue;                                        /* (*it).size() >= 2 */
        if (!((float)dist[2 * q] < (float)dist[2 * q + 1] * 0.7)) continue;        /* :1151, float < double */
        n_desc++;
        const int il = q + mono_left, ir = idx[2 * q] + mono_right;
        const float s1 = level_sigma2[kp_left[il].octave], s2 = level_sigma2[kp_right[ir].octave];   /* :1155 */
        float p[3] = {0.f, 0.f, 0.f};
        const float d = triangulate(ctx, il, ir, s1, s2, p);
        if (d > 0.0001f) {                                                       /* :1157 */
            l2r[il] = ir; r2l[ir] = il;
            p3d[3 * il] = p[0]; p3d[3 * il + 1] = p[1]; p3d[3 * il + 2] = p[2];
            depth[il] = d;
            n_matches++;
        }
    }
    if (desc_matches) *desc_matches = n_desc;
    return n_matches;
}

}  // extern "C"
