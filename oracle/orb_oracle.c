/*
 * orb_oracle.c -- CPU restatement of the ORB frontend hot path (see orb_oracle.h:
 * test infrastructure only, parity unpinned).
 *
 * Each function names the reference call site it stands behind and the OpenCV
 * routine whose published behaviour it restates.  Build: gcc -O2 -std=c99
 * -ffp-contract=off (no fused multiply-add: the float steps must round after
 * every operation, as scalar x86-64 baseline code does).
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const int8_t k_pattern[256 * 4] = {
#include "../include/orb_pattern_31.inc"   /* the one copy of the table, shared with the library */
};

/* Disc of radius 15 used by the intensity centroid: half-width of row |v|. */
static const int k_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

/* FAST-16 ring, (dx, dy) for k = 0..15, clockwise from (0, +3). */
static const int k_ring[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* 7-tap Gaussian, sigma 2, in 8.8 fixed point (sums to 257, not renormalised). */
static const int k_gauss7[7] = {18, 34, 49, 55, 49, 34, 18};

/* Round half to even, as cvRound does on x86 (cvtss2si / cvtsd2si). */
static int round_f(float v) { return (int)lrintf(v); }
static int round_d(double v) { return (int)lrint(v); }

void orc_default_params(orc_params* p) {
    p->nfeatures = 1000;
    p->scale_factor = 1.2f;
    p->nlevels = 8;
    p->edge_threshold = 31;
    p->first_level = 0;
    p->wta_k = 2;
    p->score_type = 1;
    p->patch_size = 31;
    p->fast_threshold = 20;
}

static int check_params(const orc_params* p) {
    if (!p) return ORC_ERR_ARG;
    if (p->nlevels < 1 || p->nlevels > ORC_MAX_LEVELS || p->nfeatures < 0) return ORC_ERR_ARG;
    if (!(p->scale_factor > 1.0f)) return ORC_ERR_ARG;
    if (p->first_level != 0 || p->wta_k != 2 || p->score_type != 1 || p->patch_size != 31)
        return ORC_ERR_UNSUPPORTED;
    if (p->edge_threshold < 31) return ORC_ERR_UNSUPPORTED; /* patch reads would leave the level */
    if (p->fast_threshold < 0 || p->fast_threshold > 255) return ORC_ERR_ARG;
    return ORC_OK;
}

/* ORB's getScale(): the factor is held as a double (cv::ORB::create takes a
 * float, main.cpp:100 passes 1.2f), raised in double, rounded to float. */
float orc_level_scale(const orc_params* p, int level) {
    return (float)pow((double)p->scale_factor, (double)(level - p->first_level));
}

int orc_pyramid_sizes(const orc_params* p, int w, int h, int* widths, int* heights) {
    int rc = check_params(p);
    if (rc) return rc;
    if (w <= 0 || h <= 0) return ORC_ERR_ARG;
    for (int l = 0; l < p->nlevels; l++) {
        float inv_scale = 1.0f / orc_level_scale(p, l);
        widths[l] = round_f((float)w * inv_scale);
        heights[l] = round_f((float)h * inv_scale);
    }
    return ORC_OK;
}

/* computeKeyPoints(): geometric split of nfeatures over the levels, fp32. */
int orc_level_quotas(const orc_params* p, int* quotas) {
    int rc = check_params(p);
    if (rc) return rc;
    float factor = (float)(1.0 / (double)p->scale_factor);
    float denom = 1 - (float)pow((double)factor, (double)p->nlevels);
    float desired = (float)p->nfeatures * (1 - factor) / denom;
    int sum = 0;
    for (int l = 0; l < p->nlevels - 1; l++) {
        quotas[l] = round_f(desired);
        sum += quotas[l];
        desired *= factor;
    }
    int last = p->nfeatures - sum;
    quotas[p->nlevels - 1] = last > 0 ? last : 0;
    return ORC_OK;
}

/* INTER_LINEAR_EXACT, one axis.  Source coordinate in double exactly as the
 * soft-float code does: scale = 1 / (dst/src); f = scale*(d + 0.5) - 0.5.
 * Weights are 8.8 fixed point, c1 = round(frac*256), c0 = 256 - c1.  A source
 * index left of 0 uses the first sample alone, one right of size-2 the last
 * sample alone.  A ratio of exactly 2 is the 2x2 box mean; with offsets 2d and
 * both weights 128 the two-pass formula below gives (a+b+c+d+2)>>2 exactly. */
int orc_linear_exact_coeffs(int src_size, int dst_size, int32_t* offsets, uint16_t* c1) {
    if (src_size <= 0 || dst_size <= 0) return ORC_ERR_ARG;
    double inv_scale = (double)dst_size / (double)src_size;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst_size; d++) {
        double f = scale * ((double)d + 0.5);
        f = f - 0.5;
        int i = (int)floor(f);
        if (i >= 0 && src_size > 1) {
            if (i < src_size - 1) {
                double frac = f - (double)i;
                int w1 = round_d(frac * 256.0);
                offsets[d] = i;
                c1[d] = (uint16_t)w1;
            } else {
                offsets[d] = src_size - 1;
                c1[d] = 0;
            }
        } else {
            offsets[d] = 0;
            c1[d] = 0;
        }
    }
    return ORC_OK;
}

static int is_exact_half(int src_w, int src_h, int dst_w, int dst_h) {
    return src_w == 2 * dst_w && src_h == 2 * dst_h;
}

int orc_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride,
                            uint8_t* dst, int dw, int dh, int dstride) {
    if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return ORC_ERR_ARG;
    int32_t* xo = (int32_t*)malloc(sizeof(int32_t) * (size_t)(dw + dh));
    uint16_t* xc = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)(dw + dh));
    uint16_t* line0 = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)dw * 2);
    if (!xo || !xc || !line0) { free(xo); free(xc); free(line0); return ORC_ERR_NOMEM; }
    int32_t* yo = xo + dw;
    uint16_t* yc = xc + dw;
    uint16_t* line1 = line0 + dw;
    if (is_exact_half(sw, sh, dw, dh)) {
        for (int d = 0; d < dw; d++) { xo[d] = 2 * d; xc[d] = 128; }
        for (int d = 0; d < dh; d++) { yo[d] = 2 * d; yc[d] = 128; }
    } else {
        orc_linear_exact_coeffs(sw, dw, xo, xc);
        orc_linear_exact_coeffs(sh, dh, yo, yc);
    }
    for (int y = 0; y < dh; y++) {
        int y0 = yo[y];
        int y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
        const uint8_t* r0 = src + (size_t)y0 * sstride;
        const uint8_t* r1 = src + (size_t)y1 * sstride;
        for (int x = 0; x < dw; x++) {
            int x0 = xo[x];
            int x1 = x0 + 1 < sw ? x0 + 1 : sw - 1;
            int w1 = xc[x], w0 = 256 - w1;
            line0[x] = (uint16_t)(r0[x0] * w0 + r0[x1] * w1);
            line1[x] = (uint16_t)(r1[x0] * w0 + r1[x1] * w1);
        }
        uint32_t m1 = yc[y], m0 = 256 - m1;
        uint8_t* o = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++)
            o[x] = (uint8_t)((line0[x] * m0 + line1[x] * m1 + 32768u) >> 16);
    }
    free(xo); free(xc); free(line0);
    return ORC_OK;
}

/* ORB's image pyramid: level 0 is the input, level l >= 1 is resized from
 * level l-1 (a chain, not from level 0). */
int orc_build_pyramid(const orc_params* p, const uint8_t* img, int w, int h, int stride, uint8_t* out) {
    int ws[ORC_MAX_LEVELS], hs[ORC_MAX_LEVELS];
    int rc = orc_pyramid_sizes(p, w, h, ws, hs);
    if (rc) return rc;
    if (!img || !out || stride < w) return ORC_ERR_ARG;
    for (int y = 0; y < h; y++) memcpy(out + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
    uint8_t* prev = out;
    for (int l = 1; l < p->nlevels; l++) {
        uint8_t* cur = prev + (size_t)ws[l - 1] * hs[l - 1];
        if (ws[l] <= 0 || hs[l] <= 0) return ORC_ERR_ARG;
        rc = orc_resize_linear_exact(prev, ws[l - 1], hs[l - 1], ws[l - 1], cur, ws[l], hs[l], ws[l]);
        if (rc) return rc;
        prev = cur;
    }
    return ORC_OK;
}

/* FAST-9/16 at one pixel: returns 0 if not a corner, else the corner score:
 * the largest threshold t' for which the pixel is still a corner, i.e.
 * max(t, best arc of 9 contiguous ring pixels by min |difference|) - 1. */
static int fast_pixel_score(const uint8_t* c, int stride, int threshold) {
    int v = c[0];
    int d[16];
    for (int k = 0; k < 16; k++) d[k] = v - c[k_ring[k][1] * stride + k_ring[k][0]];
    /* corner test: 9 contiguous darker-than-(v-t) or brighter-than-(v+t) */
    int corner = 0;
    int run_dark = 0, run_bright = 0;
    for (int k = 0; k < 25 && !corner; k++) {
        int dk = d[k & 15];
        run_bright = (dk < -threshold) ? run_bright + 1 : 0; /* ring pixel brighter than v+t */
        run_dark = (dk > threshold) ? run_dark + 1 : 0;      /* ring pixel darker than v-t */
        if (run_bright > 8 || run_dark > 8) corner = 1;
    }
    if (!corner) return 0;
    int a0 = threshold;
    for (int k = 0; k < 16; k++) {
        int m = d[k];
        for (int j = 1; j < 9; j++) { int e = d[(k + j) & 15]; if (e < m) m = e; }
        if (m > a0) a0 = m;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k++) {
        int m = d[k];
        for (int j = 1; j < 9; j++) { int e = d[(k + j) & 15]; if (e > m) m = e; }
        if (m < b0) b0 = m;
    }
    return -b0 - 1;
}

int orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score) {
    if (!img || !score || w <= 0 || h <= 0 || stride < w) return ORC_ERR_ARG;
    memset(score, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++)
            score[(size_t)y * w + x] = (uint8_t)fast_pixel_score(img + (size_t)y * stride + x, stride, threshold);
    return ORC_OK;
}

/* cv::FAST(img, kps, threshold, nonmaxSuppression=true, TYPE_9_16): a corner is
 * kept when its score is strictly greater than all 8 neighbours' scores. */
int orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold,
                    int32_t* xys, int capacity, int* n) {
    if (!n) return ORC_ERR_ARG;
    *n = 0;
    if (!img || w <= 0 || h <= 0) return ORC_ERR_ARG;
    uint8_t* s = (uint8_t*)malloc((size_t)w * h);
    if (!s) return ORC_ERR_NOMEM;
    int rc = orc_fast_score_map(img, w, h, stride, threshold, s);
    if (rc) { free(s); return rc; }
    int count = 0;
    for (int y = 3; y < h - 3; y++) {
        for (int x = 3; x < w - 3; x++) {
            int v = s[(size_t)y * w + x];
            if (!v) continue;
            const uint8_t* q = s + (size_t)y * w + x;
            if (v > q[-1] && v > q[1] && v > q[-w - 1] && v > q[-w] && v > q[-w + 1] &&
                v > q[w - 1] && v > q[w] && v > q[w + 1]) {
                if (count < capacity && xys) {
                    xys[3 * count] = x; xys[3 * count + 1] = y; xys[3 * count + 2] = v;
                }
                count++;
            }
        }
    }
    free(s);
    *n = count;
    return count > capacity ? ORC_ERR_CAPACITY : ORC_OK;
}

static int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

/* GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8-bit data through the
 * separable fixed-point filter: row pass to 32-bit sums, column pass
 * saturate((sum + 2^15) >> 16). */
int orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    if (!src || !dst || w <= 0 || h <= 0) return ORC_ERR_ARG;
    int32_t* rows = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
    if (!rows) return ORC_ERR_NOMEM;
    for (int y = 0; y < h; y++) {
        const uint8_t* r = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int32_t acc = 0;
            for (int k = -3; k <= 3; k++) acc += k_gauss7[k + 3] * r[reflect101(x + k, w)];
            rows[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int32_t acc = 0;
            for (int k = -3; k <= 3; k++) acc += k_gauss7[k + 3] * rows[(size_t)reflect101(y + k, h) * w + x];
            int32_t v = (acc + 32768) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    free(rows);
    return ORC_OK;
}

/* cv::fastAtan2 (scalar form): degree-7 odd polynomial in the smaller/larger
 * ratio, result in degrees. */
float orc_fast_atan2(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ORB's ICAngles(): first-order moments over the radius-15 disc. */
int orc_ic_moments(const uint8_t* img, int stride, int x, int y, int* m01, int* m10) {
    const uint8_t* c = img + (size_t)y * stride + x;
    int s01 = 0, s10 = 0;
    for (int u = -15; u <= 15; u++) s10 += u * c[u];
    for (int v = 1; v <= 15; v++) {
        int d = k_umax[v], vsum = 0;
        for (int u = -d; u <= d; u++) {
            int plus = c[u + v * stride], minus = c[u - v * stride];
            vsum += plus - minus;
            s10 += u * (plus + minus);
        }
        s01 += v * vsum;
    }
    *m01 = s01; *m10 = s10;
    return ORC_OK;
}

/* ORB's computeOrbDescriptors() for WTA_K = 2: rotate each pattern point by the
 * keypoint angle (cos/sin in double, rounded to float; products and sums in
 * float), round to the nearest pixel, compare the two blurred samples. */
int orc_brief_descriptor(const uint8_t* blurred, int stride, int cx, int cy, float angle_deg, uint8_t* desc32) {
    float angle = angle_deg * (float)(3.14159265358979323846 / 180.0);
    float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const uint8_t* c = blurred + (size_t)cy * stride + cx;
    for (int i = 0; i < 32; i++) {
        int byte = 0;
        for (int bit = 0; bit < 8; bit++) {
            const int8_t* pp = k_pattern + (size_t)(i * 8 + bit) * 4;
            float x0 = (float)pp[0] * a - (float)pp[1] * b;
            float y0 = (float)pp[0] * b + (float)pp[1] * a;
            float x1 = (float)pp[2] * a - (float)pp[3] * b;
            float y1 = (float)pp[2] * b + (float)pp[3] * a;
            int t0 = c[round_f(y0) * stride + round_f(x0)];
            int t1 = c[round_f(y1) * stride + round_f(x1)];
            byte |= (t0 < t1) << bit;
        }
        desc32[i] = (uint8_t)byte;
    }
    return ORC_OK;
}

/* ---------------------------------------------------------------------------
 * Option beyond the reference (it has no keypoint distribution step): quadtree
 * distribution with an exact target count, after ORB-SLAM's DistributeOctTree,
 * made deterministic.  Input: the NMS survivors of one level inside the border,
 * cand[3 i] = x, y, score; region [rx0, rx1) x [ry0, ry1); target = the level's quota.
 *   - n <= target: everything is kept.
 *   - Roots: nini = max(1, round(W / H)) cells of ceil(W / nini) columns side by side.
 *   - A cell splits at x0 + ceil(w / 2), y0 + ceil(h / 2); empty children are dropped.
 *     Cells are kept in path order (root index, then quadrant 2 * lower + right per level).
 *   - Round: stop when the number of cells reaches the target.  If splitting every cell that
 *     holds more than one survivor still gives fewer cells than the target, do that and
 *     start the next round; otherwise split such cells one by one, most populated first
 *     (path order among equals), until the target is reached, and stop.
 *   - Every cell keeps its best survivor (highest score, then lowest y, then lowest x); if
 *     that is more than the target, the best `target` of them by the same order stay.
 * keep[i] = 1 for the survivors that stay. */
typedef struct { int x0, x1, y0, y1, begin, count; } qt_cell;

static int qt_better(const int32_t* cand, int a, int b) {   /* a before b in (score desc, y asc, x asc) */
    if (cand[3 * a + 2] != cand[3 * b + 2]) return cand[3 * a + 2] > cand[3 * b + 2];
    if (cand[3 * a + 1] != cand[3 * b + 1]) return cand[3 * a + 1] < cand[3 * b + 1];
    return cand[3 * a] < cand[3 * b];
}

/* Splits cell c, whose survivors are idx[c.begin .. c.begin + c.count); writes the non-empty
 * children in quadrant order to out (their index ranges into idx2, filled from *fill on). */
static int qt_split(const int32_t* cand, const qt_cell* c, const int* idx, int* idx2, int* fill, qt_cell* out) {
    const int hx = (c->x1 - c->x0 + 1) / 2, hy = (c->y1 - c->y0 + 1) / 2;
    const int mx = c->x0 + hx, my = c->y0 + hy;
    int k = 0;
    for (int q = 0; q < 4; q++) {
        qt_cell ch;
        ch.x0 = (q & 1) ? mx : c->x0; ch.x1 = (q & 1) ? c->x1 : mx;
        ch.y0 = (q & 2) ? my : c->y0; ch.y1 = (q & 2) ? c->y1 : my;
        ch.begin = *fill; ch.count = 0;
        for (int j = 0; j < c->count; j++) {
            const int i = idx[c->begin + j];
            const int qi = ((cand[3 * i + 1] >= my) << 1) | (cand[3 * i] >= mx);
            if (qi == q) { idx2[*fill] = i; (*fill)++; ch.count++; }
        }
        if (ch.count > 0) out[k++] = ch;
    }
    return k;
}

int orc_quadtree_select(const int32_t* cand, int n, int rx0, int ry0, int rx1, int ry1, int target, uint8_t* keep) {
    if (n < 0 || target < 0 || (n > 0 && (!cand || !keep)) || rx1 <= rx0 || ry1 <= ry0) return ORC_ERR_ARG;
    for (int i = 0; i < n; i++) {
        if (cand[3 * i] < rx0 || cand[3 * i] >= rx1 || cand[3 * i + 1] < ry0 || cand[3 * i + 1] >= ry1) return ORC_ERR_ARG;
        keep[i] = (uint8_t)(n <= target);
    }
    if (n <= target || target == 0) return ORC_OK;
    const int W = rx1 - rx0, H = ry1 - ry0;
    int nini = (2 * W + H) / (2 * H);
    if (nini < 1) nini = 1;
    const int rw = (W + nini - 1) / nini;
    /* two generations of cells and of index lists; a generation never holds more than n cells */
    qt_cell* cells = (qt_cell*)malloc(sizeof(qt_cell) * (size_t)(n + nini) * 2);
    int* idx = (int*)malloc(sizeof(int) * (size_t)n * 2);
    int* order = (int*)malloc(sizeof(int) * (size_t)(n + nini));
    if (!cells || !idx || !order) { free(cells); free(idx); free(order); return ORC_ERR_NOMEM; }
    qt_cell* cur = cells; qt_cell* nxt = cells + (n + nini);
    int* icur = idx; int* inxt = idx + n;
    int ncur = 0, fill = 0;
    for (int r = 0; r < nini; r++) {
        qt_cell c;
        c.x0 = rx0 + r * rw; c.x1 = c.x0 + rw < rx1 ? c.x0 + rw : rx1; c.y0 = ry0; c.y1 = ry1;
        c.begin = fill; c.count = 0;
        for (int i = 0; i < n; i++)
            if (cand[3 * i] >= c.x0 && cand[3 * i] < c.x1) { icur[fill++] = i; c.count++; }
        if (c.count > 0) cur[ncur++] = c;
    }
    for (;;) {
        if (ncur >= target) break;
        /* gain of splitting each cell that can be split */
        int total_gain = 0, nexp = 0;
        qt_cell tmp[4];
        int* scratch = inxt;
        for (int c = 0; c < ncur; c++) {
            if (cur[c].count <= 1) continue;
            int f = 0;
            total_gain += qt_split(cand, &cur[c], icur, scratch, &f, tmp) - 1;
            nexp++;
        }
        if (nexp == 0) break;
        const int all = ncur + total_gain < target;
        int nsel = 0;
        if (!all) {
            /* most populated first, path order among equals: stable insertion sort of the cell numbers */
            for (int c = 0; c < ncur; c++) {
                if (cur[c].count <= 1) continue;
                int j = nsel++;
                while (j > 0 && cur[order[j - 1]].count < cur[c].count) { order[j] = order[j - 1]; j--; }
                order[j] = c;
            }
            int have = ncur, take = 0;
            for (; take < nsel && have < target; take++) {
                int f = 0;
                have += qt_split(cand, &cur[order[take]], icur, scratch, &f, tmp) - 1;
            }
            nsel = take;
        }
        /* next generation, in path order */
        int nn = 0, f = 0;
        for (int c = 0; c < ncur; c++) {
            int split = cur[c].count > 1 && all;
            if (!all) for (int j = 0; j < nsel; j++) if (order[j] == c) split = 1;
            if (split) {
                nn += qt_split(cand, &cur[c], icur, inxt, &f, nxt + nn);
            } else {
                qt_cell k = cur[c];
                k.begin = f;
                for (int j = 0; j < cur[c].count; j++) inxt[f++] = icur[cur[c].begin + j];
                nxt[nn++] = k;
            }
        }
        { qt_cell* t = cur; cur = nxt; nxt = t; int* ti = icur; icur = inxt; inxt = ti; }
        ncur = nn;
        if (!all) break;
    }
    /* the best survivor of every cell */
    int nwin = 0;
    for (int c = 0; c < ncur; c++) {
        int b = icur[cur[c].begin];
        for (int j = 1; j < cur[c].count; j++) { const int i = icur[cur[c].begin + j]; if (qt_better(cand, i, b)) b = i; }
        order[nwin++] = b;
    }
    if (nwin > target) {
        for (int a = 1; a < nwin; a++) {       /* insertion sort by (score desc, y asc, x asc) */
            const int v = order[a];
            int j = a;
            while (j > 0 && qt_better(cand, v, order[j - 1])) { order[j] = order[j - 1]; j--; }
            order[j] = v;
        }
        nwin = target;
    }
    for (int a = 0; a < nwin; a++) keep[order[a]] = 1;
    free(cells); free(idx); free(order);
    return ORC_OK;
}

/* cv::ORB::detectAndCompute(image, noArray(), keypoints, descriptors)
 * (reference: src/frame/detector.cpp:11).  Per level: FAST with NMS on the
 * unblurred level, drop keypoints within edge_threshold of the border, keep the
 * quota best by score with ties at the cut all kept (KeyPointsFilter::retainBest),
 * angle by intensity centroid on the unblurred level, pt scaled to level 0,
 * then every level blurred and the descriptors sampled from the blurred level.
 * Canonical keypoint order: (octave, y, x) ascending. */
int orc_detect_and_compute(const orc_params* p, const uint8_t* img, int w, int h, int stride,
                           orc_keypoint* kps, uint8_t* desc, int capacity, int* n) {
    return orc_detect_and_compute_grid(p, 0, img, w, h, stride, kps, desc, capacity, n);
}

int orc_detect_and_compute_grid(const orc_params* p, int grid_cell, const uint8_t* img, int w, int h, int stride,
                                orc_keypoint* kps, uint8_t* desc, int capacity, int* n) {
    return orc_detect_and_compute_ex(p, grid_cell, 0, img, w, h, stride, kps, desc, capacity, n);
}

int orc_detect_and_compute_ex(const orc_params* p, int grid_cell, int quadtree, const uint8_t* img, int w, int h,
                              int stride, orc_keypoint* kps, uint8_t* desc, int capacity, int* n) {
    if (grid_cell > 0 && quadtree) return ORC_ERR_ARG;
    int ws[ORC_MAX_LEVELS], hs[ORC_MAX_LEVELS], quotas[ORC_MAX_LEVELS];
    if (!n) return ORC_ERR_ARG;
    *n = 0;
    int rc = orc_pyramid_sizes(p, w, h, ws, hs);
    if (rc) return rc;
    orc_level_quotas(p, quotas);
    size_t total = 0, maxlvl = 0;
    size_t off[ORC_MAX_LEVELS];
    for (int l = 0; l < p->nlevels; l++) {
        if (ws[l] <= 0 || hs[l] <= 0) return ORC_ERR_ARG;
        off[l] = total;
        size_t sz = (size_t)ws[l] * hs[l];
        total += sz;
        if (sz > maxlvl) maxlvl = sz;
    }
    uint8_t* pyr = (uint8_t*)malloc(total);
    uint8_t* blur = (uint8_t*)malloc(maxlvl);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * 3 * (maxlvl / 2 + 64));
    if (!pyr || !blur || !cand) { free(pyr); free(blur); free(cand); return ORC_ERR_NOMEM; }
    rc = orc_build_pyramid(p, img, w, h, stride, pyr);
    if (rc) { free(pyr); free(blur); free(cand); return rc; }

    int count = 0;
    int level_begin[ORC_MAX_LEVELS + 1];
    for (int l = 0; l < p->nlevels; l++) {
        level_begin[l] = count;
        const uint8_t* lv = pyr + off[l];
        int lw = ws[l], lh = hs[l], edge = p->edge_threshold;
        int ncand = 0;
        int cap = (int)(maxlvl / 2) + 64; /* NMS survivors are never adjacent: at most one per 2x2 */
        rc = orc_fast_detect(lv, lw, lh, lw, p->fast_threshold, cand, cap, &ncand);
        if (rc) { free(pyr); free(blur); free(cand); return rc; }
        /* KeyPointsFilter::runByImageBorder */
        int kept = 0;
        if (!(lh <= 2 * edge || lw <= 2 * edge)) {
            for (int i = 0; i < ncand; i++) {
                int x = cand[3 * i], y = cand[3 * i + 1];
                if (x >= edge && x < lw - edge && y >= edge && y < lh - edge) {
                    cand[3 * kept] = x; cand[3 * kept + 1] = y; cand[3 * kept + 2] = cand[3 * i + 2];
                    kept++;
                }
            }
        }
        /* Option beyond the reference: one survivor per grid cell (cells anchored at the
         * level origin): the highest score, the first in raster order among equals. */
        if (grid_cell > 0 && kept > 0) {
            const int cx_n = (lw + grid_cell - 1) / grid_cell, cy_n = (lh + grid_cell - 1) / grid_cell;
            int* best = (int*)malloc(sizeof(int) * (size_t)cx_n * cy_n);
            if (!best) { free(pyr); free(blur); free(cand); return ORC_ERR_NOMEM; }
            for (int c = 0; c < cx_n * cy_n; c++) best[c] = -1;
            for (int i = 0; i < kept; i++) {
                const int c = (cand[3 * i + 1] / grid_cell) * cx_n + cand[3 * i] / grid_cell;
                if (best[c] < 0 || cand[3 * i + 2] > cand[3 * best[c] + 2]) best[c] = i;
            }
            int k2 = 0;
            for (int i = 0; i < kept; i++) {
                const int c = (cand[3 * i + 1] / grid_cell) * cx_n + cand[3 * i] / grid_cell;
                if (best[c] != i) continue;
                cand[3 * k2] = cand[3 * i]; cand[3 * k2 + 1] = cand[3 * i + 1]; cand[3 * k2 + 2] = cand[3 * i + 2];
                k2++;
            }
            kept = k2;
            free(best);
        }
        /* KeyPointsFilter::retainBest(quota): cut at the quota-th best score, keep ties */
        int cut = 0;
        int quota = quotas[l];
        if (quadtree && kept > quota) {
            /* option beyond the reference: the quadtree distribution takes the place of the cut */
            uint8_t* keepf = (uint8_t*)malloc((size_t)kept);
            if (!keepf) { free(pyr); free(blur); free(cand); return ORC_ERR_NOMEM; }
            rc = orc_quadtree_select(cand, kept, edge, edge, lw - edge, lh - edge, quota, keepf);
            if (rc) { free(keepf); free(pyr); free(blur); free(cand); return rc; }
            int k2 = 0;
            for (int i = 0; i < kept; i++) {
                if (!keepf[i]) continue;
                cand[3 * k2] = cand[3 * i]; cand[3 * k2 + 1] = cand[3 * i + 1]; cand[3 * k2 + 2] = cand[3 * i + 2];
                k2++;
            }
            kept = k2;
            free(keepf);
        } else if (kept > quota) {
            if (quota == 0) kept = 0;
            else {
                int hist[256];
                memset(hist, 0, sizeof hist);
                for (int i = 0; i < kept; i++) hist[cand[3 * i + 2]]++;
                int acc = 0;
                for (cut = 255; cut > 0; cut--) { acc += hist[cut]; if (acc >= quota) break; }
            }
        }
        float sf = orc_level_scale(p, l);
        for (int i = 0; i < kept; i++) {
            int x = cand[3 * i], y = cand[3 * i + 1], s = cand[3 * i + 2];
            if (s < cut) continue;
            if (count < capacity) {
                orc_keypoint* k = &kps[count];
                int m01, m10;
                orc_ic_moments(lv, lw, x, y, &m01, &m10);
                k->x = (float)x; k->y = (float)y;
                k->size = (float)p->patch_size * sf;
                k->angle = orc_fast_atan2((float)m01, (float)m10);
                k->response = (float)s;
                k->octave = l;
                k->class_id = -1;
            }
            count++;
        }
    }
    level_begin[p->nlevels] = count;
    if (count > capacity) { free(pyr); free(blur); free(cand); *n = count; return ORC_ERR_CAPACITY; }
    /* pt *= scale */
    for (int i = 0; i < count; i++) {
        float sf = orc_level_scale(p, kps[i].octave);
        kps[i].x *= sf; kps[i].y *= sf;
    }
    /* blur level by level, then describe from the blurred level */
    for (int l = 0; l < p->nlevels; l++) {
        if (level_begin[l + 1] == level_begin[l]) continue;
        orc_gaussian_blur7(pyr + off[l], ws[l], hs[l], ws[l], blur, ws[l]);
        float inv = 1.f / orc_level_scale(p, l);
        for (int i = level_begin[l]; i < level_begin[l + 1]; i++) {
            int cx = round_f(kps[i].x * inv), cy = round_f(kps[i].y * inv);
            orc_brief_descriptor(blur, ws[l], cx, cy, kps[i].angle, desc + (size_t)i * 32);
        }
    }
    free(pyr); free(blur); free(cand);
    *n = count;
    return ORC_OK;
}

/* ---------------------------------------------------------------------------
 * Input front end of the reference's main loop (main.cpp:124-129):
 * cv::cvtColor(BGR2GRAY) followed by cv::resize(Size(0,0), scale, scale) with the
 * default INTER_LINEAR.  Restated from OpenCV 4.x: grey = (3735 B + 19235 G + 9798 R +
 * 2^14) >> 15; resize with 11-bit fixed-point taps, the 8-bit vertical pass computing
 * (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
 * ------------------------------------------------------------------------- */
int orc_bgr2gray(const uint8_t* bgr, int w, int h, int sstride, uint8_t* gray, int dstride) {
    if (!bgr || !gray || w <= 0 || h <= 0) return ORC_ERR_ARG;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* p = bgr + (size_t)y * sstride + 3 * x;
            gray[(size_t)y * dstride + x] = (uint8_t)((3735 * p[0] + 19235 * p[1] + 9798 * p[2] + (1 << 14)) >> 15);
        }
    return ORC_OK;
}

int orc_resize_scale_size(int src_w, int src_h, float scale, int* dst_w, int* dst_h) {
    if (!(scale > 0.f) || !dst_w || !dst_h) return ORC_ERR_ARG;
    *dst_w = round_d((double)src_w * (double)scale);   /* saturate_cast<int>(double) rounds half to even */
    *dst_h = round_d((double)src_h * (double)scale);
    return (*dst_w > 0 && *dst_h > 0) ? ORC_OK : ORC_ERR_ARG;
}

/* Source offset and the two 11-bit taps of every destination index of one axis.
 * clamp_frac: the horizontal pass clamps the fraction at the borders, the vertical
 * pass clamps the row index instead and keeps the fraction. */
int orc_linear_taps(int src_size, int dst_size, double inv_scale, int clamp_frac, int32_t* ofs, int16_t* taps) {
    if (src_size <= 0 || dst_size <= 0 || !(inv_scale > 0) || !ofs || !taps) return ORC_ERR_ARG;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst_size; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s0 = (int)floorf(f);
        f -= (float)s0;
        if (clamp_frac) {
            if (s0 < 0) { f = 0.f; s0 = 0; }
            if (s0 >= src_size - 1) { f = 0.f; s0 = src_size - 1; }
        }
        ofs[d] = s0;
        taps[2 * d] = (int16_t)round_f((1.f - f) * 2048.f);
        taps[2 * d + 1] = (int16_t)round_f(f * 2048.f);
    }
    return ORC_OK;
}

int orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride, float scale,
                      uint8_t* dst, int dw, int dh, int dstride) {
    int ew, eh;
    int rc = orc_resize_scale_size(sw, sh, scale, &ew, &eh);
    if (rc) return rc;
    if (!src || !dst || ew != dw || eh != dh) return ORC_ERR_ARG;
    int32_t* xo = (int32_t*)malloc(sizeof(int32_t) * dw); int16_t* xa = (int16_t*)malloc(sizeof(int16_t) * 2 * dw);
    int32_t* yo = (int32_t*)malloc(sizeof(int32_t) * dh); int16_t* ya = (int16_t*)malloc(sizeof(int16_t) * 2 * dh);
    if (!xo || !xa || !yo || !ya) { free(xo); free(xa); free(yo); free(ya); return ORC_ERR_NOMEM; }
    orc_linear_taps(sw, dw, (double)scale, 1, xo, xa);
    orc_linear_taps(sh, dh, (double)scale, 0, yo, ya);
    for (int y = 0; y < dh; y++) {
        int y0 = yo[y], y1 = yo[y] + 1;
        y0 = y0 < 0 ? 0 : (y0 > sh - 1 ? sh - 1 : y0);
        y1 = y1 < 0 ? 0 : (y1 > sh - 1 ? sh - 1 : y1);
        const uint8_t* r0 = src + (size_t)y0 * sstride;
        const uint8_t* r1 = src + (size_t)y1 * sstride;
        for (int x = 0; x < dw; x++) {
            const int x0 = xo[x], x1 = x0 + 1 < sw ? x0 + 1 : sw - 1;   /* second tap is 0 where x0 is the last column */
            const int S0 = r0[x0] * xa[2 * x] + r0[x1] * xa[2 * x + 1];
            const int S1 = r1[x0] * xa[2 * x] + r1[x1] * xa[2 * x + 1];
            const int v = (((ya[2 * y] * (S0 >> 4)) >> 16) + ((ya[2 * y + 1] * (S1 >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)y * dstride + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xo); free(xa); free(yo); free(ya);
    return ORC_OK;
}

int orc_frontend(const uint8_t* bgr, int w, int h, int sstride, int target_width,
                 uint8_t* gray_out, int dw, int dh, int dstride) {
    if (!bgr || !gray_out || w <= 0 || h <= 0 || target_width <= 0) return ORC_ERR_ARG;
    uint8_t* g = (uint8_t*)malloc((size_t)w * h);
    if (!g) return ORC_ERR_NOMEM;
    int rc = orc_bgr2gray(bgr, w, h, sstride, g, w);
    if (!rc) rc = orc_resize_linear(g, w, h, w, (float)target_width / (float)w, gray_out, dw, dh, dstride);
    free(g);
    return rc;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int orc_detect_and_compute_batch(const orc_params* p, const uint8_t* imgs, int n_frames, size_t frame_pitch,
                                 int w, int h, int stride, orc_keypoint* kps, uint8_t* desc,
                                 int capacity, int* counts) {
    if (!imgs || !kps || !desc || !counts || n_frames < 0) return ORC_ERR_ARG;
    int rc_all = ORC_OK;
#pragma omp parallel for schedule(dynamic, 1)
    for (int f = 0; f < n_frames; f++) {
        int rc = orc_detect_and_compute(p, imgs + (size_t)f * frame_pitch, w, h, stride,
                                        kps + (size_t)f * capacity, desc + (size_t)f * capacity * 32,
                                        capacity, &counts[f]);
        if (rc != ORC_OK) {
#pragma omp critical
            rc_all = rc;
        }
    }
    return rc_all;
}

int orc_match_gated_mt(const uint8_t* qd, const orc_keypoint* qkp, int nq,
                       const uint8_t* td, const orc_keypoint* tkp, int nt,
                       float max_dist, float area, int max_level,
                       orc_dmatch* out, int* n_out) {
    if (!n_out || nq < 0 || nt < 0) return ORC_ERR_ARG;
    /* one record per query (trainIdx -1 = none), compacted afterwards in query order */
    orc_dmatch* tmp = (orc_dmatch*)malloc(sizeof(orc_dmatch) * (size_t)(nq > 0 ? nq : 1));
    if (!tmp) return ORC_ERR_NOMEM;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nq; q++) {
        int n = 0;
        orc_dmatch one;
        orc_match_gated(qd + (size_t)q * 32, qkp + q, 1, td, tkp, nt, max_dist, area, max_level, &one, &n);
        tmp[q].trainIdx = -1;
        if (n) { tmp[q] = one; tmp[q].queryIdx = q; }
    }
    int m = 0;
    for (int q = 0; q < nq; q++)
        if (tmp[q].trainIdx >= 0) out[m++] = tmp[q];
    free(tmp);
    *n_out = m;
    return ORC_OK;
}

int orc_hamming256(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int i = 0; i < 32; i++) {
        unsigned v = (unsigned)(a[i] ^ b[i]);
        while (v) { d += (int)(v & 1u); v >>= 1; }
    }
    return d;
}

/* Gate of Matcher::_filterMatches (src/frame/matcher.cpp:144-157): a candidate
 * is skipped only when BOTH octaves exceed max_level; the window is a strict
 * square |dx| < area && |dy| < area; -1 switches a gate off. */
static int gate_pass(const orc_keypoint* q, const orc_keypoint* t, float area, int max_level) {
    if (max_level != -1 && q->octave > max_level && t->octave > max_level) return 0;
    if (area == -1) return 1;
    float dx = q->x - t->x, dy = q->y - t->y;
    return fabsf(dx) < area && fabsf(dy) < area;
}

typedef struct { int32_t train; int32_t dist; } cand_t;

static int cand_cmp(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->train < y->train ? -1 : (x->train > y->train);
}

int orc_radius_match_filter(const uint8_t* qd, const orc_keypoint* qkp, int nq,
                            const uint8_t* td, const orc_keypoint* tkp, int nt,
                            float max_dist, float area, int max_level,
                            orc_dmatch* out, int* n_out) {
    if (!n_out || nq < 0 || nt < 0) return ORC_ERR_ARG;
    *n_out = 0;
    if (nq == 0 || nt == 0) return ORC_OK;
    cand_t* list = (cand_t*)malloc(sizeof(cand_t) * (size_t)nt);
    if (!list) return ORC_ERR_NOMEM;
    int m = 0;
    for (int q = 0; q < nq; q++) {
        /* radiusMatch: every train row with distance <= max_dist, then sorted by distance */
        int len = 0;
        for (int t = 0; t < nt; t++) {
            int d = orc_hamming256(qd + (size_t)q * 32, td + (size_t)t * 32);
            if ((float)d <= max_dist) { list[len].train = t; list[len].dist = d; len++; }
        }
        qsort(list, (size_t)len, sizeof(cand_t), cand_cmp);
        /* _filterMatches: first candidate that passes the gates */
        for (int i = 0; i < len; i++) {
            if (!gate_pass(&qkp[q], &tkp[list[i].train], area, max_level)) continue;
            out[m].queryIdx = q; out[m].trainIdx = list[i].train; out[m].imgIdx = 0;
            out[m].distance = (float)list[i].dist;
            m++;
            break;
        }
    }
    free(list);
    *n_out = m;
    return ORC_OK;
}

int orc_match_gated(const uint8_t* qd, const orc_keypoint* qkp, int nq,
                    const uint8_t* td, const orc_keypoint* tkp, int nt,
                    float max_dist, float area, int max_level,
                    orc_dmatch* out, int* n_out) {
    if (!n_out || nq < 0 || nt < 0) return ORC_ERR_ARG;
    int m = 0;
    for (int q = 0; q < nq; q++) {
        int best = 1 << 30, best_t = -1;
        const uint64_t* a = (const uint64_t*)(const void*)(qd + (size_t)q * 32);
        for (int t = 0; t < nt; t++) {
            const uint64_t* b = (const uint64_t*)(const void*)(td + (size_t)t * 32);
            int d = __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
                    __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
            if (d >= best) continue;
            if (!((float)d <= max_dist)) continue;
            if (!gate_pass(&qkp[q], &tkp[t], area, max_level)) continue;
            best = d; best_t = t;
        }
        if (best_t >= 0) {
            out[m].queryIdx = q; out[m].trainIdx = best_t; out[m].imgIdx = 0; out[m].distance = (float)best;
            m++;
        }
    }
    *n_out = m;
    return ORC_OK;
}

/* Three fullest bins, scanned in ascending bin order with strict comparisons, as
 * ORB-SLAM's ComputeThreeMaxima does; a bin below a tenth of the fullest is dropped. */
static void three_maxima(const int* histo, int len, int* i1, int* i2, int* i3) {
    int max1 = 0, max2 = 0, max3 = 0;
    *i1 = *i2 = *i3 = -1;
    for (int i = 0; i < len; i++) {
        const int s = histo[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *i3 = *i2; *i2 = *i1; *i1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *i3 = *i2; *i2 = i; }
        else if (s > max3) { max3 = s; *i3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { *i2 = -1; *i3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { *i3 = -1; }
}

static int rotation_bin(float angle_q, float angle_t, int len) {
    float rot = angle_q - angle_t;
    if (rot < 0.0f) rot += 360.0f;
    const float factor = (float)len / 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin >= len) bin = 0;
    return bin;
}

int orc_match_gated_ex(const uint8_t* qd, const orc_keypoint* qkp, int nq,
                       const uint8_t* td, const orc_keypoint* tkp, int nt,
                       float max_dist, float area, int max_level,
                       float ratio, int histo_length,
                       orc_dmatch* out, int* n_out) {
    if (!n_out || nq < 0 || nt < 0 || histo_length > 64) return ORC_ERR_ARG;
    int m = 0;
    for (int q = 0; q < nq; q++) {
        int best = 1 << 30, second = 1 << 30, best_t = -1;
        for (int t = 0; t < nt; t++) {
            if (!gate_pass(&qkp[q], &tkp[t], area, max_level)) continue;
            const int d = orc_hamming256(qd + (size_t)q * 32, td + (size_t)t * 32);
            if (d < best) { second = best; best = d; best_t = t; }
            else if (d < second) second = d;
        }
        if (best_t < 0 || !((float)best <= max_dist)) continue;
        if (ratio > 0.0f && second < (1 << 30) && !((float)best < ratio * (float)second)) continue;
        out[m].queryIdx = q; out[m].trainIdx = best_t; out[m].imgIdx = 0; out[m].distance = (float)best;
        m++;
    }
    if (histo_length > 0 && m > 0) {
        int histo[64];
        memset(histo, 0, sizeof histo);
        for (int i = 0; i < m; i++)
            histo[rotation_bin(qkp[out[i].queryIdx].angle, tkp[out[i].trainIdx].angle, histo_length)]++;
        int i1, i2, i3;
        three_maxima(histo, histo_length, &i1, &i2, &i3);
        int k = 0;
        for (int i = 0; i < m; i++) {
            const int b = rotation_bin(qkp[out[i].queryIdx].angle, tkp[out[i].trainIdx].angle, histo_length);
            if (b == i1 || b == i2 || b == i3) out[k++] = out[i];
        }
        m = k;
    }
    *n_out = m;
    return ORC_OK;
}

int orc_frame_match(const uint8_t* d1, const orc_keypoint* kp1, int n1,
                    const uint8_t* d2, const orc_keypoint* kp2, int n2,
                    const int32_t* ids, int n_ids,
                    float max_dist, float area, int max_level,
                    orc_dmatch* out, int* n_out) {
    if (!n_out) return ORC_ERR_ARG;
    if (!ids || n_ids == 0)
        return orc_match_gated(d1, kp1, n1, d2, kp2, n2, max_dist, area, max_level, out, n_out);
    uint8_t* gd = (uint8_t*)malloc((size_t)n_ids * 32);
    orc_keypoint* gk = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)n_ids);
    if (!gd || !gk) { free(gd); free(gk); return ORC_ERR_NOMEM; }
    for (int i = 0; i < n_ids; i++) {
        if (ids[i] < 0 || ids[i] >= n1) { free(gd); free(gk); return ORC_ERR_ARG; }
        memcpy(gd + (size_t)i * 32, d1 + (size_t)ids[i] * 32, 32);
        gk[i] = kp1[ids[i]];
    }
    int rc = orc_match_gated(gd, gk, n_ids, d2, kp2, n2, max_dist, area, max_level, out, n_out);
    if (rc == ORC_OK)
        for (int i = 0; i < *n_out; i++) out[i].queryIdx = ids[out[i].queryIdx];
    free(gd); free(gk);
    return rc;
}

/* cv::projectPoints with the pose of the target keyframe
 * (src/frame/matcher.cpp:163-184); all arithmetic in double, result float. */
int orc_project_points(const double* Rt, const double* K4, const double* dist5,
                       const float* xyz, int n, float* uv) {
    if (!Rt || !K4 || !xyz || !uv || n < 0) return ORC_ERR_ARG;
    double k1 = 0, k2 = 0, p1 = 0, p2 = 0, k3 = 0;
    if (dist5) { k1 = dist5[0]; k2 = dist5[1]; p1 = dist5[2]; p2 = dist5[3]; k3 = dist5[4]; }
    for (int i = 0; i < n; i++) {
        double X = xyz[3 * i], Y = xyz[3 * i + 1], Z = xyz[3 * i + 2];
        double x = Rt[0] * X + Rt[1] * Y + Rt[2] * Z + Rt[3];
        double y = Rt[4] * X + Rt[5] * Y + Rt[6] * Z + Rt[7];
        double z = Rt[8] * X + Rt[9] * Y + Rt[10] * Z + Rt[11];
        z = z != 0 ? 1.0 / z : 1.0;
        x *= z; y *= z;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6;
        double xd = x * cdist + p1 * a1 + p2 * a2;
        double yd = y * cdist + p1 * a3 + p2 * a1;
        uv[2 * i] = (float)(xd * K4[0] + K4[2]);
        uv[2 * i + 1] = (float)(yd * K4[1] + K4[3]);
    }
    return ORC_OK;
}

/* projectionError (src/map/mappoint.cpp:83-101): distance between the keypoint and the
 * projection of the point, cv::norm(Point2f) accumulating in double. */
static double projection_error(const double* Rt, const double* K4, const double* dist5,
                               const float* xyz, float tx, float ty) {
    float uv[2];
    orc_project_points(Rt, K4, dist5, xyz, 1, uv);
    const float dx = tx - uv[0], dy = ty - uv[1];
    return sqrt((double)dx * dx + (double)dy * dy);
}

/* isOutlier (src/map/mappoint.cpp:54-81) for n triangulated points: point i belongs to
 * matches[i] (queryIdx into kp_q, trainIdx into kp_t).  Differences are float (cv::Mat
 * CV_32F subtraction), dot products and norms of the 3-vectors accumulate in double. */
int orc_outlier_flags(const float* xyz, const orc_dmatch* matches, int n,
                      const orc_keypoint* kp_q, int nq, const orc_keypoint* kp_t, int nt,
                      const float* center_q, const float* center_t,
                      const double* Rt_q, const double* K4_q, const double* dist5_q,
                      const double* Rt_t, const double* K4_t, const double* dist5_t,
                      uint8_t* flags) {
    if (!xyz || !matches || !kp_q || !kp_t || !center_q || !center_t || !flags || n < 0) return ORC_ERR_ARG;
    for (int i = 0; i < n; i++) {
        const float* P = xyz + 3 * i;
        const int qi = matches[i].queryIdx, ti = matches[i].trainIdx;
        if (qi < 0 || qi >= nq || ti < 0 || ti >= nt) return ORC_ERR_ARG;
        flags[i] = 1;
        if (isnan(P[0])) continue;
        float a[3], b[3];
        for (int k = 0; k < 3; k++) { a[k] = P[k] - center_q[k]; b[k] = P[k] - center_t[k]; }
        double dot = 0, na = 0, nb = 0;
        for (int k = 0; k < 3; k++) {
            dot += (double)a[k] * b[k]; na += (double)a[k] * a[k]; nb += (double)b[k] * b[k];
        }
        na = sqrt(na); nb = sqrt(nb);
        const double par = dot / (na * nb);
        if (par < 0.0f || par > 0.999f) continue;
        if (fabs(na) < 1E-6 || fabs(nb) < 1E-6) continue;
        if (projection_error(Rt_q, K4_q, dist5_q, P, kp_q[qi].x, kp_q[qi].y) > 1) continue;
        if (projection_error(Rt_t, K4_t, dist5_t, P, kp_t[ti].x, kp_t[ti].y) > 1) continue;
        flags[i] = 0;
    }
    return ORC_OK;
}

int orc_duplicate_pairs(const uint8_t* d1, const float* p1, int n1,
                        const uint8_t* d2, const float* p2, int n2,
                        int desc_dist, double point_dist, uint8_t* flag) {
    if (!flag || n1 < 0 || n2 < 0) return ORC_ERR_ARG;
    for (int i = 0; i < n1; i++) {
        for (int j = 0; j < n2; j++) {
            int d = orc_hamming256(d1 + (size_t)i * 32, d2 + (size_t)j * 32);
            int dup = 0;
            if ((float)d <= (float)desc_dist) {
                /* cv::norm(Point3f) accumulates in double */
                double dx = (double)(p1[3 * i] - p2[3 * j]);
                double dy = (double)(p1[3 * i + 1] - p2[3 * j + 1]);
                double dz = (double)(p1[3 * i + 2] - p2[3 * j + 2]);
                dup = sqrt(dx * dx + dy * dy + dz * dz) < point_dist;
            }
            flag[(size_t)i * n2 + j] = (uint8_t)dup;
        }
    }
    return ORC_OK;
}

int orc_frame_scale_tables(float scale_factor, int nlevels, float* scales, float* sigma, float* inv_sigma) {
    if (nlevels < 1 || !scales || !sigma || !inv_sigma) return ORC_ERR_ARG;
    scales[0] = 1.0f; sigma[0] = 1.0f; inv_sigma[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        scales[i] = scales[i - 1] * scale_factor;
        sigma[i] = scales[i] * scales[i];
        inv_sigma[i] = 1.0f / sigma[i];
    }
    return ORC_OK;
}
