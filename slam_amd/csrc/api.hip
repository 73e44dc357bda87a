// C ABI of the gfx950 ORB frontend: contexts, geometry, launch sequencing.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "common.h"
#include "detmath.h"
#include "kernels.h"

namespace sa {

// ---- error plumbing -----------------------------------------------------------
static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    g_last_error = std::string(what) + " failed: " + hipGetErrorString(e) + " (" + file + ":" +
                   std::to_string(line) + ")";
    (void)hipGetLastError();
    return SLAM_AMD_ERR_HIP;
}

// ---- per-stage event timing (process-wide, off by default) ----------------------
struct ProfState {
    bool on = false;
    std::vector<hipEvent_t> ev;   // 2 per mark: begin, end
    std::vector<int> stage;       // stage of mark i
    size_t used = 0;              // marks begun
};
static ProfState g_prof;

void prof_mark(int stage, bool begin, hipStream_t st) {
    if (!g_prof.on) return;
    if (begin) {
        if (g_prof.used >= g_prof.stage.size()) return;  // out of slots: stop recording
        g_prof.stage[g_prof.used] = stage;
        (void)hipEventRecord(g_prof.ev[2 * g_prof.used], st);
        g_prof.used++;
    } else {
        // the matching begin is the latest mark of this stage
        for (size_t i = g_prof.used; i-- > 0;) {
            if (g_prof.stage[i] == stage) { (void)hipEventRecord(g_prof.ev[2 * i + 1], st); break; }
            if (g_prof.used - i > 4) break;
        }
    }
}

// ---- a device buffer that grows, never shrinks ---------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return SLAM_AMD_OK;
        if (p) { SA_HIP(hipFree(p)); p = nullptr; cap = 0; }
        size_t want = std::max(bytes, (size_t)4096);
        SA_HIP(hipMalloc(&p, want));
        cap = want;
        return SLAM_AMD_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

static bool stream_capturing(hipStream_t st) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &status) != hipSuccess) { (void)hipGetLastError(); return false; }
    return status == hipStreamCaptureStatusActive;
}

static int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        SA_FAIL(SLAM_AMD_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    }
    if (device < 0 || device >= n) SA_FAIL(SLAM_AMD_ERR_ARG, "device index out of range");
    SA_HIP(hipSetDevice(device));
    return SLAM_AMD_OK;
}

}  // namespace sa

using namespace sa;

// =============================================================================
// Extractor
// =============================================================================
struct slam_amd_extractor {
    int grid_shift = 0;                // log2 of the grid-distribution cell, 0 = off (the reference)
    int quadtree = 0;                  // 1: quadtree distribution with an exact per-level count (not in the reference)
    slam_amd_orb_params params;
    int width = 0, height = 0, max_batch = 0, max_keypoints = 0, device = 0;
    PyrGeom g;
    // pyramid: per level (>= 1) coefficient tables on the device and the launch plan (groups of
    // levels chained inside one kernel, regions per tile of each group's last level)
    DevBuf tables;
    const uint32_t* xtab[kMaxLevels] = {};
    const uint32_t* ytab[kMaxLevels] = {};
    struct PyrGroupPlan {
        int first, last, ntx, nty, tab_bytes, buf_even, buf_odd, x_entries;
        short xat[kMaxLevels + 1] = {}, yat[kMaxLevels + 1] = {};
        const PyrSpan* spanx; const PyrSpan* spany;
        size_t lds_bytes;
    };
    std::vector<PyrGroupPlan> pyr_groups;
    DevBuf pyr, blur, hist, tilecount, tilelist, sel, kpxy;
    DevBuf tiledesc;                   // level and tile coordinates of every FAST tile of a frame
    DevBuf lanegeom;                   // level geometry as a table in device memory: one level per lane of the describe kernel
    int select_keycap = 4096;          // kept survivors of a level ranked from LDS (debug knob)
    bool hist_clean = false;           // the score histograms are all zero once the work enqueued so far has run: zeroed at creation,
                                       // and by the describe kernel after the selection has read them; a call that fails
                                       // in between enqueues the fill itself before it returns
    // host-call staging
    DevBuf h_img, h_kp, h_desc, h_count;
    int h_img_pitch = 0;
    int h_last_n = -1;                 // keypoints of the last host call still in h_kp / h_desc (-1: none)
    hipStream_t stream = nullptr;
    // geometry of the last batch (debug accessor)
    const uint8_t* last_img0 = nullptr;
    size_t last_frame_pitch = 0;
    int last_row_stride = 0;
};

static int check_orb_params(const slam_amd_orb_params* p) {
    if (!p) SA_FAIL(SLAM_AMD_ERR_ARG, "params is null");
    if (p->nlevels < 1 || p->nlevels > kMaxLevels) SA_FAIL(SLAM_AMD_ERR_ARG, "nlevels out of range");
    if (p->nfeatures < 0) SA_FAIL(SLAM_AMD_ERR_ARG, "nfeatures < 0");
    if (!(p->scale_factor > 1.0f)) SA_FAIL(SLAM_AMD_ERR_ARG, "scale_factor must be > 1");
    if (p->fast_threshold < 0 || p->fast_threshold > 255) SA_FAIL(SLAM_AMD_ERR_ARG, "fast_threshold out of range");
    if (p->first_level != 0) SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "first_level != 0 is not implemented");
    if (p->wta_k != 2) SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "WTA_K != 2 is not implemented");
    if (p->score_type != 1) SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "only FAST_SCORE is implemented");
    if (p->patch_size != 31) SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "patch_size != 31 is not implemented");
    if (p->edge_threshold < 31) SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "edge_threshold < 31 is not implemented");
    return SLAM_AMD_OK;
}

// One axis of the bit-exact bilinear resize: source offset and 8.8 weight of the
// second tap for every destination index.  Double arithmetic, one rounding per
// operation: scale = 1 / (dst / src); f = scale * (d + 0.5) - 0.5.
static void linear_exact_axis(int src, int dst, bool exact_half, int32_t* off, uint16_t* c1) {
    if (exact_half) {
        for (int d = 0; d < dst; d++) { off[d] = 2 * d; c1[d] = 128; }
        return;
    }
    const double ratio = (double)dst / (double)src;
    const double scale = 1.0 / ratio;
    for (int d = 0; d < dst; d++) {
        double f = scale * ((double)d + 0.5);
        f -= 0.5;
        const int i = (int)floor(f);
        if (i < 0 || src <= 1) { off[d] = 0; c1[d] = 0; }
        else if (i >= src - 1) { off[d] = src - 1; c1[d] = 0; }
        else { off[d] = i; c1[d] = (uint16_t)lrint((f - (double)i) * 256.0); }
    }
}

#ifndef SA_SEL_BANDS
#define SA_SEL_BANDS 16    // bands of tile rows per level where the selection is banded (at most 16)
#endif
static int build_geometry_of(const slam_amd_orb_params& p, int width, int height, PyrGeom& g);
static int build_geometry(slam_amd_extractor* ex) {
    const int rc = build_geometry_of(ex->params, ex->width, ex->height, ex->g);
    // an extractor for one or two frames at a time has too few levels to fill the chip with one selection
    // workgroup per level: bands from 128 tiles on
    if (rc == SLAM_AMD_OK && ex->max_batch <= 2 && ex->g.lv[0].tiles_x * div_up(ex->g.lv[0].h, kFastTileH) >= 128) ex->g.bands = SA_SEL_BANDS;
    return rc;
}

static int build_geometry_of(const slam_amd_orb_params& p, int width, int height, PyrGeom& g) {
    memset(&g, 0, sizeof g);
    g.nlevels = p.nlevels;
    g.edge = p.edge_threshold;
    g.fast_threshold = p.fast_threshold;
    // per-level quotas: geometric series in float, as ORB computes them
    const float factor = (float)(1.0 / (double)p.scale_factor);
    float desired = (float)p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)p.nlevels));
    int sum = 0;
    unsigned off = 0;
    int rows = 0, tiles = 0;
    for (int l = 0; l < p.nlevels; l++) {
        LevelGeom& L = g.lv[l];
        L.scale = (float)pow((double)p.scale_factor, (double)l);
        L.inv_scale = 1.0f / L.scale;
        L.w = (int)lrintf((float)width * L.inv_scale);
        L.h = (int)lrintf((float)height * L.inv_scale);
        if (L.w <= 0 || L.h <= 0) SA_FAIL(SLAM_AMD_ERR_ARG, "image too small for the number of levels");
        L.size = (float)p.patch_size * L.scale;
        if (l < p.nlevels - 1) {
            L.quota = (int)lrintf(desired);
            sum += L.quota;
            desired *= factor;
        } else {
            L.quota = std::max(p.nfeatures - sum, 0);
        }
        L.tiles_x = div_up(L.w, kFastTileW);
        L.pitch = L.tiles_x * kFastTileW;
        L.off = off;
        L.row_base = rows;
        L.tile_base = tiles;
        off += (unsigned)L.pitch * (unsigned)((L.h + kBlurTileH - 1) / kBlurTileH * kBlurTileH);   // whole tile rows of the blurred copy
        rows += L.h;
        tiles += L.tiles_x * div_up(L.h, kFastTileH);
    }
    // (the descriptor kernel's 64-byte-wide fetches may run up to three tiles past the last one it needs)
    g.frame_bytes = (off + 3u * 128u + 255u) & ~255u;
    // long survivor lists (large frames) are selected by several workgroups per level, each with a band of tile rows
    g.bands = g.lv[0].tiles_x * div_up(g.lv[0].h, kFastTileH) > 1024 ? SA_SEL_BANDS : 1;
    g.total_rows = rows;
    g.total_tiles = tiles;
    return SLAM_AMD_OK;
}

// ---- pyramid plan ------------------------------------------------------------------------
// Regions of one axis for a group of levels [first, last] and `nt` tiles of `T` pixels of level
// `last`.  slot 0 is the source level first - 1, slot k the level first + k - 1.  Every tile owns
// [o0, o1) of each level (the tiles' shares partition the level) and computes [c0, c1): its share
// plus whatever the next level's computed range reads.  off[l][d] is the source index of
// destination index d of level l (padded by replication past the level size).
struct AxisPlan {
    std::vector<PyrSpan> span;   // [nt][slots]
    int max_len_even = 0, max_len_odd = 0;
    std::vector<int> max_len_slot;   // [slots]: longest computed range of any tile at that slot
};

static AxisPlan plan_axis(const std::vector<std::vector<int32_t>>& off, const int* size, int first, int last,
                          int T, int nt, bool is_x) {
    const int slots = last - first + 2;
    AxisPlan ap;
    ap.span.resize((size_t)nt * slots);
    ap.max_len_slot.assign(slots, 0);
    auto up4 = [](int v) { return (v + 3) & ~3; };
    for (int i = 0; i < nt; i++) {
        PyrSpan* sp = &ap.span[(size_t)i * slots];
        int need0 = 0, need1 = 0;   // needed range of the level above (slot k + 1), exact
        for (int k = slots - 1; k >= 0; k--) {
            const int l = first + k - 1;
            const int n = size[l];
            const int padded = is_x ? up4(n) : n;     // x: whole dwords are produced and stored
            int o0 = 0, o1 = 0;
            if (k >= 1) {
                auto bound = [&](int t) {
                    if (t <= 0) return 0;
                    if (t >= nt) return padded;
                    long long v = ((long long)t * T * n + size[last] / 2) / size[last];
                    int r = is_x ? (int)(v + 2) & ~3 : (int)v;
                    return std::min(std::max(r, 0), padded);
                };
                o0 = bound(i); o1 = bound(i + 1);
            }
            int lo, hi;
            if (k == slots - 1) { lo = o0; hi = o1; }
            else {
                // what the next level's needed range reads: both taps of its first and last index
                lo = off[l + 1][need0];
                hi = off[l + 1][need1 - 1] + 2;
                if (k >= 1 && o1 > o0) { lo = std::min(lo, o0); hi = std::max(hi, o1); }
            }
            if (hi <= lo) { lo = 0; hi = 2; }   // a tile without a share still stages a stub
            need0 = lo; need1 = hi;
            // computed range: the needed one widened to whole dwords in x (the extra positions are
            // produced from clamped source indices and never used)
            const int c0 = is_x ? (lo & ~3) : lo, c1 = is_x ? up4(hi) : hi;
            sp[k].c0 = (short)c0; sp[k].c1 = (short)c1; sp[k].o0 = (short)o0; sp[k].o1 = (short)o1;
            const int len = c1 - c0;
            if (k & 1) ap.max_len_odd = std::max(ap.max_len_odd, len); else ap.max_len_even = std::max(ap.max_len_even, len);
            ap.max_len_slot[k] = std::max(ap.max_len_slot[k], len);
        }
    }
    return ap;
}

// A chain buffer holds whole staged rows; the horizontal pass reads three dwords from the dword of a
// group's first tap, up to 11 bytes past the last one it needs: 16 spare bytes.
static size_t pyr_buffer_bytes(size_t area) { return ((area + 15) & ~(size_t)15) + 16; }

// The coefficient slices of a tile lie packed in LDS: every level's x slice (whole dwords of pixels, so
// multiples of four entries of 4 bytes; a level's slice starts where the longest slices of the levels before
// it end, the same place for every tile), then every level's y slice (entries of 2 bytes: 9 bits of weight
// and the source row within the staged region, which must fit 7 bits); the chain buffers that follow
// start on 16 bytes.
static size_t pyr_table_bytes(const AxisPlan& ax, const AxisPlan& ay) {
    size_t nx = 0, ny = 0;
    for (size_t k = 1; k < ax.max_len_slot.size(); k++) { nx += ax.max_len_slot[k]; ny += ay.max_len_slot[k]; }
    return 2 * kMaxLevels * sizeof(PyrSpan) + ((nx * 4 + ny * 2 + 15) & ~(size_t)15);   // the tile's spans come first
}

struct HostGroup { int first, last, T, ntx, nty; AxisPlan ax, ay; size_t lds; };
struct HostPyramidPlan {
    std::vector<std::vector<uint32_t>> xt, yt;   // per level
    std::vector<HostGroup> groups;
};

static int make_pyramid_plan(const PyrGeom& g, HostPyramidPlan& plan) {
    plan.xt.assign(g.nlevels, {}); plan.yt.assign(g.nlevels, {}); plan.groups.clear();
    if (g.nlevels < 2) return SLAM_AMD_OK;
    std::vector<std::vector<uint32_t>>& xt = plan.xt;
    std::vector<std::vector<uint32_t>>& yt = plan.yt;
    std::vector<HostGroup>& groups = plan.groups;
    // per level l >= 1: (source index | 8.8 weight of the second tap << 16) per destination index,
    // padded by replicating the last entry so that whole dwords of a row can be produced
    std::vector<std::vector<int32_t>> xo(g.nlevels), yo(g.nlevels);
    int ws[kMaxLevels], hs[kMaxLevels];
    for (int l = 0; l < g.nlevels; l++) {
        ws[l] = g.lv[l].w; hs[l] = g.lv[l].h;
        // the tile spans (PyrSpan) hold coordinates of a level, rounded up to whole dwords, in 16-bit signed fields
        if (((ws[l] + 3) & ~3) > 32767 || ((hs[l] + 3) & ~3) > 32767)
            SA_FAIL(SLAM_AMD_ERR_ARG, "image too large: a pyramid level may be at most 32764 pixels wide and high");
    }
    constexpr int kPad = 8;
    for (int l = 1; l < g.nlevels; l++) {
        const int sw = ws[l - 1], sh = hs[l - 1], dw = ws[l], dh = hs[l];
        const bool half = (sw == 2 * dw) && (sh == 2 * dh);
        std::vector<int32_t> o(std::max(dw, dh));
        std::vector<uint16_t> c(std::max(dw, dh));
        auto fill = [&](int src, int dst, std::vector<int32_t>& offs, std::vector<uint32_t>& tab) {
            linear_exact_axis(src, dst, half, o.data(), c.data());
            offs.resize((size_t)dst + kPad); tab.resize((size_t)dst + kPad);
            for (int d = 0; d < dst + kPad; d++) {
                const int e = std::min(d, dst - 1);
                if (o[e] > 0xffff) return false;
                offs[d] = o[e];
                tab[d] = (uint32_t)o[e] | ((uint32_t)c[e] << 16);
            }
            return true;
        };
        if (!fill(sw, dw, xo[l], xt[l]) || !fill(sh, dh, yo[l], yt[l])) SA_FAIL(SLAM_AMD_ERR_ARG, "image too large");
        // the kernel takes the taps of four neighbouring destination columns from eight source bytes
        for (size_t d = 0; d + 3 < xo[l].size(); d += 4)
            if (xo[l][d + 3] - xo[l][d] > 6) SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "scale factor too large for the pyramid kernel");
    }
    // groups of consecutive levels whose regions fit the LDS budget, tiles of 32 (else 16, 8) pixels
#ifndef SA_PYR_LDS_KB
#define SA_PYR_LDS_KB 48
#endif
    constexpr size_t kLdsBudget = SA_PYR_LDS_KB * 1024;
    auto try_group = [&](int first, int last, int T, HostGroup& out) {
        out.first = first; out.last = last; out.T = T;
        out.ntx = div_up(ws[last], T); out.nty = div_up(hs[last], T);
        out.ax = plan_axis(xo, ws, first, last, T, out.ntx, true);
        out.ay = plan_axis(yo, hs, first, last, T, out.nty, false);
        const size_t even = (size_t)out.ax.max_len_even * out.ay.max_len_even, odd = (size_t)out.ax.max_len_odd * out.ay.max_len_odd;
        out.lds = pyr_table_bytes(out.ax, out.ay) + pyr_buffer_bytes(even) + pyr_buffer_bytes(odd);
        const int widest = std::max(out.ax.max_len_even, out.ax.max_len_odd);
        const int tallest = std::max(out.ay.max_len_even, out.ay.max_len_odd);
        return out.lds <= kLdsBudget && widest <= 1020 && tallest <= 128;
    };
    for (int first = 1; first < g.nlevels;) {
        HostGroup best;
        bool found = false;
#ifndef SA_PYR_TILES
#define SA_PYR_TILES {32, 16, 8}
#endif
        for (int T : SA_PYR_TILES) {
            // the longest run of levels from `first` that fits with this tile size; larger tiles
            // are preferred (fewer, larger workgroups: less recomputation at the tile borders)
            int last = first;
            HostGroup cand;
#ifndef SA_PYR_MAX_GROUP
#define SA_PYR_MAX_GROUP kMaxLevels
#endif
            while (last < g.nlevels && last - first < SA_PYR_MAX_GROUP && try_group(first, last, T, cand)) { best = cand; found = true; last++; }
            if (found) break;
        }
        if (!found) SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "scale factor too large for the pyramid kernel");
        groups.push_back(best);
        first = best.last + 1;
    }
    return SLAM_AMD_OK;
}

static int build_tables(slam_amd_extractor* ex) {
    const PyrGeom& g = ex->g;
    ex->pyr_groups.clear();
    HostPyramidPlan plan;
    int prc = make_pyramid_plan(g, plan);
    if (prc) return prc;
    if (g.nlevels < 2) return SLAM_AMD_OK;
    std::vector<std::vector<uint32_t>>& xt = plan.xt;
    std::vector<std::vector<uint32_t>>& yt = plan.yt;
    std::vector<HostGroup>& groups = plan.groups;
    // one device allocation: tables of every level, then the spans of every group
    std::vector<uint8_t> host;
    auto put = [&](const void* src, size_t bytes) {
        const size_t o = (host.size() + 15) & ~(size_t)15;
        host.resize(o + bytes);
        memcpy(host.data() + o, src, bytes);
        return o;
    };
    size_t o_xt[kMaxLevels] = {}, o_yt[kMaxLevels] = {};
    for (int l = 1; l < g.nlevels; l++) {
        o_xt[l] = put(xt[l].data(), xt[l].size() * 4);
        o_yt[l] = put(yt[l].data(), yt[l].size() * 4);
    }
    std::vector<size_t> o_sx(groups.size()), o_sy(groups.size());
    for (size_t i = 0; i < groups.size(); i++) {
        o_sx[i] = put(groups[i].ax.span.data(), groups[i].ax.span.size() * sizeof(PyrSpan));
        o_sy[i] = put(groups[i].ay.span.data(), groups[i].ay.span.size() * sizeof(PyrSpan));
    }
    int rc = ex->tables.reserve(host.size());
    if (rc) return rc;
    SA_HIP(hipMemcpy(ex->tables.p, host.data(), host.size(), hipMemcpyHostToDevice));
    const uint8_t* base = ex->tables.as<uint8_t>();
    for (int l = 1; l < g.nlevels; l++) {
        ex->xtab[l] = reinterpret_cast<const uint32_t*>(base + o_xt[l]);
        ex->ytab[l] = reinterpret_cast<const uint32_t*>(base + o_yt[l]);
    }
    for (size_t i = 0; i < groups.size(); i++) {
        const HostGroup& hg = groups[i];
        slam_amd_extractor::PyrGroupPlan pg;
        pg.first = hg.first; pg.last = hg.last; pg.ntx = hg.ntx; pg.nty = hg.nty;
        pg.tab_bytes = (int)pyr_table_bytes(hg.ax, hg.ay);
        int xat = 0, yat = 0;
        for (int k = 1; k < (int)hg.ax.max_len_slot.size(); k++) {
            pg.xat[k] = (short)xat; pg.yat[k] = (short)yat;
            xat += hg.ax.max_len_slot[k]; yat += hg.ay.max_len_slot[k];
        }
        pg.x_entries = xat;
        pg.buf_even = (int)pyr_buffer_bytes((size_t)hg.ax.max_len_even * hg.ay.max_len_even);
        pg.buf_odd = (int)pyr_buffer_bytes((size_t)hg.ax.max_len_odd * hg.ay.max_len_odd);
        pg.spanx = reinterpret_cast<const PyrSpan*>(base + o_sx[i]);
        pg.spany = reinterpret_cast<const PyrSpan*>(base + o_sy[i]);
        pg.lds_bytes = hg.lds;
        ex->pyr_groups.push_back(pg);
    }
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_extractor_create(const slam_amd_orb_params* params, int width, int height,
                                         int max_batch, int max_keypoints, int device,
                                         slam_amd_extractor** out) {
    if (!out) SA_FAIL(SLAM_AMD_ERR_ARG, "out is null");
    *out = nullptr;
    int rc = check_orb_params(params);
    if (rc) return rc;
    if (width <= 0 || height <= 0 || width > 16383 || height > 16383)
        SA_FAIL(SLAM_AMD_ERR_ARG, "image size out of range");
    if (max_batch <= 0 || max_batch > 65535) SA_FAIL(SLAM_AMD_ERR_ARG, "max_batch out of range");
    if (max_keypoints <= 0) SA_FAIL(SLAM_AMD_ERR_ARG, "max_keypoints must be positive");
    rc = use_device(device);
    if (rc) return rc;
    slam_amd_extractor* ex = new (std::nothrow) slam_amd_extractor();
    if (!ex) SA_FAIL(SLAM_AMD_ERR_NOMEM, "out of host memory");
    ex->params = *params;
    ex->width = width; ex->height = height;
    ex->max_batch = max_batch; ex->max_keypoints = max_keypoints; ex->device = device;
    auto fail = [&](int code) { slam_amd_extractor_destroy(ex); return code; };
    if ((rc = build_geometry(ex))) return fail(rc);
    if ((rc = build_tables(ex))) return fail(rc);
    const PyrGeom& g = ex->g;
    const size_t B = (size_t)max_batch;
    if ((rc = ex->pyr.reserve(B * g.frame_bytes))) return fail(rc);
    if ((rc = ex->blur.reserve(B * g.frame_bytes))) return fail(rc);
    // (padded to 256 bytes: the per-call fill is rounded up to whole 256-byte blocks)
    if ((rc = ex->hist.reserve(((B * (size_t)g.nlevels * g.bands * 256 * sizeof(unsigned)) + 255) & ~(size_t)255))) return fail(rc);
    if ((rc = ex->tilecount.reserve(B * (size_t)g.total_tiles * sizeof(unsigned)))) return fail(rc);
    if ((rc = ex->tilelist.reserve(B * (size_t)g.total_tiles * kTileListCap * sizeof(uint32_t)))) return fail(rc);
    if ((rc = ex->sel.reserve(B * sizeof(FrameSelect)))) return fail(rc);
    if ((rc = ex->kpxy.reserve(B * (size_t)max_keypoints * sizeof(uint32_t)))) return fail(rc);
    {
        // the FAST kernel finds its tile by one table read instead of a search over the levels and a division
        // The table's order is the order of dispatch.  Without whole frames per XCD, XCD k takes the k-th
        // eighth of the table: the tile rows of all levels are dealt to eight bins in turn and the bins put
        // one after the other, so that every eighth holds rows of every level (level 0 first would give one
        // XCD nothing but full interior tiles and another the small top levels).
        std::vector<uint32_t> bins[8];
        int row_number = 0;
        for (int l = 0; l < g.nlevels; l++) {
            const int nty = div_up(g.lv[l].h, kFastTileH);
            if (g.lv[l].tiles_x > 0xfff || nty > 0xfff) return fail((set_error("level too large for the tile table"), SLAM_AMD_ERR_ARG));
            for (int ty = 0; ty < nty; ty++, row_number++)
                for (int tx = 0; tx < g.lv[l].tiles_x; tx++)
                    bins[row_number & 7].push_back((uint32_t)l | ((uint32_t)tx << 4) | ((uint32_t)ty << 16) | ((uint32_t)(ty * g.bands / nty) << 28));
        }
        // two orders, one after the other: levels, rows, columns (whole frames per XCD: the heavy tiles of a frame
        // first, the light ones at the end of the launch), then the bins
        std::vector<uint32_t> td;
        for (int k = 0; k < 8; k++) td.insert(td.end(), bins[k].begin(), bins[k].end());
        std::vector<uint32_t> natural(td);
        std::sort(natural.begin(), natural.end(), [](uint32_t x, uint32_t y) {
            const uint32_t kx = ((x & 15u) << 24) | (((x >> 16) & 0xfffu) << 12) | ((x >> 4) & 0xfffu);
            const uint32_t ky = ((y & 15u) << 24) | (((y >> 16) & 0xfffu) << 12) | ((y >> 4) & 0xfffu);
            return kx < ky;
        });
        td.insert(td.begin(), natural.begin(), natural.end());
        // an entry carries the tile's number in the frame as well (levels, rows, columns), so that the kernel needs
        // nothing but the entry to know where its results go
        std::vector<uint2> td2(td.size());
        for (size_t i = 0; i < td.size(); i++) {
            const uint32_t e = td[i];
            const LevelGeom& L = g.lv[e & 15u];
            td2[i] = make_uint2(e, (uint32_t)(L.tile_base + (int)((e >> 16) & 0xfffu) * L.tiles_x + (int)((e >> 4) & 0xfffu)));
        }
        if ((rc = ex->tiledesc.reserve(td2.size() * sizeof(uint2)))) return fail(rc);
        if (hipMemcpy(ex->tiledesc.p, td2.data(), td2.size() * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            set_error("upload of the tile table failed");
            return fail(SLAM_AMD_ERR_HIP);
        }
    }
    {
        // per level: width, height, pitch, offset, scale (bit pattern), three zeros
        int32_t lg[8 * kMaxLevels] = {};
        for (int l = 0; l < g.nlevels; l++) {
            lg[8 * l + 0] = g.lv[l].w; lg[8 * l + 1] = g.lv[l].h; lg[8 * l + 2] = g.lv[l].pitch;
            lg[8 * l + 3] = (int32_t)g.lv[l].off; memcpy(&lg[8 * l + 4], &g.lv[l].scale, 4);
        }
        if ((rc = ex->lanegeom.reserve(sizeof lg))) return fail(rc);
        if (hipMemcpy(ex->lanegeom.p, lg, sizeof lg, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            set_error("upload of the level table failed");
            return fail(SLAM_AMD_ERR_HIP);
        }
    }
    // packed positions start out as zeros: a slot that was never written decodes to (0, 0, level 0),
    // which the describe kernel rejects, instead of an arbitrary address
    if (hipMemset(ex->kpxy.p, 0, ex->kpxy.cap) != hipSuccess || hipMemset(ex->sel.p, 0, ex->sel.cap) != hipSuccess ||
        hipMemset(ex->hist.p, 0, ex->hist.cap) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMemset of the selection buffers failed");
        return fail(SLAM_AMD_ERR_HIP);
    }
    ex->hist_clean = true;
    hipError_t e = hipStreamCreateWithFlags(&ex->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__));
    *out = ex;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_extractor_destroy(slam_amd_extractor* ex) {
    if (!ex) return SLAM_AMD_OK;
    (void)hipSetDevice(ex->device);
    if (ex->stream) { (void)hipStreamSynchronize(ex->stream); (void)hipStreamDestroy(ex->stream); }
    for (DevBuf* b : {&ex->tables, &ex->pyr, &ex->blur, &ex->hist, &ex->tilecount, &ex->tilelist, &ex->sel, &ex->kpxy, &ex->lanegeom, &ex->tiledesc,
                      &ex->h_img, &ex->h_kp, &ex->h_desc, &ex->h_count})
        b->release();
    delete ex;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_extractor_levels(const slam_amd_extractor* ex, int* nlevels, int* widths,
                                         int* heights, int* quotas, float* scales) {
    if (!ex) SA_FAIL(SLAM_AMD_ERR_ARG, "extractor is null");
    if (nlevels) *nlevels = ex->g.nlevels;
    for (int l = 0; l < ex->g.nlevels; l++) {
        if (widths) widths[l] = ex->g.lv[l].w;
        if (heights) heights[l] = ex->g.lv[l].h;
        if (quotas) quotas[l] = ex->g.lv[l].quota;
        if (scales) scales[l] = ex->g.lv[l].scale;
    }
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_extract_batch_async(slam_amd_extractor* ex, const uint8_t* d_images, int batch,
                                            size_t frame_pitch, int row_stride,
                                            slam_amd_keypoint* d_kp, uint8_t* d_desc, int32_t* d_count,
                                            int capacity, void* stream) {
    if (!ex || !d_images || !d_kp || !d_desc) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    if (batch <= 0 || batch > ex->max_batch) SA_FAIL(SLAM_AMD_ERR_ARG, "batch exceeds the context's max_batch");
    if (capacity <= 0 || capacity > ex->max_keypoints)
        SA_FAIL(SLAM_AMD_ERR_ARG, "capacity exceeds the context's max_keypoints");
    if (row_stride < ex->width) SA_FAIL(SLAM_AMD_ERR_ARG, "row_stride smaller than the image width");
    if (batch > 1 && frame_pitch < (size_t)row_stride * (ex->height - 1) + ex->width)
        SA_FAIL(SLAM_AMD_ERR_ARG, "frame_pitch smaller than one frame");
    if ((reinterpret_cast<uintptr_t>(d_desc) & 15) || (reinterpret_cast<uintptr_t>(d_kp) & 3))
        SA_FAIL(SLAM_AMD_ERR_ARG, "d_desc must be 16-byte aligned, d_kp 4-byte aligned");
    SA_HIP(hipSetDevice(ex->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const PyrGeom& g = ex->g;
    const int aligned = ((reinterpret_cast<uintptr_t>(d_images) & 3) == 0) && (row_stride % 4 == 0) &&
                        (batch == 1 || frame_pitch % 4 == 0);
    ex->last_img0 = d_images; ex->last_frame_pitch = frame_pitch; ex->last_row_stride = row_stride;

    // The histograms must be all zero when this call's FAST kernel runs.  Every completed call leaves them so (the
    // describe kernel zeroes them), so does creation; only a call that failed half way and could not even enqueue the
    // fill leaves the flag down.  Under stream capture nothing runs now: the flag describes the buffer, not the
    // recording, and stays as it is (a recording made while it is down carries its own fill).
    const bool capturing = stream_capturing(st);
    if (!ex->hist_clean) {
        ProfScope ps(kStageMemset, st);
        SA_HIP(hipMemsetAsync(ex->hist.p, 0, ex->hist.cap, st));
    }
    struct Restore {        // an eager call that returns early puts the histograms back to zero behind whatever it enqueued
        slam_amd_extractor* ex; hipStream_t st; bool armed;
        ~Restore() {
            if (!armed) return;
            ex->hist_clean = hipMemsetAsync(ex->hist.p, 0, ex->hist.cap, st) == hipSuccess;
            if (!ex->hist_clean) (void)hipGetLastError();
        }
    } restore = {ex, st, !capturing};
    if (!capturing) ex->hist_clean = false;     // until this call's describe kernel is in the stream
    int rc;
    prof_mark(kStageResize, true, st);
    for (const slam_amd_extractor::PyrGroupPlan& pg : ex->pyr_groups) {
        PyramidArgs a;
        memset(&a, 0, sizeof a);
        a.src0 = d_images; a.src0_frame_pitch = frame_pitch; a.src0_pitch = row_stride; a.src0_aligned = aligned;
        a.pyr = ex->pyr.as<uint8_t>(); a.g = g;
        a.first = pg.first; a.last = pg.last; a.ntx = pg.ntx; a.nty = pg.nty;
        a.spanx = pg.spanx; a.spany = pg.spany;
        for (int l = 1; l < g.nlevels; l++) { a.xtab[l] = ex->xtab[l]; a.ytab[l] = ex->ytab[l]; }
        a.tab_bytes = pg.tab_bytes; a.x_entries = pg.x_entries;
        for (int k = 0; k <= kMaxLevels; k++) { a.xat[k] = pg.xat[k]; a.yat[k] = pg.yat[k]; } a.buf_even = pg.buf_even; a.buf_odd = pg.buf_odd;
        if ((rc = launch_pyramid(a, batch, pg.lds_bytes, st))) return rc;
    }
    prof_mark(kStageResize, false, st);
    FastArgs fa;
    fa.img0 = d_images; fa.img0_frame_pitch = frame_pitch; fa.img0_pitch = row_stride; fa.img0_aligned = aligned;
    fa.pyr = ex->pyr.as<uint8_t>(); fa.blur = ex->blur.as<uint8_t>(); fa.hist = ex->hist.as<unsigned>();
    fa.tilecount = ex->tilecount.as<unsigned>(); fa.tilelist = ex->tilelist.as<uint32_t>(); fa.g = g;
    fa.grid_shift = ex->grid_shift; fa.tiledesc = ex->tiledesc.as<uint2>(); fa.tiles_eighth = 0; fa.frames_by_xcd = 0;
    { ProfScope ps(kStageFast, st); if ((rc = launch_fast(fa, batch, st))) return rc; }
    SelectCompactArgs sc;
    sc.hist = ex->hist.as<unsigned>(); sc.tilecount = ex->tilecount.as<unsigned>(); sc.tilelist = ex->tilelist.as<uint32_t>();
    sc.keycap = ex->select_keycap; sc.quadtree = ex->quadtree; sc.sel = ex->sel.as<FrameSelect>(); sc.d_count = d_count;
    sc.kp = d_kp; sc.kpxy = ex->kpxy.as<uint32_t>(); sc.capacity = capacity; sc.g = g;
    { ProfScope ps(kStageSelect, st); if ((rc = launch_select_compact(sc, batch, st))) return rc; }
    DescribeArgs da;
    da.img0 = d_images; da.img0_frame_pitch = frame_pitch; da.img0_pitch = row_stride;
    da.img0_aligned = aligned; da.pyr = ex->pyr.as<uint8_t>(); da.blur = ex->blur.as<uint8_t>(); da.sel = ex->sel.as<FrameSelect>(); da.kp = d_kp;
    da.kpxy = ex->kpxy.as<uint32_t>(); da.lanegeom = ex->lanegeom.as<int32_t>(); da.desc = d_desc; da.capacity = capacity; da.batch = batch; da.g = g;
    da.hist_reset = ex->hist.as<unsigned>();
    { ProfScope ps(kStageDescribe, st); if ((rc = launch_describe(da, batch, st))) return rc; }
    restore.armed = false;
    if (!capturing) ex->hist_clean = true;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_extract(slam_amd_extractor* ex, const uint8_t* gray, int width, int height,
                                int stride, slam_amd_keypoint* kp_out, uint8_t* desc_out,
                                int capacity, int* n_out) {
    if (!ex || !n_out) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    *n_out = 0;
    if (!gray || width <= 0 || height <= 0) return SLAM_AMD_OK;  // empty image -> empty result (frame.cpp:24)
    if (width != ex->width || height != ex->height)
        SA_FAIL(SLAM_AMD_ERR_ARG, "image size differs from the size the extractor was created for");
    if (stride < width) SA_FAIL(SLAM_AMD_ERR_ARG, "stride smaller than width");
    if (capacity < 0 || (capacity > 0 && (!kp_out || !desc_out))) SA_FAIL(SLAM_AMD_ERR_ARG, "null output");
    SA_HIP(hipSetDevice(ex->device));
    int rc;
    ex->h_img_pitch = (width + 63) & ~63;
    if ((rc = ex->h_img.reserve((size_t)ex->h_img_pitch * height))) return rc;
    const int cap = ex->max_keypoints;
    if ((rc = ex->h_kp.reserve((size_t)cap * sizeof(slam_amd_keypoint)))) return rc;
    if ((rc = ex->h_desc.reserve((size_t)cap * 32))) return rc;
    if ((rc = ex->h_count.reserve(sizeof(int32_t)))) return rc;
    ex->h_last_n = -1;
    SA_HIP(hipMemcpy2DAsync(ex->h_img.p, ex->h_img_pitch, gray, stride, width, height,
                            hipMemcpyHostToDevice, ex->stream));
    rc = slam_amd_extract_batch_async(ex, ex->h_img.as<uint8_t>(), 1, 0, ex->h_img_pitch,
                                      ex->h_kp.as<slam_amd_keypoint>(), ex->h_desc.as<uint8_t>(),
                                      ex->h_count.as<int32_t>(), cap, ex->stream);
    if (rc) return rc;
    int32_t n = 0;
    SA_HIP(hipMemcpyAsync(&n, ex->h_count.p, sizeof n, hipMemcpyDeviceToHost, ex->stream));
    SA_HIP(hipStreamSynchronize(ex->stream));
    *n_out = n;
    if (n > cap) SA_FAIL(SLAM_AMD_ERR_CAPACITY, "more keypoints than the extractor's max_keypoints");
    ex->h_last_n = n;
    return slam_amd_extract_fetch(ex, kp_out, desc_out, capacity, n_out);
}

extern "C" int slam_amd_extract_fetch(slam_amd_extractor* ex, slam_amd_keypoint* kp_out, uint8_t* desc_out,
                                      int capacity, int* n_out) {
    if (!ex || !n_out) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    if (ex->h_last_n < 0) SA_FAIL(SLAM_AMD_ERR_ARG, "no complete slam_amd_extract result is held by this context");
    const int n = ex->h_last_n;
    *n_out = n;
    if (capacity < 0 || (capacity > 0 && (!kp_out || !desc_out))) SA_FAIL(SLAM_AMD_ERR_ARG, "null output");
    if (n > capacity) SA_FAIL(SLAM_AMD_ERR_CAPACITY, "more keypoints than the caller's capacity");
    SA_HIP(hipSetDevice(ex->device));
    if (n > 0) {
        SA_HIP(hipMemcpyAsync(kp_out, ex->h_kp.p, (size_t)n * sizeof(slam_amd_keypoint), hipMemcpyDeviceToHost, ex->stream));
        SA_HIP(hipMemcpyAsync(desc_out, ex->h_desc.p, (size_t)n * 32, hipMemcpyDeviceToHost, ex->stream));
        SA_HIP(hipStreamSynchronize(ex->stream));
    }
    return SLAM_AMD_OK;
}

static int debug_level(slam_amd_extractor* ex, int frame, int level, bool blurred,
                       uint8_t* host_out, int host_stride) {
    if (!ex || !host_out) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    if (level < 0 || level >= ex->g.nlevels || frame < 0 || frame >= ex->max_batch)
        SA_FAIL(SLAM_AMD_ERR_ARG, "level or frame out of range");
    const LevelGeom& L = ex->g.lv[level];
    if (host_stride < L.w) SA_FAIL(SLAM_AMD_ERR_ARG, "host_stride too small");
    SA_HIP(hipSetDevice(ex->device));
    SA_HIP(hipDeviceSynchronize());
    if (level == 0 && !blurred) {
        if (!ex->last_img0) SA_FAIL(SLAM_AMD_ERR_ARG, "no batch has been processed yet");
        SA_HIP(hipMemcpy2D(host_out, host_stride, ex->last_img0 + (size_t)frame * ex->last_frame_pitch,
                           ex->last_row_stride, L.w, L.h, hipMemcpyDeviceToHost));
    } else if (blurred) {
        // stored in tiles: fetch the level's whole tile rows and put the pixels back in rows
        const int rows8 = (L.h + kBlurTileH - 1) / kBlurTileH * kBlurTileH;
        std::vector<uint8_t> tiled((size_t)L.pitch * rows8);
        SA_HIP(hipMemcpy(tiled.data(), ex->blur.as<uint8_t>() + (size_t)frame * ex->g.frame_bytes + L.off, tiled.size(),
                         hipMemcpyDeviceToHost));
        for (int y = 0; y < L.h; y++)
            for (int x = 0; x < L.w; x++)
                host_out[(size_t)y * host_stride + x] = tiled[blur_tiled_offset((unsigned)L.pitch, (unsigned)x, (unsigned)y)];
    } else {
        SA_HIP(hipMemcpy2D(host_out, host_stride, ex->pyr.as<uint8_t>() + (size_t)frame * ex->g.frame_bytes + L.off, L.pitch,
                           L.w, L.h, hipMemcpyDeviceToHost));
    }
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_extractor_set_grid(slam_amd_extractor* ex, int cell) {
    if (!ex) SA_FAIL(SLAM_AMD_ERR_ARG, "extractor is null");
    if (cell != 0 && cell != 8 && cell != 16 && cell != 32)
        SA_FAIL(SLAM_AMD_ERR_ARG, "grid cell must be 0 (off), 8, 16 or 32");
    if (cell && ex->quadtree) SA_FAIL(SLAM_AMD_ERR_ARG, "the grid and the quadtree distribution exclude each other");
    ex->grid_shift = cell == 8 ? 3 : cell == 16 ? 4 : cell == 32 ? 5 : 0;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_extractor_set_quadtree(slam_amd_extractor* ex, int on) {
    if (!ex) SA_FAIL(SLAM_AMD_ERR_ARG, "extractor is null");
    if (on && ex->grid_shift) SA_FAIL(SLAM_AMD_ERR_ARG, "the grid and the quadtree distribution exclude each other");
    if (on) {
        // path codes are 32 bits (root index, then two bits per split) and positions 12 bits per axis
        const PyrGeom& g = ex->g;
        for (int l = 0; l < g.nlevels; l++) {
            const int W = g.lv[l].w - 2 * g.edge, H = g.lv[l].h - 2 * g.edge;
            if (W <= 0 || H <= 0) continue;
            if (g.lv[l].w > 4096 || g.lv[l].h > 4096)
                SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "quadtree distribution: levels larger than 4096 pixels per side");
            const int nini = std::max(1, (2 * W + H) / (2 * H)), rw = (W + nini - 1) / nini;
            int dmax = 0;
            while ((1 << dmax) < std::max(rw, H)) dmax++;
            if (2 * dmax >= 32 || ((long long)(nini - 1) >> (32 - 2 * dmax)) != 0)
                SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "quadtree distribution: image too elongated for 32-bit path codes");
        }
    }
    ex->quadtree = on ? 1 : 0;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_extractor_debug_set_select_cache(slam_amd_extractor* ex, int entries) {
    if (!ex) SA_FAIL(SLAM_AMD_ERR_ARG, "extractor is null");
    if (entries < 0 || entries > 4096) SA_FAIL(SLAM_AMD_ERR_ARG, "entries out of range");
    ex->select_keycap = entries;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_extractor_debug_level(slam_amd_extractor* ex, int frame, int level,
                                              uint8_t* host_out, int host_stride) {
    return debug_level(ex, frame, level, false, host_out, host_stride);
}

extern "C" int slam_amd_extractor_debug_blurred_level(slam_amd_extractor* ex, int frame, int level,
                                                      uint8_t* host_out, int host_stride) {
    return debug_level(ex, frame, level, true, host_out, host_stride);
}

// =============================================================================
// Matcher
// =============================================================================
struct slam_amd_matcher {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    DevBuf best;                       // running keys of the async calls
    DevBuf part;                       // per-split (best, second) keys when the ratio test is on
    DevBuf sorted;                     // both sets in grid-cell order for windowed matching
    bool baked = false;                // an _async call ran under stream capture: scratch pointers live in a graph
    bool best_clean = false;           // every key of `best` is "no match" once the work enqueued so far has run: filled when it is
                                       // allocated, the finalize kernel puts back what it reads, and a call that fails
                                       // in between enqueues the fill itself before it returns
    unsigned char* static_scratch = nullptr;  // this device's copy of the file-scope scratch array (no-alloc entry point)
    int window_mode = 0;               // 0: choose by size, 1: always the full sweep, 2: cell-sorted sets and the popcount kernel, 3: cell-sorted sets and the matrix cores
    int sweep_kernel = 0;              // full sweep: 0 = matrix-core kernel (gates and ratio option included), 1 = VALU popcount kernel
    float ratio = 0.f;                 // options beyond the reference, both off by default
    int histo_length = 0;
    DevBuf ticket;                     // [kFinalTicketProblems] counters of the finalize kernel, zero between calls
    DevBuf h_d1, h_k1, h_d2, h_k2, h_ids, h_xy, h_out, h_n, h_aux;  // host-call staging
};

// Scratch of slam_amd_frame_match_noalloc_async: running keys of up to kStaticRows queries, then
// the (best, second) planes of the ratio option, then a ticket counter.  One copy per GPU, held until the process ends;
// host code reaches it only through the address hipGetSymbolAddress returns for the current device.
constexpr size_t kStaticRows = SLAM_AMD_NOALLOC_MAX_QUERIES;
constexpr size_t kStaticBestBytes = kStaticRows * sizeof(unsigned);
constexpr size_t kStaticPartBytes = kStaticRows * kMaxSecondSplits * sizeof(uint2);
constexpr size_t kStaticTicketBytes = 256;      // the ticket counter of the workgroups that turn keys into records (zero between calls)
static __device__ __align__(256) unsigned char scratch[kStaticBestBytes + kStaticPartBytes + kStaticTicketBytes];

extern "C" int slam_amd_matcher_create(int device, slam_amd_matcher** out) {
    if (!out) SA_FAIL(SLAM_AMD_ERR_ARG, "out is null");
    *out = nullptr;
    int rc = use_device(device);
    if (rc) return rc;
    slam_amd_matcher* m = new (std::nothrow) slam_amd_matcher();
    if (!m) SA_FAIL(SLAM_AMD_ERR_NOMEM, "out of host memory");
    m->device = device;
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0)
        m->n_cu = cu;
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete m; return hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__); }
    // `device` is current (use_device): this is that GPU's copy of the array
    void* sp = nullptr;
    e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !sp) {
        (void)hipStreamDestroy(m->stream);
        delete m;
        return hip_fail(e != hipSuccess ? e : hipErrorInvalidSymbol, "hipGetSymbolAddress(scratch)", __FILE__, __LINE__);
    }
    m->static_scratch = static_cast<unsigned char*>(sp);
    // ticket counters of the finalize kernel's workgroups, one per problem of a call: zero here and after every call
    rc = m->ticket.reserve(kFinalTicketProblems * sizeof(unsigned));
    if (rc == SLAM_AMD_OK && (hipMemset(m->ticket.p, 0, m->ticket.cap) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) {
        (void)hipGetLastError();
        set_error("fill of the matcher's ticket counters failed");
        rc = SLAM_AMD_ERR_HIP;
    }
    if (rc) { slam_amd_matcher_destroy(m); return rc; }
    *out = m;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_matcher_destroy(slam_amd_matcher* m) {
    if (!m) return SLAM_AMD_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    for (DevBuf* b : {&m->best, &m->part, &m->sorted, &m->ticket, &m->h_d1, &m->h_k1, &m->h_d2, &m->h_k2, &m->h_ids, &m->h_xy, &m->h_out,
                      &m->h_n, &m->h_aux})
        b->release();
    delete m;
    return SLAM_AMD_OK;
}

// Device scratch of the matcher grows on demand, except where growing would break a
// captured graph: hipFree / hipMalloc are illegal while the stream is capturing, and a
// graph instantiated earlier keeps the old pointer.  Fill bytes are rounded up to 256 so
// that every hipMemsetAsync covers whole 16-byte multiples.
static size_t fill_bytes(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

static int grow_scratch(slam_amd_matcher* m, DevBuf& b, size_t bytes, bool capturing) {
    if (bytes <= b.cap) return SLAM_AMD_OK;
    if (capturing || m->baked)
        SA_FAIL(SLAM_AMD_ERR_CAPACITY, "matcher scratch too small and a captured graph holds (or is recording) its "
                                       "address: call slam_amd_matcher_reserve with the largest sizes before capturing");
    const int rc = b.reserve(bytes);
    if (rc == SLAM_AMD_OK && &b == &m->best) {
        // fresh memory starts out as "no match" physically, not by a fill that a later call (perhaps only recorded) enqueues
        m->best_clean = hipMemset(b.p, 0xff, b.cap) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
        if (!m->best_clean) { (void)hipGetLastError(); SA_FAIL(SLAM_AMD_ERR_HIP, "fill of the matcher scratch failed"); }
    }
    return rc;
}

// Byte layout of the cell-sorted copies of both sets (set 0 = queries, 1 = train): per row
// descriptors (32 B), position (8 B), index word (4 B); a box (16 B) per 32 rows (one per 64 is used by the popcount kernel).
struct SortedLayout {
    int scap[2];
    size_t o_desc[2], o_xy[2], o_idx[2], o_box[2], total;
};
static SortedLayout sorted_layout(int batch, int nq, int n2) {
    SortedLayout L;
    size_t off = 0;
    for (int s = 0; s < 2; s++) {
        const int bound = s == 0 ? nq : n2;
        L.scap[s] = div_up(std::max(bound, 1), 64) * 64;
        const size_t r = (size_t)batch * L.scap[s];
        L.o_desc[s] = off; off += r * 32;
        L.o_xy[s] = off; off += r * 8;
        L.o_idx[s] = off; off += r * 4;
        L.o_box[s] = off; off += (r / 32) * 16;
        off = (off + 255) & ~(size_t)255;
    }
    L.total = off;
    return L;
}

extern "C" int slam_amd_matcher_reserve(slam_amd_matcher* m, int batch, int query_rows, int train_rows) {
    if (!m) SA_FAIL(SLAM_AMD_ERR_ARG, "matcher is null");
    if (batch <= 0 || query_rows < 0 || train_rows < 0) SA_FAIL(SLAM_AMD_ERR_ARG, "bad size");
    SA_HIP(hipSetDevice(m->device));
    const size_t rows = std::max((size_t)batch * (size_t)query_rows, (size_t)1);
    int rc;
    if ((rc = grow_scratch(m, m->best, fill_bytes(rows * sizeof(unsigned)), false))) return rc;
    if (m->ratio > 0.f && (rc = grow_scratch(m, m->part, fill_bytes(rows * kMaxSecondSplits * sizeof(uint2)), false))) return rc;
    if (train_rows > 0 &&
        (rc = grow_scratch(m, m->sorted, sorted_layout(batch, query_rows, train_rows).total, false))) return rc;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_matcher_debug_set_window_mode(slam_amd_matcher* m, int mode) {
    if (!m || mode < 0 || mode > 3) SA_FAIL(SLAM_AMD_ERR_ARG, "mode must be 0, 1, 2 or 3");
    m->window_mode = mode;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_matcher_debug_set_sweep_kernel(slam_amd_matcher* m, int mode) {
    if (!m || mode < 0 || mode > 1) SA_FAIL(SLAM_AMD_ERR_ARG, "mode must be 0 or 1");
    m->sweep_kernel = mode;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_matcher_set_options(slam_amd_matcher* m, float ratio, int histo_length) {
    if (!m) SA_FAIL(SLAM_AMD_ERR_ARG, "matcher is null");
    if (!(ratio >= 0.f)) SA_FAIL(SLAM_AMD_ERR_ARG, "ratio must be >= 0 (0 switches the test off)");
    if (histo_length < 0 || histo_length > 64) SA_FAIL(SLAM_AMD_ERR_ARG, "histo_length must be in [0, 64]");
    m->ratio = ratio;
    m->histo_length = histo_length;
    return SLAM_AMD_OK;
}

static int dmax_of(float max_dist) {
    if (!(max_dist >= 0.f)) return -1;  // also NaN
    if (max_dist >= 256.f) return 256;
    return (int)floorf(max_dist);
}

static int match_common(slam_amd_matcher* m, int batch,
                        const uint8_t* d1, const slam_amd_keypoint* k1, int n1, const int32_t* d_n1, int cap1,
                        const uint8_t* d2, const slam_amd_keypoint* k2, int n2, const int32_t* d_n2, int cap2,
                        const int32_t* ids, int n_ids, const float* q_xy,
                        float max_dist, float area, int max_level,
                        slam_amd_dmatch* d_out, int32_t* d_n_out, hipStream_t st, bool no_alloc = false,
                        bool chain = false) {
    if (!m || !d_out || !d_n_out) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    if (n1 < 0 || n2 < 0 || n_ids < 0 || batch <= 0) SA_FAIL(SLAM_AMD_ERR_ARG, "negative size");
    if (n2 >= (1 << 20)) SA_FAIL(SLAM_AMD_ERR_ARG, "train set larger than 2^20 - 1 rows");
    if (area != area) area = -2.f;      // a NaN window passes no pair, like any negative one other than -1 (= gate off)
    const bool gated = (area != -1.f) || (max_level != -1);
    if (ids && n_ids == 0) ids = nullptr;
    if ((n1 > 0 && !d1) || (n2 > 0 && !d2)) SA_FAIL(SLAM_AMD_ERR_ARG, "null descriptor pointer");
    if (gated && ((n1 > 0 && !k1 && !(q_xy && max_level == -1)) || (n2 > 0 && !k2)))
        SA_FAIL(SLAM_AMD_ERR_ARG, "keypoints are required when a gate is on");
    if ((reinterpret_cast<uintptr_t>(d1) & 15) || (reinterpret_cast<uintptr_t>(d2) & 15))
        SA_FAIL(SLAM_AMD_ERR_ARG, "descriptor pointers must be 16-byte aligned");
    SA_HIP(hipSetDevice(m->device));
    const int nq = ids ? n_ids : n1;
    const size_t rows = std::max((size_t)batch * cap1, (size_t)1);
    const bool second = m->ratio > 0.f;
    if (m->histo_length > 0 && nq > 0 && n2 > 0 && (!k1 || !k2))
        SA_FAIL(SLAM_AMD_ERR_ARG, "the rotation-consistency option needs the keypoints of both sets");
    // full sweep: distances on the matrix cores (gates and second-best keys included)
    const bool use_mfma = m->sweep_kernel == 0;
    // With the window gate on, both sets are first reordered by grid cell (copies in the matcher's scratch), so
    // that rows that are close in the arrays are close in the image; same result either way.  Train sets that
    // fit the matrix-core kernel's index field then go through that kernel, which skips the blocks of train rows
    // a window away from a wave's queries -- most of them, once the sets are ordered; larger ones go to the
    // popcount kernel over nearby tiles once there are enough pairs to repay the reordering (about 10^9).
    const bool can_sort = area > 0.f && area <= 3.0e38f && nq > 0 && n2 > 0 && !no_alloc && m->window_mode != 1;
    bool sorted_mfma = can_sort && use_mfma && n2 <= window_mfma_max_train() && m->window_mode != 2;
    bool windowed = can_sort && !chain && !sorted_mfma;
    if (windowed && m->window_mode == 0) windowed = (long long)nq * n2 >= (1LL << 30);
    MatchPlan plan = use_mfma ? plan_match_mfma(batch, nq, n2, m->n_cu, second) : plan_match(batch, nq, n2, m->n_cu, second);
    if (windowed) plan.splits = window_splits(batch, nq, n2, m->n_cu);
    if (sorted_mfma) plan.splits = window_mfma_splits(batch, nq, n2, m->n_cu);
    int rc;
    const bool capturing = !no_alloc && stream_capturing(st);
    unsigned* best_p;
    uint2* part_p;
    if (no_alloc) {
        // the per-device static array instead of the matcher's own buffers; nothing is allocated
        if (rows > kStaticRows)
            SA_FAIL(SLAM_AMD_ERR_CAPACITY, "slam_amd_frame_match_noalloc_async: more than "
                                           "SLAM_AMD_NOALLOC_MAX_QUERIES (32768) query rows");
        best_p = reinterpret_cast<unsigned*>(m->static_scratch);
        part_p = reinterpret_cast<uint2*>(m->static_scratch + kStaticBestBytes);
    } else {
        if (second) { if ((rc = grow_scratch(m, m->part, fill_bytes(rows * plan.splits * sizeof(uint2)), capturing))) return rc; }
        else if ((rc = grow_scratch(m, m->best, fill_bytes(rows * sizeof(unsigned)), capturing))) return rc;
        best_p = m->best.as<unsigned>();
        part_p = m->part.as<uint2>();
    }
    // every (query, split) cell of the planes is written by the kernel only where the split
    // holds train rows, and the single-plane keys are merged by atomicMin: both start as "no match".
    // The matcher's own single plane is kept in that state by the finalize kernel, which puts back
    // every key it reads; it is filled whole after it was allocated or a call did not get that far.
    const bool self_clean = !no_alloc && !second;
    struct Restore {        // an eager call that returns early puts the keys back to "no match" behind whatever it enqueued,
                            // so that a graph recorded without a fill still finds them clean when it is replayed
        slam_amd_matcher* m; hipStream_t st; bool armed;
        ~Restore() {
            if (!armed) return;
            m->best_clean = hipMemsetAsync(m->best.p, 0xff, m->best.cap, st) == hipSuccess;
            if (!m->best_clean) (void)hipGetLastError();
        }
    } restore = {m, st, false};
    if (self_clean) {
        if (!m->best_clean) {
            ProfScope ps(kStageMemset, st);
            SA_HIP(hipMemsetAsync(m->best.p, 0xff, m->best.cap, st));
        }
        // Under capture nothing runs now: the flag describes the buffer, not the recording, and stays as it is.
        if (!capturing) { m->best_clean = false; restore.armed = true; }    // until this call's finalize kernel is in the stream
    } else {
        const size_t bytes = second ? fill_bytes(rows * plan.splits * sizeof(uint2)) : fill_bytes(rows * sizeof(unsigned));
        ProfScope ps(kStageMemset, st);
        SA_HIP(hipMemsetAsync(second ? static_cast<void*>(part_p) : static_cast<void*>(best_p), 0xff, bytes, st));
    }
    const int dmax = dmax_of(max_dist);
    MatchArgs a;
    memset(&a, 0, sizeof a);
    if (windowed || sorted_mfma) {
        SpatialSortArgs sa_;
        memset(&sa_, 0, sizeof sa_);
        sa_.qdesc = d1; sa_.qkp = k1; sa_.d_n1 = d_n1; sa_.n1 = n1; sa_.cap1 = cap1;
        sa_.tdesc = d2; sa_.tkp = k2; sa_.d_n2 = d_n2; sa_.n2 = n2; sa_.cap2 = cap2;
        sa_.ids = ids; sa_.n_ids = n_ids; sa_.q_xy = q_xy; sa_.area = area; sa_.max_level = max_level;
        const SortedLayout lay = sorted_layout(batch, nq, n2);
        const size_t* o_desc = lay.o_desc; const size_t* o_xy = lay.o_xy;
        const size_t* o_idx = lay.o_idx; const size_t* o_box = lay.o_box;
        sa_.scap[0] = lay.scap[0]; sa_.scap[1] = lay.scap[1];
        if ((rc = grow_scratch(m, m->sorted, lay.total, capturing))) return rc;
        uint8_t* basep = m->sorted.as<uint8_t>();
        WindowMatchArgs w;
        memset(&w, 0, sizeof w);
        for (int s = 0; s < 2; s++) {
            sa_.sdesc[s] = reinterpret_cast<uint4*>(basep + o_desc[s]);
            sa_.sxy[s] = reinterpret_cast<float2*>(basep + o_xy[s]);
            sa_.sidx[s] = reinterpret_cast<uint32_t*>(basep + o_idx[s]);
            sa_.tile_box[s] = reinterpret_cast<float4*>(basep + o_box[s]);
            w.sdesc[s] = sa_.sdesc[s]; w.sxy[s] = sa_.sxy[s]; w.sidx[s] = sa_.sidx[s]; w.scap[s] = sa_.scap[s];
        }
        sa_.batch_q = chain ? 1 : batch;
        sa_.box_rows = sorted_mfma ? 32 : 64;
        sa_.copy_desc = sorted_mfma ? 0 : 1;
        { ProfScope ps(kStageSpatialSort, st); if ((rc = launch_spatial_sort(sa_, batch, nq, n2, st))) return rc; }
        w.tile_box = sa_.tile_box[1];
        w.d_n1 = d_n1; w.n1 = n1; w.d_n2 = d_n2; w.n2 = n2; w.has_ids = ids != nullptr; w.n_ids = n_ids;
        w.cap1 = cap1; w.area = area; w.max_level = max_level; w.batch = batch;
        w.best = best_p;
        w.part = second ? part_p : nullptr;
        w.part_pitch = rows;
        w.splits = plan.splits;
        w.chain = chain ? 1 : 0;
        w.qdesc = d1; w.tdesc = d2; w.ids = ids; w.cap2 = cap2;
        ProfScope ps(kStageMatchWindow, st);
        if ((rc = sorted_mfma ? launch_match_window_mfma(w, batch, nq, st) : launch_match_window(w, batch, nq, st))) return rc;
    } else {
        a.qdesc = d1; a.qkp = k1; a.d_n1 = d_n1; a.n1 = n1;
        a.tdesc = d2; a.tkp = k2; a.d_n2 = d_n2; a.n2 = n2;
        a.ids = ids; a.n_ids = n_ids; a.q_xy = q_xy; a.cap1 = cap1; a.cap2 = cap2;
        a.area = area; a.max_level = max_level; a.dmax = dmax;
        a.best = best_p;
        a.part = second ? part_p : nullptr;
        a.part_pitch = rows;
        a.chain = chain ? 1 : 0;
        ProfScope ps(kStageMatch, st);
        rc = use_mfma ? launch_match_mfma(a, plan, batch, st) : launch_match(a, plan, batch, st);
        if (rc) return rc;
    }
    FinalizeArgs f;
    memset(&f, 0, sizeof f);
    f.best = best_p; f.d_n1 = d_n1; f.n1 = n1; f.ids = ids; f.n_ids = n_ids;
    f.cap1 = cap1; f.dmax = dmax; f.out = d_out; f.n_out = d_n_out;
    f.part = second ? part_p : nullptr; f.part_pitch = rows; f.splits = plan.splits;
    f.ratio = m->ratio; f.histo_length = m->histo_length; f.qkp = k1; f.tkp = k2; f.cap2 = cap2;
    f.chain = chain ? 1 : 0; f.d_n2 = d_n2;
    f.reset = self_clean ? best_p : nullptr;
    f.ticket = no_alloc ? reinterpret_cast<unsigned*>(m->static_scratch + kStaticBestBytes + kStaticPartBytes) : m->ticket.as<unsigned>();
    f.nq_bound = nq;
    if (capturing) m->baked = true;
    ProfScope ps(kStageFinalize, st);
    rc = launch_finalize(f, batch, st);
    if (rc == SLAM_AMD_OK && restore.armed) { restore.armed = false; m->best_clean = true; }
    return rc;
}

extern "C" int slam_amd_frame_match_async(slam_amd_matcher* m,
                                          const uint8_t* d_desc1, const slam_amd_keypoint* d_kp1, int n1,
                                          const int32_t* d_n1,
                                          const uint8_t* d_desc2, const slam_amd_keypoint* d_kp2, int n2,
                                          const int32_t* d_n2,
                                          const int32_t* d_ids, int n_ids, const float* d_q_xy,
                                          float max_dist, float area, int max_level,
                                          slam_amd_dmatch* d_out, int32_t* d_n_out, void* stream) {
    const int cap1 = std::max(d_ids ? std::max(n_ids, n1) : n1, 1);
    return match_common(m, 1, d_desc1, d_kp1, n1, d_n1, cap1, d_desc2, d_kp2, n2, d_n2, std::max(n2, 1),
                        d_ids, n_ids, d_q_xy, max_dist, area, max_level, d_out, d_n_out,
                        reinterpret_cast<hipStream_t>(stream));
}

extern "C" int slam_amd_frame_match_noalloc_async(slam_amd_matcher* m,
                                                  const uint8_t* d_desc1, const slam_amd_keypoint* d_kp1, int n1,
                                                  const int32_t* d_n1,
                                                  const uint8_t* d_desc2, const slam_amd_keypoint* d_kp2, int n2,
                                                  const int32_t* d_n2,
                                                  const int32_t* d_ids, int n_ids, const float* d_q_xy,
                                                  float max_dist, float area, int max_level,
                                                  slam_amd_dmatch* d_out, int32_t* d_n_out, void* stream) {
    const int cap1 = std::max(d_ids ? std::max(n_ids, n1) : n1, 1);
    return match_common(m, 1, d_desc1, d_kp1, n1, d_n1, cap1, d_desc2, d_kp2, n2, d_n2, std::max(n2, 1),
                        d_ids, n_ids, d_q_xy, max_dist, area, max_level, d_out, d_n_out,
                        reinterpret_cast<hipStream_t>(stream), true);
}

extern "C" int slam_amd_frame_match_batch_async(slam_amd_matcher* m, int batch,
                                                const uint8_t* d_desc1, const slam_amd_keypoint* d_kp1,
                                                int cap1, const int32_t* d_n1,
                                                const uint8_t* d_desc2, const slam_amd_keypoint* d_kp2,
                                                int cap2, const int32_t* d_n2,
                                                float max_dist, float area, int max_level,
                                                slam_amd_dmatch* d_out, int32_t* d_n_out, void* stream) {
    if (cap1 <= 0 || cap2 <= 0) SA_FAIL(SLAM_AMD_ERR_ARG, "capacities must be positive");
    if (!d_n1 || !d_n2) SA_FAIL(SLAM_AMD_ERR_ARG, "batched matching needs device counts");
    return match_common(m, batch, d_desc1, d_kp1, cap1, d_n1, cap1, d_desc2, d_kp2, cap2, d_n2, cap2,
                        nullptr, 0, nullptr, max_dist, area, max_level, d_out, d_n_out,
                        reinterpret_cast<hipStream_t>(stream));
}

extern "C" int slam_amd_frame_match_chain_async(slam_amd_matcher* m, int batch,
                                                const uint8_t* d_prev_desc, const slam_amd_keypoint* d_prev_kp,
                                                const int32_t* d_prev_n,
                                                const uint8_t* d_desc, const slam_amd_keypoint* d_kp,
                                                int cap, const int32_t* d_n,
                                                float max_dist, float area, int max_level,
                                                slam_amd_dmatch* d_out, int32_t* d_n_out, void* stream) {
    if (cap <= 0) SA_FAIL(SLAM_AMD_ERR_ARG, "capacity must be positive");
    if (!d_prev_n || !d_n) SA_FAIL(SLAM_AMD_ERR_ARG, "chained matching needs device counts");
    return match_common(m, batch, d_prev_desc, d_prev_kp, cap, d_prev_n, cap, d_desc, d_kp, cap, d_n, cap,
                        nullptr, 0, nullptr, max_dist, area, max_level, d_out, d_n_out,
                        reinterpret_cast<hipStream_t>(stream), false, true);
}

static int upload(DevBuf& b, const void* src, size_t bytes, hipStream_t st) {
    int rc = b.reserve(std::max(bytes, (size_t)16));
    if (rc) return rc;
    if (bytes) SA_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_frame_match(slam_amd_matcher* m,
                                    const uint8_t* desc1, const slam_amd_keypoint* kp1, int n1,
                                    const uint8_t* desc2, const slam_amd_keypoint* kp2, int n2,
                                    const int32_t* ids, int n_ids, const float* q_xy,
                                    float max_dist, float area, int max_level,
                                    slam_amd_dmatch* out, int* n_out) {
    if (!m || !n_out) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    *n_out = 0;
    if (n1 < 0 || n2 < 0 || n_ids < 0) SA_FAIL(SLAM_AMD_ERR_ARG, "negative size");
    if (ids && n_ids == 0) ids = nullptr;
    const int nq = ids ? n_ids : n1;
    if (nq == 0 || n2 == 0) return SLAM_AMD_OK;  // radiusMatch on an empty set yields no matches
    if (!out) SA_FAIL(SLAM_AMD_ERR_ARG, "out is null");
    if (!desc1 || !desc2) SA_FAIL(SLAM_AMD_ERR_ARG, "null descriptor pointer");
    if (ids)
        for (int i = 0; i < n_ids; i++)
            if (ids[i] < 0 || ids[i] >= n1) SA_FAIL(SLAM_AMD_ERR_ARG, "id out of range");
    SA_HIP(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    int rc;
    if ((rc = upload(m->h_d1, desc1, (size_t)n1 * 32, st))) return rc;
    if ((rc = upload(m->h_d2, desc2, (size_t)n2 * 32, st))) return rc;
    if (kp1 && (rc = upload(m->h_k1, kp1, (size_t)n1 * sizeof(slam_amd_keypoint), st))) return rc;
    if (kp2 && (rc = upload(m->h_k2, kp2, (size_t)n2 * sizeof(slam_amd_keypoint), st))) return rc;
    if (ids && (rc = upload(m->h_ids, ids, (size_t)n_ids * sizeof(int32_t), st))) return rc;
    if (q_xy && (rc = upload(m->h_xy, q_xy, (size_t)nq * 2 * sizeof(float), st))) return rc;
    if ((rc = m->h_out.reserve((size_t)std::max(nq, n1) * sizeof(slam_amd_dmatch)))) return rc;
    if ((rc = m->h_n.reserve(sizeof(int32_t)))) return rc;
    rc = slam_amd_frame_match_async(m, m->h_d1.as<uint8_t>(), kp1 ? m->h_k1.as<slam_amd_keypoint>() : nullptr, n1,
                                    nullptr, m->h_d2.as<uint8_t>(), kp2 ? m->h_k2.as<slam_amd_keypoint>() : nullptr,
                                    n2, nullptr, ids ? m->h_ids.as<int32_t>() : nullptr, n_ids,
                                    q_xy ? m->h_xy.as<float>() : nullptr, max_dist, area, max_level,
                                    m->h_out.as<slam_amd_dmatch>(), m->h_n.as<int32_t>(), st);
    if (rc) return rc;
    int32_t n = 0;
    SA_HIP(hipMemcpyAsync(&n, m->h_n.p, sizeof n, hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    if (n > 0) {
        SA_HIP(hipMemcpyAsync(out, m->h_out.p, (size_t)n * sizeof(slam_amd_dmatch), hipMemcpyDeviceToHost, st));
        SA_HIP(hipStreamSynchronize(st));
    }
    *n_out = n;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_match_pairs_async(slam_amd_matcher* m,
                                          const uint8_t* d_desc, const slam_amd_keypoint* d_kp,
                                          int n_keyframes, int n_per,
                                          const int32_t* d_pairs, int n_pairs,
                                          float max_dist, float area, int max_level,
                                          uint32_t* d_out, void* stream) {
    if (!m || !d_desc || !d_pairs || !d_out) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    if (n_keyframes <= 0 || n_per <= 0 || n_pairs < 0) SA_FAIL(SLAM_AMD_ERR_ARG, "bad size");
    if (n_per > 65535) SA_FAIL(SLAM_AMD_ERR_ARG, "n_per exceeds 65535 (16-bit train index in the result)");
    if (area != area) area = -2.f;      // as in match_common
    const bool gated = (area != -1.f) || (max_level != -1);
    if (gated && !d_kp) SA_FAIL(SLAM_AMD_ERR_ARG, "keypoints are required when a gate is on");
    if (reinterpret_cast<uintptr_t>(d_desc) & 15) SA_FAIL(SLAM_AMD_ERR_ARG, "d_desc must be 16-byte aligned");
    if (m->ratio > 0.f || m->histo_length > 0)
        SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "match_pairs does not apply the ratio / rotation options; clear them first");
    SA_HIP(hipSetDevice(m->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // With the window gate on, the keyframes are first reordered by grid cell (index words and positions in the
    // matcher's scratch) and every pair goes through the kernel that only looks at the train blocks near a wave's
    // queries; same result.
    // (every keyframe is reordered, whether a pair names it or not: not worth it for a few pairs of many keyframes)
    if (m->sweep_kernel == 0 && m->window_mode != 1 && area > 0.f && area <= 3.0e38f && n_per <= window_mfma_max_train() &&
        n_pairs > 0 && (m->window_mode == 3 || (long long)n_pairs * 8 >= n_keyframes)) {
        const SortedLayout lay = sorted_layout(n_keyframes, 0, n_per);
        int rc;
        if ((rc = grow_scratch(m, m->sorted, lay.total, stream_capturing(st)))) return rc;
        uint8_t* basep = m->sorted.as<uint8_t>();
        SpatialSortArgs sa_;
        memset(&sa_, 0, sizeof sa_);
        sa_.tdesc = d_desc; sa_.tkp = d_kp; sa_.n2 = n_per; sa_.cap2 = n_per;
        sa_.area = area; sa_.max_level = max_level;
        sa_.scap[0] = lay.scap[0]; sa_.scap[1] = lay.scap[1];
        sa_.box_rows = 32; sa_.copy_desc = 0;
        WindowMatchArgs w;
        memset(&w, 0, sizeof w);
        for (int s = 0; s < 2; s++) {
            sa_.sdesc[s] = reinterpret_cast<uint4*>(basep + lay.o_desc[s]);
            sa_.sxy[s] = reinterpret_cast<float2*>(basep + lay.o_xy[s]);
            sa_.sidx[s] = reinterpret_cast<uint32_t*>(basep + lay.o_idx[s]);
            sa_.tile_box[s] = reinterpret_cast<float4*>(basep + lay.o_box[s]);
            w.sdesc[s] = sa_.sdesc[s]; w.sxy[s] = sa_.sxy[s]; w.sidx[s] = sa_.sidx[s]; w.scap[s] = sa_.scap[s];
        }
        { ProfScope ps(kStageSpatialSort, st); if ((rc = launch_spatial_sort(sa_, n_keyframes, 0, n_per, st))) return rc; }
        w.tile_box = sa_.tile_box[1];
        w.n1 = n_per; w.n2 = n_per; w.cap1 = n_per; w.cap2 = n_per;
        w.area = area; w.max_level = max_level; w.batch = n_pairs; w.splits = 1;
        w.qdesc = d_desc; w.tdesc = d_desc;
        w.pairs = d_pairs; w.pair_out = d_out; w.dmax = dmax_of(max_dist);
        ProfScope ps(kStageMatchPairs, st);
        return launch_match_window_mfma(w, n_pairs, n_per, st);
    }
    MatchArgs a;
    memset(&a, 0, sizeof a);
    a.qdesc = d_desc; a.qkp = d_kp; a.tdesc = d_desc; a.tkp = d_kp;
    a.area = area; a.max_level = max_level; a.dmax = dmax_of(max_dist);
    a.pairs = d_pairs; a.n_per = n_per; a.pair_out = d_out;
    ProfScope ps(kStageMatchPairs, reinterpret_cast<hipStream_t>(stream));
    if (m->sweep_kernel == 0) return launch_match_pairs_mfma(a, n_pairs, reinterpret_cast<hipStream_t>(stream));
    return launch_match_pairs(a, n_pairs, reinterpret_cast<hipStream_t>(stream));
}

// =============================================================================
// Projection, duplicates
// =============================================================================
extern "C" int slam_amd_project_points_async(slam_amd_matcher* m, const double* Rt, const double* K4,
                                             const double* dist5, const float* d_xyz, int n, float* d_uv,
                                             void* stream) {
    if (!m || !Rt || !K4) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    if (n < 0) SA_FAIL(SLAM_AMD_ERR_ARG, "negative size");
    if (n == 0) return SLAM_AMD_OK;
    if (!d_xyz || !d_uv) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    SA_HIP(hipSetDevice(m->device));
    ProjectArgs a;
    memcpy(a.Rt, Rt, sizeof a.Rt);
    memcpy(a.K4, K4, sizeof a.K4);
    for (int i = 0; i < 5; i++) a.dist[i] = dist5 ? dist5[i] : 0.0;
    a.xyz = d_xyz; a.uv = d_uv; a.n = n;
    return launch_project(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int slam_amd_project_points(slam_amd_matcher* m, const double* Rt, const double* K4,
                                       const double* dist5, const float* xyz, int n, float* uv) {
    if (!m) SA_FAIL(SLAM_AMD_ERR_ARG, "matcher is null");
    if (n < 0) SA_FAIL(SLAM_AMD_ERR_ARG, "negative size");
    if (n == 0) return SLAM_AMD_OK;
    if (!xyz || !uv) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    SA_HIP(hipSetDevice(m->device));
    int rc;
    if ((rc = upload(m->h_xy, xyz, (size_t)n * 3 * sizeof(float), m->stream))) return rc;
    if ((rc = m->h_aux.reserve((size_t)n * 2 * sizeof(float)))) return rc;
    rc = slam_amd_project_points_async(m, Rt, K4, dist5, m->h_xy.as<float>(), n, m->h_aux.as<float>(), m->stream);
    if (rc) return rc;
    SA_HIP(hipMemcpyAsync(uv, m->h_aux.p, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    SA_HIP(hipStreamSynchronize(m->stream));
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_duplicate_pairs(slam_amd_matcher* m,
                                        const uint8_t* desc1, const float* pos1, int n1,
                                        const uint8_t* desc2, const float* pos2, int n2,
                                        int desc_dist, double point_dist, uint8_t* flag) {
    if (!m) SA_FAIL(SLAM_AMD_ERR_ARG, "matcher is null");
    if (n1 < 0 || n2 < 0) SA_FAIL(SLAM_AMD_ERR_ARG, "negative size");
    if (n1 == 0 || n2 == 0) return SLAM_AMD_OK;
    if (!desc1 || !desc2 || !pos1 || !pos2 || !flag) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    SA_HIP(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    int rc;
    if ((rc = upload(m->h_d1, desc1, (size_t)n1 * 32, st))) return rc;
    if ((rc = upload(m->h_d2, desc2, (size_t)n2 * 32, st))) return rc;
    if ((rc = upload(m->h_k1, pos1, (size_t)n1 * 3 * sizeof(float), st))) return rc;
    if ((rc = upload(m->h_k2, pos2, (size_t)n2 * 3 * sizeof(float), st))) return rc;
    if ((rc = m->h_aux.reserve((size_t)n1 * n2))) return rc;
    DupArgs a;
    a.d1 = m->h_d1.as<uint8_t>(); a.p1 = m->h_k1.as<float>(); a.n1 = n1;
    a.d2 = m->h_d2.as<uint8_t>(); a.p2 = m->h_k2.as<float>(); a.n2 = n2;
    a.desc_dist = desc_dist; a.point_dist = point_dist; a.flag = m->h_aux.as<uint8_t>();
    if ((rc = launch_duplicates(a, st))) return rc;
    SA_HIP(hipMemcpyAsync(flag, m->h_aux.p, (size_t)n1 * n2, hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_outlier_flags(slam_amd_matcher* m, const float* xyz, const slam_amd_dmatch* matches, int n,
                                      const slam_amd_keypoint* kp_q, int nq, const slam_amd_keypoint* kp_t, int nt,
                                      const float* center_q, const float* center_t,
                                      const double* Rt_q, const double* K4_q, const double* dist5_q,
                                      const double* Rt_t, const double* K4_t, const double* dist5_t,
                                      uint8_t* flags) {
    if (!m) SA_FAIL(SLAM_AMD_ERR_ARG, "matcher is null");
    if (n < 0 || nq < 0 || nt < 0) SA_FAIL(SLAM_AMD_ERR_ARG, "negative size");
    if (n == 0) return SLAM_AMD_OK;
    if (!xyz || !matches || !kp_q || !kp_t || !center_q || !center_t || !Rt_q || !K4_q || !Rt_t || !K4_t || !flags)
        SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    for (int i = 0; i < n; i++)
        if (matches[i].queryIdx < 0 || matches[i].queryIdx >= nq || matches[i].trainIdx < 0 || matches[i].trainIdx >= nt)
            SA_FAIL(SLAM_AMD_ERR_ARG, "match index out of range");
    SA_HIP(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    int rc;
    if ((rc = upload(m->h_xy, xyz, (size_t)n * 3 * sizeof(float), st))) return rc;
    if ((rc = upload(m->h_out, matches, (size_t)n * sizeof(slam_amd_dmatch), st))) return rc;
    if ((rc = upload(m->h_k1, kp_q, (size_t)nq * sizeof(slam_amd_keypoint), st))) return rc;
    if ((rc = upload(m->h_k2, kp_t, (size_t)nt * sizeof(slam_amd_keypoint), st))) return rc;
    if ((rc = m->h_aux.reserve((size_t)n))) return rc;
    OutlierArgs a;
    memset(&a, 0, sizeof a);
    a.xyz = m->h_xy.as<float>(); a.matches = m->h_out.as<slam_amd_dmatch>(); a.n = n;
    a.kp_q = m->h_k1.as<slam_amd_keypoint>(); a.nq = nq;
    a.kp_t = m->h_k2.as<slam_amd_keypoint>(); a.nt = nt;
    for (int k = 0; k < 3; k++) { a.cq[k] = center_q[k]; a.ct[k] = center_t[k]; }
    memcpy(a.cam_q.Rt, Rt_q, sizeof a.cam_q.Rt); memcpy(a.cam_q.K4, K4_q, sizeof a.cam_q.K4);
    memcpy(a.cam_t.Rt, Rt_t, sizeof a.cam_t.Rt); memcpy(a.cam_t.K4, K4_t, sizeof a.cam_t.K4);
    if (dist5_q) memcpy(a.cam_q.dist, dist5_q, sizeof a.cam_q.dist);
    if (dist5_t) memcpy(a.cam_t.dist, dist5_t, sizeof a.cam_t.dist);
    a.flags = m->h_aux.as<uint8_t>();
    if ((rc = launch_outliers(a, st))) return rc;
    SA_HIP(hipMemcpyAsync(flags, m->h_aux.p, (size_t)n, hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    return SLAM_AMD_OK;
}

// =============================================================================
// Input front end (BGR -> grey -> bilinear resize to the working width)
// =============================================================================
struct slam_amd_frontend {
    int device = 0;
    int sw = 0, sh = 0, channels = 3, dw = 0, dh = 0;
    hipStream_t stream = nullptr;
    DevBuf tables;       // xofs, yofs (int32), xtap, ytap (int16 pairs)
    size_t o_xofs = 0, o_yofs = 0, o_xtap = 0, o_ytap = 0;
    DevBuf h_src, h_dst; // staging of the host call
};

// Source offset and 11-bit taps of one axis, as cv::resize(INTER_LINEAR) builds them
// for 8-bit images: f = (float)((d + 0.5) / inv_scale - 0.5), taps = round((1 - f, f) * 2048).
static void linear_taps_axis(int src, int dst, double inv_scale, bool clamp_frac, int32_t* ofs, int16_t* taps) {
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s0 = (int)floorf(f);
        f -= (float)s0;
        if (clamp_frac) {
            if (s0 < 0) { f = 0.f; s0 = 0; }
            if (s0 >= src - 1) { f = 0.f; s0 = src - 1; }
        }
        ofs[d] = s0;
        taps[2 * d] = (int16_t)lrintf((1.f - f) * 2048.f);
        taps[2 * d + 1] = (int16_t)lrintf(f * 2048.f);
    }
}

extern "C" int slam_amd_frontend_create(int src_width, int src_height, int channels, int target_width,
                                        int device, slam_amd_frontend** out) {
    if (!out) SA_FAIL(SLAM_AMD_ERR_ARG, "out is null");
    *out = nullptr;
    if (src_width <= 0 || src_height <= 0 || target_width <= 0) SA_FAIL(SLAM_AMD_ERR_ARG, "sizes must be positive");
    if (src_width > 32768 || src_height > 32768 || target_width > 32768) SA_FAIL(SLAM_AMD_ERR_ARG, "size above 32768");
    if (channels != 1 && channels != 3) SA_FAIL(SLAM_AMD_ERR_UNSUPPORTED, "1 (grey) or 3 (BGR) channels");
    int rc = use_device(device);
    if (rc) return rc;
    // scale as main.cpp:125-128 computes it, destination size as cv::resize does
    const float scale = (float)target_width / (float)src_width;
    const int dw = (int)lrint((double)src_width * (double)scale);
    const int dh = (int)lrint((double)src_height * (double)scale);
    if (dw <= 0 || dh <= 0) SA_FAIL(SLAM_AMD_ERR_ARG, "the scaled image is empty");
    slam_amd_frontend* fe = new (std::nothrow) slam_amd_frontend();
    if (!fe) SA_FAIL(SLAM_AMD_ERR_NOMEM, "out of host memory");
    fe->device = device; fe->sw = src_width; fe->sh = src_height; fe->channels = channels; fe->dw = dw; fe->dh = dh;
    auto fail = [&](int code) { slam_amd_frontend_destroy(fe); return code; };
    std::vector<int32_t> xofs((size_t)dw), yofs((size_t)dh);
    std::vector<int16_t> xtap((size_t)dw * 2), ytap((size_t)dh * 2);
    linear_taps_axis(src_width, dw, (double)scale, true, xofs.data(), xtap.data());
    linear_taps_axis(src_height, dh, (double)scale, false, yofs.data(), ytap.data());
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    fe->o_xofs = 0;
    fe->o_yofs = up(fe->o_xofs + xofs.size() * 4);
    fe->o_xtap = up(fe->o_yofs + yofs.size() * 4);
    fe->o_ytap = up(fe->o_xtap + xtap.size() * 2);
    const size_t total = up(fe->o_ytap + ytap.size() * 2);
    if ((rc = fe->tables.reserve(total))) return fail(rc);
    uint8_t* t = fe->tables.as<uint8_t>();
    if (hipMemcpy(t + fe->o_xofs, xofs.data(), xofs.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t + fe->o_yofs, yofs.data(), yofs.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t + fe->o_xtap, xtap.data(), xtap.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t + fe->o_ytap, ytap.data(), ytap.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("hipMemcpy of the resize tables failed");
        return fail(SLAM_AMD_ERR_HIP);
    }
    hipError_t e = hipStreamCreateWithFlags(&fe->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { slam_amd_frontend_destroy(fe); return hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__); }
    *out = fe;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_frontend_destroy(slam_amd_frontend* fe) {
    if (!fe) return SLAM_AMD_OK;
    (void)hipSetDevice(fe->device);
    if (fe->stream) { (void)hipStreamSynchronize(fe->stream); (void)hipStreamDestroy(fe->stream); }
    fe->tables.release(); fe->h_src.release(); fe->h_dst.release();
    delete fe;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_frontend_output_size(const slam_amd_frontend* fe, int* width, int* height) {
    if (!fe || !width || !height) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    *width = fe->dw; *height = fe->dh;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_frontend_run_async(slam_amd_frontend* fe, const uint8_t* d_src, int batch,
                                           size_t src_frame_pitch, int src_row_stride,
                                           uint8_t* d_gray, size_t dst_frame_pitch, int dst_row_stride,
                                           void* stream) {
    if (!fe || !d_src || !d_gray) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    if (batch <= 0) SA_FAIL(SLAM_AMD_ERR_ARG, "batch must be positive");
    if (src_row_stride < fe->sw * fe->channels) SA_FAIL(SLAM_AMD_ERR_ARG, "source row stride smaller than a row");
    if (dst_row_stride < fe->dw) SA_FAIL(SLAM_AMD_ERR_ARG, "destination row stride smaller than a row");
    if (batch > 1 && (src_frame_pitch < (size_t)src_row_stride * (fe->sh - 1) + (size_t)fe->sw * fe->channels ||
                      dst_frame_pitch < (size_t)dst_row_stride * (fe->dh - 1) + fe->dw))
        SA_FAIL(SLAM_AMD_ERR_ARG, "frame pitch smaller than one frame");
    SA_HIP(hipSetDevice(fe->device));
    FrontendArgs a;
    memset(&a, 0, sizeof a);
    a.src = d_src; a.src_frame_pitch = src_frame_pitch; a.src_pitch = src_row_stride;
    a.sw = fe->sw; a.sh = fe->sh; a.channels = fe->channels;
    a.dst = d_gray; a.dst_frame_pitch = dst_frame_pitch; a.dst_pitch = dst_row_stride; a.dw = fe->dw; a.dh = fe->dh;
    const uint8_t* t = fe->tables.as<uint8_t>();
    a.xofs = reinterpret_cast<const int32_t*>(t + fe->o_xofs); a.yofs = reinterpret_cast<const int32_t*>(t + fe->o_yofs);
    a.xtap = reinterpret_cast<const int16_t*>(t + fe->o_xtap); a.ytap = reinterpret_cast<const int16_t*>(t + fe->o_ytap);
    return launch_frontend(a, batch, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int slam_amd_frontend_run(slam_amd_frontend* fe, const uint8_t* src, int src_row_stride,
                                     uint8_t* gray_out, int dst_row_stride) {
    if (!fe || !src || !gray_out) SA_FAIL(SLAM_AMD_ERR_ARG, "null argument");
    if (src_row_stride < fe->sw * fe->channels) SA_FAIL(SLAM_AMD_ERR_ARG, "source row stride smaller than a row");
    if (dst_row_stride < fe->dw) SA_FAIL(SLAM_AMD_ERR_ARG, "destination row stride smaller than a row");
    SA_HIP(hipSetDevice(fe->device));
    const size_t row = (size_t)fe->sw * fe->channels;
    int rc;
    if ((rc = fe->h_src.reserve(row * fe->sh))) return rc;
    if ((rc = fe->h_dst.reserve((size_t)fe->dw * fe->dh))) return rc;
    SA_HIP(hipMemcpy2DAsync(fe->h_src.p, row, src, src_row_stride, row, fe->sh, hipMemcpyHostToDevice, fe->stream));
    rc = slam_amd_frontend_run_async(fe, fe->h_src.as<uint8_t>(), 1, row * fe->sh, (int)row, fe->h_dst.as<uint8_t>(),
                                     (size_t)fe->dw * fe->dh, fe->dw, fe->stream);
    if (rc) return rc;
    SA_HIP(hipMemcpy2DAsync(gray_out, dst_row_stride, fe->h_dst.p, fe->dw, fe->dw, fe->dh, hipMemcpyDeviceToHost, fe->stream));
    SA_HIP(hipStreamSynchronize(fe->stream));
    return SLAM_AMD_OK;
}

// =============================================================================
// General
// =============================================================================
extern "C" int slam_amd_prof_enable(int max_marks) {
    if (max_marks < 0) SA_FAIL(SLAM_AMD_ERR_ARG, "max_marks < 0");
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear(); g_prof.stage.clear(); g_prof.used = 0; g_prof.on = false;
    if (max_marks == 0) return SLAM_AMD_OK;
    g_prof.ev.resize(2 * (size_t)max_marks);
    for (auto& e : g_prof.ev) SA_HIP(hipEventCreate(&e));
    g_prof.stage.assign((size_t)max_marks, -1);
    g_prof.on = true;
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_prof_collect(double* total_ms, int* calls, int n_stages) {
    if (!total_ms || !calls || n_stages < kStageCount) SA_FAIL(SLAM_AMD_ERR_ARG, "bad argument");
    for (int i = 0; i < n_stages; i++) { total_ms[i] = 0; calls[i] = 0; }
    for (size_t i = 0; i < g_prof.used; i++) {
        float ms = 0;
        SA_HIP(hipEventSynchronize(g_prof.ev[2 * i + 1]));
        SA_HIP(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        total_ms[g_prof.stage[i]] += ms;
        calls[g_prof.stage[i]]++;
    }
    g_prof.used = 0;
    return SLAM_AMD_OK;
}

extern "C" const char* slam_amd_prof_stage_name(int stage) {
    static const char* names[kStageCount] = {"pyramid (all levels)", "fast_nms", "select_compact",
        "spatial_sort", "match_window", "describe", "match", "finalize_matches", "match_pairs", "memset"};
    return (stage >= 0 && stage < kStageCount) ? names[stage] : "";
}

extern "C" int slam_amd_version(void) { return SLAM_AMD_VERSION; }

extern "C" const char* slam_amd_status_string(int status) {
    switch (status) {
        case SLAM_AMD_OK: return "ok";
        case SLAM_AMD_ERR_ARG: return "invalid argument";
        case SLAM_AMD_ERR_UNSUPPORTED: return "unsupported ORB variant";
        case SLAM_AMD_ERR_CAPACITY: return "output capacity exceeded";
        case SLAM_AMD_ERR_NOMEM: return "out of memory";
        case SLAM_AMD_ERR_HIP: return "HIP runtime error";
        case SLAM_AMD_ERR_NO_DEVICE: return "no GPU device";
        default: return "unknown status";
    }
}

extern "C" const char* slam_amd_last_error(void) { return g_last_error.c_str(); }

extern "C" int slam_amd_device_count(int* count) {
    if (!count) SA_FAIL(SLAM_AMD_ERR_ARG, "count is null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return SLAM_AMD_OK;
}

extern "C" void slam_amd_default_orb_params(slam_amd_orb_params* p) {
    if (!p) return;
    p->nfeatures = 1000; p->scale_factor = 1.2f; p->nlevels = 8; p->edge_threshold = 31;
    p->first_level = 0; p->wta_k = 2; p->score_type = 1; p->patch_size = 31; p->fast_threshold = 20;
}

extern "C" void slam_amd_sincos_f32(float angle_rad, float* cos_out, float* sin_out) {
    float c, s;
    sincos_f32(angle_rad, &c, &s);
    if (cos_out) *cos_out = c;
    if (sin_out) *sin_out = s;
}

extern "C" float slam_amd_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }

extern "C" int slam_amd_linear_taps(int src_size, int dst_size, double inv_scale, int clamp_frac,
                                    int32_t* offsets, int16_t* taps) {
    if (src_size <= 0 || dst_size <= 0 || !(inv_scale > 0) || !offsets || !taps) SA_FAIL(SLAM_AMD_ERR_ARG, "bad argument");
    linear_taps_axis(src_size, dst_size, inv_scale, clamp_frac != 0, offsets, taps);
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_debug_pyramid_plan(const slam_amd_orb_params* params, int width, int height, int group,
                                           int* n_groups, int* info6, int16_t* spanx, int16_t* spany, int span_capacity) {
    int rc = check_orb_params(params);
    if (rc) return rc;
    if (width <= 0 || height <= 0 || width > 16383 || height > 16383) SA_FAIL(SLAM_AMD_ERR_ARG, "image size out of range");
    PyrGeom g;
    if ((rc = build_geometry_of(*params, width, height, g))) return rc;
    HostPyramidPlan plan;
    if ((rc = make_pyramid_plan(g, plan))) return rc;
    if (n_groups) *n_groups = (int)plan.groups.size();
    if (group < 0 || group >= (int)plan.groups.size()) return SLAM_AMD_OK;
    const HostGroup& hg = plan.groups[(size_t)group];
    if (info6) { info6[0] = hg.first; info6[1] = hg.last; info6[2] = hg.T; info6[3] = hg.ntx; info6[4] = hg.nty; info6[5] = (int)hg.lds; }
    const size_t nx = hg.ax.span.size() * 4, ny = hg.ay.span.size() * 4;
    if ((spanx && nx > (size_t)span_capacity) || (spany && ny > (size_t)span_capacity)) SA_FAIL(SLAM_AMD_ERR_CAPACITY, "span buffer too small");
    if (spanx) memcpy(spanx, hg.ax.span.data(), nx * sizeof(int16_t));
    if (spany) memcpy(spany, hg.ay.span.data(), ny * sizeof(int16_t));
    return SLAM_AMD_OK;
}

extern "C" int slam_amd_linear_exact_coeffs(int src_size, int dst_size, int32_t* offsets, uint16_t* c1) {
    if (src_size <= 0 || dst_size <= 0 || !offsets || !c1) SA_FAIL(SLAM_AMD_ERR_ARG, "bad argument");
    linear_exact_axis(src_size, dst_size, false, offsets, c1);
    return SLAM_AMD_OK;
}
