// Host engine + extern "C" ABI (include/vors_hip.h) of libvors_hip.so.
//
// Owns device workspaces and launches the kernels of kernels.hip. No CPU compute path exists here: every compute
// entry point needs a HIP device and fails loudly (VORS_ERR_NO_DEVICE) without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

using namespace vors;

// ---------------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static vors_status fail(vors_status st, const std::string& msg) {
    g_last_error = msg;
    return st;
}
vors_status vors_set_last_error(vors_status st, const std::string& msg) { return fail(st, msg); }  // for multi.cpp
#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess)                                                                                  \
            return fail(VORS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                      \
    } while (0)

static vors_status require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(VORS_ERR_NO_DEVICE, "vors_hip: no HIP device available (this library has no CPU fallback)");
    }
    return VORS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------------------------------
// FUSED arithmetic: levels of at most this many points are evaluated in the EXACT arithmetic (engine.h Geom::fused_exact_points).
#ifndef VORS_FUSED_EXACT_POINTS_DEFAULT
#define VORS_FUSED_EXACT_POINTS_DEFAULT 2500
#endif
static vors_status build_geom(const vors_config* cfg, int rows, int cols, Geom* g) {
    if (!cfg) return fail(VORS_ERR_INVALID_ARGUMENT, "cfg is NULL");
    if (cfg->nb_levels < 1 || cfg->nb_levels > VORS_MAX_LEVELS)
        return fail(VORS_ERR_INVALID_ARGUMENT, "nb_levels must be in [1, " + std::to_string(VORS_MAX_LEVELS) + "]");
    if (rows < 2 || cols < 2 || rows > 65535 || cols > 65535) return fail(VORS_ERR_INVALID_ARGUMENT, "rows/cols must be in [2, 65535]");
    if ((long long)rows * cols > (1ll << 28))  // the kernels address a level with 32-bit byte offsets
        return fail(VORS_ERR_INVALID_ARGUMENT, "rows * cols must not exceed 2^28 pixels");
    if (cfg->candidates_mode != VORS_CANDIDATES_COARSE_TO_FINE && cfg->candidates_mode != VORS_CANDIDATES_DENSE &&
        cfg->candidates_mode != VORS_CANDIDATES_DSO)
        return fail(VORS_ERR_INVALID_ARGUMENT, "unknown candidates_mode");
    if (cfg->candidates_diff_threshold < 0 || cfg->candidates_diff_threshold > 65535)
        return fail(VORS_ERR_INVALID_ARGUMENT, "candidates_diff_threshold must fit u16");
    std::memset(g, 0, sizeof(*g));
    g->L = cfg->nb_levels;
    if (cfg->arithmetic != VORS_ARITH_EXACT && cfg->arithmetic != VORS_ARITH_FUSED && cfg->arithmetic != VORS_ARITH_REFERENCE)
        return fail(VORS_ERR_INVALID_ARGUMENT, "unknown arithmetic mode");
    g->mode = cfg->candidates_mode;
    g->arith = cfg->arithmetic;
    g->thresh = cfg->candidates_diff_threshold;
    g->depth_scale = cfg->depth_scale;
    g->idepth_variance = cfg->idepth_variance;
    g->huber_delta = cfg->huber_delta;
    g->fused_exact_points = getenv("VORS_FUSED_EXACT_POINTS") ? atoi(getenv("VORS_FUSED_EXACT_POINTS")) : VORS_FUSED_EXACT_POINTS_DEFAULT;
    g->fused_exact_step = getenv("VORS_FUSED_EXACT_STEP") ? atoi(getenv("VORS_FUSED_EXACT_STEP")) : 0;
    g->ref_inflight_x2 = 2;
    g->ref_rank = (getenv("VORS_REF_RANK") && atoi(getenv("VORS_REF_RANK")) == 0) ? 0 : 1;
    g->fused_small_warp = (getenv("VORS_FUSED_SMALL") && std::string(getenv("VORS_FUSED_SMALL")) == "exact") ? 0 : 1;
    g->S0 = rows * cols;
    int r = rows, c = cols;
    Intr k{cfg->cu, cfg->cv, cfg->fu, cfg->fv, cfg->skew};
    int img_off = 0;
    for (int l = 0; l < g->L; ++l) {
        if (l > 0) {
            r /= 2;  // multires.rs:73-77: halve returns None when a half size is 0
            c /= 2;
            if (r == 0 || c == 0)
                return fail(VORS_ERR_PYRAMID_TOO_SHORT,
                            "image too small for nb_levels (the reference panics: inverse_compositional.rs:124-125,183-189)");
            k = intr_half_res(k);  // camera.rs:106-123
        }
        g->lv[l].rows = r;
        g->lv[l].cols = c;
        g->lv[l].k = k;
        g->lv[l].inv_fu_d = 1.0 / (double)k.fu;
        g->lv[l].inv_fv_d = 1.0 / (double)k.fv;
        g->lv[l].inv_fu = (float)g->lv[l].inv_fu_d;
        g->lv[l].inv_fv = (float)g->lv[l].inv_fv_d;
        g->lv[l].s_fuv = (float)((double)k.skew / ((double)k.fu * (double)k.fv));
        if (l == 0) {
            g->lv[l].img_off = -1;
        } else {
            g->lv[l].img_off = img_off;
            img_off += (r * c + 15) & ~15;
        }
    }
    g->upper_stride = std::max(img_off, 16);
    g->root_rows = g->lv[g->L - 1].rows;
    g->root_cols = g->lv[g->L - 1].cols;
    const long n_roots = (long)g->root_rows * g->root_cols;
    long slot_off = 0;
    const bool dense = g->mode == VORS_CANDIDATES_DENSE;
    const bool generic = g->mode == VORS_CANDIDATES_DSO;
    for (int l = 0; l < g->L; ++l) {
        long n = dense ? (long)g->lv[l].rows * g->lv[l].cols : n_roots * (1L << (g->L - 1 - l));
        // generic-mask modes: compacted candidate lists with a fixed capacity per level (the DSO selector aims at 2000 points
        // and re-runs when it gets more than 4x that; 65536 leaves a wide margin, excess candidates would be dropped)
        if (generic) n = std::min((long)g->lv[l].rows * g->lv[l].cols, 65536L);
        if (slot_off + n > 0x7fffffffL) return fail(VORS_ERR_UNSUPPORTED, "too many candidate slots");
        g->lv[l].n_slots = (int)n;
        if (dense && l == 0) {  // dense level 0 stores nothing per point (recomputed on the fly by the LM kernel)
            g->lv[l].slot_off = -1;
            continue;
        }
        g->lv[l].slot_off = (int)slot_off;
        slot_off += (n + 3) & ~3L;
    }
    g->slots_total = (int)std::max(slot_off, 4L);
    return VORS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// device-resident batch engine
// ---------------------------------------------------------------------------------------------------------------
struct vors_batch {
    vors_config cfg;
    Geom g;
    int max_pairs = 0;
    int device = 0;          // HIP device the workspaces live on (vors_batch_create_on); every entry point switches to it
    int prepared_pairs = 0;  // n_pairs of the last prepare_keyframes: track_current may not ask for more
    int current_pairs = 0;   // n_pairs of the last track_current: the current-frame pyramid slots that hold an image
    uint8_t* kf_upper = nullptr;
    uint8_t* cur_upper = nullptr;
    const uint8_t* kf_level0 = nullptr;   // caller's buffer of the last prepare_keyframes
    const uint8_t* cur_level0 = nullptr;  // caller's buffer of the last track_current
    const uint16_t* kf_depth = nullptr;   // caller's depth buffer of the last prepare_keyframes (read by the dense LM kernel)
    Records rec{};
    LmSplitWs split{};
    uint64_t bytes = 0;
    int lm_block = 256;  // threads per frame pair in the LM kernel (256 / 512 / 1024)
    bool owns_sort_tmp = false;  // rec.sort_tmp is an allocation of its own (DSO mode, REFERENCE arithmetic), not the staging grid
    // generic-mask (DSO) mode workspaces
    DsoWs dso{};
    PixelPlanes pp{};
    uint8_t* mask0 = nullptr;
    // Per-stage HIP-event ring (stage: 0 keyframe pyramid, 1 keyframe precompute, 2 current pyramid, 3 LM kernel).
    // Events are only RECORDED on the caller's stream during a step (non-blocking); elapsed times are read afterwards.
    int ring = 0;
    std::vector<hipEvent_t> ev0[4], ev1[4];
    long count[4] = {0, 0, 0, 0};
};

template <class T>
static hipError_t dmalloc(T** p, size_t n, uint64_t* bytes) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T));
    if (e == hipSuccess) *bytes += n * sizeof(T);
    return e;
}

static void batch_free(vors_batch* b) {
    if (!b) return;
    if (b->kf_upper) (void)hipFree(b->kf_upper);
    if (b->cur_upper) (void)hipFree(b->cur_upper);
    if (b->rec.A) (void)hipFree(b->rec.A);
    if (b->rec.B) (void)hipFree(b->rec.B);
    if (b->rec.C) (void)hipFree(b->rec.C);
    if (b->rec.XY) (void)hipFree(b->rec.XY);
    if (b->rec.IZ) (void)hipFree(b->rec.IZ);
    if (b->rec.V) (void)hipFree(b->rec.V);
    if (b->rec.LUT) (void)hipFree(const_cast<float2*>(b->rec.LUT));
    if (b->owns_sort_tmp && b->rec.sort_tmp) (void)hipFree(b->rec.sort_tmp);
    if (b->split.side_stream) (void)hipStreamDestroy(b->split.side_stream);
    if (b->split.ev_fork) (void)hipEventDestroy(b->split.ev_fork);
    if (b->split.ev_join) (void)hipEventDestroy(b->split.ev_join);
    if (b->split.side_list) (void)hipFree(b->split.side_list);
    if (b->split.join_list) (void)hipFree(b->split.join_list);
    void* handoff[] = {b->rec.handoff.state, b->rec.handoff.list, b->rec.handoff.counters};
    for (void* p : handoff)
        if (p) (void)hipFree(p);
    void* planes[] = {b->rec.dense_t.recs, b->rec.dense_t.n_valid, b->rec.dense_t.cur0, b->rec.dense_t.curu};
    for (void* p : planes)
        if (p) (void)hipFree(p);
    void* extra[] = {b->dso.gmag, b->dso.median, b->dso.thresh, b->dso.max_g, b->dso.max_pos, b->dso.mask1, b->dso.picked, b->dso.state, b->dso.pick_list,
                     b->mask0, b->pp.iz, b->pp.v, b->pp.counts, b->rec.n_used, b->rec.S, b->rec.stage, b->rec.region_cnt,
                     b->split.state, b->split.partials, b->split.list[0], b->split.list[1], b->split.count};
    for (void* p : extra)
        if (p) (void)hipFree(p);
    for (int st = 0; st < 4; ++st) {
        for (auto e : b->ev0[st])
            if (e) (void)hipEventDestroy(e);
        for (auto e : b->ev1[st])
            if (e) (void)hipEventDestroy(e);
    }
    delete b;
}

// Entry points run on the handle's device whatever the caller's current device is, and restore the caller's on exit.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
// The stream work is enqueued on must belong to the handle's device (a stream of another device would silently run nothing useful).
static vors_status check_stream(const vors_batch* b, hipStream_t s);

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T>
    T* as() { return static_cast<T*>(p); }
};

#define STAGE_BEGIN(b, st, s) \
    do {                      \
        if ((b)->ring > 0) HIP_TRY(hipEventRecord((b)->ev0[st][(b)->count[st] % (b)->ring], s)); \
    } while (0)
#define STAGE_END(b, st, s)   \
    do {                      \
        if ((b)->ring > 0) {  \
            HIP_TRY(hipEventRecord((b)->ev1[st][(b)->count[st] % (b)->ring], s)); \
            (b)->count[st] += 1; \
        }                     \
    } while (0)

extern "C" {

const char* vors_last_error(void) { return g_last_error.c_str(); }
int vors_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
int vors_abi_version(void) { return 5; }

vors_status vors_selfcheck_isqrt(int* mismatches) {
    vors_status st = require_device();
    if (st != VORS_OK) return st;
    if (!mismatches) return fail(VORS_ERR_INVALID_ARGUMENT, "mismatches is null");
    const int n = vors::count_isqrt_u16_mismatches(nullptr);
    if (n < 0) return fail(VORS_ERR_HIP, "isqrt self-check could not run");
    *mismatches = n;
    return VORS_OK;
}

vors_status vors_device_info(int device, int* clock_khz, int* compute_units, uint64_t* memory_bytes) {
    vors_status st = require_device();
    if (st != VORS_OK) return st;
    if (device < 0 || device >= vors_device_count()) return fail(VORS_ERR_INVALID_ARGUMENT, "device index out of range");
    int v = 0;
    if (clock_khz) {
        HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, device));
        *clock_khz = v;
    }
    if (compute_units) {
        HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device));
        *compute_units = v;
    }
    if (memory_bytes) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        *memory_bytes = (uint64_t)prop.totalGlobalMem;
    }
    return VORS_OK;
}

vors_status vors_batch_create(const vors_config* cfg, int max_pairs, int rows, int cols, vors_batch** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        dev = 0;
    }
    return vors_batch_create_on(dev, cfg, max_pairs, rows, cols, out);
}

vors_status vors_batch_create_on(int device, const vors_config* cfg, int max_pairs, int rows, int cols, vors_batch** out) {
    if (!out) return fail(VORS_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (max_pairs < 1) return fail(VORS_ERR_INVALID_ARGUMENT, "max_pairs must be >= 1");
    Geom g;
    vors_status st = build_geom(cfg, rows, cols, &g);
    if (st != VORS_OK) return st;
    if ((st = require_device()) != VORS_OK) return st;
    if (device < 0 || device >= vors_device_count()) return fail(VORS_ERR_INVALID_ARGUMENT, "device index out of range");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(VORS_ERR_HIP, "hipSetDevice failed");
    vors_batch* b = new vors_batch();
    b->device = device;
    b->cfg = *cfg;
    b->g = g;
    b->max_pairs = max_pairs;
    // Threads per frame pair in the LM kernel. Few pairs: one big workgroup per CU (latency); many pairs: 256-thread
    // workgroups, several per CU, so that pairs with different iteration counts balance (measured).
    // (sparse modes, many pairs: 128 threads — the candidate lists are short, small workgroups waste fewer lanes at the coarse levels and
    // more of them are resident: coarse-to-fine LM stage 1.61 -> 1.28 ms per 4096 pairs)
    // Round 3: the thresholds below come from tools/speed_sweep.py over 320x240 / 640x480 / 1280x960 x 64 ... 4096 pairs (the round-2 ones
    // were fitted at 640x480 with 256 and 4096 pairs and cost 15-25 % at 512 coarse-to-fine pairs, 2.6x at 64 dense 1280x960 pairs):
    // the best size depends on the BATCH, hardly on the shape — the chip wants ~100 k resident threads whatever a pair is made of.
    if (g.mode == VORS_CANDIDATES_DENSE) b->lm_block = max_pairs >= 512 ? 256 : 1024;
    else if (g.mode == VORS_CANDIDATES_DSO) b->lm_block = max_pairs <= 768 ? 512 : 256;  // (lists of ~2000 candidates per level)
    else b->lm_block = max_pairs <= 768 ? 512 : (max_pairs < 1536 ? 256 : 128);
    if (const char* e = getenv("VORS_LM_BLOCK")) {  // tuning knob (256 / 512 / 1024)
        const int v = atoi(e);
        if (v != 64 && v != 128 && v != 256 && v != 512 && v != 1024) {
            delete b;
            return fail(VORS_ERR_INVALID_ARGUMENT, "VORS_LM_BLOCK must be 64, 128, 256, 512 or 1024");
        }
        b->lm_block = v;
    }
    const size_t np = (size_t)max_pairs;
    const size_t slots = np * (size_t)g.slots_total;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = dmalloc(&b->kf_upper, np * g.upper_stride, &b->bytes);
    if (e == hipSuccess) e = dmalloc(&b->cur_upper, np * g.upper_stride, &b->bytes);
    if (g.mode == VORS_CANDIDATES_DENSE) {
        if (e == hipSuccess) e = dmalloc(&b->rec.IZ, slots, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->rec.V, slots, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->rec.n_used, np * VORS_MAX_LEVELS, &b->bytes);
        // column-major records + current pyramid (engine.h RefDensePlanes). Their index arithmetic (i / rows through one multiply-high,
        // lm_reference.hip RefDenseTSrc) is exact while pixels x rows < 2^32 — up to 1920x1080 and beyond; larger frames keep the gathering
        // source on the row-major planes (correct, slow).
        // A level of ONE row (320x240 with 8 levels, 64x32 with 6) has no multiply-high divisor — floor(2^32 / 1) + 1 wraps to 0 and every
        // pixel of the level would decode as (0, i) instead of (i, 0): such pyramids keep the gathering source as well.
        bool one_row_level = false;
        for (int l = 0; l < g.L; ++l) one_row_level = one_row_level || g.lv[l].rows < 2;
        if (g.arith == VORS_ARITH_REFERENCE && !one_row_level && (unsigned long long)g.S0 * (unsigned long long)g.lv[0].rows < (1ull << 32)) {
            RefDensePlanes& t = b->rec.dense_t;
            if (e == hipSuccess) e = dmalloc(&t.recs, np * ((size_t)g.S0 + g.upper_stride), &b->bytes);
            if (e == hipSuccess) e = dmalloc(&t.n_valid, np * VORS_MAX_LEVELS, &b->bytes);
            if (e == hipSuccess) e = dmalloc(&t.cur0, np * g.S0, &b->bytes);
            if (e == hipSuccess) e = dmalloc(&t.curu, np * g.upper_stride, &b->bytes);
        }
    } else {  // sparse modes: compact 12-byte candidate lists (+ the keyframe kernel's staging grid in coarse-to-fine mode)
        if (e == hipSuccess) e = dmalloc(&b->rec.S, slots, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->rec.n_used, np * VORS_MAX_LEVELS, &b->bytes);
        if (g.mode == VORS_CANDIDATES_COARSE_TO_FINE) {
            keyframe_region_geometry(g, &b->rec.kf_r, &b->rec.n_regions);
            if (e == hipSuccess) e = dmalloc(&b->rec.stage, slots, &b->bytes);
            if (e == hipSuccess) e = dmalloc(&b->rec.region_cnt, np * VORS_MAX_LEVELS * (size_t)b->rec.n_regions, &b->bytes);
            b->rec.sort_tmp = b->rec.stage;  // (free once the regions have been compacted)
        } else if (g.arith == VORS_ARITH_REFERENCE) {
            if (e == hipSuccess) e = dmalloc(&b->rec.sort_tmp, slots, &b->bytes);
            b->owns_sort_tmp = true;
        }
    }
    if (e == hipSuccess && g.arith == VORS_ARITH_REFERENCE) {  // straggler hand-over of large batches (engine.h RefHandoff)
        if (e == hipSuccess) e = dmalloc(&b->rec.handoff.state, np, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->rec.handoff.list, np, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->rec.handoff.counters, 2, &b->bytes);
    }
    if (e == hipSuccess && g.mode == VORS_CANDIDATES_DSO) {
        const int rr = (rows + 31) / 32, rc = (cols + 31) / 32;
        b->dso.n_regions = rr * rc;
        b->dso.max_stride = g.S0 + g.S0 / 4 + g.S0 / 16 + 64;  // worst case: base block size 1
        b->dso.mask_stride = g.S0 + g.S0 / 4 + g.S0 / 16 + 64;
        int off = 0;
        for (int l = 0; l < g.L; ++l) {  // level 0 is not stored (mask + depth are read instead)
            b->pp.off[l] = off;
            if (l >= 1) off += (g.lv[l].rows * g.lv[l].cols + 3) & ~3;
        }
        b->pp.stride = off > 0 ? off : 4;
        int coff = 0;
        for (int l = 0; l < g.L; ++l) {
            b->pp.chunk_off[l] = coff;
            coff += (g.lv[l].rows * g.lv[l].cols + VORS_CHUNK_PX - 1) / VORS_CHUNK_PX;
        }
        b->pp.chunk_off[g.L] = b->pp.chunks_total = coff;
        if (e == hipSuccess) e = dmalloc(&b->dso.gmag, np * g.S0, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->dso.median, np * b->dso.n_regions, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->dso.thresh, np * b->dso.n_regions, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->dso.max_g, np * b->dso.max_stride, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->dso.max_pos, np * b->dso.max_stride, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->dso.mask1, np * b->dso.mask_stride, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->dso.picked, np * g.S0, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->dso.state, np, &b->bytes);
        if (e == hipSuccess) e = hipMemset(b->dso.state, 0, np * sizeof(DsoState));  // epoch 0: the first selection clears the stamp plane
        // picks of one selection round: every block of the first round's three levels at most (a later round with smaller blocks may
        // exceed it: the list then overflows and the pair falls back to the scan of the stamp plane)
        b->dso.list_cap = (g.S0 / 16 + g.S0 / 64 + g.S0 / 256 + 1024 + 3) & ~3;
        if (e == hipSuccess) e = dmalloc(&b->dso.pick_list, np * (size_t)b->dso.list_cap, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->mask0, np * g.S0, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->pp.iz, np * b->pp.stride, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->pp.v, np * b->pp.stride, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->pp.counts, np * b->pp.chunks_total, &b->bytes);
    }
    if (e == hipSuccess && g.mode == VORS_CANDIDATES_DENSE && !(getenv("VORS_LM_SPLIT") && atoi(getenv("VORS_LM_SPLIT")) == 0)) {
        // evaluation rounds on the finest levels (lm_kernels.hip): chunks per pair so that large batches get ~16 workgroups per pair and
        // small ones (down to the single tracker) still spread one evaluation over the chip.
        // `chunks` = partial-sum slots per pair = the late-round cut (at least 512 pixels each); the full rounds use a quarter of it
        // (ONE count for every handle below 512 pairs: the chunk count fixes the order of the f32 partial sums, and a vors_tracker (N = 1)
        // must stay bit-identical to a sequence of a lock-step handle of up to 511 sequences at every image size — the S0 / 2400 cap below
        // only happened to equalise 128 and 256 up to 640x480)
        int chunks = max_pairs >= 1024 ? 64 : (max_pairs >= 512 ? 128 : 256);
        chunks = std::max(4, std::min(chunks, g.S0 / 2400));  // (at least ~2400 pixels per chunk: 320x240 wants 32, not 64-150)
        if (const char* ev = getenv("VORS_LM_CHUNKS")) chunks = std::max(4, atoi(ev));
        b->split.chunks = chunks;
        // levels worth a chip-wide launch per evaluation: at least 64 Ki pixels (640x480: levels 0 and 1; 1280x960: 0, 1, 2)
        int n_split = 0;
        for (int l = 0; l < g.L; ++l)
            if ((long long)g.lv[l].rows * g.lv[l].cols >= 65536) n_split = l + 1;
        b->split.n_split = getenv("VORS_LM_SPLIT_LEVELS") ? atoi(getenv("VORS_LM_SPLIT_LEVELS")) : std::max(1, n_split);
        b->split.n_split = std::max(1, std::min(b->split.n_split, g.L));
        // FUSED: a level of at most fused_exact_points pixels is evaluated in the EXACT arithmetic (include/vors_hip.h) — the per-pair kernel
        // applies that rule, the evaluation rounds do not, so such levels are never solved by rounds (tiny images: no rounds at all)
        if (g.arith == VORS_ARITH_FUSED)
            while (b->split.n_split > 0 && g.lv[b->split.n_split - 1].rows * g.lv[b->split.n_split - 1].cols <= g.fused_exact_points) b->split.n_split -= 1;
        // rounds before the per-pair finish: a level solved by rounds needs >= 2 of them per evaluation pattern, so the count follows the
        // number of such levels (1280x960 has three: 10 rounds left 64 pairs 2.6x slower than 16)
        const int ns = b->split.n_split;
        b->split.rounds = getenv("VORS_LM_SPLIT_ROUNDS") ? atoi(getenv("VORS_LM_SPLIT_ROUNDS")) : (max_pairs >= 512 ? (ns <= 1 ? 12 : 26) : 4 * ns + 8);
        // (a lone dense pair would be 8 % faster with 2 * ns rounds, but a vors_tracker must stay bit-identical to a sequence of a
        // lock-step handle of up to 511 sequences: the same count for every handle below 512 pairs)
        if (b->split.n_split == 0) b->split.chunks = 0;  // (split path off: launch_lm_track then runs the per-pair kernel for every level)
        if (e == hipSuccess) e = dmalloc(&b->split.state, np, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->split.partials, np * chunks * 32, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->split.list[0], np, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->split.list[1], np, &b->bytes);
        if (e == hipSuccess) e = dmalloc(&b->split.count, (size_t)SPLIT_COUNT_INTS, &b->bytes);
        b->split.cap = max_pairs;
        b->split.side_round = -1;
        // side lane for the level-1 stragglers of a LARGE batch (engine.h LmSplitWs): two levels solved by rounds, >= 2048 pairs — the rounds
        // of a smaller batch are short enough for the stragglers to keep up (measured: 512 pairs 2.21 -> 2.35 ms, 1024 pairs 3.76 -> 3.81 ms
        // with it, 4096 pairs 12.3 -> 11.9 ms); VORS_LM_SIDE=0 turns it off, VORS_LM_SIDE=1 forces it from 512 pairs on
        const int side_from = (getenv("VORS_LM_SIDE") && atoi(getenv("VORS_LM_SIDE")) == 1) ? 512 : 2048;
        if (b->split.chunks > 0 && b->split.n_split == 2 && max_pairs >= side_from && !(getenv("VORS_LM_SIDE") && atoi(getenv("VORS_LM_SIDE")) == 0)) {
            if (e == hipSuccess) e = dmalloc(&b->split.side_list, np, &b->bytes);
            if (e == hipSuccess) e = dmalloc(&b->split.join_list, np, &b->bytes);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->split.side_stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&b->split.ev_fork, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&b->split.ev_join, hipEventDisableTiming);
            if (e == hipSuccess) {
                b->split.side_round = 1;
                if (!getenv("VORS_LM_SPLIT_ROUNDS")) b->split.rounds = 10;  // (the long tail of rounds was theirs; measured 8 / 10 / 12 / 16 / 26 at 4096 pairs)
            }
        }
    }
    float2* lut = nullptr;
    if (e == hipSuccess && g.mode == VORS_CANDIDATES_DENSE) {
        e = dmalloc(&lut, (size_t)65536, &b->bytes);
        b->rec.LUT = lut;
    }
    if (e != hipSuccess) {
        batch_free(b);
        return fail(VORS_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    if (lut) launch_build_depth_lut(g.depth_scale, lut, nullptr);
    if (g.mode == VORS_CANDIDATES_DENSE) b->g.fast_idepth = (!getenv("VORS_NO_FASTDIV") && verify_fast_idepth(g.depth_scale, nullptr)) ? 1 : 0;
    // Fast exact division by the focal lengths: proven per divisor by exhaustive enumeration on the device, else disabled.
    // Levels halve the focal lengths exactly (camera.rs:119-120), so the level-0 proof covers every level.
    {
        const float fu0 = g.lv[0].k.fu, fv0 = g.lv[0].k.fv;
        const bool ok_u = !getenv("VORS_NO_FASTDIV") && verify_fastdiv(fu0, 1.0f / fu0, nullptr);
        const bool ok_v = !getenv("VORS_NO_FASTDIV") && verify_fastdiv(fv0, 1.0f / fv0, nullptr);
        for (int l = 0; l < g.L; ++l) {
            const float fu = b->g.lv[l].k.fu, fv = b->g.lv[l].k.fv;
            const bool pow2_u = (fu * (float)(1 << l) == fu0), pow2_v = (fv * (float)(1 << l) == fv0);
            b->g.lv[l].fu = FastDiv{fu, 1.0f / fu, (ok_u && pow2_u) ? 1 : 0};
            b->g.lv[l].fv = FastDiv{fv, 1.0f / fv, (ok_v && pow2_v) ? 1 : 0};
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) {
        batch_free(b);
        return fail(VORS_ERR_HIP, "device error while initialising the batch handle");
    }
    *out = b;
    return VORS_OK;
}

void vors_batch_destroy(vors_batch* b) {
    if (!b) return;
    DeviceGuard guard(b->device);
    batch_free(b);
}
vors_status vors_batch_device(const vors_batch* b, int* device) {
    if (!b || !device) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    *device = b->device;
    return VORS_OK;
}

vors_status vors_batch_workspace_bytes(const vors_batch* b, uint64_t* bytes) {
    if (!b || !bytes) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    *bytes = b->bytes;
    return VORS_OK;
}

vors_status vors_batch_enable_kernel_timing(vors_batch* b, int ring) {
    if (!b) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL handle");
    if (ring < 0 || ring > 4096) return fail(VORS_ERR_INVALID_ARGUMENT, "ring must be in [0, 4096]");
    DeviceGuard guard(b->device);  // events belong to the device that is current when they are created: the handle's, not the caller's
    if (!guard.ok) return fail(VORS_ERR_HIP, "hipSetDevice failed");
    b->ring = 0;  // stays off if an event cannot be created below
    for (int st = 0; st < 4; ++st) {
        for (auto e : b->ev0[st])
            if (e) (void)hipEventDestroy(e);
        for (auto e : b->ev1[st])
            if (e) (void)hipEventDestroy(e);
        b->ev0[st].assign(ring, nullptr);
        b->ev1[st].assign(ring, nullptr);
        b->count[st] = 0;
        for (int k = 0; k < ring; ++k) {
            HIP_TRY(hipEventCreate(&b->ev0[st][k]));
            HIP_TRY(hipEventCreate(&b->ev1[st][k]));
        }
    }
    b->ring = ring;
    return VORS_OK;
}

static vors_status check_stream(const vors_batch* b, hipStream_t s) {
    if (!s) return VORS_OK;  // the default stream of the handle's device (the guard has switched to it)
    hipDevice_t d;
    if (hipStreamGetDevice(s, &d) != hipSuccess) {
        (void)hipGetLastError();
        return VORS_OK;  // cannot tell: let the launch report
    }
    if ((int)d != b->device)
        return fail(VORS_ERR_INVALID_ARGUMENT, "the stream belongs to device " + std::to_string((int)d) + " but the handle lives on device " +
                                                   std::to_string(b->device));
    return VORS_OK;
}
static vors_status check_n(const vors_batch* b, int n_pairs) {
    if (!b) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL handle");
    if (n_pairs < 1 || n_pairs > b->max_pairs) return fail(VORS_ERR_INVALID_ARGUMENT, "n_pairs out of range for this handle");
    return VORS_OK;
}

vors_status vors_batch_prepare_keyframes(vors_batch* b, int n_pairs, const uint8_t* d_kf_gray, const uint16_t* d_kf_depth,
                                         void* hip_stream) {
    vors_status st = check_n(b, n_pairs);
    if (st != VORS_OK) return st;
    if (!d_kf_gray || !d_kf_depth) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL image pointer");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    DeviceGuard guard(b->device);
    if ((st = check_stream(b, s)) != VORS_OK) return st;
    b->kf_level0 = d_kf_gray;
    b->kf_depth = d_kf_depth;
    b->prepared_pairs = n_pairs;
    Pyramid kf{d_kf_gray, b->kf_upper};
    STAGE_BEGIN(b, 0, s);
    launch_pyramid(b->g, kf, n_pairs, s);
    STAGE_END(b, 0, s);
    STAGE_BEGIN(b, 1, s);
    if (b->g.mode == VORS_CANDIDATES_DSO) {
        launch_keyframe_dso(b->g, kf, d_kf_depth, b->dso, b->mask0, b->pp, b->rec, n_pairs, s);
    } else {
        launch_keyframe(b->g, kf, d_kf_depth, b->rec, n_pairs, s);
    }
    if (b->g.arith == VORS_ARITH_REFERENCE) {  // extract_z's order (inverse_compositional.rs:260-279): sorted lists / column-major planes
        launch_sort_colmajor(b->g, b->rec, n_pairs, s);
        launch_ref_dense_planes_keyframe(b->g, kf, d_kf_depth, b->rec, n_pairs, s);
    }
    STAGE_END(b, 1, s);
    HIP_TRY(hipGetLastError());
    return VORS_OK;
}

static vors_status batch_track_current(vors_batch* b, int n_pairs, const uint8_t* d_cur_gray, const float* d_prev_poses7,
                                       const float* d_kf_poses7, float* d_out_poses7, int32_t* d_out_status,
                                       vors_pair_stats* d_out_stats, hipStream_t s) {
    if (b->prepared_pairs <= 0) return fail(VORS_ERR_INVALID_ARGUMENT, "track_current called before prepare_keyframes");
    if (n_pairs > b->prepared_pairs)
        return fail(VORS_ERR_INVALID_ARGUMENT, "track_current: n_pairs (" + std::to_string(n_pairs) + ") exceeds the " +
                                                   std::to_string(b->prepared_pairs) + " keyframes prepared on this handle");
    b->cur_level0 = d_cur_gray;
    b->current_pairs = n_pairs;
    Pyramid cur{d_cur_gray, b->cur_upper};
    STAGE_BEGIN(b, 2, s);
    launch_pyramid(b->g, cur, n_pairs, s);
    if (b->g.arith == VORS_ARITH_REFERENCE) launch_ref_dense_planes_current(b->g, cur, b->rec, n_pairs, s);
    STAGE_END(b, 2, s);
    STAGE_BEGIN(b, 3, s);
    if (b->g.arith == VORS_ARITH_REFERENCE)
        launch_lm_track_reference(b->g, cur, Pyramid{b->kf_level0, b->kf_upper}, b->kf_depth, b->rec, d_prev_poses7, d_kf_poses7, d_out_poses7, d_out_status, d_out_stats, n_pairs, s);
    else
        launch_lm_track(b->g, cur, Pyramid{b->kf_level0, b->kf_upper}, b->kf_depth, b->rec, d_prev_poses7, d_kf_poses7, d_out_poses7, d_out_status, d_out_stats, n_pairs, b->lm_block, b->split, s);
    STAGE_END(b, 3, s);
    HIP_TRY(hipGetLastError());
    return VORS_OK;
}

vors_status vors_batch_track_current(vors_batch* b, int n_pairs, const uint8_t* d_cur_gray, const float* d_prev_poses7,
                                     float* d_out_poses7, int32_t* d_out_status, vors_pair_stats* d_out_stats,
                                     void* hip_stream) {
    vors_status st = check_n(b, n_pairs);
    if (st != VORS_OK) return st;
    if (!d_cur_gray || !d_out_poses7 || !d_out_status) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL pointer");
    DeviceGuard guard(b->device);
    if ((st = check_stream(b, static_cast<hipStream_t>(hip_stream))) != VORS_OK) return st;
    return batch_track_current(b, n_pairs, d_cur_gray, d_prev_poses7, nullptr, d_out_poses7, d_out_status, d_out_stats,
                               static_cast<hipStream_t>(hip_stream));
}

vors_status vors_batch_track_pairs(vors_batch* b, int n_pairs, const uint8_t* d_kf_gray, const uint16_t* d_kf_depth,
                                   const uint8_t* d_cur_gray, const float* d_prev_poses7, float* d_out_poses7,
                                   int32_t* d_out_status, vors_pair_stats* d_out_stats, void* hip_stream) {
    vors_status st = vors_batch_prepare_keyframes(b, n_pairs, d_kf_gray, d_kf_depth, hip_stream);
    if (st != VORS_OK) return st;
    return vors_batch_track_current(b, n_pairs, d_cur_gray, d_prev_poses7, d_out_poses7, d_out_status, d_out_stats, hip_stream);
}

vors_status vors_batch_kernel_times(vors_batch* b, int stage, float* ms_out, int capacity, int* n_out) {
    if (!b || !n_out || stage < 0 || stage > 3 || (capacity > 0 && !ms_out)) return fail(VORS_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(b->device);
    const int n = (int)std::min<long>(b->count[stage], b->ring);
    *n_out = n;
    for (int k = 0; k < n && k < capacity; ++k) {
        // oldest first
        const long idx = (b->count[stage] - n + k) % b->ring;
        HIP_TRY(hipEventSynchronize(b->ev1[stage][idx]));
        HIP_TRY(hipEventElapsedTime(&ms_out[k], b->ev0[stage][idx], b->ev1[stage][idx]));
    }
    return VORS_OK;
}

vors_status vors_batch_last_kernel_ms(vors_batch* b, float* lm_ms, float* keyframe_ms, float* pyramid_ms) {
    if (!b) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL handle");
    DeviceGuard guard(b->device);
    float v[4] = {-1.f, -1.f, -1.f, -1.f};
    for (int st = 0; st < 4; ++st)
        if (b->ring > 0 && b->count[st] > 0) {
            const long idx = (b->count[st] - 1) % b->ring;
            HIP_TRY(hipEventSynchronize(b->ev1[st][idx]));
            HIP_TRY(hipEventElapsedTime(&v[st], b->ev0[st][idx], b->ev1[st][idx]));
        }
    if (lm_ms) *lm_ms = v[3];
    if (keyframe_ms) *keyframe_ms = v[1];
    if (pyramid_ms) *pyramid_ms = (v[0] < 0.f && v[2] < 0.f) ? -1.f : std::max(v[0], 0.f) + std::max(v[2], 0.f);
    return VORS_OK;
}

static vors_status get_image(vors_batch* b, const uint8_t* level0, const uint8_t* upper, int pair, int level, uint8_t* out,
                             int* rows, int* cols) {
    if (!b || !out) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (pair < 0 || pair >= b->max_pairs || level < 0 || level >= b->g.L) return fail(VORS_ERR_INVALID_ARGUMENT, "pair/level out of range");
    if (!level0) return fail(VORS_ERR_INVALID_ARGUMENT, "no image has been submitted yet");
    DeviceGuard guard(b->device);
    const LevelGeom& lg = b->g.lv[level];
    const uint8_t* src = level == 0 ? level0 + (size_t)pair * b->g.S0 : upper + (size_t)pair * b->g.upper_stride + lg.img_off;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, src, (size_t)lg.rows * lg.cols, hipMemcpyDeviceToHost));
    if (rows) *rows = lg.rows;
    if (cols) *cols = lg.cols;
    return VORS_OK;
}
vors_status vors_batch_get_keyframe_image(vors_batch* b, int pair, int level, uint8_t* out, int* rows, int* cols) {
    return get_image(b, b ? b->kf_level0 : nullptr, b ? b->kf_upper : nullptr, pair, level, out, rows, cols);
}
vors_status vors_batch_get_current_image(vors_batch* b, int pair, int level, uint8_t* out, int* rows, int* cols) {
    return get_image(b, b ? b->cur_level0 : nullptr, b ? b->cur_upper : nullptr, pair, level, out, rows, cols);
}

vors_status vors_batch_get_points(vors_batch* b, int pair, int level, int capacity, int32_t* xy, float* idepth, float* jac,
                                  uint8_t* tmpl, int* n_out) {
    if (!b || !n_out) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (pair < 0 || pair >= b->max_pairs || level < 0 || level >= b->g.L) return fail(VORS_ERR_INVALID_ARGUMENT, "pair/level out of range");
    const LevelGeom& lg = b->g.lv[level];
    size_t n = (size_t)lg.n_slots;
    DeviceGuard guard(b->device);
    HIP_TRY(hipDeviceSynchronize());
    const bool dense = b->g.mode == VORS_CANDIDATES_DENSE;
    if (!b->kf_level0 || !b->kf_depth)
        return fail(VORS_ERR_INVALID_ARGUMENT, b->prepared_pairs > 0 ? "keyframe inspection is not available on a trackers-owned batch in the candidate-list modes (the handle keeps records, not frames)" : "no keyframe has been prepared yet");
    if (!dense) {  // sparse modes: compact lists
        int used = 0;
        HIP_TRY(hipMemcpy(&used, b->rec.n_used + (size_t)pair * VORS_MAX_LEVELS + level, sizeof(int), hipMemcpyDeviceToHost));
        n = (size_t)std::min(std::max(used, 0), lg.n_slots);
    }
    std::vector<float4> A(n), B(n);
    std::vector<float2> C(n);
    std::vector<uint32_t> XY(n);
    std::vector<float> IZ(n);
    {
        // no mode keeps full records: materialise this level with the exact arithmetic of the reference's precompute
        DevBuf dA, dB, dC, dXY, dIZ;
        HIP_TRY(dA.alloc(n * 16));
        HIP_TRY(dB.alloc(n * 16));
        HIP_TRY(dC.alloc(n * 8));
        HIP_TRY(dXY.alloc(n * 4));
        HIP_TRY(dIZ.alloc(n * 4));
        Records out{dA.as<float4>(), dB.as<float4>(), dC.as<float2>(), dXY.as<uint32_t>(), dIZ.as<float>(), nullptr, nullptr, nullptr};
        if (dense) launch_dense_materialize(b->g, level, pair, Pyramid{b->kf_level0, b->kf_upper}, b->kf_depth, b->rec, out, nullptr);
        else launch_slim_materialize(b->g, level, pair, b->rec, (int)n, out, nullptr);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(A.data(), dA.p, n * sizeof(float4), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(B.data(), dB.p, n * sizeof(float4), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(C.data(), dC.p, n * sizeof(float2), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(XY.data(), dXY.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(IZ.data(), dIZ.p, n * sizeof(float), hipMemcpyDeviceToHost));
    }
    int cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!(A[i].w >= 0.f)) continue;
        if (cnt < capacity) {
            if (xy) {
                xy[2 * cnt] = (int32_t)(XY[i] & 0xffffu);
                xy[2 * cnt + 1] = (int32_t)(XY[i] >> 16);
            }
            if (idepth) idepth[cnt] = IZ[i];
            if (jac) {
                jac[6 * cnt] = B[i].x; jac[6 * cnt + 1] = B[i].y; jac[6 * cnt + 2] = B[i].z; jac[6 * cnt + 3] = B[i].w;
                jac[6 * cnt + 4] = C[i].x; jac[6 * cnt + 5] = C[i].y;
            }
            if (tmpl) tmpl[cnt] = (uint8_t)A[i].w;
        }
        ++cnt;
    }
    *n_out = cnt;
    return VORS_OK;
}

vors_status vors_batch_eval_level(vors_batch* b, int pair, int level, const float model7[7], int arithmetic, float sums29[29]) {
    if (!b || !model7 || !sums29) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (pair < 0 || pair >= std::min(b->prepared_pairs, b->current_pairs) || level < 0 || level >= b->g.L)
        return fail(VORS_ERR_INVALID_ARGUMENT, "pair/level out of range (pair must be < the n_pairs of the last prepare_keyframes AND track_current)");
    if (!b->kf_level0 || !b->cur_level0)
        return fail(VORS_ERR_INVALID_ARGUMENT, (b->prepared_pairs > 0 && b->current_pairs > 0 && !b->kf_level0) ? "keyframe inspection is not available on a trackers-owned batch in the candidate-list modes (the handle keeps records, not frames)"
                                                                                                                 : "eval_level needs prepare_keyframes and track_current first");
    if (arithmetic != VORS_ARITH_EXACT && arithmetic != VORS_ARITH_FUSED && arithmetic != VORS_ARITH_REFERENCE)
        return fail(VORS_ERR_INVALID_ARGUMENT, "unknown arithmetic mode");
    DeviceGuard guard(b->device);
    DevBuf d_model, d_out;
    HIP_TRY(d_model.alloc(7 * sizeof(float)));
    HIP_TRY(d_out.alloc(32 * sizeof(float)));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(d_model.p, model7, 7 * sizeof(float), hipMemcpyHostToDevice));
    const Pyramid cur{b->cur_level0, b->cur_upper}, kf{b->kf_level0, b->kf_upper};
    if (arithmetic == VORS_ARITH_REFERENCE)  // sequential sums in the order of the handle's lists (column-major iff the handle itself is REFERENCE)
        launch_lm_eval_level_reference(b->g, cur, kf, b->kf_depth, b->rec, pair, level, d_model.as<float>(), d_out.as<float>(), nullptr);
    else if (arithmetic == VORS_ARITH_FUSED)
        launch_lm_eval_level_fused(b->g, cur, kf, b->kf_depth, b->rec, pair, level, d_model.as<float>(), d_out.as<float>(), nullptr);
    else
        launch_lm_eval_level_exact(b->g, cur, kf, b->kf_depth, b->rec, pair, level, d_model.as<float>(), d_out.as<float>(), nullptr);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(sums29, d_out.p, 29 * sizeof(float), hipMemcpyDeviceToHost));
    return VORS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// host-buffer batch entry
// ---------------------------------------------------------------------------------------------------------------
// Upload an image batch and convert to the row-major device layout when the caller's layout is column-major.
static vors_status upload_u8(const uint8_t* host, int n, int rows, int cols, int layout, DevBuf& dst, DevBuf& tmp, hipStream_t s) {
    const size_t bytes = (size_t)n * rows * cols;
    if (layout == VORS_ROW_MAJOR) {
        HIP_TRY(hipMemcpyAsync(dst.p, host, bytes, hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(hipMemcpyAsync(tmp.p, host, bytes, hipMemcpyHostToDevice, s));
        launch_transpose_u8(tmp.as<uint8_t>(), dst.as<uint8_t>(), rows, cols, n, s);
    }
    return VORS_OK;
}
static vors_status upload_u16(const uint16_t* host, int n, int rows, int cols, int layout, DevBuf& dst, DevBuf& tmp, hipStream_t s) {
    const size_t bytes = (size_t)n * rows * cols * 2;
    if (layout == VORS_ROW_MAJOR) {
        HIP_TRY(hipMemcpyAsync(dst.p, host, bytes, hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(hipMemcpyAsync(tmp.p, host, bytes, hipMemcpyHostToDevice, s));
        launch_transpose_u16(tmp.as<uint16_t>(), dst.as<uint16_t>(), rows, cols, n, s);
    }
    return VORS_OK;
}

vors_status vors_track_pairs(const vors_config* cfg, int n_pairs, const uint8_t* kf_gray, const uint16_t* kf_depth,
                             const uint8_t* cur_gray, int rows, int cols, int layout, const float* prev_poses7,
                             float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats) {
    if (!kf_gray || !kf_depth || !cur_gray || !out_poses7 || !out_status) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL pointer");
    if (layout != VORS_ROW_MAJOR && layout != VORS_COL_MAJOR) return fail(VORS_ERR_INVALID_ARGUMENT, "bad layout");
    if (n_pairs < 1) return fail(VORS_ERR_INVALID_ARGUMENT, "n_pairs must be >= 1");
    vors_batch* b = nullptr;
    vors_status st = vors_batch_create(cfg, n_pairs, rows, cols, &b);
    if (st != VORS_OK) return st;
    struct Guard {
        vors_batch* b;
        ~Guard() { vors_batch_destroy(b); }
    } guard{b};
    const size_t S = (size_t)rows * cols, n = (size_t)n_pairs;
    DevBuf d_kf, d_dep, d_cur, d_tmp, d_prev, d_pose, d_stat, d_stats;
    HIP_TRY(d_kf.alloc(n * S));
    HIP_TRY(d_dep.alloc(n * S * 2));
    HIP_TRY(d_cur.alloc(n * S));
    if (layout == VORS_COL_MAJOR) HIP_TRY(d_tmp.alloc(n * S * 2));
    HIP_TRY(d_pose.alloc(n * 7 * sizeof(float)));
    HIP_TRY(d_stat.alloc(n * sizeof(int32_t)));
    HIP_TRY(d_stats.alloc(n * sizeof(vors_pair_stats)));
    hipStream_t s = nullptr;
    if ((st = upload_u8(kf_gray, n_pairs, rows, cols, layout, d_kf, d_tmp, s)) != VORS_OK) return st;
    if ((st = upload_u16(kf_depth, n_pairs, rows, cols, layout, d_dep, d_tmp, s)) != VORS_OK) return st;
    if ((st = upload_u8(cur_gray, n_pairs, rows, cols, layout, d_cur, d_tmp, s)) != VORS_OK) return st;
    if (prev_poses7) {
        HIP_TRY(d_prev.alloc(n * 7 * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(d_prev.p, prev_poses7, n * 7 * sizeof(float), hipMemcpyHostToDevice, s));
    }
    st = vors_batch_track_pairs(b, n_pairs, d_kf.as<uint8_t>(), d_dep.as<uint16_t>(), d_cur.as<uint8_t>(),
                                prev_poses7 ? d_prev.as<float>() : nullptr, d_pose.as<float>(), d_stat.as<int32_t>(),
                                d_stats.as<vors_pair_stats>(), s);
    if (st != VORS_OK) return st;
    HIP_TRY(hipMemcpyAsync(out_poses7, d_pose.p, n * 7 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_status, d_stat.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (out_stats) HIP_TRY(hipMemcpyAsync(out_stats, d_stats.p, n * sizeof(vors_pair_stats), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return VORS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Tracker: one sequence (Config::init / Tracker::track / Tracker::current_frame)
// ---------------------------------------------------------------------------------------------------------------
// The single sequence is the N = 1 case of the lock-step engine below (vors_trackers_*): poses, the keyframe test and the promotion of
// the current frame stay on the device, so a frame is ONE chain of stream-ordered work and ONE synchronisation — upload (pinned staging,
// the depth map on a second stream: it is only read by a promotion, after the LM stage), Tracker::track, read-back of pose / status /
// diagnostics. The host mirrors only what current_frame() / keyframe() report.
struct PinnedBuf {
    void* p = nullptr;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault); }
    template <class T>
    T* as() { return static_cast<T*>(p); }
};
struct vors_tracker {
    vors_config cfg;
    int rows = 0, cols = 0, layout = 0, device = 0;
    vors_trackers* seq = nullptr;  // n_sequences = 1
    DevBuf gray, depth, tmp8, tmp16;   // the frame on the device (row-major); tmp*: column-major uploads before the transpose
    PinnedBuf h_gray, h_depth, h_out;  // staging: frame in; pose7 + status + keyframe index + vors_pair_stats out
    hipStream_t s_main = nullptr, s_copy = nullptr;
    hipEvent_t ev_depth = nullptr, ev_frame_done = nullptr, ev_result = nullptr;
    // State of inverse_compositional.rs:52-60 as the host reports it
    double keyframe_depth_timestamp = 0, keyframe_img_timestamp = 0;
    Iso keyframe_pose = iso_identity();
    double current_frame_depth_timestamp = 0, current_frame_img_timestamp = 0;
    Iso current_frame_pose = iso_identity();
    vors_pair_stats last{};
    bool has_last = false;
    ~vors_tracker() {
        vors_trackers_destroy(seq);
        if (ev_depth) (void)hipEventDestroy(ev_depth);
        if (ev_frame_done) (void)hipEventDestroy(ev_frame_done);
        if (ev_result) (void)hipEventDestroy(ev_result);
        if (s_main) (void)hipStreamDestroy(s_main);
        if (s_copy) (void)hipStreamDestroy(s_copy);
    }
};
struct TrackerOut {  // layout of vors_tracker::h_out
    float pose[7];
    int32_t status, kf_index;
    vors_pair_stats stats;
};

// Frame -> device (row-major). The caller's buffers are pageable: they are copied into pinned staging first, so that the transfers are
// truly asynchronous (the depth map travels on its own stream under the LM stage).
// Two halves, so that vors_tracker_track can stage the depth map (the larger copy, on the CPU) WHILE the device already runs the pyramid
// and the LM stage of the frame: only the promotion at the end of the frame reads it.
static vors_status tracker_upload_gray(vors_tracker* t, const uint8_t* gray) {
    const size_t S = (size_t)t->rows * t->cols;
    std::memcpy(t->h_gray.p, gray, S);  // (the previous upload has completed: its results were waited for)
    if (t->layout == VORS_ROW_MAJOR) {
        HIP_TRY(hipMemcpyAsync(t->gray.p, t->h_gray.p, S, hipMemcpyHostToDevice, t->s_main));
    } else {
        HIP_TRY(hipMemcpyAsync(t->tmp8.p, t->h_gray.p, S, hipMemcpyHostToDevice, t->s_main));
        launch_transpose_u8(t->tmp8.as<uint8_t>(), t->gray.as<uint8_t>(), t->rows, t->cols, 1, t->s_main);
    }
    return VORS_OK;
}
static vors_status tracker_upload_depth(vors_tracker* t, const uint16_t* depth) {
    const size_t S = (size_t)t->rows * t->cols;
    HIP_TRY(hipEventSynchronize(t->ev_depth));  // (track() returns once the RESULTS are back: the previous depth upload may still be reading the staging buffer)
    std::memcpy(t->h_depth.p, depth, S * 2);
    // the previous frame's promotion may still read t->depth: the copy stream first waits for the end of the previous frame
    HIP_TRY(hipStreamWaitEvent(t->s_copy, t->ev_frame_done, 0));
    if (t->layout == VORS_ROW_MAJOR) {
        HIP_TRY(hipMemcpyAsync(t->depth.p, t->h_depth.p, S * 2, hipMemcpyHostToDevice, t->s_copy));
    } else {
        HIP_TRY(hipMemcpyAsync(t->tmp16.p, t->h_depth.p, S * 2, hipMemcpyHostToDevice, t->s_copy));
        launch_transpose_u16(t->tmp16.as<uint16_t>(), t->depth.as<uint16_t>(), t->rows, t->cols, 1, t->s_copy);
    }
    HIP_TRY(hipEventRecord(t->ev_depth, t->s_copy));
    return VORS_OK;
}
static vors_status tracker_upload(vors_tracker* t, const uint8_t* gray, const uint16_t* depth) {
    vors_status st = tracker_upload_gray(t, gray);
    return st != VORS_OK ? st : tracker_upload_depth(t, depth);
}

extern "C" {

vors_status vors_tracker_create(const vors_config* cfg, double depth_time, const uint16_t* depth, double img_time,
                                const uint8_t* gray, int rows, int cols, int layout, vors_tracker** out) {
    if (!out) return fail(VORS_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!depth || !gray) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL image pointer");
    if (layout != VORS_ROW_MAJOR && layout != VORS_COL_MAJOR) return fail(VORS_ERR_INVALID_ARGUMENT, "bad layout");
    vors_trackers* seq = nullptr;
    vors_status st = vors_trackers_create(cfg, 1, rows, cols, &seq);
    if (st != VORS_OK) return st;
    vors_tracker* t = new vors_tracker();
    t->seq = seq;
    t->cfg = *cfg;
    t->rows = rows;
    t->cols = cols;
    t->layout = layout;
    struct Guard {
        vors_tracker* t;
        ~Guard() { delete t; }
    } guard{t};
    if (hipGetDevice(&t->device) != hipSuccess) t->device = 0;  // the tracker lives on the device that is current at creation
    const size_t S = (size_t)rows * cols;
    HIP_TRY(t->gray.alloc(S));
    HIP_TRY(t->depth.alloc(S * 2));
    if (layout == VORS_COL_MAJOR) {
        HIP_TRY(t->tmp8.alloc(S));
        HIP_TRY(t->tmp16.alloc(S * 2));
    }
    HIP_TRY(t->h_gray.alloc(S));
    HIP_TRY(t->h_depth.alloc(S * 2));
    HIP_TRY(t->h_out.alloc(sizeof(TrackerOut)));
    HIP_TRY(hipStreamCreateWithFlags(&t->s_main, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&t->s_copy, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&t->ev_depth, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&t->ev_frame_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&t->ev_result, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(t->ev_frame_done, t->s_main));
    if ((st = tracker_upload(t, gray, depth)) != VORS_OK) return st;
    HIP_TRY(hipStreamWaitEvent(t->s_main, t->ev_depth, 0));
    st = vors_trackers_init(t->seq, t->gray.as<uint8_t>(), t->depth.as<uint16_t>(), t->s_main);  // (synchronises s_main)
    if (st != VORS_OK) return st;
    HIP_TRY(hipEventRecord(t->ev_frame_done, t->s_main));
    t->keyframe_depth_timestamp = depth_time;
    t->keyframe_img_timestamp = img_time;
    t->current_frame_depth_timestamp = depth_time;
    t->current_frame_img_timestamp = img_time;
    guard.t = nullptr;
    *out = t;
    return VORS_OK;
}

// The two halves of vors_trackers_track (internal; defined with the lock-step engine below): Tracker::track up to the keyframe test,
// and the promotion of the sequences that switch — the only reader of the depth map, which may arrive on another stream (depth_ready).
static vors_status trackers_track_lm(vors_trackers* t, const uint8_t* d_gray, hipStream_t s);
static vors_status trackers_promote(vors_trackers* t, const uint8_t* d_gray, const uint16_t* d_depth, hipEvent_t depth_ready, hipStream_t s);

vors_status vors_tracker_track(vors_tracker* t, double depth_time, const uint16_t* depth, double img_time, const uint8_t* gray,
                               int* track_status) {
    if (!t || !depth || !gray) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    DeviceGuard guard(t->device);
    // Tracker::track (inverse_compositional.rs:170-240) incl. the keyframe switch, all on the device. Order on the host: grey image up,
    // pyramid + LM + keyframe test enqueued, THEN the depth map staged and sent on the copy stream (under the LM stage), then the
    // promotion, which waits for it.
    // ... and the host returns as soon as the RESULTS are back (a packed record the device stores into pinned host memory), while the
    // promotion of a switching frame still runs: the next call is ordered behind it on the stream.
    const float* d_pose = nullptr;
    const int32_t *d_status = nullptr, *d_kf = nullptr;
    const vors_pair_stats* d_stats = nullptr;
    (void)vors_trackers_state(t->seq, &d_pose, nullptr, &d_status, &d_kf, &d_stats);
    TrackerOut* o = t->h_out.as<TrackerOut>();
    vors_status st = VORS_OK;
    // (A HIP graph of this per-frame sequence — it has no per-frame argument any more: the frame index lives on the device — was built and
    // measured in round 4: 0.172 vs 0.174 ms per frame. The launches are enqueued ahead of the device anyway; what the frame waits for is
    // the LM kernel's chain of ~35 dependent evaluations. Not kept.)
    if ((st = tracker_upload_gray(t, gray)) != VORS_OK) return st;
    if ((st = trackers_track_lm(t->seq, t->gray.as<uint8_t>(), t->s_main)) != VORS_OK) return st;
    launch_tracker_pack_out(d_pose, d_status, d_kf, d_stats, o, t->s_main);
    HIP_TRY(hipGetLastError());  // (a failed launch is reported against THIS frame, not against whatever touches the stream next)
    HIP_TRY(hipEventRecord(t->ev_result, t->s_main));
    if ((st = tracker_upload_depth(t, depth)) != VORS_OK) return st;
    if ((st = trackers_promote(t->seq, t->gray.as<uint8_t>(), t->depth.as<uint16_t>(), t->ev_depth, t->s_main)) != VORS_OK) return st;
    HIP_TRY(hipEventRecord(t->ev_frame_done, t->s_main));
    HIP_TRY(hipEventSynchronize(t->ev_result));
    t->last = o->stats;
    t->has_last = true;
    // inverse_compositional.rs:203-208
    t->current_frame_depth_timestamp = depth_time;
    t->current_frame_img_timestamp = img_time;
    t->current_frame_pose = iso_load(o->pose);  // == previous pose when the optimizer failed
    // inverse_compositional.rs:224-239 (the device has already promoted the frame)
    if (t->last.change_keyframe) {
        t->keyframe_depth_timestamp = depth_time;
        t->keyframe_img_timestamp = img_time;
        t->keyframe_pose = t->current_frame_pose;
    }
    if (track_status) *track_status = o->status;
    return VORS_OK;
}

vors_status vors_tracker_track_checked(vors_tracker* t, double depth_time, const uint16_t* depth, double img_time, const uint8_t* gray,
                                       int rows, int cols, int* track_status) {
    if (!t) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (rows != t->rows || cols != t->cols)
        return fail(VORS_ERR_INVALID_ARGUMENT, "frame is " + std::to_string(rows) + " x " + std::to_string(cols) + " but the tracker was created for " +
                                                   std::to_string(t->rows) + " x " + std::to_string(t->cols));
    return vors_tracker_track(t, depth_time, depth, img_time, gray, track_status);
}

vors_status vors_tracker_current_frame(const vors_tracker* t, double* timestamp, float pose7[7]) {
    if (!t || !timestamp || !pose7) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    *timestamp = t->current_frame_depth_timestamp;  // the DEPTH timestamp: inverse_compositional.rs:243-247
    iso_store(t->current_frame_pose, pose7);
    return VORS_OK;
}
vors_status vors_tracker_keyframe(const vors_tracker* t, double* timestamp, float pose7[7]) {
    if (!t || !timestamp || !pose7) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    *timestamp = t->keyframe_depth_timestamp;
    iso_store(t->keyframe_pose, pose7);
    return VORS_OK;
}
vors_status vors_tracker_last_stats(const vors_tracker* t, vors_pair_stats* stats) {
    if (!t || !stats) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!t->has_last) return fail(VORS_ERR_INVALID_ARGUMENT, "no frame has been tracked yet");
    *stats = t->last;
    return VORS_OK;
}
void vors_tracker_destroy(vors_tracker* t) {
    if (!t) return;
    DeviceGuard guard(t->device);  // the buffers are freed on the device they live on
    if (t->s_main) (void)hipStreamSynchronize(t->s_main);
    if (t->s_copy) (void)hipStreamSynchronize(t->s_copy);
    delete t;
}

// ---------------------------------------------------------------------------------------------------------------
// N sequences in lock-step, device resident (vors_trackers_*): the state machine of Tracker::track
// (inverse_compositional.rs:170-240) for every sequence without a host round trip — initial guess from the poses kept on the device,
// LM, pose composition, keyframe test, and the promotion of the current frame to keyframe for exactly the sequences whose optical flow
// reached the threshold (masked launches of the keyframe stage over the list trackers_advance_kernel builds).
// ---------------------------------------------------------------------------------------------------------------
}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Throughput mode: a ring of batch handles, each with its own internal stream. A step's tail is latency-bound (straggler rounds of the
// dense LM stage, the tree descent, the last workgroups of the per-pair kernel) and its body VALU- or bandwidth-bound; with the steps
// of a continuous feed alternating between two handles on two streams the GPU fills one with the other.
// ---------------------------------------------------------------------------------------------------------------
struct vors_pipeline {
    int device = 0, depth = 0;
    std::vector<vors_batch*> slot;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> done;      // completion of the last step submitted to the slot
    std::vector<long> ticket_of;       // ticket of that step (-1: none yet)
    hipEvent_t ready = nullptr;        // "the caller's stream has reached the submit" (re-recorded per submit)
    long next = 0;
};
static void pipeline_free(vors_pipeline* p) {
    for (hipStream_t s : p->stream)
        if (s) (void)hipStreamSynchronize(s);  // nothing may still be running on a handle that is about to go
    for (vors_batch* b : p->slot) vors_batch_destroy(b);
    for (hipStream_t s : p->stream)
        if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : p->done)
        if (e) (void)hipEventDestroy(e);
    if (p->ready) (void)hipEventDestroy(p->ready);
    delete p;
}

extern "C" {

vors_status vors_pipeline_create(int device, const vors_config* cfg, int depth, int max_pairs, int rows, int cols, vors_pipeline** out) {
    if (!out) return fail(VORS_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (depth < 1 || depth > 8) return fail(VORS_ERR_INVALID_ARGUMENT, "depth must be 1..8");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) {
            (void)hipGetLastError();
            device = 0;
        }
    }
    vors_pipeline* p = new vors_pipeline();
    p->device = device;
    p->depth = depth;
    struct Guard {
        vors_pipeline* p;
        ~Guard() {
            if (p) pipeline_free(p);
        }
    } cleanup{p};
    for (int k = 0; k < depth; ++k) {
        vors_batch* b = nullptr;
        vors_status st = vors_batch_create_on(device, cfg, max_pairs, rows, cols, &b);
        if (st != VORS_OK) return st;
        // a slot of a ring shares the chip with its neighbours' steps (engine.h Geom::ref_inflight_x2): measured at 512 pairs per step, ring of 3,
        // REFERENCE: coarse-to-fine 0.556 ms per step with the lone step's 5 wavefronts per pair, 0.458 with 4; DSO 0.889 / 0.752
        if (depth >= 2 && !(getenv("VORS_PIPELINE_INFLIGHT") && atoi(getenv("VORS_PIPELINE_INFLIGHT")) == 0)) b->g.ref_inflight_x2 = 3;
        p->slot.push_back(b);
    }
    DeviceGuard on_device(device);
    if (!on_device.ok) return fail(VORS_ERR_HIP, "hipSetDevice failed");
    for (int k = 0; k < depth; ++k) {
        hipStream_t s = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        p->stream.push_back(s);
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->done.push_back(e);
        p->ticket_of.push_back(-1);
    }
    HIP_TRY(hipEventCreateWithFlags(&p->ready, hipEventDisableTiming));
    cleanup.p = nullptr;
    *out = p;
    return VORS_OK;
}

vors_status vors_pipeline_submit(vors_pipeline* p, int n_pairs, const uint8_t* d_kf_gray, const uint16_t* d_kf_depth, const uint8_t* d_cur_gray,
                                 const float* d_prev_poses7, float* d_out_poses7, int32_t* d_out_status, vors_pair_stats* d_out_stats,
                                 void* hip_stream, int64_t* ticket) {
    if (!p) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL handle");
    DeviceGuard on_device(p->device);
    const int k = (int)(p->next % p->depth);
    vors_status st = check_stream(p->slot[k], static_cast<hipStream_t>(hip_stream));
    if (st != VORS_OK) return st;
    // the inputs (and the output buffers' previous readers) are ordered on the caller's stream: the slot's stream waits for it
    HIP_TRY(hipEventRecord(p->ready, static_cast<hipStream_t>(hip_stream)));
    HIP_TRY(hipStreamWaitEvent(p->stream[k], p->ready, 0));
    st = vors_batch_track_pairs(p->slot[k], n_pairs, d_kf_gray, d_kf_depth, d_cur_gray, d_prev_poses7, d_out_poses7, d_out_status, d_out_stats,
                                p->stream[k]);
    // (on failure part of the step may be enqueued: record the event all the same, so that wait / drain cover whatever runs)
    HIP_TRY(hipEventRecord(p->done[k], p->stream[k]));
    p->ticket_of[k] = p->next;
    if (ticket) *ticket = p->next;
    ++p->next;
    return st;
}

vors_status vors_pipeline_wait(vors_pipeline* p, int64_t ticket, void* hip_stream, int host_sync) {
    if (!p) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL handle");
    if (ticket < 0 || ticket >= p->next) return fail(VORS_ERR_INVALID_ARGUMENT, "no such ticket");
    DeviceGuard on_device(p->device);
    const int k = (int)(ticket % p->depth);
    // A slot's stream runs its steps in order: the event of a LATER step of the same slot covers this one too.
    if (host_sync) HIP_TRY(hipEventSynchronize(p->done[k]));
    else HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), p->done[k], 0));
    return VORS_OK;
}

vors_status vors_pipeline_drain(vors_pipeline* p, void* hip_stream, int host_sync) {
    if (!p) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL handle");
    DeviceGuard on_device(p->device);
    for (int k = 0; k < p->depth; ++k) {
        if (p->ticket_of[k] < 0) continue;
        if (host_sync) HIP_TRY(hipEventSynchronize(p->done[k]));
        else HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), p->done[k], 0));
    }
    return VORS_OK;
}

void vors_pipeline_destroy(vors_pipeline* p) {
    if (!p) return;
    DeviceGuard on_device(p->device);
    pipeline_free(p);
}

}  // extern "C"

struct vors_trackers {
    vors_batch* batch = nullptr;
    int n_seq = 0;
    int frame_index = 0;  // index of the last frame submitted (0 = the init frame)
    bool initialised = false;
    DevBuf cur_poses, kf_poses, out_poses, status, stats, kf_frame, promo_list, promo_count, frame_counter;
    DevBuf own_gray, own_depth;  // dense mode: the keyframes' level 0 and depth maps (re-read by every evaluation) live in the handle
    ~vors_trackers() { vors_batch_destroy(batch); }
};

extern "C" {

vors_status vors_trackers_create(const vors_config* cfg, int n_sequences, int rows, int cols, vors_trackers** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        dev = 0;
    }
    return vors_trackers_create_on(dev, cfg, n_sequences, rows, cols, out);
}

vors_status vors_trackers_create_on(int device, const vors_config* cfg, int n_sequences, int rows, int cols, vors_trackers** out) {
    if (!out) return fail(VORS_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    vors_batch* b = nullptr;
    vors_status st = vors_batch_create_on(device, cfg, n_sequences, rows, cols, &b);
    if (st != VORS_OK) return st;
    DeviceGuard on_device(device);  // the state buffers live on the handle's device
    vors_trackers* t = new vors_trackers();
    t->batch = b;
    t->n_seq = n_sequences;
    struct Guard {
        vors_trackers* t;
        ~Guard() { delete t; }
    } guard{t};
    const size_t n = (size_t)n_sequences, S = (size_t)rows * cols;
    HIP_TRY(t->cur_poses.alloc(n * 7 * sizeof(float)));
    HIP_TRY(t->kf_poses.alloc(n * 7 * sizeof(float)));
    HIP_TRY(t->out_poses.alloc(n * 7 * sizeof(float)));
    HIP_TRY(t->status.alloc(n * sizeof(int32_t)));
    HIP_TRY(t->stats.alloc(n * sizeof(vors_pair_stats)));
    HIP_TRY(t->kf_frame.alloc(n * sizeof(int32_t)));
    HIP_TRY(t->promo_list.alloc(n * sizeof(int)));
    HIP_TRY(t->promo_count.alloc(sizeof(int)));
    HIP_TRY(t->frame_counter.alloc(sizeof(int)));
    if (b->g.mode == VORS_CANDIDATES_DENSE) {
        HIP_TRY(t->own_gray.alloc(n * S));
        HIP_TRY(t->own_depth.alloc(n * S * 2));
    }
    guard.t = nullptr;
    *out = t;
    return VORS_OK;
}

void vors_trackers_destroy(vors_trackers* t) {
    if (!t) return;
    DeviceGuard guard(t->batch ? t->batch->device : 0);
    delete t;
}

int vors_trackers_count(const vors_trackers* t) { return t ? t->n_seq : 0; }

vors_status vors_trackers_init(vors_trackers* t, const uint8_t* d_gray, const uint16_t* d_depth, void* hip_stream) {
    if (!t || !d_gray || !d_depth) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    vors_batch* b = t->batch;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    DeviceGuard guard(b->device);
    vors_status st = check_stream(b, s);
    if (st != VORS_OK) return st;
    const size_t n = (size_t)t->n_seq, S = (size_t)b->g.S0;
    const uint8_t* kf_gray = d_gray;
    const uint16_t* kf_depth = d_depth;
    if (b->g.mode == VORS_CANDIDATES_DENSE) {  // the handle's own copies (zero copy is impossible: keyframes outlive the caller's frames)
        HIP_TRY(hipMemcpyAsync(t->own_gray.p, d_gray, n * S, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(t->own_depth.p, d_depth, n * S * 2, hipMemcpyDeviceToDevice, s));
        kf_gray = t->own_gray.as<uint8_t>();
        kf_depth = t->own_depth.as<uint16_t>();
    }
    st = vors_batch_prepare_keyframes(b, t->n_seq, kf_gray, kf_depth, s);
    if (st != VORS_OK) return st;
    if (b->g.mode != VORS_CANDIDATES_DENSE) {
        // Sparse modes: everything later stages need is in the records; the caller may reuse or free its frames, and keyframe promotion
        // (trackers_promote) rebuilds the records from later frames. The handle must not keep pointers into frames it does not own:
        // the keyframe-inspection entry points (vors_batch_get_keyframe_image / get_points) are not available on a trackers-owned batch.
        b->kf_level0 = nullptr;
        b->kf_depth = nullptr;
    }
    // first frame: keyframe_pose = current_frame_pose = identity (inverse_compositional.rs:86-99)
    launch_identity_poses(t->cur_poses.as<float>(), t->kf_poses.as<float>(), t->n_seq, s);  // (on the device: init only enqueues work, like track)
    HIP_TRY(hipMemsetAsync(t->kf_frame.p, 0, n * sizeof(int32_t), s));
    HIP_TRY(hipMemsetAsync(t->status.p, 0, n * sizeof(int32_t), s));
    HIP_TRY(hipMemsetAsync(t->frame_counter.p, 0, sizeof(int), s));
    HIP_TRY(hipGetLastError());
    t->frame_index = 0;
    t->initialised = true;
    return VORS_OK;
}

vors_status vors_trackers_track(vors_trackers* t, const uint8_t* d_gray, const uint16_t* d_depth, void* hip_stream) {
    if (!d_depth) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    vors_status st = trackers_track_lm(t, d_gray, s);
    return st != VORS_OK ? st : trackers_promote(t, d_gray, d_depth, nullptr, s);
}

static vors_status trackers_track_lm(vors_trackers* t, const uint8_t* d_gray, hipStream_t s) {
    if (!t || !d_gray) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!t->initialised) return fail(VORS_ERR_INVALID_ARGUMENT, "vors_trackers_track called before vors_trackers_init");
    vors_batch* b = t->batch;
    DeviceGuard guard(b->device);
    vors_status st = check_stream(b, s);
    if (st != VORS_OK) return st;
    const int n = t->n_seq;
    t->frame_index += 1;
    // Tracker::track up to the keyframe test (inverse_compositional.rs:177-224), all sequences
    st = batch_track_current(b, n, d_gray, t->cur_poses.as<float>(), t->kf_poses.as<float>(), t->out_poses.as<float>(), t->status.as<int32_t>(),
                             t->stats.as<vors_pair_stats>(), s);
    if (st != VORS_OK) return st;
    // :203-208 and :224-239 on the device: poses forward, promotion list
    launch_trackers_advance(n, t->frame_counter.as<int>(), t->out_poses.as<float>(), t->stats.as<vors_pair_stats>(), t->cur_poses.as<float>(),
                            t->kf_poses.as<float>(), t->kf_frame.as<int32_t>(), t->promo_list.as<int>(), t->promo_count.as<int>(), s);
    HIP_TRY(hipGetLastError());
    return VORS_OK;
}

// depth_ready (nullable): an event after which d_depth holds this frame's depth map (uploaded on another stream).
static vors_status trackers_promote(vors_trackers* t, const uint8_t* d_gray, const uint16_t* d_depth, hipEvent_t depth_ready, hipStream_t s) {
    vors_batch* b = t->batch;
    DeviceGuard guard(b->device);
    const int n = t->n_seq;
    // precompute_multires_data (:230-235) for the promoted sequences only: the pyramid of the current frame is reused, the depth map is
    // the one that came with it
    Geom gm = b->g;
    gm.sel_list = t->promo_list.as<int>();
    gm.sel_count = t->promo_count.as<int>();
    if (depth_ready) HIP_TRY(hipStreamWaitEvent(s, depth_ready, 0));
    STAGE_BEGIN(b, 1, s);
    if (b->g.mode == VORS_CANDIDATES_DENSE) {
        const size_t S = (size_t)b->g.S0;
        launch_promote_copy(gm, d_gray, S, t->own_gray.p, S, S, n, s);
        launch_promote_copy(gm, d_depth, 2 * S, t->own_depth.p, 2 * S, 2 * S, n, s);
        launch_promote_copy(gm, b->cur_upper, (size_t)b->g.upper_stride, b->kf_upper, (size_t)b->g.upper_stride, (size_t)b->g.upper_stride, n, s);
        launch_keyframe(gm, Pyramid{t->own_gray.as<uint8_t>(), b->kf_upper}, t->own_depth.as<uint16_t>(), b->rec, n, s);
    } else if (b->g.mode == VORS_CANDIDATES_DSO) {
        launch_keyframe_dso(gm, Pyramid{d_gray, b->cur_upper}, d_depth, b->dso, b->mask0, b->pp, b->rec, n, s);
    } else {
        launch_keyframe(gm, Pyramid{d_gray, b->cur_upper}, d_depth, b->rec, n, s);
    }
    if (b->g.arith == VORS_ARITH_REFERENCE) {
        launch_sort_colmajor(gm, b->rec, n, s);
        if (b->g.mode == VORS_CANDIDATES_DENSE)
            launch_ref_dense_planes_keyframe(gm, Pyramid{t->own_gray.as<uint8_t>(), b->kf_upper}, t->own_depth.as<uint16_t>(), b->rec, n, s);
    }
    STAGE_END(b, 1, s);
    HIP_TRY(hipGetLastError());
    return VORS_OK;
}

vors_status vors_trackers_state(const vors_trackers* t, const float** d_current_poses7, const float** d_keyframe_poses7,
                                const int32_t** d_status, const int32_t** d_keyframe_index, const vors_pair_stats** d_stats) {
    if (!t) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL handle");
    if (d_current_poses7) *d_current_poses7 = static_cast<const float*>(t->cur_poses.p);
    if (d_keyframe_poses7) *d_keyframe_poses7 = static_cast<const float*>(t->kf_poses.p);
    if (d_status) *d_status = static_cast<const int32_t*>(t->status.p);
    if (d_keyframe_index) *d_keyframe_index = static_cast<const int32_t*>(t->kf_frame.p);
    if (d_stats) *d_stats = static_cast<const vors_pair_stats*>(t->stats.p);
    return VORS_OK;
}

vors_status vors_trackers_current_frames(vors_trackers* t, float* poses7, int32_t* status, int32_t* keyframe_index, void* hip_stream) {
    if (!t) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL handle");
    if (!t->initialised) return fail(VORS_ERR_INVALID_ARGUMENT, "vors_trackers_current_frames called before vors_trackers_init");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    DeviceGuard guard(t->batch->device);
    const size_t n = (size_t)t->n_seq;
    if (poses7) HIP_TRY(hipMemcpyAsync(poses7, t->cur_poses.p, n * 7 * sizeof(float), hipMemcpyDeviceToHost, s));
    if (status) HIP_TRY(hipMemcpyAsync(status, t->status.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (keyframe_index) HIP_TRY(hipMemcpyAsync(keyframe_index, t->kf_frame.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return VORS_OK;
}

vors_status vors_trackers_last_stats(vors_trackers* t, vors_pair_stats* stats, void* hip_stream) {
    if (!t || !stats) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (t->frame_index < 1) return fail(VORS_ERR_INVALID_ARGUMENT, "no frame has been tracked yet");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    DeviceGuard guard(t->batch->device);
    HIP_TRY(hipMemcpyAsync(stats, t->stats.p, (size_t)t->n_seq * sizeof(vors_pair_stats), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return VORS_OK;
}

vors_status vors_trackers_enable_kernel_timing(vors_trackers* t, int ring) {
    if (!t) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL handle");
    return vors_batch_enable_kernel_timing(t->batch, ring);
}
vors_status vors_trackers_kernel_times(vors_trackers* t, int stage, float* ms_out, int capacity, int* n_out) {
    if (!t) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL handle");
    return vors_batch_kernel_times(t->batch, stage, ms_out, capacity, n_out);
}

// ---------------------------------------------------------------------------------------------------------------
// operator level
// ---------------------------------------------------------------------------------------------------------------
struct ObsDev {
    DevBuf tmpl, img, xy, iz, jac, A, B, C, XY, IZ, model, out, res;
    Records rec{};
    Intr k;
};
static vors_status upload_obs(const vors_obs* o, const float model7[7], ObsDev& d, bool want_res, hipStream_t s) {
    if (!o || !model7) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (o->rows < 2 || o->cols < 2 || o->n < 0) return fail(VORS_ERR_INVALID_ARGUMENT, "bad observation shape");
    if (o->arithmetic != VORS_ARITH_EXACT && o->arithmetic != VORS_ARITH_REFERENCE)
        return fail(VORS_ERR_INVALID_ARGUMENT, "vors_obs.arithmetic must be VORS_ARITH_EXACT or VORS_ARITH_REFERENCE");
    if (!o->template_ || !o->image || (o->n > 0 && (!o->coordinates || !o->_z_candidates || !o->jacobians)))
        return fail(VORS_ERR_INVALID_ARGUMENT, "NULL observation array");
    vors_status st = require_device();
    if (st != VORS_OK) return st;
    const size_t S = (size_t)o->rows * o->cols, n = (size_t)o->n;
    for (size_t i = 0; i < n; ++i) {
        const int x = o->coordinates[2 * i], y = o->coordinates[2 * i + 1];
        if (x < 0 || y < 0 || x >= o->cols || y >= o->rows) return fail(VORS_ERR_INVALID_ARGUMENT, "coordinate outside the template");
    }
    HIP_TRY(d.tmpl.alloc(S));
    HIP_TRY(d.img.alloc(S));
    HIP_TRY(d.xy.alloc(n * 8));
    HIP_TRY(d.iz.alloc(n * 4));
    HIP_TRY(d.jac.alloc(n * 24));
    HIP_TRY(d.A.alloc(n * 16));
    HIP_TRY(d.B.alloc(n * 16));
    HIP_TRY(d.C.alloc(n * 8));
    HIP_TRY(d.XY.alloc(n * 4));
    HIP_TRY(d.IZ.alloc(n * 4));
    HIP_TRY(d.model.alloc(7 * 4));
    HIP_TRY(d.out.alloc(64 * 4));
    if (want_res) HIP_TRY(d.res.alloc(n * 4));
    HIP_TRY(hipMemcpyAsync(d.tmpl.p, o->template_, S, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d.img.p, o->image, S, hipMemcpyHostToDevice, s));
    if (n) {
        HIP_TRY(hipMemcpyAsync(d.xy.p, o->coordinates, n * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d.iz.p, o->_z_candidates, n * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d.jac.p, o->jacobians, n * 24, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(d.model.p, model7, 28, hipMemcpyHostToDevice, s));
    d.k = Intr{o->cu, o->cv, o->fu, o->fv, o->skew};
    d.rec = Records{d.A.as<float4>(), d.B.as<float4>(), d.C.as<float2>(), d.XY.as<uint32_t>(), d.IZ.as<float>(), nullptr, nullptr, nullptr};
    launch_records_from_obs(d.k, o->rows, o->cols, d.tmpl.as<uint8_t>(), o->n, d.xy.as<int32_t>(), d.iz.as<float>(),
                            d.jac.as<float>(), d.rec, s);
    return VORS_OK;
}

vors_status vors_lm_eval(const vors_obs* obs, const float model7[7], float* energy, int32_t* n_inside, float g[6], float H[36],
                         float* residuals) {
    if (!energy || !n_inside || !g || !H) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL output");
    ObsDev d;
    hipStream_t s = nullptr;
    vors_status st = upload_obs(obs, model7, d, residuals != nullptr, s);
    if (st != VORS_OK) return st;
    if (obs->arithmetic == VORS_ARITH_REFERENCE)
        launch_lm_eval_obs_reference(d.k, obs->rows, obs->cols, d.img.as<uint8_t>(), obs->n, d.rec, obs->huber_delta, d.model.as<float>(),
                                     d.out.as<float>(), residuals ? d.res.as<float>() : nullptr, s);
    else
        launch_lm_eval_obs(d.k, obs->rows, obs->cols, d.img.as<uint8_t>(), obs->n, d.rec, obs->huber_delta, d.model.as<float>(),
                           d.out.as<float>(), residuals ? d.res.as<float>() : nullptr, s);
    float out[44];
    HIP_TRY(hipMemcpyAsync(out, d.out.p, sizeof(out), hipMemcpyDeviceToHost, s));
    if (residuals && obs->n) HIP_TRY(hipMemcpyAsync(residuals, d.res.p, (size_t)obs->n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    *energy = out[0];
    *n_inside = (int32_t)out[1];
    std::memcpy(g, out + 2, 24);
    std::memcpy(H, out + 8, 144);
    return VORS_OK;
}

vors_status vors_lm_solve(const vors_obs* obs, const float model7[7], float out_model7[7], int32_t* nb_iter, float* energy,
                          float* lm_coef, int* solve_status) {
    if (!out_model7 || !nb_iter || !energy || !lm_coef || !solve_status) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL output");
    ObsDev d;
    hipStream_t s = nullptr;
    vors_status st = upload_obs(obs, model7, d, false, s);
    if (st != VORS_OK) return st;
    if (obs->arithmetic == VORS_ARITH_REFERENCE)
        launch_lm_solve_obs_reference(d.k, obs->rows, obs->cols, d.img.as<uint8_t>(), obs->n, d.rec, obs->huber_delta, d.model.as<float>(),
                                      d.out.as<float>(), s);
    else
        launch_lm_solve_obs(d.k, obs->rows, obs->cols, d.img.as<uint8_t>(), obs->n, d.rec, obs->huber_delta, d.model.as<float>(),
                            d.out.as<float>(), s);
    float out[11];
    HIP_TRY(hipMemcpyAsync(out, d.out.p, sizeof(out), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    std::memcpy(out_model7, out, 28);
    *nb_iter = (int32_t)out[7];
    *energy = out[8];
    *lm_coef = out[9];
    *solve_status = out[10] != 0.f ? VORS_TRACK_OPTIMIZER_FAILED_POSE_KEPT : VORS_TRACK_OK;
    return VORS_OK;
}

vors_status vors_lm_step(const float H[36], const float g[6], const float model7[7], float lm_coef, float out_model7[7],
                         int* chol_ok) {
    if (!H || !g || !model7 || !out_model7 || !chol_ok) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    Iso out;
    const bool ok = lm_step(H, g, iso_load(model7), lm_coef, &out);
    *chol_ok = ok ? 1 : 0;
    if (ok) iso_store(out, out_model7);
    return VORS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Lie helpers (host arithmetic)
// ---------------------------------------------------------------------------------------------------------------
void vors_se3_exp(const float xi[6], float out_iso7[7]) { iso_store(se3_exp(xi), out_iso7); }
void vors_ref_sincos(const float* x, int n, float* sin_out, float* cos_out) {
    for (int i = 0; i < n; ++i) {
        if (sin_out) sin_out[i] = ref_sinf(x[i]);
        if (cos_out) cos_out[i] = ref_cosf(x[i]);
    }
}
void vors_se3_log(const float iso7[7], float out_xi[6]) { se3_log(iso_load(iso7), out_xi); }
void vors_so3_exp(const float w[3], float out_q4[4]) {
    const Quat q = so3_exp(w);
    out_q4[0] = q.i; out_q4[1] = q.j; out_q4[2] = q.k; out_q4[3] = q.w;
}
void vors_so3_log(const float q4[4], float out_w[3]) { so3_log(Quat{q4[0], q4[1], q4[2], q4[3]}, out_w); }
void vors_iso_mul(const float a7[7], const float b7[7], float out7[7]) { iso_store(iso_mul(iso_load(a7), iso_load(b7)), out7); }
void vors_iso_inverse(const float a7[7], float out7[7]) { iso_store(iso_inverse(iso_load(a7)), out7); }

// ---------------------------------------------------------------------------------------------------------------
// synthetic scenes
// ---------------------------------------------------------------------------------------------------------------
vors_status vors_synth_render_pairs(uint64_t seed0, int n_pairs, int rows, int cols, const double cam5[5], double motion_scale,
                                    int invalid_percent, uint8_t* d_kf_gray, uint16_t* d_kf_depth, uint8_t* d_cur_gray,
                                    uint16_t* d_cur_depth, float* d_gt_models7, void* hip_stream) {
    if (!cam5 || !d_kf_gray || !d_kf_depth || !d_cur_gray) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_pairs < 1 || rows < 1 || cols < 1) return fail(VORS_ERR_INVALID_ARGUMENT, "bad shape");
    vors_status st = require_device();
    if (st != VORS_OK) return st;
    launch_synth_pairs(seed0, n_pairs, rows, cols, cam5, motion_scale, invalid_percent, d_kf_gray, d_kf_depth, d_cur_gray,
                       d_cur_depth, d_gt_models7, static_cast<hipStream_t>(hip_stream));
    HIP_TRY(hipGetLastError());
    return VORS_OK;
}

vors_status vors_synth_render_frames(int n_frames, const uint64_t* seeds, const uint64_t* salts, const double* xi6, int rows, int cols,
                                     const double cam5[5], int invalid_percent, uint8_t* d_gray, uint16_t* d_depth, void* hip_stream) {
    if (!seeds || !salts || !xi6 || !cam5 || !d_gray || !d_depth) return fail(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_frames < 1 || rows < 1 || cols < 1) return fail(VORS_ERR_INVALID_ARGUMENT, "bad shape");
    vors_status st = require_device();
    if (st != VORS_OK) return st;
    struct Frame {
        uint64_t seed, salt;
        double xi[6];
    };
    std::vector<Frame> h((size_t)n_frames);
    for (int f = 0; f < n_frames; ++f) {
        h[f].seed = seeds[f];
        h[f].salt = salts[f];
        for (int q = 0; q < 6; ++q) h[f].xi[q] = xi6[6 * f + q];
    }
    DevBuf d;
    HIP_TRY(d.alloc(h.size() * sizeof(Frame)));
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(Frame), hipMemcpyHostToDevice, s));
    launch_synth_frames(d.p, n_frames, rows, cols, cam5, invalid_percent, d_gray, d_depth, s);
    HIP_TRY(hipStreamSynchronize(s));  // (the table is freed on return)
    HIP_TRY(hipGetLastError());
    return VORS_OK;
}

}  // extern "C"
