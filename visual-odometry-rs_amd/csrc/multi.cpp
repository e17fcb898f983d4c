// Multi-GPU entry points of the C ABI (include/vors_hip.h §2b): ONE process drives several devices.
//
// Frame pairs are independent (reference: a Tracker is self-contained, src/core/track/inverse_compositional.rs:31-34), so a batch
// shards by contiguous blocks of pairs, one vors_batch per device on its own stream, no data-path exchange; the only collective is ONE
// all-gather of 8 f32 per pair (pose 7 + status) — RCCL ncclAllGather over xGMI (16 KiB per GPU for 4096 pairs on 8 GPUs: far below
// any link bound) — after which every device holds all results. This is what a Rust host without torch calls; bench.py's N > 1 path
// does the same thing with one process per GPU through torch.distributed (backend "nccl" = RCCL).
//
// RCCL is bound at run time (dlopen "librccl.so.1" / "librccl.so"): the library has no link-time dependency on it, single-device use
// never touches it, and a process that already loaded an RCCL (e.g. through torch) shares that one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/vors_hip.h"

extern "C" const char* vors_last_error(void);
vors_status vors_set_last_error(vors_status st, const std::string& msg);  // capi.cpp

namespace {

typedef struct ncclComm* ncclComm_t;  // opaque (rccl.h)
constexpr int kNcclFloat = 7;         // ncclFloat32 (rccl.h:466)
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    bool load(std::string* err) {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) {
            *err = std::string("cannot load RCCL (librccl.so): ") + dlerror();
            return false;
        }
#define VORS_SYM(field, sym)                                              \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, sym));           \
    if (!field) {                                                         \
        *err = std::string("RCCL lacks symbol ") + sym;                   \
        return false;                                                     \
    }
        VORS_SYM(CommInitAll, "ncclCommInitAll")
        VORS_SYM(CommDestroy, "ncclCommDestroy")
        VORS_SYM(AllGather, "ncclAllGather")
        VORS_SYM(GroupStart, "ncclGroupStart")
        VORS_SYM(GroupEnd, "ncclGroupEnd")
        VORS_SYM(GetErrorString, "ncclGetErrorString")
        VORS_SYM(GetVersion, "ncclGetVersion")
#undef VORS_SYM
        return true;
    }
};
Rccl g_rccl;

struct Shard {
    int device = 0;
    vors_batch* batch = nullptr;
    hipStream_t stream = nullptr;
    float* poses = nullptr;     // [per, 7]
    int32_t* status = nullptr;  // [per]
    float* pack = nullptr;      // [per, 8]  pose 7 + status bits
    float* all = nullptr;       // [n_dev * per, 8]
    uint8_t *kf = nullptr, *cur = nullptr;  // staging for the host-buffer entry
    uint16_t* depth = nullptr;
    ncclComm_t comm = nullptr;
};

}  // namespace

struct vors_multi {
    vors_config cfg{};
    int rows = 0, cols = 0, per = 0;  // per = max pairs per device
    std::vector<Shard> sh;
    bool staged = false;
};

#define MHIP(expr)                                                                                       \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) return vors_set_last_error(VORS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

extern "C" {

void vors_multi_destroy(vors_multi* m) {
    if (!m) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (Shard& s : m->sh) {
        (void)hipSetDevice(s.device);
        if (s.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(s.comm);
        vors_batch_destroy(s.batch);
        for (void* p : {(void*)s.poses, (void*)s.status, (void*)s.pack, (void*)s.all, (void*)s.kf, (void*)s.cur, (void*)s.depth})
            if (p) (void)hipFree(p);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    (void)hipSetDevice(prev);
    delete m;
}

vors_status vors_multi_create(const vors_config* cfg, int n_devices, const int* device_ids, int max_pairs_per_device, int rows, int cols,
                              vors_multi** out) {
    if (!out) return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!cfg || max_pairs_per_device < 1) return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "bad argument");
    const int visible = vors_device_count();
    if (visible < 1) return vors_set_last_error(VORS_ERR_NO_DEVICE, "vors_hip: no HIP device available (this library has no CPU fallback)");
    if (n_devices <= 0) n_devices = visible;
    if (n_devices > visible) return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "more devices requested than visible");
    vors_multi* m = new vors_multi();
    m->cfg = *cfg;
    m->rows = rows;
    m->cols = cols;
    m->per = max_pairs_per_device;
    m->sh.resize(n_devices);
    int prev = 0;
    (void)hipGetDevice(&prev);
    struct Guard {
        vors_multi* m;
        int prev;
        ~Guard() {
            if (m) vors_multi_destroy(m);
            (void)hipSetDevice(prev);
        }
    } guard{m, prev};
    std::vector<int> ids(n_devices);
    for (int k = 0; k < n_devices; ++k) {
        ids[k] = device_ids ? device_ids[k] : k;
        if (ids[k] < 0 || ids[k] >= visible) return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "device id out of range");
        for (int j = 0; j < k; ++j)
            if (ids[j] == ids[k]) return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "duplicate device id");
    }
    const size_t per = (size_t)max_pairs_per_device;
    for (int k = 0; k < n_devices; ++k) {
        Shard& s = m->sh[k];
        s.device = ids[k];
        MHIP(hipSetDevice(s.device));
        vors_status st = vors_batch_create_on(s.device, cfg, max_pairs_per_device, rows, cols, &s.batch);
        if (st != VORS_OK) return st;
        MHIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        MHIP(hipMalloc(reinterpret_cast<void**>(&s.poses), per * 7 * sizeof(float)));
        MHIP(hipMalloc(reinterpret_cast<void**>(&s.status), per * sizeof(int32_t)));
        MHIP(hipMalloc(reinterpret_cast<void**>(&s.pack), per * 8 * sizeof(float)));
        MHIP(hipMalloc(reinterpret_cast<void**>(&s.all), per * 8 * sizeof(float) * n_devices));
    }
    if (n_devices > 1) {
        std::string err;
        if (!g_rccl.load(&err)) return vors_set_last_error(VORS_ERR_UNSUPPORTED, err);
        std::vector<ncclComm_t> comms(n_devices);
        const int rc = g_rccl.CommInitAll(comms.data(), n_devices, ids.data());
        if (rc != 0) return vors_set_last_error(VORS_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc));
        for (int k = 0; k < n_devices; ++k) m->sh[k].comm = comms[k];
    }
    guard.m = nullptr;
    *out = m;
    return VORS_OK;
}

int vors_multi_device_count(const vors_multi* m) { return m ? (int)m->sh.size() : 0; }

int vors_multi_rccl_version(const vors_multi* m) {
    int v = 0;
    if (!m || m->sh.size() < 2 || !g_rccl.GetVersion || g_rccl.GetVersion(&v) != 0) return 0;
    return v;
}

vors_status vors_multi_shard(const vors_multi* m, int n_pairs_total, int k, int* first, int* count) {
    if (!m || k < 0 || k >= (int)m->sh.size() || !first || !count) return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "bad argument");
    const int nd = (int)m->sh.size(), per = (n_pairs_total + nd - 1) / nd;  // pair i -> device floor(i / ceil(n / G))
    const int lo = std::min(k * per, n_pairs_total), hi = std::min(lo + per, n_pairs_total);
    *first = lo;
    *count = hi - lo;
    return VORS_OK;
}

vors_status vors_multi_track_pairs(vors_multi* m, int n_pairs_total, const uint8_t* const* d_kf_gray, const uint16_t* const* d_kf_depth,
                                   const uint8_t* const* d_cur_gray, float* out_poses7, int32_t* out_status) {
    if (!m || !d_kf_gray || !d_kf_depth || !d_cur_gray || !out_poses7 || !out_status)
        return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    const int nd = (int)m->sh.size();
    const int per = (n_pairs_total + nd - 1) / nd;
    if (n_pairs_total < 1 || per > m->per) return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "n_pairs_total out of range for this handle");
    int prev = 0;
    (void)hipGetDevice(&prev);
    // Whatever happens below, no stream of the handle has work in flight when this call returns: on an error path the other devices
    // may already be running kernels that read the caller's buffers, so every shard stream is drained before the error is reported.
    struct Restore {
        vors_multi* m;
        int d;
        ~Restore() {
            for (Shard& s : m->sh)
                if (hipSetDevice(s.device) == hipSuccess) (void)hipStreamSynchronize(s.stream);
            (void)hipSetDevice(d);
        }
    } restore{m, prev};
    // 1. every device tracks its block (concurrently: one stream per device, nothing is synchronised here)
    for (int k = 0; k < nd; ++k) {
        Shard& s = m->sh[k];
        int lo, cnt;
        (void)vors_multi_shard(m, n_pairs_total, k, &lo, &cnt);
        MHIP(hipSetDevice(s.device));
        MHIP(hipMemsetAsync(s.pack, 0xff, (size_t)per * 8 * sizeof(float), s.stream));  // padding rows: status bits = -1
        if (cnt > 0) {
            if (!d_kf_gray[k] || !d_kf_depth[k] || !d_cur_gray[k]) return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "NULL shard pointer");
            vors_status st = vors_batch_track_pairs(s.batch, cnt, d_kf_gray[k], d_kf_depth[k], d_cur_gray[k], nullptr, s.poses, s.status, nullptr, s.stream);
            if (st != VORS_OK) return st;
            // pack (pose 7 | status bits) -> 8 f32 per pair
            MHIP(hipMemcpy2DAsync(s.pack, 32, s.poses, 28, 28, (size_t)cnt, hipMemcpyDeviceToDevice, s.stream));
            MHIP(hipMemcpy2DAsync(reinterpret_cast<char*>(s.pack) + 28, 32, s.status, 4, 4, (size_t)cnt, hipMemcpyDeviceToDevice, s.stream));
        }
    }
    // 2. the single collective: all-gather of per * 8 f32 per device
    if (nd > 1) {
        int rc = g_rccl.GroupStart();
        for (int k = 0; k < nd && rc == 0; ++k) {
            Shard& s = m->sh[k];
            rc = g_rccl.AllGather(s.pack, s.all, (size_t)per * 8, kNcclFloat, s.comm, s.stream);
        }
        const int rc2 = g_rccl.GroupEnd();
        if (rc != 0 || rc2 != 0) return vors_set_last_error(VORS_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(rc ? rc : rc2));
    } else {
        Shard& s = m->sh[0];
        MHIP(hipSetDevice(s.device));
        MHIP(hipMemcpyAsync(s.all, s.pack, (size_t)per * 8 * sizeof(float), hipMemcpyDeviceToDevice, s.stream));
    }
    // 3. results to the host from device 0 (every device holds them all); trim the padding
    std::vector<float> host((size_t)nd * per * 8);
    MHIP(hipSetDevice(m->sh[0].device));
    MHIP(hipMemcpyAsync(host.data(), m->sh[0].all, host.size() * sizeof(float), hipMemcpyDeviceToHost, m->sh[0].stream));
    for (int k = 0; k < nd; ++k) {
        MHIP(hipSetDevice(m->sh[k].device));
        MHIP(hipStreamSynchronize(m->sh[k].stream));
    }
    for (int k = 0; k < nd; ++k) {
        int lo, cnt;
        (void)vors_multi_shard(m, n_pairs_total, k, &lo, &cnt);
        for (int i = 0; i < cnt; ++i) {
            const float* src = &host[((size_t)k * per + i) * 8];
            for (int q = 0; q < 7; ++q) out_poses7[(size_t)(lo + i) * 7 + q] = src[q];
            int32_t st;
            __builtin_memcpy(&st, src + 7, 4);
            out_status[lo + i] = st;
        }
    }
    return VORS_OK;
}

vors_status vors_multi_track_pairs_host(vors_multi* m, int n_pairs_total, const uint8_t* kf_gray, const uint16_t* kf_depth,
                                        const uint8_t* cur_gray, float* out_poses7, int32_t* out_status) {
    if (!m || !kf_gray || !kf_depth || !cur_gray) return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "NULL argument");
    const int nd = (int)m->sh.size();
    const size_t S = (size_t)m->rows * m->cols;
    int prev = 0;
    (void)hipGetDevice(&prev);
    struct Restore {
        int d;
        ~Restore() { (void)hipSetDevice(d); }
    } restore{prev};
    std::vector<const uint8_t*> kf(nd), cur(nd);
    std::vector<const uint16_t*> dep(nd);
    for (int k = 0; k < nd; ++k) {
        Shard& s = m->sh[k];
        int lo, cnt;
        vors_status st = vors_multi_shard(m, n_pairs_total, k, &lo, &cnt);
        if (st != VORS_OK) return st;
        if (cnt > m->per) return vors_set_last_error(VORS_ERR_INVALID_ARGUMENT, "n_pairs_total out of range for this handle");
        MHIP(hipSetDevice(s.device));
        if (!s.kf) {
            MHIP(hipMalloc(reinterpret_cast<void**>(&s.kf), (size_t)m->per * S));
            MHIP(hipMalloc(reinterpret_cast<void**>(&s.cur), (size_t)m->per * S));
            MHIP(hipMalloc(reinterpret_cast<void**>(&s.depth), (size_t)m->per * S * 2));
        }
        if (cnt > 0) {
            MHIP(hipMemcpyAsync(s.kf, kf_gray + (size_t)lo * S, (size_t)cnt * S, hipMemcpyHostToDevice, s.stream));
            MHIP(hipMemcpyAsync(s.depth, kf_depth + (size_t)lo * S, (size_t)cnt * S * 2, hipMemcpyHostToDevice, s.stream));
            MHIP(hipMemcpyAsync(s.cur, cur_gray + (size_t)lo * S, (size_t)cnt * S, hipMemcpyHostToDevice, s.stream));
        }
        kf[k] = s.kf;
        cur[k] = s.cur;
        dep[k] = s.depth;
    }
    return vors_multi_track_pairs(m, n_pairs_total, kf.data(), dep.data(), cur.data(), out_poses7, out_status);
}

}  // extern "C"
