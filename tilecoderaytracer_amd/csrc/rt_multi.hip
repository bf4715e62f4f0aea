/*
 * rt_multi.hip -- rt_render_multi(): the reference's static partitioning
 * (PARTIONING_STRATEGY 1, src/RayTracer.cpp:904-923 with CORE_NUM > 1: each
 * rank renders one contiguous strip and the strips meet in shared memory) done
 * across the GPUs of one node from a single process.
 *
 * Partition: contiguous x-strips.  The framebuffer is x-major
 * (pixels[x][z], src/RayTracer.h:44), so strip g is one contiguous block and
 * rank order equals memory order: ncclGather(root 0) drops every strip in
 * place, no repacking.  dx/dz are computed from the global x and W, H, so the
 * gathered image is bit-identical to a single-GPU render.
 *
 * RCCL is bound lazily (dlopen of librccl.so) so that the single-GPU entry
 * points carry no RCCL dependency; per-process multi-GPU (bench.py, one rank
 * per GPU) uses torch.distributed's RCCL instead and never calls this.
 */
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/rt_capi.h"

namespace {

typedef void *ncclComm_t;
typedef int ncclResult_t;            /* ncclSuccess == 0                       */
enum { kNcclFloat = 7 };             /* ncclFloat32, rccl.h ncclDataType_t     */

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

thread_local std::string g_multi_error;

bool load_rccl(Rccl &r, std::string &err) {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) { err = std::string("dlopen librccl: ") + dlerror(); return false; }
#define BIND(field, sym)                                                              \
    do {                                                                              \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym));          \
        if (!r.field) { err = std::string("dlsym ") + sym + " failed"; return false; } \
    } while (0)
    BIND(CommInitAll, "ncclCommInitAll");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(Gather, "ncclGather");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    return true;
}

} // namespace

/* defined in rt_capi.hip: stores the message for rt_last_error() */
extern "C" int rt_internal_set_error(int code, const char *msg);

extern "C" int rt_strip_bounds(int W, int ngpu, int g, int *x0, int *x1) {
    if (W <= 0 || ngpu <= 0 || g < 0 || g >= ngpu) return 0;
    const int strip = (W + ngpu - 1) / ngpu;
    if (x0) *x0 = (long long)g * strip < W ? g * strip : W;
    if (x1) *x1 = ((long long)g + 1) * strip < W ? (g + 1) * strip : W;
    return strip;
}

extern "C" int rt_render_multi(const rt_scene_desc *desc, const rt_camera_desc *cam, int W, int H,
                               int max_depth, int ngpu, float *out_rgb) {
    if (!desc || !cam || !out_rgb) return rt_internal_set_error(RT_ERR_INVALID, "desc/cam/out_rgb is NULL");
    if (W <= 0 || H <= 0 || ngpu <= 0) return rt_internal_set_error(RT_ERR_INVALID, "W, H, ngpu must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return rt_internal_set_error(RT_ERR_NO_DEVICE, "no HIP device (this library has no CPU path)");
    if (ngpu > ndev) return rt_internal_set_error(RT_ERR_INVALID, "ngpu exceeds the visible devices");

    /* equal strips of ceil(W / ngpu) columns; trailing strips may be short or
     * empty.  Because only trailing strips are short, columns [0, W) are
     * contiguous at the start of the gathered buffer. */
    const int strip = rt_strip_bounds(W, ngpu, 0, nullptr, nullptr);
    const size_t strip_floats = (size_t)strip * (size_t)H * 3;

    std::vector<rt_scene *> scenes((size_t)ngpu, nullptr);
    std::vector<void *> d_strip((size_t)ngpu, nullptr);
    std::vector<hipStream_t> streams((size_t)ngpu, nullptr);
    std::vector<ncclComm_t> comms((size_t)ngpu, nullptr);
    void *d_full = nullptr;
    Rccl rccl;
    int rc = RT_OK;
    std::string err;

    auto cleanup = [&]() {
        for (int g = 0; g < ngpu; ++g) {
            (void)hipSetDevice(g);
            if (comms[(size_t)g] && rccl.CommDestroy) rccl.CommDestroy(comms[(size_t)g]);
            if (streams[(size_t)g]) (void)hipStreamDestroy(streams[(size_t)g]);
            if (d_strip[(size_t)g]) (void)hipFree(d_strip[(size_t)g]);
            if (scenes[(size_t)g]) rt_scene_destroy(scenes[(size_t)g]);
        }
        if (d_full) { (void)hipSetDevice(0); (void)hipFree(d_full); }
        if (rccl.handle) dlclose(rccl.handle);
    };
#define HIP_OR_BAIL(expr)                                                             \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            cleanup();                                                                \
            return rt_internal_set_error(RT_ERR_HIP, err.c_str());                    \
        }                                                                             \
    } while (0)
#define NCCL_OR_BAIL(expr)                                                            \
    do {                                                                              \
        ncclResult_t r_ = (expr);                                                     \
        if (r_ != 0) {                                                                \
            err = std::string(#expr) + ": " + rccl.GetErrorString(r_);                \
            cleanup();                                                                \
            return rt_internal_set_error(RT_ERR_RCCL, err.c_str());                   \
        }                                                                             \
    } while (0)

    for (int g = 0; g < ngpu; ++g) {
        rc = rt_scene_create(desc, g, &scenes[(size_t)g]);
        if (rc) { cleanup(); return rc; }
        HIP_OR_BAIL(hipSetDevice(g));
        HIP_OR_BAIL(hipStreamCreate(&streams[(size_t)g]));
        HIP_OR_BAIL(hipMalloc(&d_strip[(size_t)g], strip_floats * sizeof(float)));
    }
    HIP_OR_BAIL(hipSetDevice(0));
    HIP_OR_BAIL(hipMalloc(&d_full, strip_floats * sizeof(float) * (size_t)ngpu));

    if (ngpu > 1) {
        if (!load_rccl(rccl, err)) { cleanup(); return rt_internal_set_error(RT_ERR_RCCL, err.c_str()); }
        std::vector<int> devs((size_t)ngpu);
        for (int g = 0; g < ngpu; ++g) devs[(size_t)g] = g;
        NCCL_OR_BAIL(rccl.CommInitAll(comms.data(), ngpu, devs.data()));
    }

    /* render: every GPU its strip, concurrently, each on its own stream */
    for (int g = 0; g < ngpu; ++g) {
        int x0 = 0, x1 = 0;
        (void)rt_strip_bounds(W, ngpu, g, &x0, &x1);
        rc = rt_render_device(scenes[(size_t)g], cam, W, H, x0, x1, max_depth, d_strip[(size_t)g],
                              streams[(size_t)g]);
        if (rc) { cleanup(); return rc; }
    }
    /* gather to rank 0 over xGMI, enqueued behind each strip's kernel */
    if (ngpu > 1) {
        NCCL_OR_BAIL(rccl.GroupStart());
        for (int g = 0; g < ngpu; ++g) {
            HIP_OR_BAIL(hipSetDevice(g));
            NCCL_OR_BAIL(rccl.Gather(d_strip[(size_t)g], g == 0 ? d_full : nullptr, strip_floats, kNcclFloat, 0,
                                     comms[(size_t)g], streams[(size_t)g]));
        }
        NCCL_OR_BAIL(rccl.GroupEnd());
    }
    for (int g = 0; g < ngpu; ++g) {
        HIP_OR_BAIL(hipSetDevice(g));
        HIP_OR_BAIL(hipStreamSynchronize(streams[(size_t)g]));
    }
    HIP_OR_BAIL(hipSetDevice(0));
    const size_t image_bytes = (size_t)W * (size_t)H * 3 * sizeof(float);
    HIP_OR_BAIL(hipMemcpy(out_rgb, ngpu > 1 ? d_full : d_strip[0], image_bytes, hipMemcpyDeviceToHost));
    cleanup();
    return RT_OK;
#undef HIP_OR_BAIL
#undef NCCL_OR_BAIL
}
