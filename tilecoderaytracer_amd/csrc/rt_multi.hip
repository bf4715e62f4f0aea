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
 * Within a frame every strip is rendered and sent in column CHUNKS: chunk k
 * travels to device 0 on the device's communication stream while chunk k+1 is
 * rendered on its compute stream -- the reference's ranks, too, write their
 * pixels into the shared image while they render (src/RayTracer.cpp:904-923,
 * 1188-1193); there is no serial "then gather" phase.  rt_multi_create() keeps
 * the scenes, streams, buffers and the communicator across frames.
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
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
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
    BIND(Send, "ncclSend");
    BIND(Recv, "ncclRecv");
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

extern "C" int rt_chunk_bounds(int x0, int x1, int chunks, int k, int align, int *a, int *b) {
    if (x0 < 0 || x1 < x0 || chunks <= 0 || k < 0 || k >= chunks || align <= 0) return 1;
    const long long n = (long long)x1 - x0, units = (n + align - 1) / align;
    const long long lo = (units * k / chunks) * align, hi = (units * (k + 1) / chunks) * align;
    if (a) *a = x0 + (int)(lo < n ? lo : n);
    if (b) *b = x0 + (int)(hi < n ? hi : n);
    return 0;
}

/* the multi-GPU handle: everything that survives from frame to frame */
struct rt_multi {
    int ngpu = 0;
    std::vector<rt_scene *> scenes;
    std::vector<hipStream_t> compute, comm;          /* per device: the render kernels' stream and the transfers' */
    std::vector<std::vector<hipEvent_t>> rendered;   /* per device, per chunk: that chunk's kernel is done */
    std::vector<void *> d_strip;                     /* per device g >= 1: its strip, strip_floats floats */
    std::vector<size_t> strip_bytes;
    std::vector<ncclComm_t> comms;
    void *d_full = nullptr;                          /* device 0: the whole image; device 0 renders its strip in place */
    size_t full_bytes = 0;
    Rccl rccl;
};

namespace {

int multi_fail(int code, const std::string &msg) { return rt_internal_set_error(code, msg.c_str()); }

#define HIP_OR_FAIL(expr)                                                             \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) return multi_fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NCCL_OR_FAIL(m, expr)                                                         \
    do {                                                                              \
        ncclResult_t r_ = (expr);                                                     \
        if (r_ != 0) return multi_fail(RT_ERR_RCCL, std::string(#expr) + ": " + (m)->rccl.GetErrorString(r_)); \
    } while (0)

constexpr int kMaxChunks = 64;

} // namespace

extern "C" int rt_multi_destroy(rt_multi *m) {
    if (!m) return RT_OK;
    for (int g = 0; g < m->ngpu; ++g) {
        (void)hipSetDevice(g);
        if ((size_t)g < m->comms.size() && m->comms[(size_t)g] && m->rccl.CommDestroy) m->rccl.CommDestroy(m->comms[(size_t)g]);
        if ((size_t)g < m->rendered.size())
            for (hipEvent_t e : m->rendered[(size_t)g]) (void)hipEventDestroy(e);
        if ((size_t)g < m->compute.size() && m->compute[(size_t)g]) (void)hipStreamDestroy(m->compute[(size_t)g]);
        if ((size_t)g < m->comm.size() && m->comm[(size_t)g]) (void)hipStreamDestroy(m->comm[(size_t)g]);
        if ((size_t)g < m->d_strip.size() && m->d_strip[(size_t)g]) (void)hipFree(m->d_strip[(size_t)g]);
        if ((size_t)g < m->scenes.size() && m->scenes[(size_t)g]) rt_scene_destroy(m->scenes[(size_t)g]);
    }
    if (m->d_full) { (void)hipSetDevice(0); (void)hipFree(m->d_full); }
    if (m->rccl.handle) dlclose(m->rccl.handle);
    delete m;
    return RT_OK;
}

extern "C" int rt_multi_create(const rt_scene_desc *desc, int ngpu, rt_multi **out) {
    if (!desc || !out) return multi_fail(RT_ERR_INVALID, "desc/out is NULL");
    *out = nullptr;
    if (ngpu <= 0) return multi_fail(RT_ERR_INVALID, "ngpu must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return multi_fail(RT_ERR_NO_DEVICE, "no HIP device (this library has no CPU path)");
    if (ngpu > ndev) return multi_fail(RT_ERR_INVALID, "ngpu exceeds the visible devices");
    rt_multi *m = new (std::nothrow) rt_multi();
    if (!m) return multi_fail(RT_ERR_INVALID, "out of memory");
    m->ngpu = ngpu;
    m->scenes.assign((size_t)ngpu, nullptr);
    m->compute.assign((size_t)ngpu, nullptr);
    m->comm.assign((size_t)ngpu, nullptr);
    m->rendered.assign((size_t)ngpu, {});
    m->d_strip.assign((size_t)ngpu, nullptr);
    m->strip_bytes.assign((size_t)ngpu, 0);
    m->comms.assign((size_t)ngpu, nullptr);
    auto bail = [&](int rc) { rt_multi_destroy(m); return rc; };
    for (int g = 0; g < ngpu; ++g) {
        int rc = rt_scene_create(desc, g, &m->scenes[(size_t)g]);
        if (rc) return bail(rc);
        hipError_t e = hipSetDevice(g);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->compute[(size_t)g], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->comm[(size_t)g], hipStreamNonBlocking);
        if (e != hipSuccess) return bail(multi_fail(RT_ERR_HIP, std::string("stream setup: ") + hipGetErrorString(e)));
    }
    if (ngpu > 1) {
        std::string err;
        if (!load_rccl(m->rccl, err)) return bail(multi_fail(RT_ERR_RCCL, err));
        std::vector<int> devs((size_t)ngpu);
        for (int g = 0; g < ngpu; ++g) devs[(size_t)g] = g;
        ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), ngpu, devs.data());
        if (r != 0) return bail(multi_fail(RT_ERR_RCCL, std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(r)));
    }
    *out = m;
    return RT_OK;
}

extern "C" int rt_multi_set_option(rt_multi *m, const char *key, int value) {
    if (!m) return multi_fail(RT_ERR_INVALID, "handle is NULL");
    for (rt_scene *s : m->scenes) {
        int rc = rt_set_option(s, key, value);
        if (rc) return rc;
    }
    return RT_OK;
}

extern "C" int rt_multi_render(rt_multi *m, const rt_camera_desc *cam, int W, int H, int max_depth, int chunks, float *out_rgb) {
    if (!m || !cam || !out_rgb) return multi_fail(RT_ERR_INVALID, "handle/cam/out_rgb is NULL");
    if (W <= 0 || H <= 0) return multi_fail(RT_ERR_INVALID, "W and H must be positive");
    if (chunks <= 0 || chunks > kMaxChunks) return multi_fail(RT_ERR_INVALID, "chunks must be in [1, 64]");
    const int ngpu = m->ngpu;
    /* equal strips of ceil(W / ngpu) columns; trailing strips may be short or empty */
    const int strip = rt_strip_bounds(W, ngpu, 0, nullptr, nullptr);
    const size_t column_floats = (size_t)H * 3;
    const size_t image_bytes = (size_t)W * column_floats * sizeof(float);
    HIP_OR_FAIL(hipSetDevice(0));
    if (image_bytes > m->full_bytes) {
        if (m->d_full) { HIP_OR_FAIL(hipFree(m->d_full)); m->d_full = nullptr; m->full_bytes = 0; }
        HIP_OR_FAIL(hipMalloc(&m->d_full, image_bytes));
        m->full_bytes = image_bytes;
    }
    for (int g = 0; g < ngpu; ++g) {
        HIP_OR_FAIL(hipSetDevice(g));
        const size_t need = (size_t)strip * column_floats * sizeof(float);
        if (g > 0 && need > m->strip_bytes[(size_t)g]) {
            if (m->d_strip[(size_t)g]) { HIP_OR_FAIL(hipFree(m->d_strip[(size_t)g])); m->d_strip[(size_t)g] = nullptr; m->strip_bytes[(size_t)g] = 0; }
            HIP_OR_FAIL(hipMalloc(&m->d_strip[(size_t)g], need));
            m->strip_bytes[(size_t)g] = need;
        }
        while ((int)m->rendered[(size_t)g].size() < chunks) {
            hipEvent_t e;
            HIP_OR_FAIL(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            m->rendered[(size_t)g].push_back(e);
        }
    }
    /* chunk boundaries fall on multiples of 16 columns from the strip's first (the widest wavefront tile) */
    const int align = 16;
    for (int k = 0; k < chunks; ++k) {
        /* every GPU renders chunk k of its strip (device 0 straight into the image) ... */
        for (int g = 0; g < ngpu; ++g) {
            int x0 = 0, x1 = 0, a = 0, b = 0;
            (void)rt_strip_bounds(W, ngpu, g, &x0, &x1);
            (void)rt_chunk_bounds(x0, x1, chunks, k, align, &a, &b);
            float *dst = g == 0 ? static_cast<float *>(m->d_full) + (size_t)a * column_floats
                                : static_cast<float *>(m->d_strip[(size_t)g]) + (size_t)(a - x0) * column_floats;
            int rc = rt_render_device(m->scenes[(size_t)g], cam, W, H, a, b, max_depth, dst, m->compute[(size_t)g]);
            if (rc) return rc;
            HIP_OR_FAIL(hipSetDevice(g));
            HIP_OR_FAIL(hipEventRecord(m->rendered[(size_t)g][(size_t)k], m->compute[(size_t)g]));
            HIP_OR_FAIL(hipStreamWaitEvent(m->comm[(size_t)g], m->rendered[(size_t)g][(size_t)k], 0));
        }
        /* ... and chunk k goes to device 0 over xGMI, behind its kernel, while chunk k + 1 is rendered */
        if (ngpu > 1) {
            NCCL_OR_FAIL(m, m->rccl.GroupStart());
            for (int g = 1; g < ngpu; ++g) {
                int x0 = 0, x1 = 0, a = 0, b = 0;
                (void)rt_strip_bounds(W, ngpu, g, &x0, &x1);
                (void)rt_chunk_bounds(x0, x1, chunks, k, align, &a, &b);
                if (b <= a) continue;
                const size_t count = (size_t)(b - a) * column_floats;
                NCCL_OR_FAIL(m, m->rccl.Send(static_cast<float *>(m->d_strip[(size_t)g]) + (size_t)(a - x0) * column_floats, count, kNcclFloat, 0,
                                             m->comms[(size_t)g], m->comm[(size_t)g]));
                NCCL_OR_FAIL(m, m->rccl.Recv(static_cast<float *>(m->d_full) + (size_t)a * column_floats, count, kNcclFloat, g,
                                             m->comms[0], m->comm[0]));
            }
            NCCL_OR_FAIL(m, m->rccl.GroupEnd());
        }
    }
    for (int g = 0; g < ngpu; ++g) {
        HIP_OR_FAIL(hipSetDevice(g));
        HIP_OR_FAIL(hipStreamSynchronize(m->compute[(size_t)g]));
        HIP_OR_FAIL(hipStreamSynchronize(m->comm[(size_t)g]));
    }
    HIP_OR_FAIL(hipSetDevice(0));
    HIP_OR_FAIL(hipMemcpy(out_rgb, m->d_full, image_bytes, hipMemcpyDeviceToHost));
    /* what the kernels may have had to tell the host (a HELP wait that timed out: the image is exact, the caller is told) */
    for (int g = 0; g < ngpu; ++g) {
        rt_timing tm;
        int rc = rt_get_timing(m->scenes[(size_t)g], &tm);
        if (rc) return rc;
    }
    return RT_OK;
}

extern "C" int rt_render_multi(const rt_scene_desc *desc, const rt_camera_desc *cam, int W, int H,
                               int max_depth, int ngpu, float *out_rgb) {
    if (!desc || !cam || !out_rgb) return multi_fail(RT_ERR_INVALID, "desc/cam/out_rgb is NULL");
    if (W <= 0 || H <= 0 || ngpu <= 0) return multi_fail(RT_ERR_INVALID, "W, H, ngpu must be positive");
    rt_multi *m = nullptr;
    int rc = rt_multi_create(desc, ngpu, &m);
    if (rc) return rc;
    rc = rt_multi_render(m, cam, W, H, max_depth, ngpu > 1 ? 4 : 1, out_rgb);
    rt_multi_destroy(m);
    return rc;
}
