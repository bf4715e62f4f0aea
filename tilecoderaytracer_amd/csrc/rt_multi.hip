/*
 * rt_multi.hip -- rt_render_multi(): the reference's static partitioning
 * (PARTIONING_STRATEGY 1, src/RayTracer.cpp:904-923 with CORE_NUM > 1: each
 * rank renders one contiguous strip and the strips meet in shared memory) done
 * across the GPUs of one node from a single process.
 *
 * Partition: contiguous x-strips.  The framebuffer is x-major
 * (pixels[x][z], src/RayTracer.h:44), so strip g is one contiguous block of the
 * image and its columns land in place on device 0 (ncclSend on the strip's GPU,
 * ncclRecv on device 0, point to point over the GPU's own xGMI link), no
 * repacking.  dx/dz are computed from the global x and W, H, so the gathered
 * image is bit-identical to a single-GPU render.
 *
 * The strips are cut by MEASURED cost (rt_multi_render with chunks = 0, which is
 * what rt_render_multi and bin/tcrt_raytracer --gpus G use): a warm-up frame on
 * equal strips gives every GPU's kernel time and the time the transfers take on
 * their own; device 0, which receives and sends nothing, then gets as many
 * columns as it renders in the time a peer needs to render AND ship its own
 * (rt_balance_strips: the arithmetic of tilecoderaytracer_amd/distributed.py's
 * balanced_bounds, which bench.py's one-process-per-GPU path uses).
 *
 * Within a frame every strip is rendered and sent in column CHUNKS: chunk k
 * travels to device 0 on the device's communication stream while chunk k+1 is
 * rendered on its compute stream -- the reference's ranks, too, write their
 * pixels into the shared image while they render (src/RayTracer.cpp:904-923,
 * 1188-1193); there is no serial "then gather" phase.  How many chunks follows
 * from the same measurement (rt_suggest_chunks).  rt_multi_create() keeps the
 * scenes, streams, buffers and the communicator across frames.
 *
 * RCCL is bound lazily (dlopen of librccl.so) so that the single-GPU entry
 * points carry no RCCL dependency; per-process multi-GPU (bench.py, one rank
 * per GPU) uses torch.distributed's RCCL instead and never calls this.
 */
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_capi_tuning.h"

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

/* LOOPBACK (testing aid, TCRT_MULTI_ONE_DEVICE=2): with all of a handle's GPUs on ONE device there is no RCCL -- it refuses
 * two ranks on a device -- and the strip-buffer transport (chunks, events, the two streams per GPU, the measured cut with its
 * send term, the trial against the direct stores) would never run on a one-GPU box.  These stand in for the six RCCL calls
 * multi_frame() makes: a Send remembers its buffer and stream, the Recv that follows it in the same group copies device to
 * device on the SENDER's communication stream -- where ncclSend's work would be queued, behind the chunk's kernel. */
struct LoopbackSend { const void *src = nullptr; size_t count = 0; hipStream_t stream = nullptr; };
thread_local LoopbackSend g_loopback;
ncclResult_t loopback_group() { return 0; }
ncclResult_t loopback_destroy(ncclComm_t) { return 0; }
ncclResult_t loopback_send(const void *p, size_t n, int, int, ncclComm_t, hipStream_t s) {
    g_loopback.src = p; g_loopback.count = n; g_loopback.stream = s;
    return 0;
}
ncclResult_t loopback_recv(void *p, size_t n, int, int, ncclComm_t, hipStream_t) {
    if (n != g_loopback.count || !g_loopback.src) return 1;
    return hipMemcpyAsync(p, g_loopback.src, n * sizeof(float), hipMemcpyDeviceToDevice, g_loopback.stream) == hipSuccess ? 0 : 1;
}
const char *loopback_error(ncclResult_t) { return "loopback transfer failed"; }

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

/* ---- the partition's arithmetic: pure functions, the same on every host (tests compare them with distributed.py) ---- */

extern "C" int rt_suggest_chunks(double kernel_ms, double send_ms, int most) {
    if (most < 1) most = 1;
    if (!(kernel_ms > 0.0)) return send_ms > 0.0 ? most : 1;
    const double k = std::nearbyint(4.0 * send_ms / kernel_ms);           /* half to even, like Python's round() */
    return (int)std::min((double)most, std::max(1.0, k));
}

extern "C" int rt_balance_strips(int W, int ngpu, const int *measured_bounds, const double *kernel_ms,
                                 double send_ms_per_column, int chunks, int *out_bounds) {
    if (W <= 0 || ngpu <= 0 || !measured_bounds || !kernel_ms || !out_bounds) return 1;
    if (measured_bounds[0] != 0 || measured_bounds[ngpu] != W) return 1;
    for (int g = 0; g < ngpu; ++g)
        if (measured_bounds[g] > measured_bounds[g + 1] || !(kernel_ms[g] == kernel_ms[g])) return 1;
    /* a column costs what its strip took, spread evenly (prefix sums, accumulated in column order) */
    std::vector<double> prefix((size_t)W + 1, 0.0);
    for (int g = 0; g < ngpu; ++g) {
        const int a = measured_bounds[g], b = measured_bounds[g + 1];
        if (b <= a) continue;
        const double per_column = std::max(kernel_ms[g], 0.0) / (double)(b - a);
        for (int x = a; x < b; ++x) prefix[(size_t)x + 1] = per_column;
    }
    for (int x = 0; x < W; ++x) prefix[(size_t)x + 1] = prefix[(size_t)x] + prefix[(size_t)x + 1];
    const double g_send = std::max(send_ms_per_column, 0.0);
    const int K = std::max(chunks, 1);
    /* the time GPU r needs for columns [x, x1): device 0 renders only; a peer renders and sends, chunk k on its way while
     * chunk k + 1 is rendered, so only the first chunk of the slower activity is not covered by the other */
    auto rank_time = [&](int r, int x, int x1) {
        const double render = prefix[(size_t)x1] - prefix[(size_t)x];
        if (r == 0 || g_send <= 0.0) return render;
        const double send = g_send * (double)(x1 - x);
        if (K <= 1) return render + send;
        return std::max(render, send) + std::min(render, send) / (double)K;
    };
    /* greedy fill under a time limit, bisection on the limit */
    auto fill = [&](double limit, int *bounds) {
        int x = 0;
        for (int r = 0; r < ngpu; ++r) {
            int lo = x, hi = W;
            while (lo < hi) {
                const int mid = (lo + hi + 1) / 2;
                if (rank_time(r, x, mid) <= limit) lo = mid; else hi = mid - 1;
            }
            bounds[r] = x;
            x = lo;
        }
        bounds[ngpu] = x;
        return x;
    };
    std::vector<int> trial((size_t)ngpu + 1);
    double lo = 0.0, hi = prefix[(size_t)W] + g_send * (double)W + 1e-9;
    for (int it = 0; it < 60; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (fill(mid, trial.data()) >= W) hi = mid; else lo = mid;
    }
    (void)fill(hi, out_bounds);
    out_bounds[ngpu] = W;                                  /* numerical corner: the rest goes to the last GPU */
    return 0;
}

/* the multi-GPU handle: everything that survives from frame to frame */
struct rt_multi {
    int ngpu = 0;
    std::vector<rt_scene *> scenes;
    std::vector<hipStream_t> compute, comm;          /* per device: the render kernels' stream and the transfers' */
    std::vector<std::vector<hipEvent_t>> rendered;   /* per device, per chunk: that chunk's kernel is done */
    std::vector<void *> d_strip;                     /* per device g >= 1: its strip */
    std::vector<size_t> strip_bytes;
    std::vector<ncclComm_t> comms;
    void *d_full = nullptr;                          /* device 0: the whole image; device 0 renders its strip in place */
    size_t full_bytes = 0;
    Rccl rccl;
    /* the partition in use: ngpu + 1 column bounds for images `bounds_W` wide (empty, or another width: equal strips) */
    std::vector<int> bounds;
    int bounds_W = 0, bounds_chunks = 1;
    bool bounds_explicit = false;                    /* rt_multi_set_bounds: never re-measured */
    /* what the automatic partition was measured for */
    bool have_key = false;
    int key[3] = {0, 0, 0};                          /* W, H, max_depth */
    rt_camera_desc key_cam{};
    bool broken = false;                             /* an RCCL call failed mid-frame: the communicator's state is unknown */
    /* DIRECT: every GPU's kernel stores its strip straight into the image on device 0 (peer access over xGMI) -- the reference's
     * ranks writing into the one `pixels` array, src/RayTracer.cpp:904-923, 1188-1196; no strip buffer, no transfer.  Possible
     * when every GPU may address device 0's memory; whether it is FASTER than strip buffers + RCCL is measured (multi_balance). */
    std::vector<int> dev;                            /* HIP device of GPU g of this handle: g -- or 0 for all of them under the testing aid
                                                      * TCRT_MULTI_ONE_DEVICE=1 (a box with one GPU: the strips' arithmetic, the direct stores
                                                      * and the measurement run as for ngpu GPUs, on one device; no RCCL, which refuses that) */
    bool have_rccl = false;                          /* strip buffers + ncclSend / ncclRecv are available */
    bool peer_ok = false;                            /* every GPU g >= 1 may store into device 0's memory */
    int transport_wanted = RT_MULTI_TRANSPORT_AUTO;  /* rt_multi_set_option("transport") */
    bool direct = false;                             /* the transport of the partition in use */
    rt_multi_info info{};
};

namespace {

int multi_fail(int code, const std::string &msg) { return rt_internal_set_error(code, msg.c_str()); }

constexpr int kMaxChunks = 64;
constexpr int kAlign = 16;          /* chunk boundaries fall on multiples of 16 columns from the strip's first (the widest wavefront tile) */

void strip_of(const rt_multi *m, int W, int g, int *x0, int *x1) {
    if (!m->bounds.empty() && m->bounds_W == W) { *x0 = m->bounds[(size_t)g]; *x1 = m->bounds[(size_t)g + 1]; return; }
    (void)rt_strip_bounds(W, m->ngpu, g, x0, x1);
}

/* after a failure somewhere in a frame: nothing of it may still be queued when the caller gets the handle back */
void drain_all(rt_multi *m) {
    for (int g = 0; g < m->ngpu; ++g) {
        if (hipSetDevice(m->dev[(size_t)g]) != hipSuccess) continue;
        if (m->compute[(size_t)g]) (void)hipStreamSynchronize(m->compute[(size_t)g]);
        if (m->comm[(size_t)g]) (void)hipStreamSynchronize(m->comm[(size_t)g]);
    }
}

/* One frame on the partition in use: every GPU renders its strip in `chunks` column chunks (device 0 straight into the image)
 * and -- transfers -- chunk k goes to device 0 behind its kernel while chunk k + 1 is rendered; returns when everything is on
 * device 0.  render = false: the transfers alone (the strips' columns as they are).  Any failure leaves no work queued and
 * no RCCL group open. */
int multi_frame(rt_multi *m, const rt_camera_desc *cam, int W, int H, int max_depth, int chunks, bool render, bool transfers,
                bool direct = false) {
    const int ngpu = m->ngpu;
    if (direct) transfers = false;                   /* the pixels are where they belong when the kernels are done */
    if (ngpu > 1 && transfers && !m->have_rccl) return multi_fail(RT_ERR_RCCL, "this handle has no RCCL communicator");
    const size_t column_floats = (size_t)H * 3;
    bool group_open = false;
    auto body = [&]() -> int {
#define HIP_STEP(expr)                                                                 \
        do {                                                                           \
            hipError_t e_ = (expr);                                                    \
            if (e_ != hipSuccess) return multi_fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        } while (0)
#define NCCL_STEP(expr)                                                                \
        do {                                                                           \
            ncclResult_t r_ = (expr);                                                  \
            if (r_ != 0) { m->broken = true; return multi_fail(RT_ERR_RCCL, std::string(#expr) + ": " + m->rccl.GetErrorString(r_)); } \
        } while (0)
        for (int k = 0; k < chunks; ++k) {
            for (int g = 0; g < ngpu && render; ++g) {
                int x0 = 0, x1 = 0, a = 0, b = 0;
                strip_of(m, W, g, &x0, &x1);
                (void)rt_chunk_bounds(x0, x1, chunks, k, kAlign, &a, &b);
                float *dst = (g == 0 || direct) ? static_cast<float *>(m->d_full) + (size_t)a * column_floats
                                                : static_cast<float *>(m->d_strip[(size_t)g]) + (size_t)(a - x0) * column_floats;
                int rc = rt_render_device(m->scenes[(size_t)g], cam, W, H, a, b, max_depth, dst, m->compute[(size_t)g]);
                if (rc) return rc;
                if (g == 0 || !transfers) continue;           /* device 0 sends nothing: its receipts wait for no kernel of its own */
                HIP_STEP(hipSetDevice(m->dev[(size_t)g]));
                HIP_STEP(hipEventRecord(m->rendered[(size_t)g][(size_t)k], m->compute[(size_t)g]));
                HIP_STEP(hipStreamWaitEvent(m->comm[(size_t)g], m->rendered[(size_t)g][(size_t)k], 0));
            }
            if (ngpu > 1 && transfers) {
                NCCL_STEP(m->rccl.GroupStart());
                group_open = true;
                for (int g = 1; g < ngpu; ++g) {
                    int x0 = 0, x1 = 0, a = 0, b = 0;
                    strip_of(m, W, g, &x0, &x1);
                    (void)rt_chunk_bounds(x0, x1, chunks, k, kAlign, &a, &b);
                    if (b <= a) continue;
                    const size_t count = (size_t)(b - a) * column_floats;
                    NCCL_STEP(m->rccl.Send(static_cast<float *>(m->d_strip[(size_t)g]) + (size_t)(a - x0) * column_floats, count, kNcclFloat, 0,
                                           m->comms[(size_t)g], m->comm[(size_t)g]));
                    NCCL_STEP(m->rccl.Recv(static_cast<float *>(m->d_full) + (size_t)a * column_floats, count, kNcclFloat, g,
                                           m->comms[0], m->comm[0]));
                }
                group_open = false;
                NCCL_STEP(m->rccl.GroupEnd());
            }
        }
        for (int g = 0; g < ngpu; ++g) {
            HIP_STEP(hipSetDevice(m->dev[(size_t)g]));
            HIP_STEP(hipStreamSynchronize(m->compute[(size_t)g]));
            HIP_STEP(hipStreamSynchronize(m->comm[(size_t)g]));
        }
        return RT_OK;
#undef HIP_STEP
#undef NCCL_STEP
    };
    const int rc = body();
    if (rc != RT_OK) {
        const std::string why = rt_last_error();             /* (the clean-up below must not replace the message) */
        if (group_open) (void)m->rccl.GroupEnd();
        drain_all(m);
        return multi_fail(rc, why);
    }
    return RT_OK;
}

#define HIP_OR_FAIL(expr)                                                             \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) return multi_fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

/* device memory and events for frames of this shape on the partition in use */
int multi_reserve(rt_multi *m, int W, int H, int chunks) {
    const size_t column_floats = (size_t)H * 3;
    const size_t image_bytes = (size_t)W * column_floats * sizeof(float);
    HIP_OR_FAIL(hipSetDevice(m->dev[0]));
    if (image_bytes > m->full_bytes) {
        if (m->d_full) { HIP_OR_FAIL(hipFree(m->d_full)); m->d_full = nullptr; m->full_bytes = 0; }
        HIP_OR_FAIL(hipMalloc(&m->d_full, image_bytes));
        m->full_bytes = image_bytes;
    }
    for (int g = 0; g < m->ngpu; ++g) {
        HIP_OR_FAIL(hipSetDevice(m->dev[(size_t)g]));
        int x0 = 0, x1 = 0;
        strip_of(m, W, g, &x0, &x1);
        const size_t need = (size_t)std::max(x1 - x0, 1) * column_floats * sizeof(float);
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
    return RT_OK;
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

/* Correctness before speed, in the automatic choice of transport: sampled COLUMNS of the image on device 0 -- the first and last
 * column of every strip of the cut in use and every 64th column -- against the same columns rendered by GPU 0 alone (`reference`:
 * filled by the first call, compared by later ones).  *same = every sampled byte equal.  TCRT_MULTI_CORRUPT=rccl|direct|both
 * (testing aid) spoils one pixel of that transport's image before it is looked at. */
int sample_columns_check(rt_multi *m, const rt_camera_desc *cam, int W, int H, int max_depth, const std::vector<int> &columns,
                         std::vector<float> &reference, const char *transport, bool *same) {
    const size_t column_floats = (size_t)H * 3;
    HIP_OR_FAIL(hipSetDevice(m->dev[0]));
    const char *spoil = std::getenv("TCRT_MULTI_CORRUPT");
    if (spoil && (std::strcmp(spoil, transport) == 0 || std::strcmp(spoil, "both") == 0) && !columns.empty()) {
        const float bad = -12345.0f;
        HIP_OR_FAIL(hipMemcpy(static_cast<float *>(m->d_full) + (size_t)columns[0] * column_floats, &bad, sizeof bad, hipMemcpyHostToDevice));
    }
    if (reference.empty()) {                              /* GPU 0 renders the sampled columns by itself, one at a time */
        void *d_column = nullptr;
        HIP_OR_FAIL(hipMalloc(&d_column, column_floats * sizeof(float)));
        reference.resize(columns.size() * column_floats);
        int rc = RT_OK;
        for (size_t i = 0; i < columns.size() && rc == RT_OK; ++i) {
            rc = rt_render_device(m->scenes[0], cam, W, H, columns[i], columns[i] + 1, max_depth, d_column, m->compute[0]);
            if (rc == RT_OK && hipStreamSynchronize(m->compute[0]) != hipSuccess) rc = multi_fail(RT_ERR_HIP, "hipStreamSynchronize failed");
            if (rc == RT_OK && hipMemcpy(reference.data() + i * column_floats, d_column, column_floats * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
                rc = multi_fail(RT_ERR_HIP, "hipMemcpy of a reference column failed");
        }
        (void)hipFree(d_column);
        if (rc != RT_OK) { reference.clear(); return rc; }
    }
    std::vector<float> got(column_floats);
    *same = true;
    for (size_t i = 0; i < columns.size(); ++i) {
        HIP_OR_FAIL(hipMemcpy(got.data(), static_cast<float *>(m->d_full) + (size_t)columns[i] * column_floats, column_floats * sizeof(float),
                              hipMemcpyDeviceToHost));
        if (std::memcmp(got.data(), reference.data() + i * column_floats, column_floats * sizeof(float)) != 0) { *same = false; break; }
    }
    return RT_OK;
}

/* The measured-cost partition for frames of this shape: one frame on equal strips to warm everything up (code objects, the
 * links' first use), then the kernels alone (every GPU's time for its equal strip) and the transfers alone (the time the
 * strips need to reach device 0, all peers at once, each over its own link); rt_suggest_chunks and rt_balance_strips turn that
 * into the chunk count and the strips.  What tilecoderaytracer_amd/distributed.py's measure_and_balance does for bench.py. */
int multi_balance(rt_multi *m, const rt_camera_desc *cam, int W, int H, int max_depth) {
    const int ngpu = m->ngpu;
    m->bounds.clear();
    m->bounds_chunks = 1;
    m->have_key = false;
    m->info.balanced = 0;
    std::vector<int> equal((size_t)ngpu + 1, W);
    for (int g = 0; g < ngpu; ++g) { int a, b; (void)rt_strip_bounds(W, ngpu, g, &a, &b); equal[(size_t)g] = a; }
    std::vector<double> kernel_ms((size_t)ngpu, 0.0);
    double gather_ms = 0.0;
    int chunks = 1;
    std::vector<int> cut = equal;
    const bool try_rccl = ngpu > 1 && m->have_rccl && m->transport_wanted != RT_MULTI_TRANSPORT_DIRECT;
    const bool try_direct = ngpu > 1 && m->peer_ok && (m->transport_wanted != RT_MULTI_TRANSPORT_RCCL || !m->have_rccl);
    if (ngpu > 1) {
        int rc = multi_reserve(m, W, H, 1);
        if (rc) return rc;
    }
    if (try_rccl) {
        int rc = multi_frame(m, cam, W, H, max_depth, 1, true, true);
        if (rc) return rc;
        for (int g = 0; g < ngpu; ++g) { rc = rt_reset_timing(m->scenes[(size_t)g]); if (rc) return rc; }
        rc = multi_frame(m, cam, W, H, max_depth, 1, true, false);
        if (rc) return rc;
        for (int g = 0; g < ngpu; ++g) {
            rt_timing tm;
            rc = rt_get_timing(m->scenes[(size_t)g], &tm);
            if (rc) return rc;
            kernel_ms[(size_t)g] = tm.sum_kernel_ms;
        }
        const double t0 = now_ms();
        rc = multi_frame(m, cam, W, H, max_depth, 1, false, true);
        if (rc) return rc;
        gather_ms = now_ms() - t0;
        double mean = 0.0;
        for (double k : kernel_ms) mean += k;
        mean /= (double)ngpu;
        chunks = rt_suggest_chunks(mean, gather_ms, 8);
        const double per_column = gather_ms / (double)std::max(equal[1] - equal[0], 1);
        if (rt_balance_strips(W, ngpu, equal.data(), kernel_ms.data(), per_column, chunks, cut.data()))
            return multi_fail(RT_ERR_INVALID, "rt_balance_strips refused the measured times");
    }
    /* DIRECT (see struct rt_multi): the same measurement without transfers -- every GPU's kernel time with its equal strip stored
     * into device 0's image (a peer's includes what its link made of the stores), strips of equal measured time -- and then one
     * whole frame of each transport on its own cut, by the host clock: the faster one is kept.  "transport" 1 / 2 skip the trial. */
    m->direct = false;
    m->info.transport = RT_MULTI_TRANSPORT_RCCL;
    m->info.trial_frame_ms[0] = m->info.trial_frame_ms[1] = 0.0;
    m->info.trial_image_ok[0] = m->info.trial_image_ok[1] = -1;
    if (try_direct) {
        std::vector<double> direct_ms((size_t)ngpu, 0.0);
        std::vector<int> direct_cut = equal;
        int rc = multi_frame(m, cam, W, H, max_depth, 1, true, false, true);          /* first touch of the peer mappings */
        if (rc) return rc;
        for (int g = 0; g < ngpu; ++g) { rc = rt_reset_timing(m->scenes[(size_t)g]); if (rc) return rc; }
        rc = multi_frame(m, cam, W, H, max_depth, 1, true, false, true);
        if (rc) return rc;
        for (int g = 0; g < ngpu; ++g) {
            rt_timing tm;
            rc = rt_get_timing(m->scenes[(size_t)g], &tm);
            if (rc) return rc;
            direct_ms[(size_t)g] = tm.sum_kernel_ms;
        }
        if (rt_balance_strips(W, ngpu, equal.data(), direct_ms.data(), 0.0, 1, direct_cut.data()))
            return multi_fail(RT_ERR_INVALID, "rt_balance_strips refused the measured times");
        bool use_direct = !try_rccl;
        if (try_rccl) {                                  /* both are possible and neither was asked for: one frame of each */
            m->bounds = cut; m->bounds_W = W;
            rc = multi_reserve(m, W, H, chunks);
            if (rc) return rc;
            double best[2] = {1e300, 1e300};
            bool right[2] = {true, true};                    /* the transport's image equals GPU 0's own rendering on the sampled columns */
            std::vector<int> columns;
            for (int x = 0; x < W; x += 64) columns.push_back(x);
            for (int g = 0; g < ngpu; ++g)
                for (const std::vector<int> *b : {&cut, &direct_cut})
                    if ((*b)[(size_t)g + 1] > (*b)[(size_t)g]) { columns.push_back((*b)[(size_t)g]); columns.push_back((*b)[(size_t)g + 1] - 1); }
            std::sort(columns.begin(), columns.end());
            columns.erase(std::unique(columns.begin(), columns.end()), columns.end());
            std::vector<float> reference;
            for (int round = 0; round < 2; ++round) {        /* the faster of two frames each: a single frame's host time is noisy */
                m->bounds = cut;
                double t0 = now_ms();
                rc = multi_frame(m, cam, W, H, max_depth, chunks, true, true);
                if (rc) return rc;
                best[0] = std::min(best[0], now_ms() - t0);
                if (round == 1) { rc = sample_columns_check(m, cam, W, H, max_depth, columns, reference, "rccl", &right[0]); if (rc) return rc; }
                m->bounds = direct_cut;
                t0 = now_ms();
                rc = multi_frame(m, cam, W, H, max_depth, 1, true, false, true);
                if (rc) return rc;
                best[1] = std::min(best[1], now_ms() - t0);
                if (round == 1) { rc = sample_columns_check(m, cam, W, H, max_depth, columns, reference, "direct", &right[1]); if (rc) return rc; }
            }
            m->info.trial_frame_ms[0] = best[0];
            m->info.trial_frame_ms[1] = best[1];
            m->info.trial_image_ok[0] = right[0] ? 1 : 0;
            m->info.trial_image_ok[1] = right[1] ? 1 : 0;
            if (!right[0] && !right[1])
                return multi_fail(RT_ERR_HIP, "neither transport delivered the image GPU 0 renders by itself (sampled columns differ)");
            /* correctness before speed: the faster transport only if its image is right */
            use_direct = right[1] && (best[1] < best[0] || !right[0]);
        }
        if (use_direct) {
            cut = direct_cut;
            chunks = 1;
            kernel_ms = direct_ms;
            m->direct = true;
            m->info.transport = RT_MULTI_TRANSPORT_DIRECT;
        }
    }
    m->bounds = cut;
    m->bounds_W = W;
    m->bounds_chunks = chunks;
    m->have_key = true;
    m->key[0] = W; m->key[1] = H; m->key[2] = max_depth;
    m->key_cam = *cam;
    m->info.balanced = ngpu > 1 ? 1 : 0;
    m->info.measured_gather_ms = gather_ms;
    for (int g = 0; g < ngpu && g < RT_MULTI_MAX_GPUS; ++g) m->info.measured_kernel_ms[g] = kernel_ms[(size_t)g];
    return RT_OK;
}

} // namespace

/* ---- the shared image: the reference's one `pixels` array (src/RayTracer.h:44) for one process per GPU ---- */

static_assert(sizeof(hipIpcMemHandle_t) == RT_SHARED_HANDLE_BYTES, "RT_SHARED_HANDLE_BYTES is the size of a HIP IPC memory handle");

extern "C" int rt_shared_image_create(int device, uint64_t bytes, void **d_image, unsigned char handle[RT_SHARED_HANDLE_BYTES]) {
    if (!d_image || !handle) return multi_fail(RT_ERR_INVALID, "d_image/handle is NULL");
    *d_image = nullptr;
    if (bytes == 0) return multi_fail(RT_ERR_INVALID, "bytes must be positive");
    HIP_OR_FAIL(hipSetDevice(device));
    void *p = nullptr;
    HIP_OR_FAIL(hipMalloc(&p, (size_t)bytes));
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return multi_fail(RT_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e) +
                                          " (this pool's driver needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment)");
    }
    std::memcpy(handle, &h, RT_SHARED_HANDLE_BYTES);
    *d_image = p;
    return RT_OK;
}

extern "C" int rt_shared_image_open(int device, const unsigned char handle[RT_SHARED_HANDLE_BYTES], void **d_image) {
    if (!d_image || !handle) return multi_fail(RT_ERR_INVALID, "d_image/handle is NULL");
    *d_image = nullptr;
    HIP_OR_FAIL(hipSetDevice(device));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, RT_SHARED_HANDLE_BYTES);
    void *p = nullptr;
    HIP_OR_FAIL(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *d_image = p;
    return RT_OK;
}

extern "C" int rt_shared_image_close(int device, void *d_image) {
    if (!d_image) return RT_OK;
    HIP_OR_FAIL(hipSetDevice(device));
    HIP_OR_FAIL(hipIpcCloseMemHandle(d_image));
    return RT_OK;
}

extern "C" int rt_shared_image_destroy(int device, void *d_image) {
    if (!d_image) return RT_OK;
    HIP_OR_FAIL(hipSetDevice(device));
    HIP_OR_FAIL(hipFree(d_image));
    return RT_OK;
}

extern "C" int rt_multi_destroy(rt_multi *m) {
    if (!m) return RT_OK;
    for (int g = 0; g < m->ngpu; ++g) {
        (void)hipSetDevice((size_t)g < m->dev.size() ? m->dev[(size_t)g] : g);
        if ((size_t)g < m->comms.size() && m->comms[(size_t)g] && m->rccl.CommDestroy) m->rccl.CommDestroy(m->comms[(size_t)g]);
        if ((size_t)g < m->rendered.size())
            for (hipEvent_t e : m->rendered[(size_t)g]) (void)hipEventDestroy(e);
        if ((size_t)g < m->compute.size() && m->compute[(size_t)g]) (void)hipStreamDestroy(m->compute[(size_t)g]);
        if ((size_t)g < m->comm.size() && m->comm[(size_t)g]) (void)hipStreamDestroy(m->comm[(size_t)g]);
        if ((size_t)g < m->d_strip.size() && m->d_strip[(size_t)g]) (void)hipFree(m->d_strip[(size_t)g]);
        if ((size_t)g < m->scenes.size() && m->scenes[(size_t)g]) rt_scene_destroy(m->scenes[(size_t)g]);
    }
    if (m->d_full) { (void)hipSetDevice(m->dev.empty() ? 0 : m->dev[0]); (void)hipFree(m->d_full); }
    if (m->rccl.handle) dlclose(m->rccl.handle);
    delete m;
    return RT_OK;
}

extern "C" int rt_multi_create(const rt_scene_desc *desc, int ngpu, rt_multi **out) {
    if (!desc || !out) return multi_fail(RT_ERR_INVALID, "desc/out is NULL");
    *out = nullptr;
    if (ngpu <= 0) return multi_fail(RT_ERR_INVALID, "ngpu must be positive");
    if (ngpu > RT_MULTI_MAX_GPUS) return multi_fail(RT_ERR_INVALID, "ngpu exceeds RT_MULTI_MAX_GPUS");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return multi_fail(RT_ERR_NO_DEVICE, "no HIP device (this library has no CPU path)");
    const char *one = std::getenv("TCRT_MULTI_ONE_DEVICE");
    const bool loopback = one && one[0] == '2';              /* one device AND the strip-buffer transport, its transfers as local copies */
    const bool one_device = one && (one[0] == '1' || loopback);
    if (ngpu > ndev && !one_device) return multi_fail(RT_ERR_INVALID, "ngpu exceeds the visible devices");
    rt_multi *m = new (std::nothrow) rt_multi();
    if (!m) return multi_fail(RT_ERR_INVALID, "out of memory");
    m->ngpu = ngpu;
    m->dev.assign((size_t)ngpu, 0);
    for (int g = 0; g < ngpu && !one_device; ++g) m->dev[(size_t)g] = g;
    m->scenes.assign((size_t)ngpu, nullptr);
    m->compute.assign((size_t)ngpu, nullptr);
    m->comm.assign((size_t)ngpu, nullptr);
    m->rendered.assign((size_t)ngpu, {});
    m->d_strip.assign((size_t)ngpu, nullptr);
    m->strip_bytes.assign((size_t)ngpu, 0);
    m->comms.assign((size_t)ngpu, nullptr);
    m->info.ngpu = ngpu;
    auto bail = [&](int rc) { rt_multi_destroy(m); return rc; };
    for (int g = 0; g < ngpu; ++g) {
        int rc = rt_scene_create(desc, m->dev[(size_t)g], &m->scenes[(size_t)g]);
        if (rc) return bail(rc);
        hipError_t e = hipSetDevice(m->dev[(size_t)g]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->compute[(size_t)g], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->comm[(size_t)g], hipStreamNonBlocking);
        if (e != hipSuccess) return bail(multi_fail(RT_ERR_HIP, std::string("stream setup: ") + hipGetErrorString(e)));
    }
    m->peer_ok = ngpu > 1;
    for (int g = 1; g < ngpu && !one_device; ++g) {     /* may GPU g's kernels store into device 0's memory? */
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, g, 0) != hipSuccess || !can) { m->peer_ok = false; break; }
        hipError_t e = hipSetDevice(g);
        if (e == hipSuccess) e = hipDeviceEnablePeerAccess(0, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
        if (e != hipSuccess) { (void)hipGetLastError(); m->peer_ok = false; break; }
    }
    if (ngpu > 1 && loopback) {
        m->rccl.GroupStart = loopback_group; m->rccl.GroupEnd = loopback_group;
        m->rccl.Send = loopback_send; m->rccl.Recv = loopback_recv;
        m->rccl.CommDestroy = loopback_destroy; m->rccl.GetErrorString = loopback_error;
        m->have_rccl = true;
    }
    if (ngpu > 1 && !one_device) {
        /* RCCL: needed where a GPU cannot address device 0's memory; with peer access everywhere its absence leaves the direct stores */
        std::string err;
        if (load_rccl(m->rccl, err)) {
            std::vector<int> devs((size_t)ngpu);
            for (int g = 0; g < ngpu; ++g) devs[(size_t)g] = g;
            ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), ngpu, devs.data());
            if (r == 0) m->have_rccl = true;
            else err = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(r);
        }
        if (!m->have_rccl && !m->peer_ok) return bail(multi_fail(RT_ERR_RCCL, err));
    }
    *out = m;
    return RT_OK;
}

extern "C" int rt_multi_set_option(rt_multi *m, const char *key, int value) {
    if (!m) return multi_fail(RT_ERR_INVALID, "handle is NULL");
    if (key && std::strcmp(key, "transport") == 0) {           /* the handle's own option: how the strips reach device 0 */
        if (value < RT_MULTI_TRANSPORT_AUTO || value > RT_MULTI_TRANSPORT_DIRECT)
            return multi_fail(RT_ERR_INVALID, "transport: 0 automatic (measured), 1 strip buffers + RCCL, 2 direct stores into device 0's image");
        if (value == RT_MULTI_TRANSPORT_DIRECT && m->ngpu > 1 && !m->peer_ok)
            return multi_fail(RT_ERR_INVALID, "transport 2: not every GPU of this handle has peer access to device 0");
        if (value == RT_MULTI_TRANSPORT_RCCL && m->ngpu > 1 && !m->have_rccl)
            return multi_fail(RT_ERR_INVALID, "transport 1: this handle has no RCCL communicator");
        m->transport_wanted = value;
        m->have_key = false;                                     /* the next automatic frame measures again */
        return RT_OK;
    }
    for (rt_scene *s : m->scenes) {
        int rc = rt_set_option(s, key, value);
        if (rc) return rc;
    }
    return RT_OK;
}

extern "C" int rt_multi_set_bounds(rt_multi *m, int W, const int *bounds, int chunks) {
    if (!m) return multi_fail(RT_ERR_INVALID, "handle is NULL");
    if (!bounds) {                                     /* back to equal strips / the automatic partition */
        m->bounds.clear();
        m->bounds_explicit = false;
        m->have_key = false;
        m->bounds_chunks = 1;
        return RT_OK;
    }
    if (W <= 0 || chunks < 1 || chunks > kMaxChunks) return multi_fail(RT_ERR_INVALID, "need W > 0 and chunks in [1, 64]");
    if (bounds[0] != 0 || bounds[m->ngpu] != W) return multi_fail(RT_ERR_INVALID, "the strips must cover [0, W)");
    for (int g = 0; g < m->ngpu; ++g)
        if (bounds[g] > bounds[g + 1]) return multi_fail(RT_ERR_INVALID, "the strips must be in GPU order");
    m->bounds.assign(bounds, bounds + m->ngpu + 1);
    m->bounds_W = W;
    m->bounds_chunks = chunks;
    m->bounds_explicit = true;
    m->have_key = false;
    return RT_OK;
}

extern "C" int rt_multi_get_info(const rt_multi *m, rt_multi_info *out) {
    if (!m || !out) return multi_fail(RT_ERR_INVALID, "handle/out is NULL");
    *out = m->info;
    return RT_OK;
}

extern "C" int rt_multi_render(rt_multi *m, const rt_camera_desc *cam, int W, int H, int max_depth, int chunks, float *out_rgb) {
    if (!m || !cam || !out_rgb) return multi_fail(RT_ERR_INVALID, "handle/cam/out_rgb is NULL");
    if (W <= 0 || H <= 0) return multi_fail(RT_ERR_INVALID, "W and H must be positive");
    if (chunks < 0 || chunks > kMaxChunks) return multi_fail(RT_ERR_INVALID, "chunks must be in [0, 64] (0: automatic)");
    if (m->broken)
        return multi_fail(RT_ERR_RCCL, "an RCCL call failed in an earlier frame of this handle: destroy it and create another");
    const int ngpu = m->ngpu;
    /* the caller's own strips or chunk count: the transport is the one asked for ("transport" 2: direct stores), else RCCL */
    bool direct = ngpu > 1 && m->peer_ok && (m->transport_wanted == RT_MULTI_TRANSPORT_DIRECT || !m->have_rccl);
    if (chunks == 0) {
        /* automatic: the strips of rt_multi_set_bounds if there are any, else the measured-cost partition for this shape */
        if (m->bounds_explicit && m->bounds_W == W) {
            chunks = m->bounds_chunks;
        } else {
            const bool same = m->have_key && m->key[0] == W && m->key[1] == H && m->key[2] == max_depth &&
                              std::memcmp(&m->key_cam, cam, sizeof(*cam)) == 0;
            if (!same) {
                int rc = multi_balance(m, cam, W, H, max_depth);
                if (rc) return rc;
            }
            chunks = m->bounds_chunks;
            direct = m->direct;                          /* what the measurement chose */
        }
    } else if (!m->bounds_explicit && !m->bounds.empty()) {
        m->bounds.clear();                               /* an explicit chunk count on a measured partition: equal strips, as asked */
        m->have_key = false;
        m->direct = false;
        m->info.balanced = 0;
    }
    if (direct) chunks = 1;                              /* nothing to overlap: one launch per strip */
    int rc = multi_reserve(m, W, H, chunks);
    if (rc) return rc;
    for (int g = 0; g < ngpu; ++g) { rc = rt_reset_timing(m->scenes[(size_t)g]); if (rc) return rc; }
    const double t0 = now_ms();
    rc = multi_frame(m, cam, W, H, max_depth, chunks, true, true, direct);
    if (rc) return rc;
    m->info.frame_ms = now_ms() - t0;
    m->info.transport = direct ? RT_MULTI_TRANSPORT_DIRECT : RT_MULTI_TRANSPORT_RCCL;
    m->info.chunks = chunks;
    const size_t image_bytes = (size_t)W * (size_t)H * 3 * sizeof(float);
    HIP_OR_FAIL(hipSetDevice(m->dev[0]));
    HIP_OR_FAIL(hipMemcpy(out_rgb, m->d_full, image_bytes, hipMemcpyDeviceToHost));
    /* per GPU: its kernels' time in this frame; and what the kernels may have had to tell the host (a HELP wait that timed
     * out: the image is exact, the caller is told) */
    for (int g = 0; g < ngpu; ++g) {
        rt_timing tm;
        rc = rt_get_timing(m->scenes[(size_t)g], &tm);
        if (rc) return rc;
        int x0 = 0, x1 = 0;
        strip_of(m, W, g, &x0, &x1);
        m->info.bounds[g] = x0;
        m->info.bounds[g + 1] = x1;
        m->info.kernel_ms[g] = tm.sum_kernel_ms;
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
    rc = rt_multi_render(m, cam, W, H, max_depth, 0, out_rgb);      /* strips and chunks by measurement */
    rt_multi_destroy(m);
    return rc;
}
