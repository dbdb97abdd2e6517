// stx_api.cpp — host side of the C ABI declared in include/stitching_amd.h.
// Context / stream / caching allocator / profiler, device images, the Warper entry points
// (ProjectorBase::setCameraParams, ROI finalisation) and the Blender state machines.
// Compiled with -ffp-contract=off: the fp32 host arithmetic below restates OpenCV's baseline
// (non-FMA) evaluation order.
#include <algorithm>
#include <array>
#include <limits>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "stx_blend_kernels.h"
#include "stx_internal.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void stx_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int stx_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

STX_EXPORT const char* stx_last_error(void) { return g_err; }
STX_EXPORT int stx_version(void) { return STX_VERSION; }

STX_EXPORT int stx_device_count(int* out_n)
{
    if (!out_n) return stx_fail(STX_ERR_INVALID, "out_n is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_n = 0;
        return stx_fail(STX_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *out_n = n;
    return STX_OK;
}

// ---------------------------------------------------------------------------------------------
// context, allocator
// ---------------------------------------------------------------------------------------------
int stx_set_device(stx_ctx* ctx)
{
    STX_HIP(hipSetDevice(ctx->device));
    return STX_OK;
}

STX_EXPORT int stx_ctx_create(int device, stx_ctx** out)
{
    if (!out) return stx_fail(STX_ERR_INVALID, "out is null");
    *out = nullptr;
    int n = 0;
    STX_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return stx_fail(STX_ERR_INVALID, "device %d out of range (have %d)", device, n);
    STX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    STX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return stx_fail(STX_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                        prop.gcnArchName);
    stx_ctx* ctx = new stx_ctx();
    ctx->device = device;
    ctx->pinned_bytes = 1 << 16;
    ctx->stage_bytes = 1 << 20;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocCoherent | hipHostMallocMapped);  // kernels write ROI results into it
    if (e == hipSuccess) memset(ctx->pinned, 0, ctx->pinned_bytes);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->stage, ctx->stage_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&ctx->aux_scratch, ctx->pinned_bytes);
    if (e != hipSuccess) {
        stx_ctx_destroy(ctx);  // releases whatever was created
        return stx_fail(STX_ERR_HIP, "context set-up failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return STX_OK;
}

STX_EXPORT int stx_ctx_sync(stx_ctx* ctx)
{
    if (!ctx) return stx_fail(STX_ERR_INVALID, "ctx is null");
    STX_TRY(stx_set_device(ctx));
    STX_HIP(hipStreamSynchronize(ctx->stream));
    return STX_OK;
}

STX_EXPORT int stx_ctx_destroy(stx_ctx* ctx)
{
    if (!ctx) return STX_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) hipStreamSynchronize(ctx->aux_stream);
    for (auto& kv : ctx->block_size) hipFree(kv.first);
    for (auto& p : ctx->prof_pending) { hipEventDestroy(p.start); hipEventDestroy(p.stop); }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    for (auto e : ctx->marks) if (e) hipEventDestroy(e);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    if (ctx->stage) hipHostFree(ctx->stage);
    for (hipEvent_t e : ctx->stage_ev) if (e) hipEventDestroy(e);
    if (ctx->aux_scratch) hipFree(ctx->aux_scratch);
    if (ctx->aux_stream) hipStreamDestroy(ctx->aux_stream);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return STX_OK;
}

static size_t bucket_of(size_t bytes)
{
    if (bytes < 256) return 256;
    if (bytes >= (1u << 20)) return (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
    size_t b = 256;
    while (b < bytes) b <<= 1;
    return b;
}

// Stream-ordered caching allocator: every kernel of a ctx runs on ctx->stream, so a block
// returned here can be handed out again immediately — its next user is enqueued after its last.
int stx_dev_alloc(stx_ctx* ctx, size_t bytes, void** out)
{
    size_t b = bucket_of(bytes + 64);  // +64: kernels may over-read up to 12 bytes past a row
    std::lock_guard<std::mutex> lock(ctx->alloc_mutex);
    auto it = ctx->free_blocks.find(b);
    if (it != ctx->free_blocks.end() && !it->second.empty()) {
        *out = it->second.back();
        it->second.pop_back();
        return STX_OK;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, b);
    if (e != hipSuccess) {
        // release the cache and retry once
        hipStreamSynchronize(ctx->stream);
        for (auto& kv : ctx->free_blocks) {
            for (void* q : kv.second) { ctx->bytes_allocated -= ctx->block_size[q]; ctx->block_size.erase(q); hipFree(q); }
            kv.second.clear();
        }
        e = hipMalloc(&p, b);
        if (e != hipSuccess) return stx_fail(STX_ERR_OOM, "hipMalloc(%zu) failed: %s", b, hipGetErrorString(e));
    }
    ctx->block_size[p] = b;
    ctx->bytes_allocated += b;
    *out = p;
    return STX_OK;
}

void stx_dev_free(stx_ctx* ctx, void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(ctx->alloc_mutex);
    auto it = ctx->block_size.find(p);
    if (it == ctx->block_size.end()) return;
    ctx->free_blocks[it->second].push_back(p);
}

int stx_stage_upload(stx_ctx* ctx, void* d, const void* h, size_t bytes)
{
    if (bytes == 0) return STX_OK;
    const size_t seg = ctx->stage_bytes / STX_STAGE_SEGS;
    if (bytes > seg) {  // larger than a segment of the ring: plain synchronous copy
        STX_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
        STX_HIP(hipStreamSynchronize(ctx->stream));
        return STX_OK;
    }
    size_t off = ctx->stage_off;
    int cur = ctx->stage_seg;  // the segment that holds the previous upload
    if (off + bytes > (size_t)(cur + 1) * seg) {  // does not fit into the rest of it: on to the next segment
        const int next = (cur + 1) % STX_STAGE_SEGS;
        if (!ctx->stage_ev[cur]) STX_HIP(hipEventCreateWithFlags(&ctx->stage_ev[cur], hipEventDisableTiming));
        STX_HIP(hipEventRecord(ctx->stage_ev[cur], ctx->stream));  // behind the last copy out of segment `cur`
        ctx->stage_ev_set[cur] = true;
        if (ctx->stage_ev_set[next]) STX_HIP(hipEventSynchronize(ctx->stage_ev[next]));  // its copies of the previous lap
        off = (size_t)next * seg;
        cur = next;
    }
    ctx->stage_seg = cur;
    uint8_t* slot = ctx->stage + off;
    memcpy(slot, h, bytes);
    ctx->stage_off = off + ((bytes + 255) & ~(size_t)255);
    STX_HIP(hipMemcpyAsync(d, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    return STX_OK;
}

// ---------------------------------------------------------------------------------------------
// profiler: HIP events around each launch, on the stream the kernel is launched on
// ---------------------------------------------------------------------------------------------
static hipEvent_t take_event(stx_ctx* ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

StxProfScope::StxProfScope(stx_ctx* c, const char* name, double algo_bytes, hipStream_t on, bool attach)
    : ctx(c), stream(on ? on : c->stream), attached(attach)
{
    if (!ctx->prof_on) return;
    auto it = ctx->prof_index.find(name);
    int idx;
    if (it == ctx->prof_index.end()) {
        idx = (int)ctx->prof.size();
        ctx->prof.push_back(StxProfEntry());
        ctx->prof.back().name = name;
        ctx->prof_index[name] = idx;
    } else {
        idx = it->second;
    }
    ctx->prof[idx].calls += 1;
    ctx->prof[idx].algo_bytes += algo_bytes;
    StxPendingEvent pe;
    pe.start = take_event(ctx);
    pe.stop = take_event(ctx);
    pe.entry = idx;
    if (!attached) hipEventRecord(pe.start, stream);
    ctx->prof_pending.push_back(pe);
    pending = (int)ctx->prof_pending.size() - 1;
}

StxProfScope::~StxProfScope()
{
    if (pending >= 0 && !attached) hipEventRecord(ctx->prof_pending[pending].stop, stream);
}

hipEvent_t StxProfScope::start() const { return pending >= 0 ? ctx->prof_pending[pending].start : nullptr; }
hipEvent_t StxProfScope::stop() const { return pending >= 0 ? ctx->prof_pending[pending].stop : nullptr; }

static void prof_collect(stx_ctx* ctx)
{
    if (ctx->prof_pending.empty()) return;
    hipStreamSynchronize(ctx->stream);
    hipStreamSynchronize(ctx->aux_stream);  // the ROI pass is bracketed on the side stream it runs on
    for (auto& pe : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.start, pe.stop) == hipSuccess) ctx->prof[pe.entry].total_ms += ms;
        ctx->event_pool.push_back(pe.start);
        ctx->event_pool.push_back(pe.stop);
    }
    ctx->prof_pending.clear();
}

STX_EXPORT int stx_prof_enable(stx_ctx* ctx, int on)
{
    if (!ctx) return stx_fail(STX_ERR_INVALID, "ctx is null");
    STX_TRY(stx_set_device(ctx));
    if (!on) prof_collect(ctx);
    ctx->prof_on = on != 0;
    return STX_OK;
}

STX_EXPORT int stx_prof_reset(stx_ctx* ctx)
{
    if (!ctx) return stx_fail(STX_ERR_INVALID, "ctx is null");
    STX_TRY(stx_set_device(ctx));
    prof_collect(ctx);
    ctx->prof.clear();
    ctx->prof_index.clear();
    return STX_OK;
}

STX_EXPORT int stx_prof_count(stx_ctx* ctx, int* out_n)
{
    if (!ctx || !out_n) return stx_fail(STX_ERR_INVALID, "null argument");
    STX_TRY(stx_set_device(ctx));
    prof_collect(ctx);
    *out_n = (int)ctx->prof.size();
    return STX_OK;
}

STX_EXPORT int stx_prof_get(stx_ctx* ctx, int index, char* name, int name_cap, int64_t* calls, double* total_ms,
                            double* algo_bytes)
{
    if (!ctx) return stx_fail(STX_ERR_INVALID, "ctx is null");
    if (index < 0 || index >= (int)ctx->prof.size()) return stx_fail(STX_ERR_INVALID, "profile index out of range");
    const StxProfEntry& e = ctx->prof[index];
    if (name && name_cap > 0) {
        strncpy(name, e.name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (calls) *calls = e.calls;
    if (total_ms) *total_ms = e.total_ms;
    if (algo_bytes) *algo_bytes = e.algo_bytes;
    return STX_OK;
}

STX_EXPORT int stx_mark(stx_ctx* ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= 16) return stx_fail(STX_ERR_INVALID, "bad mark slot");
    STX_TRY(stx_set_device(ctx));
    if (!ctx->marks[slot]) STX_HIP(hipEventCreate(&ctx->marks[slot]));
    STX_HIP(hipEventRecord(ctx->marks[slot], ctx->stream));
    return STX_OK;
}

STX_EXPORT int stx_mark_elapsed_ms(stx_ctx* ctx, int a, int b, double* out_ms)
{
    if (!ctx || a < 0 || a >= 16 || b < 0 || b >= 16 || !out_ms || !ctx->marks[a] || !ctx->marks[b])
        return stx_fail(STX_ERR_INVALID, "bad mark slots");
    STX_TRY(stx_set_device(ctx));
    STX_HIP(hipEventSynchronize(ctx->marks[b]));
    float ms = 0.f;
    STX_HIP(hipEventElapsedTime(&ms, ctx->marks[a], ctx->marks[b]));
    *out_ms = ms;
    return STX_OK;
}

// ---------------------------------------------------------------------------------------------
// device images
// ---------------------------------------------------------------------------------------------
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int stx_buf_new(stx_ctx* ctx, int w, int h, int c, int elem, stx_buf** out)
{
    if (w <= 0 || h <= 0 || c <= 0 || c > 4 || elem < STX_U8 || elem > STX_F32)
        return stx_fail(STX_ERR_INVALID, "bad image geometry %dx%dx%d elem %d", w, h, c, elem);
    std::unique_ptr<stx_buf> b(new stx_buf());
    b->ctx = ctx;
    b->w = w; b->h = h; b->c = c; b->elem = elem;
    // rows are 64-byte aligned and hold a whole number of 8-pixel groups (kernels store 4 or 8 px per lane)
    b->stride = align_up(align_up((size_t)w, 8) * c * stx_elem_bytes(elem), 64);
    // 64 bytes in front and 64 behind: the gather kernels read whole aligned windows around the first / last pixels of a row —
    // also where an 8-pixel group of a lane lies partly left of the image (up to 21 bytes before row 0), see mb_level0_pk_kernel
    STX_TRY(stx_dev_alloc(ctx, STX_BUF_FRONT_PAD + b->stride * h + 64, &b->base));
    b->ptr = (uint8_t*)b->base + STX_BUF_FRONT_PAD;
    *out = b.release();
    return STX_OK;
}

void stx_buf_retain(stx_buf* b) { b->refs.fetch_add(1); }

void stx_buf_release(stx_buf* b)
{
    if (!b) return;
    if (b->refs.fetch_sub(1) != 1) return;
    if (b->parent) stx_buf_release(b->parent);
    else stx_dev_free(b->ctx, b->base);
    delete b;
}

STX_EXPORT int stx_buf_alloc(stx_ctx* ctx, int w, int h, int channels, int elem, stx_buf** out)
{
    if (!ctx || !out) return stx_fail(STX_ERR_INVALID, "null argument");
    STX_TRY(stx_set_device(ctx));
    return stx_buf_new(ctx, w, h, channels, elem, out);
}

// true when [p, p + bytes) is page-locked host memory known to the runtime (stx_host_alloc, hipHostMalloc, hipHostRegister)
static bool is_pinned_host(const void* p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an unregistered pointer is an expected answer, not a sticky error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

static int buf_from_host(stx_ctx* ctx, const void* host, size_t host_stride, int w, int h, int channels, int elem, bool wait,
                         stx_buf** out)
{
    if (!ctx || !host || !out) return stx_fail(STX_ERR_INVALID, "null argument");
    STX_TRY(stx_set_device(ctx));
    size_t row = (size_t)w * channels * stx_elem_bytes(elem);
    if (host_stride < row) return stx_fail(STX_ERR_INVALID, "host stride %zu < row bytes %zu", host_stride, row);
    stx_buf* b = nullptr;
    STX_TRY(stx_buf_new(ctx, w, h, channels, elem, &b));
    hipError_t e = hipMemcpy2DAsync(b->ptr, b->stride, host, host_stride, row, h, hipMemcpyHostToDevice, ctx->stream);
    // the host buffer is only borrowed for this call — unless the caller asked for the asynchronous form and the
    // memory is page-locked (a pageable source is staged by the runtime; waiting keeps that case simple and safe)
    if (e == hipSuccess && (wait || !is_pinned_host(host))) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        stx_buf_release(b);
        return stx_fail(STX_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
    }
    if (channels == 1 && elem == STX_U8) {  // masks: remember whether every byte is 0 or 255 (packed blend kernels)
        bool binary = true;
        for (int y = 0; y < h && binary; y++) {
            const uint8_t* r = (const uint8_t*)host + (size_t)y * host_stride;
            unsigned bad = 0;
            for (int x = 0; x < w; x++) bad |= (unsigned)((r[x] + 1) & 0xfe);  // 0 -> 0, 255 -> 0, else nonzero
            binary = bad == 0;
        }
        b->mask_binary = binary ? 1 : 0;
    }
    *out = b;
    return STX_OK;
}

STX_EXPORT int stx_buf_from_host(stx_ctx* ctx, const void* host, size_t host_stride, int w, int h, int channels,
                                 int elem, stx_buf** out)
{
    return buf_from_host(ctx, host, host_stride, w, h, channels, elem, true, out);
}

STX_EXPORT int stx_buf_from_host_async(stx_ctx* ctx, const void* host, size_t host_stride, int w, int h, int channels,
                                       int elem, stx_buf** out)
{
    return buf_from_host(ctx, host, host_stride, w, h, channels, elem, false, out);
}

// Page-locked host memory for the frames a decoder produces and for read-backs: copies from / to it run at PCIe
// rate without the driver's staging through pageable memory (next row N3: staging of the source frames).
STX_EXPORT int stx_host_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0) return stx_fail(STX_ERR_INVALID, "bad argument");
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return stx_fail(STX_ERR_OOM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    *out = p;
    return STX_OK;
}

STX_EXPORT int stx_host_free(void* p)
{
    if (p) hipHostFree(p);
    return STX_OK;
}

static int buf_to_host(const stx_buf* buf, void* host, size_t host_stride, bool wait)
{
    if (!buf || !host) return stx_fail(STX_ERR_INVALID, "null argument");
    STX_TRY(stx_set_device(buf->ctx));
    size_t row = (size_t)buf->w * buf->c * stx_elem_bytes(buf->elem);
    if (host_stride < row) return stx_fail(STX_ERR_INVALID, "host stride %zu < row bytes %zu", host_stride, row);
    STX_HIP(hipMemcpy2DAsync(host, host_stride, buf->ptr, buf->stride, row, buf->h, hipMemcpyDeviceToHost,
                             buf->ctx->stream));
    if (wait || !is_pinned_host(host)) STX_HIP(hipStreamSynchronize(buf->ctx->stream));
    return STX_OK;
}

STX_EXPORT int stx_buf_to_host(const stx_buf* buf, void* host, size_t host_stride) { return buf_to_host(buf, host, host_stride, true); }

STX_EXPORT int stx_buf_to_host_async(const stx_buf* buf, void* host, size_t host_stride)
{
    return buf_to_host(buf, host, host_stride, false);
}

STX_EXPORT int stx_buf_view(const stx_buf* buf, int x, int y, int w, int h, stx_buf** out)
{
    if (!buf || !out) return stx_fail(STX_ERR_INVALID, "null argument");
    if (x < 0 || y < 0 || w <= 0 || h <= 0 || x + w > buf->w || y + h > buf->h)
        return stx_fail(STX_ERR_INVALID, "view (%d,%d,%d,%d) outside %dx%d", x, y, w, h, buf->w, buf->h);
    stx_buf* root = const_cast<stx_buf*>(buf);
    stx_buf* v = new stx_buf();
    v->ctx = buf->ctx;
    v->base = buf->base;
    v->ptr = buf->ptr + (size_t)y * buf->stride + (size_t)x * buf->c * stx_elem_bytes(buf->elem);
    v->w = w; v->h = h; v->c = buf->c; v->elem = buf->elem;
    v->stride = buf->stride;
    v->parent = root;
    v->mask_binary = buf->mask_binary;
    stx_buf_retain(root);
    *out = v;
    return STX_OK;
}

STX_EXPORT int stx_buf_info(const stx_buf* buf, int64_t info[6])
{
    if (!buf || !info) return stx_fail(STX_ERR_INVALID, "null argument");
    info[0] = buf->w; info[1] = buf->h; info[2] = buf->c; info[3] = buf->elem;
    info[4] = (int64_t)buf->stride; info[5] = buf->ctx->device;
    return STX_OK;
}

STX_EXPORT int stx_buf_flags(const stx_buf* buf, int* out_flags)
{
    if (!buf || !out_flags) return stx_fail(STX_ERR_INVALID, "null argument");
    *out_flags = buf->mask_binary ? STX_CONTRIB_U8_BINARY : 0;
    return STX_OK;
}

STX_EXPORT int stx_buf_device_ptr(const stx_buf* buf, void** out)
{
    if (!buf || !out) return stx_fail(STX_ERR_INVALID, "null argument");
    *out = buf->ptr;
    return STX_OK;
}

STX_EXPORT int stx_buf_free(stx_buf* buf)
{
    stx_buf_release(buf);
    return STX_OK;
}

// ---------------------------------------------------------------------------------------------
// "next" rows: exposure gain between warp and feed (N1), timelapse frames (N4)
// ---------------------------------------------------------------------------------------------
STX_EXPORT int stx_gain_apply(stx_ctx* ctx, stx_buf* img, const float gains_bgr[3])
{
    if (!ctx || !img || !gains_bgr) return stx_fail(STX_ERR_INVALID, "null argument");
    if (img->elem != STX_U8 || img->c != 3) return stx_fail(STX_ERR_INVALID, "gain apply needs a u8x3 image");
    if (img->ctx != ctx) return stx_fail(STX_ERR_INVALID, "image belongs to another context");
    STX_TRY(stx_set_device(ctx));
    return stx_launch_gain_apply(ctx, img, gains_bgr);
}

STX_EXPORT int stx_timelapse_frame(stx_ctx* ctx, const stx_buf* img, int tlx, int tly, const int dst_roi_xywh[4], stx_buf** out_frame)
{
    if (!ctx || !img || !dst_roi_xywh || !out_frame) return stx_fail(STX_ERR_INVALID, "null argument");
    if (img->ctx != ctx) return stx_fail(STX_ERR_INVALID, "image belongs to another context");
    const int rx = dst_roi_xywh[0], ry = dst_roi_xywh[1], rw = dst_roi_xywh[2], rh = dst_roi_xywh[3];
    if (rw <= 0 || rh <= 0) return stx_fail(STX_ERR_INVALID, "empty timelapse roi %dx%d", rw, rh);
    STX_TRY(stx_set_device(ctx));
    stx_buf* f = nullptr;
    STX_TRY(stx_buf_new(ctx, rw, rh, img->c, img->elem, &f));
    // Timelapser::process: dst_.setTo(0); img.copyTo(dst_(Rect(tl - dst_roi_.tl(), img.size()))), clipped to the roi
    hipError_t e = hipMemsetAsync(f->ptr, 0, f->stride * (size_t)rh, ctx->stream);
    const int x0 = std::max(tlx, rx), y0 = std::max(tly, ry);
    const int x1 = std::min(tlx + img->w, rx + rw), y1 = std::min(tly + img->h, ry + rh);
    const size_t px = (size_t)img->c * stx_elem_bytes(img->elem);
    if (e == hipSuccess && x1 > x0 && y1 > y0)
        e = hipMemcpy2DAsync(f->ptr + (size_t)(y0 - ry) * f->stride + (size_t)(x0 - rx) * px, f->stride,
                             img->ptr + (size_t)(y0 - tly) * img->stride + (size_t)(x0 - tlx) * px, img->stride,
                             (size_t)(x1 - x0) * px, (size_t)(y1 - y0), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) { stx_buf_release(f); return stx_fail(STX_ERR_HIP, "timelapse frame: %s", hipGetErrorString(e)); }
    *out_frame = f;
    return STX_OK;
}

// interpolationLinear<ufixedpoint16>::getCoeffs of cv::resize(INTER_LINEAR_EXACT) [OCV-MEM]: all in double precision
// (softdouble upstream = IEEE double; this file is compiled with -ffp-contract=off)
static void linear_exact_table(int src_n, int dst_n, std::vector<int>& t)
{
    t.resize(2 * (size_t)dst_n);
    const double inv_scale = (double)dst_n / (double)src_n;
    const double scale = 1.0 / inv_scale;
    for (int v = 0; v < dst_n; v++) {
        const double fval = scale * ((double)v + 0.5) - 0.5;
        const int ival = (int)std::floor(fval);
        int ofs = 0, c1 = 0, interior = 0;
        if (ival >= 0 && src_n > 1) {
            if (ival < src_n - 1) { ofs = ival; c1 = (int)std::nearbyint((fval - (double)ival) * 256.0); interior = 1; }
            else ofs = src_n - 1;
        }
        t[2 * (size_t)v] = ofs;
        t[2 * (size_t)v + 1] = c1 | (interior << 16);
    }
}

// small host array -> device through the context's pinned ring (asynchronous); the caller frees *d_out
static int upload_small(stx_ctx* ctx, const void* h, size_t bytes, void** d_out)
{
    void* d = nullptr;
    STX_TRY(stx_dev_alloc(ctx, std::max<size_t>(bytes, 4), &d));
    const int rc = stx_stage_upload(ctx, d, h, bytes);
    if (rc != STX_OK) { stx_dev_free(ctx, d); return rc; }
    *d_out = d;
    return STX_OK;
}

static int resize_impl(stx_ctx* ctx, const stx_buf* src, int dw, int dh, bool dilate, const stx_buf* andmask, stx_buf** out)
{
    if (src->elem != STX_U8 || (src->c != 1 && src->c != 3)) return stx_fail(STX_ERR_UNSUPPORTED, "resize needs a u8x1 or u8x3 image");
    if (dw <= 0 || dh <= 0) return stx_fail(STX_ERR_INVALID, "resize to %dx%d", dw, dh);
    if (src->ctx->device != ctx->device) return stx_fail(STX_ERR_INVALID, "image lives on another device");
    std::vector<int> xt, yt;
    linear_exact_table(src->w, dw, xt);
    linear_exact_table(src->h, dh, yt);
    const size_t nx = xt.size();
    xt.resize((nx + 7) & ~(size_t)7, 0);  // entries in whole groups of 4 columns (the 4-pixel seam kernel reads 4 at once), 32-byte rows
    std::vector<int> both(xt);
    both.insert(both.end(), yt.begin(), yt.end());
    void* d_tab = nullptr;
    STX_TRY(upload_small(ctx, both.data(), both.size() * sizeof(int), &d_tab));
    stx_buf* dst = nullptr;
    int rc = stx_buf_new(ctx, dw, dh, src->c, STX_U8, &dst);
    if (rc == STX_OK) rc = stx_launch_resize_exact(ctx, src, dst, (const int*)d_tab, (const int*)d_tab + xt.size(), dilate, andmask);
    stx_dev_free(ctx, d_tab);  // stream-ordered reuse
    if (rc != STX_OK) { stx_buf_release(dst); return rc; }
    *out = dst;
    return STX_OK;
}

// coefficient set-up of cv::resize(INTER_LINEAR) for CV_32F [OCV-MEM]: f = (float)((d + 0.5) * scale - 0.5), s = floor(f),
// f -= s; horizontal offsets are clamped with f = 0 at both ends, vertical ones are not (rows are clamped when fetched)
static void linear_f32_table(int src_n, int dst_n, bool clamp_offsets, std::vector<int>& t)
{
    t.resize(2 * (size_t)dst_n);
    const double scale = 1.0 / ((double)dst_n / (double)src_n);
    for (int d = 0; d < dst_n; d++) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        int sidx = (int)std::floor(f);
        f = f - (float)sidx;
        if (clamp_offsets) {
            if (sidx < 0) { sidx = 0; f = 0.f; }
            else if (sidx >= src_n - 1) { sidx = src_n - 1; f = 0.f; }
        }
        int bits;
        memcpy(&bits, &f, 4);
        t[2 * (size_t)d] = sidx;
        t[2 * (size_t)d + 1] = bits;
    }
}

// one image through the one-pixel-per-lane kernel: any alignment (views), any gain-map size
static int block_gain_plain(stx_ctx* ctx, stx_buf* img, const stx_buf* gain_map)
{
    std::vector<int> xt, yt;
    linear_f32_table(gain_map->w, img->w, true, xt);
    linear_f32_table(gain_map->h, img->h, false, yt);
    const size_t nx = xt.size();
    xt.resize((nx + 7) & ~(size_t)7, 0);  // entries in whole groups of 4 columns (the 4-pixel seam kernel reads 4 at once), 32-byte rows
    std::vector<int> both(xt);
    both.insert(both.end(), yt.begin(), yt.end());
    void* d_tab = nullptr;
    STX_TRY(upload_small(ctx, both.data(), both.size() * sizeof(int), &d_tab));
    const int rc = stx_launch_block_gain(ctx, img, gain_map, (const int*)d_tab, (const int*)d_tab + xt.size());
    stx_dev_free(ctx, d_tab);
    return rc;
}

static int block_gain_check(stx_ctx* ctx, const stx_buf* img, const stx_buf* gain_map)
{
    if (!gain_map) return stx_fail(STX_ERR_INVALID, "null argument");
    if (img && (img->elem != STX_U8 || img->c != 3)) return stx_fail(STX_ERR_INVALID, "block gain apply needs a u8x3 image");
    if (gain_map->elem != STX_F32 || (gain_map->c != 1 && gain_map->c != 3))
        return stx_fail(STX_ERR_INVALID, "the gain map must be f32x1 (gain_blocks) or f32x3 (channel_blocks)");
    if ((img && img->ctx != ctx) || gain_map->ctx->device != ctx->device) return stx_fail(STX_ERR_INVALID, "buffers belong to another context");
    return STX_OK;
}

// flags_or_null[i] & STX_GAIN_MAP_BOUNDED: the caller has checked that every gain of map i is finite and |g| < 2^31 / 255 (then no
// product p * g can leave the int range, and the kernel drops the cvRound overflow test)
STX_EXPORT int stx_block_gain_apply_batch(stx_ctx* ctx, int n, stx_buf* const* imgs, const stx_buf* const* gain_maps,
                                          const int* full_wh_xy0, const int* flags_or_null)
{
    if (!ctx || n < 0 || (n > 0 && (!imgs || !gain_maps))) return stx_fail(STX_ERR_INVALID, "bad argument");
    if (n == 0) return STX_OK;
    STX_TRY(stx_set_device(ctx));
    for (int i = 0; i < n; i++) {
        if (!imgs[i]) return stx_fail(STX_ERR_INVALID, "null argument");
        STX_TRY(block_gain_check(ctx, imgs[i], gain_maps[i]));
        if (full_wh_xy0) {
            const int* q = full_wh_xy0 + 4 * i;
            if (q[2] < 0 || q[3] < 0 || q[2] + imgs[i]->w > q[0] || q[3] + imgs[i]->h > q[1])
                return stx_fail(STX_ERR_INVALID, "image %d: rectangle (%d,%d,%dx%d) outside the full image %dx%d", i, q[2], q[3], imgs[i]->w, imgs[i]->h, q[0], q[1]);
        }
    }
    // the batched kernels want whole buffers of the library's own (dword rows, 4-pixel groups) and gain maps of block size (their
    // horizontally interpolated rows are kept: gh x w floats); anything else takes the plain kernel, one image at a time
    std::vector<stx_buf*> bi;
    std::vector<const stx_buf*> bg;
    std::vector<int> sub, fast, plain;
    size_t scratch = 0;
    std::vector<size_t> offH, offY;
    // classify and validate EVERY image before anything is launched: the product is written in place, so a call that fails must not
    // have multiplied some of its images already (a caller could not retry it)
    for (int i = 0; i < n; i++) {
        const stx_buf* im = imgs[i];
        const bool whole = !im->parent && ((uintptr_t)im->ptr & 3) == 0 && (im->stride & 3) == 0 && (size_t)((im->w + 3) & ~3) * 3 <= im->stride;
        const size_t hbytes = (size_t)gain_maps[i]->h * ((im->w + 3) & ~3) * gain_maps[i]->c * sizeof(float);
        int first_c = -1;
        for (int j = 0; j < i && first_c < 0; j++)
            if (std::find(plain.begin(), plain.end(), j) == plain.end()) first_c = gain_maps[j]->c;
        const bool same_c = first_c < 0 || gain_maps[i]->c == first_c;
        if (!whole || hbytes > ((size_t)64 << 20) || !same_c) {
            if (full_wh_xy0 && (full_wh_xy0[4 * i] != im->w || full_wh_xy0[4 * i + 1] != im->h))
                return stx_fail(STX_ERR_UNSUPPORTED, "image %d: a rectangle of a larger image must be a whole buffer with a block-sized gain map", i);
            plain.push_back(i);
        }
    }
    for (int i : plain) STX_TRY(block_gain_plain(ctx, imgs[i], gain_maps[i]));
    for (int i = 0; i < n; i++) {
        if (std::find(plain.begin(), plain.end(), i) != plain.end()) continue;
        const stx_buf* im = imgs[i];
        const size_t hbytes = (size_t)gain_maps[i]->h * ((im->w + 3) & ~3) * gain_maps[i]->c * sizeof(float);
        bi.push_back(imgs[i]); bg.push_back(gain_maps[i]);
        for (int k = 0; k < 4; k++) sub.push_back(full_wh_xy0 ? full_wh_xy0[4 * i + k] : (k == 0 ? im->w : (k == 1 ? im->h : 0)));
        fast.push_back(flags_or_null && (flags_or_null[i] & STX_GAIN_MAP_BOUNDED) ? 1 : 0);
        offH.push_back(scratch); scratch += align_up(hbytes, 256);
        offY.push_back(scratch); scratch += align_up((size_t)im->h * 8, 256);
    }
    if (bi.empty()) return STX_OK;
    void* d = nullptr;
    STX_TRY(stx_dev_alloc(ctx, scratch, &d));
    std::vector<float*> Hs(bi.size());
    std::vector<void*> yts(bi.size());
    for (size_t i = 0; i < bi.size(); i++) { Hs[i] = (float*)((uint8_t*)d + offH[i]); yts[i] = (uint8_t*)d + offY[i]; }
    const int rc = stx_launch_block_gain_batch(ctx, (int)bi.size(), bi.data(), bg.data(), sub.data(), Hs.data(), yts.data(), fast.data());
    stx_dev_free(ctx, d);  // stream-ordered reuse
    return rc;
}

STX_EXPORT int stx_block_gain_apply(stx_ctx* ctx, stx_buf* img, const stx_buf* gain_map)
{
    if (!ctx) return stx_fail(STX_ERR_INVALID, "null argument");
    return stx_block_gain_apply_batch(ctx, 1, &img, &gain_map, nullptr, nullptr);
}

STX_EXPORT int stx_resize_linear_exact(stx_ctx* ctx, const stx_buf* src, int dst_w, int dst_h, stx_buf** out)
{
    if (!ctx || !src || !out) return stx_fail(STX_ERR_INVALID, "null argument");
    STX_TRY(stx_set_device(ctx));
    return resize_impl(ctx, src, dst_w, dst_h, false, nullptr, out);
}

STX_EXPORT int stx_seam_mask_resize(stx_ctx* ctx, const stx_buf* seam_mask, const stx_buf* final_mask, stx_buf** out)
{
    if (!ctx || !seam_mask || !final_mask || !out) return stx_fail(STX_ERR_INVALID, "null argument");
    if (seam_mask->c != 1 || seam_mask->elem != STX_U8 || final_mask->c != 1 || final_mask->elem != STX_U8)
        return stx_fail(STX_ERR_INVALID, "seam masks are u8x1");
    STX_TRY(stx_set_device(ctx));
    if (seam_mask->ctx->device == ctx->device && final_mask->ctx->device == ctx->device) {  // the one-launch form first
        stx_buf* d = nullptr;
        STX_TRY(stx_buf_new(ctx, final_mask->w, final_mask->h, 1, STX_U8, &d));
        bool done = false;
        const int rc = stx_launch_seam_resize_lds(ctx, 1, &seam_mask, &final_mask, &d, nullptr, &done);
        if (rc == STX_OK && done) { *out = d; return STX_OK; }
        stx_buf_release(d);
        if (rc != STX_OK) return rc;
    }
    return resize_impl(ctx, seam_mask, final_mask->w, final_mask->h, true, final_mask, out);
}

// SeamFinder.resize for all images of a panorama: one table upload, one dilate launch and one resize launch per 16 images.
// Falls back to the per-image call when a buffer does not meet the 4-pixel kernel's alignment needs.
// sub: null -> final_masks[i] is the whole final mask; else {full_w, full_h, x0, y0} per image: final_masks[i] is the
// rectangle at (x0, y0) of a final mask of size full_w x full_h (the seam mask is enlarged to THAT size, only the
// rectangle is produced; x0 a multiple of 4)
static int seam_resize_batch_impl(stx_ctx* ctx, int n, const stx_buf* const* seam_masks, const stx_buf* const* final_masks,
                                  const int* sub, stx_buf** outs)
{
    if (!ctx || n < 0 || (n > 0 && (!seam_masks || !final_masks || !outs))) return stx_fail(STX_ERR_INVALID, "null argument");
    STX_TRY(stx_set_device(ctx));
    bool fast = true;
    for (int i = 0; i < n && sub; i++) {
        const int* q = sub + 4 * i;
        if (!final_masks[i] || q[2] < 0 || q[3] < 0 || (q[2] & 3) || q[2] + final_masks[i]->w > q[0] || q[3] + final_masks[i]->h > q[1])
            return stx_fail(STX_ERR_INVALID, "seam mask rectangle %d outside its final mask (or x0 not a multiple of 4)", i);
    }
    for (int i = 0; i < n; i++) {
        const stx_buf *s = seam_masks[i], *m = final_masks[i];
        if (!s || !m) return stx_fail(STX_ERR_INVALID, "null argument");
        if (s->c != 1 || s->elem != STX_U8 || m->c != 1 || m->elem != STX_U8) return stx_fail(STX_ERR_INVALID, "seam masks are u8x1");
        if (s->ctx->device != ctx->device || m->ctx->device != ctx->device) return stx_fail(STX_ERR_INVALID, "image lives on another device");
        fast = fast && ((uintptr_t)m->ptr & 3) == 0 && (m->stride & 3) == 0 && (size_t)((m->w + 3) & ~3) <= m->stride;
    }
    if (n == 0) return STX_OK;
    // final masks the 4-pixel kernel cannot read in place (a view that starts on an odd byte, a pitch that is not a multiple
    // of 4): whole masks go through the per-image call; rectangles (sub) are first copied into aligned buffers of their own
    std::vector<stx_buf*> aligned;  // released on every path below
    std::vector<const stx_buf*> fm(final_masks, final_masks + n);
    struct ReleaseAll { std::vector<stx_buf*>& v; ~ReleaseAll() { for (stx_buf* b : v) stx_buf_release(b); } } release_aligned{aligned};
    if (!fast && !sub) {
        for (int i = 0; i < n; i++) {
            const int rc1 = stx_seam_mask_resize(ctx, seam_masks[i], final_masks[i], &outs[i]);
            if (rc1 != STX_OK) {  // hand nothing out: release what the earlier iterations produced
                for (int j = 0; j < i; j++) { stx_buf_release(outs[j]); outs[j] = nullptr; }
                return rc1;
            }
        }
        return STX_OK;
    }
    if (!fast) {
        for (int i = 0; i < n; i++) {
            const stx_buf* m = final_masks[i];
            if (((uintptr_t)m->ptr & 3) == 0 && (m->stride & 3) == 0 && (size_t)((m->w + 3) & ~3) <= m->stride) continue;
            stx_buf* c = nullptr;
            STX_TRY(stx_buf_new(ctx, m->w, m->h, 1, STX_U8, &c));
            aligned.push_back(c);
            STX_HIP(hipMemcpy2DAsync(c->ptr, c->stride, m->ptr, m->stride, (size_t)m->w, (size_t)m->h, hipMemcpyDeviceToDevice, ctx->stream));
            c->mask_binary = m->mask_binary;
            fm[i] = c;
        }
    }
    final_masks = fm.data();
    {
        // the one-launch form: nothing to upload, no scratch (whole buffers of the library's own qualify; anything else: the tables below)
        std::vector<stx_buf*> d1(n, nullptr);
        int rc1 = STX_OK;
        for (int i = 0; i < n && rc1 == STX_OK; i++) rc1 = stx_buf_new(ctx, final_masks[i]->w, final_masks[i]->h, 1, STX_U8, &d1[i]);
        bool done = false;
        if (rc1 == STX_OK) rc1 = stx_launch_seam_resize_lds(ctx, n, seam_masks, final_masks, d1.data(), sub, &done);
        if (rc1 != STX_OK || !done) {
            for (stx_buf* d : d1) stx_buf_release(d);
            if (rc1 != STX_OK) return rc1;
        } else {
            for (int i = 0; i < n; i++) outs[i] = d1[i];
            return STX_OK;
        }
    }
    // tables of all images in one upload: per image xt (dw rounded up to 4 entries) then yt
    std::vector<int> all;
    std::vector<size_t> xoff(n), yoff(n);
    for (int i = 0; i < n; i++) {
        std::vector<int> xt, yt;
        linear_exact_table(seam_masks[i]->w, sub ? sub[4 * i] : final_masks[i]->w, xt);
        linear_exact_table(seam_masks[i]->h, sub ? sub[4 * i + 1] : final_masks[i]->h, yt);
        if (sub) {  // the rectangle's slice of the tables (2 ints per destination column / row)
            const int x0 = sub[4 * i + 2], y0 = sub[4 * i + 3];
            xt = std::vector<int>(xt.begin() + 2 * (size_t)x0, xt.begin() + 2 * (size_t)(x0 + final_masks[i]->w));
            yt = std::vector<int>(yt.begin() + 2 * (size_t)y0, yt.begin() + 2 * (size_t)(y0 + final_masks[i]->h));
        }
        xt.resize((xt.size() + 7) & ~(size_t)7, 0);
        xoff[i] = all.size();
        all.insert(all.end(), xt.begin(), xt.end());
        yoff[i] = all.size();
        all.insert(all.end(), yt.begin(), yt.end());
        all.resize((all.size() + 7) & ~(size_t)7, 0);  // keep every table 32-byte aligned
    }
    void* d_tab = nullptr;
    STX_TRY(upload_small(ctx, all.data(), all.size() * sizeof(int), &d_tab));
    std::vector<stx_buf*> dsts(n, nullptr);
    std::vector<void*> tmps(n, nullptr);
    std::vector<uint8_t*> tptr(n);
    std::vector<size_t> tstride(n);
    std::vector<const int*> dx(n), dy(n);
    int rc = STX_OK;
    for (int i = 0; i < n && rc == STX_OK; i++) {
        rc = stx_buf_new(ctx, final_masks[i]->w, final_masks[i]->h, 1, STX_U8, &dsts[i]);
        tstride[i] = ((size_t)seam_masks[i]->w + 63) & ~(size_t)63;
        if (rc == STX_OK) rc = stx_dev_alloc(ctx, tstride[i] * seam_masks[i]->h, &tmps[i]);
        tptr[i] = (uint8_t*)tmps[i];
        dx[i] = (const int*)d_tab + xoff[i];
        dy[i] = (const int*)d_tab + yoff[i];
    }
    if (rc == STX_OK) rc = stx_launch_seam_resize_batch(ctx, n, seam_masks, final_masks, dsts.data(), dx.data(), dy.data(), tptr.data(), tstride.data());
    for (void* t : tmps) stx_dev_free(ctx, t);  // stream-ordered reuse
    stx_dev_free(ctx, d_tab);
    if (rc != STX_OK) {
        for (stx_buf* d : dsts) stx_buf_release(d);
        return rc;
    }
    for (int i = 0; i < n; i++) outs[i] = dsts[i];
    return STX_OK;
}

STX_EXPORT int stx_seam_mask_resize_batch(stx_ctx* ctx, int n, const stx_buf* const* seam_masks, const stx_buf* const* final_masks,
                                          stx_buf** outs)
{
    return seam_resize_batch_impl(ctx, n, seam_masks, final_masks, nullptr, outs);
}

STX_EXPORT int stx_seam_mask_resize_batch_sub(stx_ctx* ctx, int n, const stx_buf* const* seam_masks, const stx_buf* const* final_masks,
                                              const int* full_wh_xy0, stx_buf** outs)
{
    if (!full_wh_xy0 && n > 0) return stx_fail(STX_ERR_INVALID, "null argument");
    return seam_resize_batch_impl(ctx, n, seam_masks, final_masks, full_wh_xy0, outs);
}

// ---------------------------------------------------------------------------------------------
// projector: ProjectorBase::setCameraParams, AffineWarper::getRTfromHomogeneous
// ---------------------------------------------------------------------------------------------
static void inv3x3_f32(const float* m, float* o)
{
    // cv::invert for a 3x3 CV_32F matrix: cofactors and determinant in double, cast to float
    auto M = [&](int i, int j) { return (double)m[i * 3 + j]; };
    double d = m[0] * (M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1)) - m[1] * (M(1, 0) * M(2, 2) - M(1, 2) * M(2, 0)) +
               m[2] * (M(1, 0) * M(2, 1) - M(1, 1) * M(2, 0));
    if (d == 0.) {
        for (int i = 0; i < 9; i++) o[i] = 0.f;
        return;
    }
    d = 1. / d;
    o[0] = (float)((M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1)) * d);
    o[1] = (float)((M(0, 2) * M(2, 1) - M(0, 1) * M(2, 2)) * d);
    o[2] = (float)((M(0, 1) * M(1, 2) - M(0, 2) * M(1, 1)) * d);
    o[3] = (float)((M(1, 2) * M(2, 0) - M(1, 0) * M(2, 2)) * d);
    o[4] = (float)((M(0, 0) * M(2, 2) - M(0, 2) * M(2, 0)) * d);
    o[5] = (float)((M(0, 2) * M(1, 0) - M(0, 0) * M(1, 2)) * d);
    o[6] = (float)((M(1, 0) * M(2, 1) - M(1, 1) * M(2, 0)) * d);
    o[7] = (float)((M(0, 1) * M(2, 0) - M(0, 0) * M(2, 1)) * d);
    o[8] = (float)((M(0, 0) * M(1, 1) - M(0, 1) * M(1, 0)) * d);
}

static void mul3x3_f32(const float* a, const float* b, float* d)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float t = a[i * 3] * b[j];
            t = t + a[i * 3 + 1] * b[3 + j];
            t = t + a[i * 3 + 2] * b[6 + j];
            d[i * 3 + j] = t;
        }
}

// ---------------------------------------------------------------------------------------------
// trig mode: which sinf / cosf the projectors follow (stx_device_math.h).  Process-wide like the libm it stands for; initialised
// from STITCHING_AMD_TRIG = exact | glibc | glibc-nofma on first use.
// ---------------------------------------------------------------------------------------------
static std::atomic<int> g_trig_mode{-1};

static int trig_mode_now()
{
    int m = g_trig_mode.load();
    if (m >= 0) return m;
    const char* e = getenv("STITCHING_AMD_TRIG");
    m = STX_TRIG_EXACT;
    if (e && !strcmp(e, "glibc")) m = STX_TRIG_GLIBC;
    else if (e && !strcmp(e, "glibc-nofma")) m = STX_TRIG_GLIBC_NOFMA;
    else if (e && *e && strcmp(e, "exact")) fprintf(stderr, "[stitching_amd] STITCHING_AMD_TRIG=%s is not one of exact, glibc, glibc-nofma: using exact\n", e);
    int expected = -1;
    g_trig_mode.compare_exchange_strong(expected, m);
    return g_trig_mode.load();
}

STX_EXPORT int stx_get_trig_mode(void) { return trig_mode_now(); }

STX_EXPORT int stx_set_trig_mode(int mode)
{
    if (mode < STX_TRIG_EXACT || mode > STX_TRIG_GLIBC_NOFMA) return stx_fail(STX_ERR_INVALID, "trig mode %d", mode);
    g_trig_mode.store(mode);  // (ROI cache entries carry their mode in the key)
    return STX_OK;
}

// remap mode: the interpolation model of the image samples (include/stitching_amd.h); process-wide, initialised from
// STITCHING_AMD_REMAP = q15 | float | float-fma on first use
static std::atomic<int> g_remap_mode{-1};

static int remap_mode_now()
{
    int m = g_remap_mode.load();
    if (m >= 0) return m;
    const char* e = getenv("STITCHING_AMD_REMAP");
    m = STX_REMAP_Q15;
    if (e && !strcmp(e, "float")) m = STX_REMAP_FLOAT;
    else if (e && !strcmp(e, "float-fma")) m = STX_REMAP_FLOAT_FMA;
    else if (e && *e && strcmp(e, "q15")) fprintf(stderr, "[stitching_amd] STITCHING_AMD_REMAP=%s is not one of q15, float, float-fma: using q15\n", e);
    int expected = -1;
    g_remap_mode.compare_exchange_strong(expected, m);
    return g_remap_mode.load();
}

STX_EXPORT int stx_get_remap_mode(void) { return remap_mode_now(); }

STX_EXPORT int stx_set_remap_mode(int mode)
{
    if (mode < STX_REMAP_Q15 || mode > STX_REMAP_FLOAT_FMA) return stx_fail(STX_ERR_INVALID, "remap mode %d", mode);
    g_remap_mode.store(mode);
    return STX_OK;
}

// pyrDown order of the fp32 weight pyramids (include/stitching_amd.h STX_PYRDOWN_*): process-wide, from STITCHING_AMD_PYRDOWN
// ("simd-hv", "simd-v-fma:8", ...) on first use; packed as mode | lanes << 8
static std::atomic<int> g_pyrdown{-1};

static int pyrdown_now()
{
    int m = g_pyrdown.load();
    if (m >= 0) return m;
    const char* e = getenv("STITCHING_AMD_PYRDOWN");
    int mode = STX_PYRDOWN_SCALAR, lanes = 4;
    if (e && *e) {
        std::string v(e);
        const size_t colon = v.find(':');
        if (colon != std::string::npos) { lanes = atoi(v.c_str() + colon + 1); v.resize(colon); }
        if (v == "simd-v") mode = STX_PYRDOWN_SIMD_V;
        else if (v == "simd-hv") mode = STX_PYRDOWN_SIMD_HV;
        else if (v == "simd-v-fma") mode = STX_PYRDOWN_SIMD_V | STX_PYRDOWN_FMA;
        else if (v == "simd-hv-fma") mode = STX_PYRDOWN_SIMD_HV | STX_PYRDOWN_FMA;
        else if (v != "scalar") fprintf(stderr, "[stitching_amd] STITCHING_AMD_PYRDOWN=%s is not scalar, simd-v, simd-hv, simd-v-fma or simd-hv-fma: using scalar\n", e);
        if (lanes != 4 && lanes != 8 && lanes != 16) { fprintf(stderr, "[stitching_amd] STITCHING_AMD_PYRDOWN lanes %d: using 4\n", lanes); lanes = 4; }
    }
    int expected = -1;
    g_pyrdown.compare_exchange_strong(expected, mode | (lanes << 8));
    return g_pyrdown.load();
}

STX_EXPORT int stx_get_pyrdown_mode(int* out_lanes)
{
    const int m = pyrdown_now();
    if (out_lanes) *out_lanes = m >> 8;
    return m & 255;
}

STX_EXPORT int stx_set_pyrdown_mode(int mode, int lanes)
{
    const bool known = mode == STX_PYRDOWN_SCALAR || (mode & ~STX_PYRDOWN_FMA) == STX_PYRDOWN_SIMD_V || (mode & ~STX_PYRDOWN_FMA) == STX_PYRDOWN_SIMD_HV;
    if (!known || (lanes != 4 && lanes != 8 && lanes != 16)) return stx_fail(STX_ERR_INVALID, "pyrDown mode %d, lanes %d", mode, lanes);
    g_pyrdown.store(mode | (lanes << 8));
    return STX_OK;
}

int stx_make_projector(int type, float scale, const float* K, const float* R, StxProjector* p)
{
    if (type < STX_WARP_PLANE || type >= STX_WARP_TYPE_COUNT)
        return stx_fail(STX_ERR_UNSUPPORTED, "warper type id %d is not implemented by this back end", type);
    if (!K || !R) return stx_fail(STX_ERR_INVALID, "K and R must be 3x3 fp32");
    for (int i = 0; i < 9; i++)
        if (!std::isfinite(K[i]) || !std::isfinite(R[i])) return stx_fail(STX_ERR_INVALID, "K/R contain non-finite values");
    p->type = type;
    p->scale = scale;
    p->trig = trig_mode_now();
    p->remap = remap_mode_now();
    // PyRotationWarper's constructor: "compressedPlaneA2B1" -> CompressedRectilinearWarper(2.0f, 1.0f), "...A1.5B1" -> (1.5f, 1.0f), ...
    static const struct { int family; float a; } kTypes[STX_WARP_TYPE_COUNT] = {
        {STX_F_PLANE, 1.f}, {STX_F_PLANE, 1.f}, {STX_F_CYLINDRICAL, 1.f}, {STX_F_SPHERICAL, 1.f}, {STX_F_FISHEYE, 1.f},
        {STX_F_STEREOGRAPHIC, 1.f}, {STX_F_CRECT, 2.0f}, {STX_F_CRECT, 1.5f}, {STX_F_CRECT_PORTRAIT, 2.0f},
        {STX_F_CRECT_PORTRAIT, 1.5f}, {STX_F_PANINI, 2.0f}, {STX_F_PANINI, 1.5f}, {STX_F_PANINI_PORTRAIT, 2.0f},
        {STX_F_PANINI_PORTRAIT, 1.5f}, {STX_F_MERCATOR, 1.f}, {STX_F_TRANSVERSE_MERCATOR, 1.f}};
    p->family = kTypes[type].family;
    p->a = kTypes[type].a;
    p->b = 1.0f;
    float Rm[9], T[3] = {0.f, 0.f, 0.f};
    if (type == STX_WARP_AFFINE) {
        // R' = (H with H[0,2] = H[1,2] = 0)^T ; T' = -(R' * (H[0,2], H[1,2], 0)); the caller's scale is kept
        // (cv::AffineWarper::create(scale) -> detail::AffineWarper(scale) : PlaneWarper(scale))
        float H[9];
        memcpy(H, R, sizeof(H));
        const float t0 = H[2], t1 = H[5];
        H[2] = 0.f;
        H[5] = 0.f;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rm[i * 3 + j] = H[j * 3 + i];
        for (int i = 0; i < 3; i++) {
            float v = Rm[i * 3] * t0;
            v = v + Rm[i * 3 + 1] * t1;
            v = v + Rm[i * 3 + 2] * 0.f;
            T[i] = v * -1.f;
        }
    } else {
        memcpy(Rm, R, sizeof(Rm));
    }
    memcpy(p->k, K, sizeof(p->k));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) p->rinv[i * 3 + j] = Rm[j * 3 + i];
    float kinv[9];
    inv3x3_f32(K, kinv);
    mul3x3_f32(Rm, kinv, p->r_kinv);
    mul3x3_f32(K, p->rinv, p->k_rinv);
    p->t[0] = T[0]; p->t[1] = T[1]; p->t[2] = T[2];
    return STX_OK;
}

// (int)float as x86 cvttss2si
static int trunc_i32(float v)
{
    if (!(v >= -2147483648.f && v < 2147483648.f)) return INT_MIN;
    return (int)v;
}

// PlaneProjector::mapForward for the 4 corners (PlaneWarper::detectResultRoi)
static void plane_forward(const StxProjector& p, float x, float y, float& u, float& v)
{
    const float* rk = p.r_kinv;
    float x_ = rk[0] * x;
    x_ = x_ + rk[1] * y;
    x_ = x_ + rk[2];
    float y_ = rk[3] * x;
    y_ = y_ + rk[4] * y;
    y_ = y_ + rk[5];
    float z_ = rk[6] * x;
    z_ = z_ + rk[7] * y;
    z_ = z_ + rk[8];
    float q = x_ / z_;
    q = q * (1 - p.t[2]);
    x_ = p.t[0] + q;
    q = y_ / z_;
    q = q * (1 - p.t[2]);
    y_ = p.t[1] + q;
    u = p.scale * x_;
    v = p.scale * y_;
}

static void finish_roi(const StxProjector& p, int w, int h, const float* mm, int* out_xywh)
{
    int tlx = trunc_i32(mm[0]), tly = trunc_i32(mm[1]), brx = trunc_i32(mm[2]), bry = trunc_i32(mm[3]);
    if (p.type == STX_WARP_SPHERICAL) {
        // SphericalWarper::detectResultRoi: include the poles when they project inside the image
        float tl_uf = (float)tlx, tl_vf = (float)tly, br_uf = (float)brx, br_vf = (float)bry;
        float x = p.rinv[1], y = p.rinv[4], z = p.rinv[7];
        if (y > 0.f) {
            float a = p.k[0] * x;
            a = a + p.k[1] * y;
            float x_ = a / z + p.k[2];
            float y_ = p.k[4] * y / z + p.k[5];
            if (x_ > 0.f && x_ < w && y_ > 0.f && y_ < h) {
                float pv = static_cast<float>(3.14159265358979323846 * p.scale);
                tl_uf = std::min(tl_uf, 0.f); tl_vf = std::min(tl_vf, pv);
                br_uf = std::max(br_uf, 0.f); br_vf = std::max(br_vf, pv);
            }
        }
        y = -p.rinv[4];
        if (y > 0.f) {
            float a = p.k[0] * x;
            a = a + p.k[1] * y;
            float x_ = a / z + p.k[2];
            float y_ = p.k[4] * y / z + p.k[5];
            if (x_ > 0.f && x_ < w && y_ > 0.f && y_ < h) {
                tl_uf = std::min(tl_uf, 0.f); tl_vf = std::min(tl_vf, 0.f);
                br_uf = std::max(br_uf, 0.f); br_vf = std::max(br_vf, 0.f);
            }
        }
        tlx = trunc_i32(tl_uf); tly = trunc_i32(tl_vf); brx = trunc_i32(br_uf); bry = trunc_i32(br_vf);
    }
    out_xywh[0] = tlx; out_xywh[1] = tly;
    out_xywh[2] = brx - tlx + 1; out_xywh[3] = bry - tly + 1;
}

static int rois_impl(stx_ctx* ctx, int n, const StxProjector* projs, const int* sizes_wh, int* out_xywh)
{
    std::vector<float> mm(4 * (size_t)n);
    std::vector<int> dev_idx;
    for (int i = 0; i < n; i++) {
        const int w = sizes_wh[2 * i], h = sizes_wh[2 * i + 1];
        if (w <= 0 || h <= 0) return stx_fail(STX_ERR_INVALID, "image size %dx%d", w, h);
        if (projs[i].type == STX_WARP_PLANE || projs[i].type == STX_WARP_AFFINE) {
            float mn_u = std::numeric_limits<float>::max(), mn_v = mn_u, mx_u = -mn_u, mx_v = -mn_u, u, v;
            const float xs[2] = {0.f, (float)(w - 1)}, ys[2] = {0.f, (float)(h - 1)};
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 2; b++) {
                    plane_forward(projs[i], xs[a], ys[b], u, v);
                    mn_u = std::min(mn_u, u); mn_v = std::min(mn_v, v);
                    mx_u = std::max(mx_u, u); mx_v = std::max(mx_v, v);
                }
            mm[4 * i] = mn_u; mm[4 * i + 1] = mn_v; mm[4 * i + 2] = mx_u; mm[4 * i + 3] = mx_v;
        } else {
            dev_idx.push_back(i);
        }
    }
    if (!dev_idx.empty()) {
        const int m = (int)dev_idx.size();
        std::vector<StxProjector> dp(m);
        std::vector<int> dsz(2 * (size_t)m);
        std::vector<float> dmm(4 * (size_t)m);
        for (int j = 0; j < m; j++) {
            dp[j] = projs[dev_idx[j]];
            dsz[2 * j] = sizes_wh[2 * dev_idx[j]];
            dsz[2 * j + 1] = sizes_wh[2 * dev_idx[j] + 1];
        }
        STX_TRY(stx_launch_roi_minmax(ctx, m, dp.data(), dsz.data(), dmm.data()));
        for (int j = 0; j < m; j++) memcpy(&mm[4 * dev_idx[j]], &dmm[4 * j], 16);
    }
    for (int i = 0; i < n; i++) finish_roi(projs[i], sizes_wh[2 * i], sizes_wh[2 * i + 1], &mm[4 * i], out_xywh + 4 * i);
    return STX_OK;
}

// ROI cache: the reference recomputes detectResultRoi inside every warp()/warpRoi() call
// (stitching/warper.py:44,59,80 build three warpers per image); we compute it once per camera.
struct RoiKey {
    int type, w, h, trig;
    float scale, K[9], R[9];
    bool operator<(const RoiKey& o) const { return memcmp(this, &o, sizeof(RoiKey)) < 0; }
};
static thread_local std::map<RoiKey, std::array<int, 4>>* g_roi_cache = nullptr;

static RoiKey make_key(int type, float scale, const float* K, const float* R, int w, int h)
{
    RoiKey k;
    memset(&k, 0, sizeof(k));
    k.type = type; k.w = w; k.h = h; k.scale = scale;
    k.trig = trig_mode_now();  // the forward maps of the per-pixel projector families call sinf / cosf
    memcpy(k.K, K, 36);
    memcpy(k.R, R, 36);
    return k;
}

static int roi_cached(stx_ctx* ctx, int type, float scale, const float* K, const float* R, int w, int h,
                      const StxProjector& proj, int* out)
{
    if (!g_roi_cache) g_roi_cache = new std::map<RoiKey, std::array<int, 4>>();
    RoiKey key = make_key(type, scale, K, R, w, h);
    auto it = g_roi_cache->find(key);
    if (it != g_roi_cache->end()) {
        memcpy(out, it->second.data(), 16);
        return STX_OK;
    }
    int sz[2] = {w, h};
    STX_TRY(rois_impl(ctx, 1, &proj, sz, out));
    if (g_roi_cache->size() > 8192) g_roi_cache->clear();
    (*g_roi_cache)[key] = {out[0], out[1], out[2], out[3]};
    return STX_OK;
}

STX_EXPORT int stx_warp_roi(stx_ctx* ctx, int type, float scale, const float K[9], const float R[9], int w, int h,
                            int out_xywh[4])
{
    if (!ctx || !out_xywh) return stx_fail(STX_ERR_INVALID, "null argument");
    STX_TRY(stx_set_device(ctx));
    StxProjector p;
    STX_TRY(stx_make_projector(type, scale, K, R, &p));
    if (w <= 0 || h <= 0) return stx_fail(STX_ERR_INVALID, "image size %dx%d", w, h);
    return roi_cached(ctx, type, scale, K, R, w, h, p, out_xywh);
}

STX_EXPORT int stx_warp_rois(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s,
                             const int* sizes_wh, int* out_xywh)
{
    if (!ctx || !K9s || !R9s || !sizes_wh || !out_xywh || n < 0) return stx_fail(STX_ERR_INVALID, "bad argument");
    if (n == 0) return STX_OK;
    STX_TRY(stx_set_device(ctx));
    std::vector<StxProjector> ps(n);
    for (int i = 0; i < n; i++) STX_TRY(stx_make_projector(type, scale, K9s + 9 * i, R9s + 9 * i, &ps[i]));
    STX_TRY(rois_impl(ctx, n, ps.data(), sizes_wh, out_xywh));
    if (!g_roi_cache) g_roi_cache = new std::map<RoiKey, std::array<int, 4>>();
    if (g_roi_cache->size() > 8192) g_roi_cache->clear();
    for (int i = 0; i < n; i++)
        (*g_roi_cache)[make_key(type, scale, K9s + 9 * i, R9s + 9 * i, sizes_wh[2 * i], sizes_wh[2 * i + 1])] = {
            out_xywh[4 * i], out_xywh[4 * i + 1], out_xywh[4 * i + 2], out_xywh[4 * i + 3]};
    return STX_OK;
}

static int warp_impl(stx_ctx* ctx, int type, float scale, const float* K, const float* R, const stx_buf* src, int sw,
                     int sh, bool want_img, bool want_mask, bool nearest_src, stx_buf** out_img, stx_buf** out_mask,
                     int* out_xywh)
{
    StxProjector p;
    STX_TRY(stx_make_projector(type, scale, K, R, &p));
    int roi[4];
    STX_TRY(roi_cached(ctx, type, scale, K, R, sw, sh, p, roi));
    if (roi[2] <= 0 || roi[3] <= 0 || (long long)roi[2] * roi[3] > (1ll << 33))
        return stx_fail(STX_ERR_INVALID, "degenerate warp roi %dx%d (camera parameters?)", roi[2], roi[3]);
    stx_buf *bi = nullptr, *bm = nullptr;
    if (want_img) STX_TRY(stx_buf_new(ctx, roi[2], roi[3], nearest_src ? 1 : 3, STX_U8, &bi));
    if (want_mask) {
        int rc = stx_buf_new(ctx, roi[2], roi[3], 1, STX_U8, &bm);
        if (rc != STX_OK) { stx_buf_release(bi); return rc; }
    }
    StxWarpLaunch L;
    L.proj = p;
    L.tlx = roi[0]; L.tly = roi[1]; L.dw = roi[2]; L.dh = roi[3];
    L.src = src ? src->ptr : nullptr;
    L.sw = sw; L.sh = sh;
    L.sstride = src ? src->stride : 0;
    L.src_channels = src ? src->c : 0;
    L.nearest_src = nearest_src ? 1 : 0;
    if (nearest_src) {  // generic INTER_NEAREST warp of a u8x1 source: the "mask" path writes the image
        L.dimg = nullptr; L.dimg_stride = 0;
        L.dmask = bi->ptr; L.dmask_stride = bi->stride;
    } else {
        L.dimg = bi ? bi->ptr : nullptr; L.dimg_stride = bi ? bi->stride : 0;
        L.dmask = bm ? bm->ptr : nullptr; L.dmask_stride = bm ? bm->stride : 0;
    }
    int rc = stx_launch_warp(ctx, L);
    if (rc != STX_OK) { stx_buf_release(bi); stx_buf_release(bm); return rc; }
    if (bm) bm->mask_binary = 1;  // remapNearest of a 255-filled source with a constant-0 border
    if (out_img) *out_img = bi;
    if (out_mask) *out_mask = bm;
    if (out_xywh) memcpy(out_xywh, roi, 16);
    return STX_OK;
}

STX_EXPORT int stx_warp(stx_ctx* ctx, int type, float scale, const float K[9], const float R[9], const stx_buf* src,
                        int interp, int border, stx_buf** out, int out_tl[2])
{
    if (!ctx || !src || !out) return stx_fail(STX_ERR_INVALID, "null argument");
    STX_TRY(stx_set_device(ctx));
    if (src->elem != STX_U8) return stx_fail(STX_ERR_INVALID, "warp source must be 8-bit");
    int roi[4];
    if (interp == STX_INTER_LINEAR && border == STX_BORDER_REFLECT) {
        if (src->c != 3) return stx_fail(STX_ERR_UNSUPPORTED, "INTER_LINEAR warp needs a 3-channel u8 image");
        STX_TRY(warp_impl(ctx, type, scale, K, R, src, src->w, src->h, true, false, false, out, nullptr, roi));
    } else if (interp == STX_INTER_NEAREST && border == STX_BORDER_CONSTANT) {
        if (src->c != 1) return stx_fail(STX_ERR_UNSUPPORTED, "INTER_NEAREST warp needs a 1-channel u8 mask");
        STX_TRY(warp_impl(ctx, type, scale, K, R, src, src->w, src->h, true, false, true, out, nullptr, roi));
    } else {
        return stx_fail(STX_ERR_UNSUPPORTED,
                        "only (INTER_LINEAR, BORDER_REFLECT) and (INTER_NEAREST, BORDER_CONSTANT) are on the path "
                        "(stitching/warper.py:49-50,65-66)");
    }
    if (out_tl) { out_tl[0] = roi[0]; out_tl[1] = roi[1]; }
    return STX_OK;
}

// rects: null -> the destination rectangle of image i is its ROI (found here, cached); else the caller's rectangle in warp
// coordinates (any sub-rectangle of the ROI gives exactly the ROI warp's pixels there: every pixel is mapped on its own)
static int warp_batch_impl(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s,
                           const stx_buf* const* srcs, const int* rects, stx_buf** out_imgs, stx_buf** out_masks, int* out_xywh,
                           const stx_buf* const* gains = nullptr, const int* gflags = nullptr, bool fresh_rois = false)
{
    if (!ctx || !K9s || !R9s || !srcs || n < 0) return stx_fail(STX_ERR_INVALID, "bad argument");
    if (!out_imgs && !out_masks) return stx_fail(STX_ERR_INVALID, "nothing requested");
    if (n == 0) return STX_OK;
    STX_TRY(stx_set_device(ctx));
    std::vector<StxProjector> ps(n);
    std::vector<int> rois(4 * (size_t)n), sizes(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        if (!srcs[i] || srcs[i]->elem != STX_U8 || srcs[i]->c != 3) return stx_fail(STX_ERR_INVALID, "warp source %d must be u8x3", i);
        // sources may live in another context of the same device (long-lived read-only inputs shared by several streams)
        if (srcs[i]->ctx->device != ctx->device) return stx_fail(STX_ERR_INVALID, "warp source %d lives on another device", i);
        STX_TRY(stx_make_projector(type, scale, K9s + 9 * i, R9s + 9 * i, &ps[i]));
        sizes[2 * i] = srcs[i]->w;
        sizes[2 * i + 1] = srcs[i]->h;
    }
    // ROIs: cached ones as they are, all missing ones in ONE device pass (one synchronisation)
    if (!g_roi_cache) g_roi_cache = new std::map<RoiKey, std::array<int, 4>>();
    std::vector<int> miss;
    // with gains the ROI of every image is needed even under caller-given rectangles: the gain map lies over the WHOLE warped image
    const bool need_rois = !rects || gains;
    for (int i = 0; i < n && need_rois; i++) {
        if (fresh_rois) { miss.push_back(i); continue; }  // the ROI pass belongs to this call (stx_warp_batch_with_rois)
        auto it = g_roi_cache->find(make_key(type, scale, K9s + 9 * i, R9s + 9 * i, sizes[2 * i], sizes[2 * i + 1]));
        if (it != g_roi_cache->end()) memcpy(&rois[4 * i], it->second.data(), 16);
        else miss.push_back(i);
    }
    if (!miss.empty()) {
        const int m = (int)miss.size();
        std::vector<StxProjector> mp(m);
        std::vector<int> msz(2 * (size_t)m), mroi(4 * (size_t)m);
        for (int j = 0; j < m; j++) { mp[j] = ps[miss[j]]; msz[2 * j] = sizes[2 * miss[j]]; msz[2 * j + 1] = sizes[2 * miss[j] + 1]; }
        STX_TRY(rois_impl(ctx, m, mp.data(), msz.data(), mroi.data()));
        if (g_roi_cache->size() > 8192) g_roi_cache->clear();
        for (int j = 0; j < m; j++) {
            const int i = miss[j];
            memcpy(&rois[4 * i], &mroi[4 * j], 16);
            (*g_roi_cache)[make_key(type, scale, K9s + 9 * i, R9s + 9 * i, sizes[2 * i], sizes[2 * i + 1])] = {
                mroi[4 * j], mroi[4 * j + 1], mroi[4 * j + 2], mroi[4 * j + 3]};
        }
    }
    std::vector<int> full_rois;
    if (gains) {
        if (!out_imgs) return stx_fail(STX_ERR_INVALID, "gains without images");
        full_rois = rois;
        for (int i = 0; i < n; i++) STX_TRY(block_gain_check(ctx, nullptr, gains[i]));
    }
    if (rects) {
        memcpy(rois.data(), rects, sizeof(int) * 4 * (size_t)n);
        for (int i = 0; i < n && gains; i++) {
            const int *r = &rois[4 * i], *f = &full_rois[4 * i];
            if (r[0] < f[0] || r[1] < f[1] || r[0] + r[2] > f[0] + f[2] || r[1] + r[3] > f[1] + f[3])
                return stx_fail(STX_ERR_INVALID, "image %d: with gains the rectangle must lie inside the warp roi", i);
        }
    }
    std::vector<stx_buf*> bi(n, nullptr), bm(n, nullptr);
    std::vector<StxWarpLaunch> Ls(n);
    int rc = STX_OK;
    for (int i = 0; i < n && rc == STX_OK; i++) {
        const int* roi = &rois[4 * i];
        if (roi[2] <= 0 || roi[3] <= 0 || (long long)roi[2] * roi[3] > (1ll << 33))
            rc = stx_fail(STX_ERR_INVALID, "degenerate warp roi %dx%d (camera parameters?)", roi[2], roi[3]);
        if (rc == STX_OK && out_imgs) rc = stx_buf_new(ctx, roi[2], roi[3], 3, STX_U8, &bi[i]);
        if (rc == STX_OK && out_masks) rc = stx_buf_new(ctx, roi[2], roi[3], 1, STX_U8, &bm[i]);
        if (rc != STX_OK) break;
        StxWarpLaunch& L = Ls[i];
        L.proj = ps[i];
        L.tlx = roi[0]; L.tly = roi[1]; L.dw = roi[2]; L.dh = roi[3];
        L.src = srcs[i]->ptr; L.sw = srcs[i]->w; L.sh = srcs[i]->h; L.sstride = srcs[i]->stride; L.src_channels = srcs[i]->c;
        L.nearest_src = 0;
        L.dimg = bi[i] ? bi[i]->ptr : nullptr; L.dimg_stride = bi[i] ? bi[i]->stride : 0;
        L.dmask = bm[i] ? bm[i]->ptr : nullptr; L.dmask_stride = bm[i] ? bm[i]->stride : 0;
    }
    // Exposure gains (BlocksCompensator::apply, stitching/stitcher.py:123,219-221).  Fused into the warp's epilogue when every image runs
    // the tuned kernel and every map is a bounded single-channel one: the warped bytes leave LDS already multiplied, the 6 bytes per
    // pixel of a separate pass never move.  Anything else: warp, then stx_block_gain_apply_batch — the same bytes either way.
    std::vector<int> sub;
    void* gscratch = nullptr;
    bool fused = false;
    if (rc == STX_OK && gains) {
        for (int i = 0; i < n; i++) {
            sub.push_back(full_rois[4 * i + 2]); sub.push_back(full_rois[4 * i + 3]);
            sub.push_back(rois[4 * i] - full_rois[4 * i]); sub.push_back(rois[4 * i + 1] - full_rois[4 * i + 1]);
        }
        static const bool no_fuse = getenv("STITCHING_AMD_NO_GAIN_FUSION") != nullptr;  // diagnostic: A/B against the separate pass
        fused = !no_fuse;
        for (int i = 0; i < n && fused; i++)
            fused = gains[i]->c == 1 && gflags && (gflags[i] & STX_GAIN_MAP_BOUNDED) && stx_warp_fast_eligible(Ls[i]) &&
                    (size_t)gains[i]->h * (size_t)Ls[i].dw < ((size_t)16 << 20);
        if (fused) {
            std::vector<size_t> offH(n), offY(n);
            size_t bytes = 0;
            for (int i = 0; i < n; i++) {
                offH[i] = bytes; bytes += align_up((size_t)gains[i]->h * ((Ls[i].dw + 3) & ~3) * sizeof(float), 256);
                offY[i] = bytes; bytes += align_up((size_t)Ls[i].dh * 8, 256);
            }
            rc = stx_dev_alloc(ctx, bytes, &gscratch);
            if (rc == STX_OK) {
                std::vector<float*> Hs(n);
                std::vector<void*> yts(n);
                std::vector<int> wh(2 * (size_t)n);
                for (int i = 0; i < n; i++) {
                    Hs[i] = (float*)((uint8_t*)gscratch + offH[i]); yts[i] = (uint8_t*)gscratch + offY[i];
                    wh[2 * i] = Ls[i].dw; wh[2 * i + 1] = Ls[i].dh;
                    Ls[i].gain_H = Hs[i]; Ls[i].gain_hstride = (Ls[i].dw + 3) & ~3; Ls[i].gain_yt = yts[i]; Ls[i].gain_gh = gains[i]->h;
                }
                rc = stx_launch_gain_rows(ctx, n, wh.data(), gains, sub.data(), Hs.data(), yts.data());
            }
        }
    }
    if (rc == STX_OK) rc = stx_launch_warp_batch(ctx, Ls.data(), n);
    if (gscratch) stx_dev_free(ctx, gscratch);  // stream-ordered reuse
    if (rc == STX_OK && gains && !fused) rc = stx_block_gain_apply_batch(ctx, n, bi.data(), gains, sub.data(), gflags);
    if (rc != STX_OK) {
        for (int i = 0; i < n; i++) { stx_buf_release(bi[i]); stx_buf_release(bm[i]); }
        return rc;
    }
    for (int i = 0; i < n; i++) {
        if (bm[i]) bm[i]->mask_binary = 1;
        if (out_imgs) out_imgs[i] = bi[i];
        if (out_masks) out_masks[i] = bm[i];
    }
    if (out_xywh) memcpy(out_xywh, rois.data(), sizeof(int) * 4 * (size_t)n);
    return STX_OK;
}

STX_EXPORT int stx_warp_batch(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s,
                              const stx_buf* const* srcs, stx_buf** out_imgs, stx_buf** out_masks, int* out_xywh)
{
    return warp_batch_impl(ctx, type, scale, n, K9s, R9s, srcs, nullptr, out_imgs, out_masks, out_xywh);
}

STX_EXPORT int stx_warp_batch_rects(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s,
                                    const stx_buf* const* srcs, const int* rects_xywh, stx_buf** out_imgs, stx_buf** out_masks)
{
    if (!rects_xywh) return stx_fail(STX_ERR_INVALID, "null argument");
    return warp_batch_impl(ctx, type, scale, n, K9s, R9s, srcs, rects_xywh, out_imgs, out_masks, nullptr);
}

STX_EXPORT int stx_warp_batch_gain(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s, const stx_buf* const* srcs,
                                   const int* rects_xywh_or_null, const stx_buf* const* gain_maps, const int* gain_flags, stx_buf** out_imgs,
                                   stx_buf** out_masks, int* out_xywh_or_null)
{
    if (!gain_maps || !out_imgs) return stx_fail(STX_ERR_INVALID, "null argument");
    return warp_batch_impl(ctx, type, scale, n, K9s, R9s, srcs, rects_xywh_or_null, out_imgs, out_masks, rects_xywh_or_null ? nullptr : out_xywh_or_null,
                           gain_maps, gain_flags);
}

STX_EXPORT int stx_warp_batch_with_rois(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s,
                                        const stx_buf* const* srcs, const stx_buf* const* gain_maps_or_null, const int* gain_flags_or_null,
                                        stx_buf** out_imgs, stx_buf** out_masks, int* out_xywh)
{
    if (!out_xywh) return stx_fail(STX_ERR_INVALID, "null argument");
    if (gain_maps_or_null && !out_imgs) return stx_fail(STX_ERR_INVALID, "gains without images");
    return warp_batch_impl(ctx, type, scale, n, K9s, R9s, srcs, nullptr, out_imgs, out_masks, out_xywh, gain_maps_or_null, gain_flags_or_null, true);
}

STX_EXPORT int stx_warp_image_and_mask(stx_ctx* ctx, int type, float scale, const float K[9], const float R[9],
                                       const stx_buf* src, stx_buf** out_img, stx_buf** out_mask, int out_xywh[4])
{
    if (!ctx || !src) return stx_fail(STX_ERR_INVALID, "null argument");
    if (!out_img && !out_mask) return stx_fail(STX_ERR_INVALID, "nothing requested");
    STX_TRY(stx_set_device(ctx));
    if (src->elem != STX_U8 || src->c != 3) return stx_fail(STX_ERR_INVALID, "warp source must be u8x3");
    return warp_impl(ctx, type, scale, K, R, src, src->w, src->h, out_img != nullptr, out_mask != nullptr, false,
                     out_img, out_mask, out_xywh);
}

STX_EXPORT int stx_warp_mask(stx_ctx* ctx, int type, float scale, const float K[9], const float R[9], int w, int h,
                             stx_buf** out_mask, int out_xywh[4])
{
    if (!ctx || !out_mask) return stx_fail(STX_ERR_INVALID, "null argument");
    if (w <= 0 || h <= 0) return stx_fail(STX_ERR_INVALID, "image size %dx%d", w, h);
    STX_TRY(stx_set_device(ctx));
    return warp_impl(ctx, type, scale, K, R, nullptr, w, h, false, true, false, nullptr, out_mask, out_xywh);
}

STX_EXPORT int stx_debug_feather_dist_cap(void) { return STX_FEATHER_DIST_CAP; }

// Test hook (include/stitching_amd_debug.h): the fp32 backward map of a warp as the device projector computes it.
STX_EXPORT int stx_debug_warp_maps(stx_ctx* ctx, int type, float scale, const float K[9], const float R[9], int w, int h, int which,
                                   const int rect_xywh[4], stx_buf** out_xmap, stx_buf** out_ymap, int out_xywh[4])
{
    if (!ctx || !K || !R || !out_xmap || !out_ymap) return stx_fail(STX_ERR_INVALID, "null argument");
    if (w <= 0 || h <= 0) return stx_fail(STX_ERR_INVALID, "image size %dx%d", w, h);
    if (which != 1 && which != 2) return stx_fail(STX_ERR_INVALID, "which = %d (1: the kernel a warp takes, 2: the generic kernel)", which);
    STX_TRY(stx_set_device(ctx));
    StxProjector p;
    STX_TRY(stx_make_projector(type, scale, K, R, &p));
    int roi[4];
    if (rect_xywh) memcpy(roi, rect_xywh, 16);
    else STX_TRY(roi_cached(ctx, type, scale, K, R, w, h, p, roi));
    if (roi[2] <= 0 || roi[3] <= 0 || (long long)roi[2] * roi[3] > (1ll << 30))
        return stx_fail(STX_ERR_INVALID, "degenerate warp roi %dx%d", roi[2], roi[3]);
    stx_buf *bx = nullptr, *by = nullptr;
    STX_TRY(stx_buf_new(ctx, roi[2], roi[3], 1, STX_F32, &bx));
    int rc = stx_buf_new(ctx, roi[2], roi[3], 1, STX_F32, &by);
    if (rc != STX_OK) { stx_buf_release(bx); return rc; }
    StxWarpLaunch L;
    L.proj = p;
    L.tlx = roi[0]; L.tly = roi[1]; L.dw = roi[2]; L.dh = roi[3];
    L.src = nullptr; L.sw = w; L.sh = h; L.sstride = 0; L.src_channels = 0;
    L.nearest_src = 0;
    L.dimg = bx->ptr; L.dimg_stride = bx->stride;
    L.dmask = by->ptr; L.dmask_stride = by->stride;
    L.debug_maps = which;
    rc = stx_launch_warp(ctx, L);
    if (rc != STX_OK) { stx_buf_release(bx); stx_buf_release(by); return rc; }
    *out_xmap = bx;
    *out_ymap = by;
    if (out_xywh) memcpy(out_xywh, roi, 16);
    return STX_OK;
}

// ---------------------------------------------------------------------------------------------
// blenders
// ---------------------------------------------------------------------------------------------
STX_EXPORT int stx_result_roi(int n, const int* corners_xy, const int* sizes_wh, int out_xywh[4])
{
    if (n <= 0 || !corners_xy || !sizes_wh || !out_xywh) return stx_fail(STX_ERR_INVALID, "bad argument");
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; i++) {
        tlx = std::min(tlx, corners_xy[2 * i]);
        tly = std::min(tly, corners_xy[2 * i + 1]);
        brx = std::max(brx, corners_xy[2 * i] + sizes_wh[2 * i]);
        bry = std::max(bry, corners_xy[2 * i + 1] + sizes_wh[2 * i + 1]);
    }
    out_xywh[0] = tlx; out_xywh[1] = tly; out_xywh[2] = brx - tlx; out_xywh[3] = bry - tly;
    return STX_OK;
}

constexpr size_t MB_FRONT_PAD = 64;  // bytes in front of every int16 pyramid / finished-level buffer (keeps 64-byte alignment)

struct stx_blender {
    stx_ctx* ctx = nullptr;
    int kind = 0, num_bands = 0;
    float sharpness = 0.02f;
    int rx = 0, ry = 0, rw = 0, rh = 0;  // dst_roi_ (padded for multiband)
    int fw = 0, fh = 0;                  // dst_roi_final_ size
    bool finished = false;
    // multiband (deferred gather)
    std::vector<StxMbImage> images;   // kept sorted by .order (the global feed order)
    std::vector<char> built;          // pyramid of images[i] exists (kind 0)
    std::vector<stx_buf*> held;
    std::vector<void*> pyr_allocs;
    StxMbImage* d_all = nullptr;      // device copy of `images` as the pyramid pass uploaded it, while it still equals `images` (else null)
    int band_x0 = 0, band_x1 = 0;     // columns of the final roi this blender produces (sharded blending)
    int next_order = 0;
    int pyr_mode = 0;                 // STX_PYRDOWN_* | lanes << 8, captured at stx_blend_create: one summation order per panorama
                                      // whatever stx_set_pyrdown_mode is called with between feed() and blend()
    // no: deferred gather over the fed images (stx_launch_no_gather)
    std::vector<NoImg> no_images;
    // feather: deferred gather as well (stx_launch_feather_weights / _gather)
    std::vector<FeatherImg> feather_images;
};

static void blender_release(stx_blender* b)
{
    for (stx_buf* h : b->held) stx_buf_release(h);
    b->held.clear();
    for (void* p : b->pyr_allocs) stx_dev_free(b->ctx, p);
    b->pyr_allocs.clear();
    b->d_all = nullptr;
    b->images.clear();
    b->built.clear();
    b->no_images.clear();
    b->feather_images.clear();
}

STX_EXPORT int stx_blend_create(stx_ctx* ctx, int kind, int num_bands, float sharpness, const int roi_xywh[4],
                                stx_blender** out)
{
    if (!roi_xywh || !out) return stx_fail(STX_ERR_INVALID, "null argument");
    // ctx == NULL: geometry-only multi-band blender (band count, feed / contribution rectangles) for
    // planning on hosts without a GPU; it cannot be fed
    if (!ctx && kind != STX_BLEND_MULTIBAND) return stx_fail(STX_ERR_INVALID, "ctx is null");
    if (ctx) STX_TRY(stx_set_device(ctx));
    if (kind < STX_BLEND_NO || kind > STX_BLEND_MULTIBAND) return stx_fail(STX_ERR_INVALID, "unknown blender kind %d", kind);
    int w = roi_xywh[2], h = roi_xywh[3];
    if (w <= 0 || h <= 0) return stx_fail(STX_ERR_INVALID, "empty destination roi %dx%d", w, h);
    std::unique_ptr<stx_blender> b(new stx_blender());
    b->ctx = ctx;
    b->kind = kind;
    b->sharpness = sharpness;
    b->fw = w; b->fh = h;
    if (kind == STX_BLEND_MULTIBAND) {
        if (num_bands < 0) return stx_fail(STX_ERR_INVALID, "num_bands %d", num_bands);  // CV_Assert(val >= 0)
        // MultiBandBlender::prepare: crop unnecessary bands, pad to a multiple of 2^bands
        double max_len = (double)std::max(w, h);
        int nb = std::min(num_bands, (int)std::ceil(std::log(max_len) / std::log(2.0)));
        if (nb > STX_MAX_BANDS) nb = STX_MAX_BANDS;
        b->num_bands = nb;
        w += ((1 << nb) - w % (1 << nb)) % (1 << nb);
        h += ((1 << nb) - h % (1 << nb)) % (1 << nb);
    }
    b->rx = roi_xywh[0]; b->ry = roi_xywh[1]; b->rw = w; b->rh = h;
    b->band_x0 = 0; b->band_x1 = b->fw;
    b->pyr_mode = pyrdown_now();
    *out = b.release();
    return STX_OK;
}

STX_EXPORT int stx_blend_num_bands(const stx_blender* b, int* out_num_bands)
{
    if (!b || !out_num_bands) return stx_fail(STX_ERR_INVALID, "null argument");
    *out_num_bands = b->num_bands;
    return STX_OK;
}

// MultiBandBlender::feed geometry: keep the image with a gap, snap to the 2^bands grid, stay inside dst_roi_.
// Returns the feed rectangle (tl_new .. br_new) relative to the padded roi.
static void mb_feed_rect(const stx_blender* b, int w, int h, int tlx, int tly, int* fx, int* fy, int* fw, int* fh)
{
    const int nb = b->num_bands;
    const int gap = 3 * (1 << nb);
    int tlnx = std::max(b->rx, tlx - gap), tlny = std::max(b->ry, tly - gap);
    int brnx = std::min(b->rx + b->rw, tlx + w + gap), brny = std::min(b->ry + b->rh, tly + h + gap);
    tlnx = b->rx + (((tlnx - b->rx) >> nb) << nb);
    tlny = b->ry + (((tlny - b->ry) >> nb) << nb);
    int width = brnx - tlnx, height = brny - tlny;
    width += ((1 << nb) - width % (1 << nb)) % (1 << nb);
    height += ((1 << nb) - height % (1 << nb)) % (1 << nb);
    brnx = tlnx + width;
    brny = tlny + height;
    const int dy = std::max(brny - (b->ry + b->rh), 0), dx = std::max(brnx - (b->rx + b->rw), 0);
    tlnx -= dx; tlny -= dy;
    *fx = tlnx - b->rx; *fy = tlny - b->ry; *fw = width; *fh = height;
}

// Region of every level that the columns [bx0, bx1) of the final panorama depend on (pyrUp halo:
// level i needs level i+1 at (x >> 1) +- 1).  Origins are multiples of 8 for the levels of the vector kernels
// (<= B - 3: a lane owns 8 adjacent samples) and multiples of 2 for the coarser levels of the per-sample kernel (the finer
// level reads them through dword-aligned windows): 8 samples of the coarsest level are 8 * 2^B panorama columns, which used
// to widen every band's region — and with it every strip another rank has to supply — by up to 256 columns at 5 bands.
static void mb_level_regions(const stx_blender* b, int bx0, int bx1, int* xb, int* xe)
{
    xb[0] = bx0; xe[0] = bx1;
    for (int i = 1; i <= b->num_bands; i++) {
        const int pw = b->rw >> i;
        const int al = i <= b->num_bands - 3 ? 7 : 1;
        xb[i] = std::max(0, (xb[i - 1] >> 1) - 1) & ~al;
        xe[i] = std::min(pw, ((((xe[i - 1] - 1) >> 1) + 2) + al) & ~al);
    }
}

// Level-0 column range [sx0, sx1) (2^bands aligned, clipped to the feed rect [fx, fx+fw)) of the
// contribution an image must supply to the rank that owns the columns [bx0, bx1).
static bool mb_contrib_range(const stx_blender* b, int fx, int fw, int bx0, int bx1, int* sx0, int* sx1)
{
    int xb[STX_MAX_BANDS + 1], xe[STX_MAX_BANDS + 1];
    mb_level_regions(b, bx0, bx1, xb, xe);
    const int nb = b->num_bands, al = (1 << nb) - 1;
    long long lo = xb[0], hi = xe[0];
    for (int i = 1; i <= nb; i++) {
        lo = std::min(lo, (long long)xb[i] << i);
        hi = std::max(hi, (long long)xe[i] << i);
    }
    lo = lo & ~(long long)al;
    hi = (hi + al) & ~(long long)al;
    lo = std::max(lo, (long long)fx);
    hi = std::min(hi, (long long)fx + fw);
    *sx0 = (int)lo; *sx1 = (int)hi;
    return hi > lo;
}

// packed layout of a contribution strip of size (w, h) at level 0: per level i the three int16 planes
// then the fp32 weights; every section starts on a 256-byte boundary
struct ContribLayout {
    size_t g_off[STX_MAX_BANDS + 1], w_off[STX_MAX_BANDS + 1];
    long long g_stride[STX_MAX_BANDS + 1], w_stride[STX_MAX_BANDS + 1];
    size_t bytes;
};
static void mb_contrib_layout(int nb, int w, int h, ContribLayout* L)
{
    size_t off = 0;
    for (int i = 0; i <= nb; i++) {
        const int lw = w >> i, lh = h >> i;
        L->g_stride[i] = (long long)align_up((size_t)std::max(lw, 1), 32);
        L->w_stride[i] = (long long)align_up((size_t)std::max(lw, 1), 16);
        L->g_off[i] = off;
        off = align_up(off + (size_t)L->g_stride[i] * std::max(lh, 1) * 3 * sizeof(short), 256);
        L->w_off[i] = off;
        off = align_up(off + (size_t)L->w_stride[i] * std::max(lh, 1) * sizeof(float), 256);
    }
    L->bytes = off;
}

static void mb_insert_sorted(stx_blender* b, const StxMbImage& im, bool is_built)
{
    size_t pos = b->images.size();
    while (pos > 0 && b->images[pos - 1].order > im.order) pos--;
    b->images.insert(b->images.begin() + pos, im);
    b->built.insert(b->built.begin() + pos, is_built ? 1 : 0);
    b->d_all = nullptr;
}

static int mb_feed(stx_blender* b, const stx_buf* img, const stx_buf* mask, int tlx, int tly, int order)
{
    stx_ctx* ctx = b->ctx;
    const int nb = b->num_bands, w = img->w, h = img->h;
    StxMbImage im;
    memset(&im, 0, sizeof(im));
    im.kind = 0;
    im.order = order;
    mb_feed_rect(b, w, h, tlx, tly, &im.fx, &im.fy, &im.fw, &im.fh);
    im.img0 = img->ptr; im.img0_stride = (long long)img->stride; im.img0_is_s16 = img->elem == STX_S16;
    im.mask0 = mask->ptr; im.mask0_stride = (long long)mask->stride; im.mask_binary = mask->mask_binary;
    im.iw = w; im.ih = h;
    im.ix = tlx - b->rx; im.iy = tly - b->ry;
    im.left = im.ix - im.fx; im.top = im.iy - im.fy;
    // the Gaussian levels of a u8 image are 0..255: stored as bytes (3 instead of 6 bytes per sample on every pyramid pass)
    im.g_u8 = img->elem == STX_U8 ? 1 : 0;
    // W_1 of a 0 / 255 mask is k / 256, k <= 256: stored as halves, exactly (StxMbImage::w1_f16).  STITCHING_AMD_W1_F32: diagnostic (fp32 as before)
    static const bool w1_f32 = getenv("STITCHING_AMD_W1_F32") != nullptr;
    im.w1_f16 = (mask->mask_binary && !w1_f32) ? 1 : 0;
    for (int i = 1; i <= nb; i++) {
        const int lw = im.fw >> i, lh = im.fh >> i;
        // rows of 64 bytes either way
        const long long gs = (long long)align_up((size_t)lw, im.g_u8 ? 64 : 32), ws = (long long)align_up((size_t)lw, 16);
        void *g = nullptr, *wt = nullptr;
        // MB_FRONT_PAD in front, 64 bytes behind: the pyrUp tap windows of the gather kernels start up to 4 bytes in front of a row
        // and end up to 8 bytes behind its last sample (up_row_window / up_row_window_u8)
        STX_TRY(stx_dev_alloc(ctx, MB_FRONT_PAD + (size_t)gs * lh * 3 * (im.g_u8 ? 1 : sizeof(short)) + 64, &g));
        b->pyr_allocs.push_back(g);
        STX_TRY(stx_dev_alloc(ctx, (size_t)ws * lh * ((i == 1 && im.w1_f16) ? sizeof(uint16_t) : sizeof(float)), &wt));
        b->pyr_allocs.push_back(wt);
        im.g[i] = (short*)((uint8_t*)g + MB_FRONT_PAD); im.g_stride[i] = gs; im.g_plane[i] = gs * lh;
        im.wt[i] = (float*)wt; im.wt_stride[i] = ws;
    }
    // deferred: the pyramids of all images are built together (one launch per level), at the first
    // export / blend() that needs them
    mb_insert_sorted(b, im, nb == 0);
    stx_buf_retain(const_cast<stx_buf*>(img));
    stx_buf_retain(const_cast<stx_buf*>(mask));
    b->held.push_back(const_cast<stx_buf*>(img));
    b->held.push_back(const_cast<stx_buf*>(mask));
    return STX_OK;
}

// upload `n` descriptors through the context's pinned ring: asynchronous, in stream order, no host wait
static int mb_upload(stx_blender* b, const StxMbImage* h, int n, StxMbImage** d_out)
{
    stx_ctx* ctx = b->ctx;
    void* d = nullptr;
    STX_TRY(stx_dev_alloc(ctx, sizeof(StxMbImage) * std::max(n, 1), &d));
    b->pyr_allocs.push_back(d);
    if (n > 0) STX_TRY(stx_stage_upload(ctx, d, h, sizeof(StxMbImage) * (size_t)n));
    *d_out = (StxMbImage*)d;
    return STX_OK;
}

static int mb_ensure_pyramids(stx_blender* b)
{
    std::vector<StxMbImage> todo;
    for (size_t i = 0; i < b->images.size(); i++)
        if (b->images[i].kind == 0 && !b->built[i]) todo.push_back(b->images[i]);
    if (todo.empty()) return STX_OK;
    // occupancy maps of the weight pyramids (StxMbImage::occ): one arena for this batch.  Only where every level is built by
    // the batched LDS kernels, which write them (int16 sources take the generic level-0 kernel).
    static const bool occ_off = getenv("STITCHING_AMD_NO_OCC") != nullptr;  // diagnostic: A/B of the bookkeeping
    const int pyr = b->pyr_mode;  // (the blender's own, fixed at creation) != scalar: the generic kernels build every level, and they keep no occupancy maps
    if (todo.size() <= 65535 && !occ_off && (pyr & 255) == STX_PYRDOWN_SCALAR) {
        const int nl = b->num_bands + 1;
        std::vector<size_t> off(todo.size() * (size_t)nl, 0);
        size_t bytes = 0;
        for (size_t t = 0; t < todo.size(); t++) {
            if (todo[t].img0_is_s16) continue;
            for (int i = 1; i < nl; i++) {
                off[t * nl + i] = bytes;
                // rows of ((fw >> i) / 64 rounded up, then to a multiple of 4) bytes; one more row = slack for the 12-byte reads
                bytes += (size_t)((((todo[t].fh >> i) + 1) >> 1) + 1) * (size_t)(((((todo[t].fw >> i) + 63) >> 6) + 3) & ~3);
            }
        }
        if (bytes > 0) {
            void* arena = nullptr;
            STX_TRY(stx_dev_alloc(b->ctx, bytes + 16, &arena));
            b->pyr_allocs.push_back(arena);
            size_t t = 0;
            for (size_t i = 0; i < b->images.size(); i++) {
                if (!(b->images[i].kind == 0 && !b->built[i])) continue;
                if (!todo[t].img0_is_s16)
                    for (int l = 1; l < nl; l++) todo[t].occ[l] = b->images[i].occ[l] = (uint8_t*)arena + off[t * nl + l];
                t++;
            }
        }
    }
    StxMbImage* d = nullptr;
    STX_TRY(mb_upload(b, todo.data(), (int)todo.size(), &d));
    // every image of the blender in this pass (the usual case): blend() reads the very same table — one upload, one copy dispatch fewer
    // between the pyramids and the collapse
    b->d_all = todo.size() == b->images.size() && memcmp(todo.data(), b->images.data(), sizeof(StxMbImage) * todo.size()) == 0 ? d : nullptr;
    STX_TRY(stx_launch_mb_pyramids(b->ctx, d, todo.data(), (int)todo.size(), b->num_bands, pyr & 255, pyr >> 8));
    for (size_t i = 0; i < b->images.size(); i++) b->built[i] = 1;
    return STX_OK;
}

static double mb_level_bytes(const stx_blender* b, const std::vector<StxMbImage>& imgs, int lv, int x0, int x1, bool emit,
                             bool with16)
{
    // algorithmic bytes: every input element of the region once, every output element once
    const int nb = b->num_bands;
    const int ph = lv == 0 ? b->fh : b->rh >> lv;
    double bytes = 0.0;
    for (const StxMbImage& im : imgs) {
        int rx = im.fx >> lv, rw = im.fw >> lv, ry = im.fy >> lv, rh = im.fh >> lv;
        if (lv == 0 && im.kind == 0) { rx = im.ix; rw = im.iw; ry = im.iy; rh = im.ih; }
        const double cols = std::max(0, std::min(rx + rw, x1) - std::max(rx, x0)), rows = std::min(ry + rh, ph) - ry;
        if (cols <= 0 || rows <= 0) continue;
        const double g3 = im.g_u8 ? 3.0 : 6.0;  // the three Gaussian planes of a sample: bytes (u8 image) or int16
        if (im.kind == 1) bytes += cols * rows * 10.0;
        else if (lv == 0) bytes += cols * rows * ((im.img0_is_s16 ? 6 : 3) + 1) + (nb > 0 ? cols * rows * g3 / 4.0 : 0.0);
        else bytes += cols * rows * (g3 + ((lv == 1 && im.w1_f16) ? 2.0 : 4.0)) + (lv < nb ? cols * rows * g3 / 4.0 : 0.0);
    }
    const double area = (double)(x1 - x0) * ph;
    if (emit) return bytes + area * 10.0;
    if (lv < nb) bytes += area * 6.0 / 4.0;
    bytes += lv == 0 ? area * (4 + (with16 ? 6 : 0)) : area * 6.0;
    return bytes;
}

STX_EXPORT int stx_blend_feed_ex(stx_blender* b, const stx_buf* img, const stx_buf* mask, int tlx, int tly, int order)
{
    if (!b || !img || !mask) return stx_fail(STX_ERR_INVALID, "null argument");
    if (b->finished) return stx_fail(STX_ERR_STATE, "feed after blend()");
    if (!b->ctx) return stx_fail(STX_ERR_STATE, "geometry-only blender (created without a context)");
    STX_TRY(stx_set_device(b->ctx));
    // CV_Assert(img.type() == CV_16SC3 [|| CV_8UC3]); CV_Assert(mask.type() == CV_8U)
    if (img->c != 3 || (img->elem != STX_U8 && img->elem != STX_S16))
        return stx_fail(STX_ERR_INVALID, "feed: image must be u8x3 or s16x3");
    if (mask->c != 1 || mask->elem != STX_U8) return stx_fail(STX_ERR_INVALID, "feed: mask must be u8x1");
    if (mask->w != img->w || mask->h != img->h)
        return stx_fail(STX_ERR_INVALID, "feed: mask %dx%d does not match image %dx%d", mask->w, mask->h, img->w, img->h);
    if (img->ctx != b->ctx || mask->ctx != b->ctx) return stx_fail(STX_ERR_INVALID, "feed: buffers belong to another context");
    // the image must lie inside the roi given to prepare() (OpenCV would write out of bounds)
    const int ux = b->kind == STX_BLEND_MULTIBAND ? b->rx + b->fw : b->rx + b->rw;
    const int uy = b->kind == STX_BLEND_MULTIBAND ? b->ry + b->fh : b->ry + b->rh;
    if (tlx < b->rx || tly < b->ry || tlx + img->w > ux || tly + img->h > uy)
        return stx_fail(STX_ERR_INVALID, "feed: image at (%d,%d) size %dx%d leaves the prepared roi (%d,%d,%d,%d)", tlx, tly,
                        img->w, img->h, b->rx, b->ry, ux - b->rx, uy - b->ry);
    if (order < 0) order = b->next_order;
    b->next_order = std::max(b->next_order, order + 1);
    if (b->kind == STX_BLEND_MULTIBAND) return mb_feed(b, img, mask, tlx, tly, order);
    if (b->kind == STX_BLEND_NO) {  // deferred: the image joins the table, the gather runs in blend()
        NoImg im;
        memset(&im, 0, sizeof(im));
        im.img = img->ptr; im.istride = (long long)img->stride; im.is_s16 = img->elem == STX_S16;
        im.mask = mask->ptr; im.mstride = (long long)mask->stride;
        im.x = tlx - b->rx; im.y = tly - b->ry; im.w = img->w; im.h = img->h;
        im.mask_binary = mask->mask_binary;
        b->no_images.push_back(im);
        stx_buf_retain(const_cast<stx_buf*>(img));
        stx_buf_retain(const_cast<stx_buf*>(mask));
        b->held.push_back(const_cast<stx_buf*>(img));
        b->held.push_back(const_cast<stx_buf*>(mask));
        return STX_OK;
    }
    // feather, deferred: the image joins the table; distance transforms, weights and the gather run in blend()
    FeatherImg im;
    memset(&im, 0, sizeof(im));
    im.img = img->ptr; im.istride = (long long)img->stride; im.is_s16 = img->elem == STX_S16;
    im.mask = mask->ptr; im.mstride = (long long)mask->stride;
    im.x = tlx - b->rx; im.y = tly - b->ry; im.w = img->w; im.h = img->h;
    im.dstride = ((long long)img->w + 15) & ~15ll;
    im.n_chunks = (img->h + STX_DT_RC - 1) / STX_DT_RC;
    void *wm = nullptr, *summ = nullptr;
    // 64 bytes in front and behind: a lane's group of 4 distances may start up to 3 samples left of a row / end 3 right of it
    STX_TRY(stx_dev_alloc(b->ctx, 64 + sizeof(uint16_t) * (size_t)im.dstride * img->h + 64, &wm));
    b->pyr_allocs.push_back(wm);
    wm = (uint8_t*)wm + 64;
    // per (chunk, column): the zero rows as a 64-bit set, then the first and the last of them
    STX_TRY(stx_dev_alloc(b->ctx, (sizeof(unsigned long long) + 2 * sizeof(int)) * (size_t)im.dstride * im.n_chunks, &summ));
    b->pyr_allocs.push_back(summ);
    im.dist = (uint16_t*)wm;
    im.zbits = (unsigned long long*)summ;
    im.first = (int*)(im.zbits + (size_t)im.dstride * im.n_chunks); im.last = im.first + (size_t)im.dstride * im.n_chunks;
    b->feather_images.push_back(im);
    stx_buf_retain(const_cast<stx_buf*>(img));
    stx_buf_retain(const_cast<stx_buf*>(mask));
    b->held.push_back(const_cast<stx_buf*>(img));
    b->held.push_back(const_cast<stx_buf*>(mask));
    return STX_OK;
}

STX_EXPORT int stx_blend_feed(stx_blender* b, const stx_buf* img, const stx_buf* mask, int tlx, int tly)
{
    return stx_blend_feed_ex(b, img, mask, tlx, tly, -1);
}

static void mb_fill_common(const stx_blender* b, MbLevelK* K, const StxMbImage* d_images, int n, int lv)
{
    memset(K, 0, sizeof(*K));
    K->images = d_images; K->n_images = n; K->level = lv; K->num_bands = b->num_bands;
    K->pw = b->rw >> lv; K->ph = b->rh >> lv;
    K->all_u8 = 1;
}

static int mb_finish(stx_blender* b, stx_buf* pano, stx_buf* pmask, stx_buf* pano16)
{
    stx_ctx* ctx = b->ctx;
    const int nb = b->num_bands, n = (int)b->images.size();
    STX_TRY(mb_ensure_pyramids(b));
    StxMbImage* d_images = b->d_all;
    if (!d_images) STX_TRY(mb_upload(b, b->images.data(), n, &d_images));
    bool all_u8 = true, has_contrib = false, pk_ok = true;
    for (const StxMbImage& im : b->images) {
        if (im.kind == 0 && im.img0_is_s16) all_u8 = false;
        if (im.kind == 1) has_contrib = true;
        if ((im.kind == 0 && im.img0_is_s16) || !im.mask_binary) pk_ok = false;
    }
    int xb[STX_MAX_BANDS + 1], xe[STX_MAX_BANDS + 1];
    mb_level_regions(b, b->band_x0, b->band_x1, xb, xe);
    std::vector<short*> out(nb + 2, nullptr);
    std::vector<long long> ostride(nb + 2, 0), oplane(nb + 2, 0);
    // The three coarsest levels go through one launch (mb_coarse_kernel: the finished levels B and B-1 live in LDS only) when
    // there are at least 3 bands, i.e. when level B-2 is not the panorama itself.  STITCHING_AMD_NO_COARSE_FUSION: diagnostic.
    static const bool no_fusion = getenv("STITCHING_AMD_NO_COARSE_FUSION") != nullptr;
    const bool fuse = nb >= 3 && !no_fusion;
    for (int lv = fuse ? nb - 2 : nb; lv >= 0; lv--) {
        MbLevelK K;
        mb_fill_common(b, &K, d_images, n, lv);
        K.all_u8 = all_u8 ? 1 : 0;
        K.has_contrib = has_contrib ? 1 : 0;
        K.pk_ok = pk_ok ? 1 : 0;
        K.x0 = xb[lv]; K.x1 = xe[lv]; K.y0 = 0; K.y1 = lv == 0 ? b->fh : b->rh >> lv;
        if (lv < nb) {
            K.up = out[lv + 1]; K.up_stride = ostride[lv + 1]; K.up_plane = oplane[lv + 1];
            K.up_x0 = xb[lv + 1]; K.up_y0 = 0;
        }
        if (lv == 0) {
            K.pano = pano->ptr; K.pano_stride = (long long)pano->stride;
            K.pmask = pmask->ptr; K.pmask_stride = (long long)pmask->stride;
            if (pano16) { K.pano16 = (short*)pano16->ptr; K.pano16_stride = (long long)pano16->stride; }
            K.pano_x0 = b->band_x0; K.pano_y0 = 0;
        } else {
            const int w = xe[lv] - xb[lv], ph = b->rh >> lv;
            const long long st = (long long)align_up((size_t)std::max(w, 1), 32);
            void* p = nullptr;
            STX_TRY(stx_dev_alloc(ctx, MB_FRONT_PAD + (size_t)st * ph * 3 * sizeof(short), &p));
            b->pyr_allocs.push_back(p);
            out[lv] = (short*)((uint8_t*)p + MB_FRONT_PAD); ostride[lv] = st; oplane[lv] = st * ph;
            K.out = out[lv]; K.out_stride = st; K.out_plane = st * ph;
            K.out_x0 = xb[lv]; K.out_y0 = 0;
        }
        if (fuse && lv == nb - 2) {
            // algorithmic bytes: the inputs of the three levels over their regions once, the finished level B-2 once
            double bytes = mb_level_bytes(b, b->images, lv, K.x0, K.x1, false, false) - (double)(K.x1 - K.x0) * (b->rh >> lv) * 6.0 / 4.0;
            for (int l2 = nb - 1; l2 <= nb; l2++)
                bytes += mb_level_bytes(b, b->images, l2, xb[l2], xe[l2], false, false) - (double)(xe[l2] - xb[l2]) * (b->rh >> l2) * (l2 < nb ? 7.5 : 6.0);
            K.up = nullptr;
            STX_TRY(stx_launch_mb_coarse(ctx, K, bytes));
            continue;
        }
        STX_TRY(stx_launch_mb_level(ctx, K, mb_level_bytes(b, b->images, lv, K.x0, K.x1, false, pano16 != nullptr)));
    }
    return STX_OK;
}

// ---- sharded multi-band blending (one blender per rank) -------------------------------
STX_EXPORT int stx_blend_set_band(stx_blender* b, int x0, int x1)
{
    if (!b) return stx_fail(STX_ERR_INVALID, "null argument");
    if (b->kind != STX_BLEND_MULTIBAND) return stx_fail(STX_ERR_UNSUPPORTED, "bands exist for the multi-band blender only");
    if (b->finished) return stx_fail(STX_ERR_STATE, "set_band after blend()");
    const int al = (1 << b->num_bands) - 1;
    if (x0 < 0 || x1 > b->fw || x1 <= x0 || (x0 & al) || ((x1 & al) && x1 != b->fw) || (x0 & 7))
        return stx_fail(STX_ERR_INVALID, "band [%d,%d) must lie in [0,%d) with edges on multiples of max(8, 2^bands)", x0, x1, b->fw);
    b->band_x0 = x0; b->band_x1 = x1;
    return STX_OK;
}

STX_EXPORT int stx_blend_contrib_rect(const stx_blender* b, int img_w, int img_h, int tlx, int tly, int band_x0, int band_x1,
                                      int out_rect_xywh[4], size_t* out_bytes)
{
    if (!b || !out_rect_xywh) return stx_fail(STX_ERR_INVALID, "null argument");
    if (b->kind != STX_BLEND_MULTIBAND) return stx_fail(STX_ERR_UNSUPPORTED, "multi-band blender only");
    int fx, fy, fw, fh, sx0, sx1;
    mb_feed_rect(b, img_w, img_h, tlx, tly, &fx, &fy, &fw, &fh);
    if (!mb_contrib_range(b, fx, fw, band_x0, band_x1, &sx0, &sx1)) {
        out_rect_xywh[0] = out_rect_xywh[1] = out_rect_xywh[2] = out_rect_xywh[3] = 0;
        if (out_bytes) *out_bytes = 0;
        return STX_OK;
    }
    out_rect_xywh[0] = sx0; out_rect_xywh[1] = fy; out_rect_xywh[2] = sx1 - sx0; out_rect_xywh[3] = fh;
    if (out_bytes) {
        ContribLayout L;
        mb_contrib_layout(b->num_bands, sx1 - sx0, fh, &L);
        *out_bytes = L.bytes;
    }
    return STX_OK;
}

// ---- image-strip sharding -------------------------------------------------------------------------------------
// Instead of per-level contributions ((short)(L W) and W: 13.3 bytes per strip pixel, plus an export pass on the sender)
// a rank can ship the COLUMNS of its warped image and mask that the other band depends on (4 bytes per pixel, copied out
// by the DMA engine) and let the receiver feed them like an image of its own.  The receiver's pyramids of the strip equal
// the owner's wherever the band looks, provided the strip holds every source column that reaches the band's region:
//   * [sx0, sx1): the level-0 columns where the band needs this image's contributions (mb_contrib_range);
//   * + gap + 2^B on both sides: a Laplacian sample depends on the bordered image within 3 * 2^B = gap columns
//     (pyrDown support 2 (2^B - 1), pyrUp of the next level 2^B more), one more 2^B for the grid snapping;
//   * where that range runs into the image's own left / right edge the border is copyMakeBorder(REFLECT): the columns the
//     reflection reads (as many as the range sticks out) must be in the strip as well.
// A cut edge of the strip is then at least gap + 2^B away from everything the band reads; what the receiver computes
// beyond it (it reflects where the owner had real pixels) is never looked at.  Rows are not cut.
static bool mb_strip_range(const stx_blender* b, int img_w, int img_h, int tlx, int tly, int bx0, int bx1, int* x0, int* x1)
{
    int fx, fy, fw, fh, sx0, sx1;
    mb_feed_rect(b, img_w, img_h, tlx, tly, &fx, &fy, &fw, &fh);
    if (!mb_contrib_range(b, fx, fw, bx0, bx1, &sx0, &sx1)) return false;
    const int nb = b->num_bands, reach = 3 * (1 << nb) + (1 << nb);
    const int ix0 = tlx - b->rx, ix1 = ix0 + img_w;
    const int qlo = std::max(sx0 - reach, fx), qhi = std::min(sx1 + reach, fx + fw);
    int lo = std::max(ix0, qlo), hi = std::min(ix1, qhi);
    if (qlo < ix0) hi = std::max(hi, std::min(ix1, 2 * ix0 - qlo));
    if (qhi > ix1) lo = std::min(lo, std::max(ix0, 2 * ix1 - qhi));
    if (img_w < 2 * reach) { lo = ix0; hi = ix1; }  // narrower than a border: several reflections, send it whole
    lo = ix0 + ((lo - ix0) & ~7);                    // 8-pixel groups, as the rows of every image buffer
    hi = std::min(ix1, ix0 + ((hi - ix0 + 7) & ~7));
    *x0 = lo - ix0; *x1 = hi - ix0;
    return hi > lo;
}

// flags & STX_STRIP_MASK_BITS: the mask rows hold one bit per pixel (0 / 255 masks only)
static void strip_layout(int w, int h, int flags, size_t* img_stride, size_t* mask_stride, size_t* bytes)
{
    *img_stride = align_up(align_up((size_t)w, 8) * 3, 64);
    *mask_stride = (flags & STX_STRIP_MASK_BITS) ? align_up(align_up((size_t)w, 8) / 8, 64) : align_up(align_up((size_t)w, 8), 64);
    *bytes = (*img_stride + *mask_stride) * (size_t)h;
}

STX_EXPORT int stx_strip_bytes(int w, int h, int flags, size_t* out_bytes)
{
    if (!out_bytes || w <= 0 || h <= 0) return stx_fail(STX_ERR_INVALID, "strip of %dx%d", w, h);
    size_t si, sm;
    strip_layout(w, h, flags, &si, &sm, out_bytes);
    return STX_OK;
}

// The same range along y (rows [y0, y1) of an image that the rows [by0, by1) of the panorama depend on): pyrDown / pyrUp and
// the feed geometry are the same along both axes, so this is mb_level_regions + mb_contrib_range + mb_strip_range with
// (ry, rh, fy, fh, tly, img_h) in the places of (rx, rw, fx, fw, tlx, img_w).  The column version's 8-sample region alignment
// (vector lanes of the kernels) is kept: a wider region is a superset.  Rows are cut to even positions only.
static bool mb_strip_range_y(const stx_blender* b, int img_w, int img_h, int tlx, int tly, int by0, int by1, int* y0, int* y1)
{
    int fx, fy, fw, fh;
    mb_feed_rect(b, img_w, img_h, tlx, tly, &fx, &fy, &fw, &fh);
    const int nb = b->num_bands;
    int yb[STX_MAX_BANDS + 1], ye[STX_MAX_BANDS + 1];
    yb[0] = by0; ye[0] = by1;
    for (int i = 1; i <= nb; i++) {
        const int ph = b->rh >> i;
        const int al = i <= nb - 3 ? 7 : 1;
        yb[i] = std::max(0, (yb[i - 1] >> 1) - 1) & ~al;
        ye[i] = std::min(ph, ((((ye[i - 1] - 1) >> 1) + 2) + al) & ~al);
    }
    const int al = (1 << nb) - 1;
    long long lo = yb[0], hi = ye[0];
    for (int i = 1; i <= nb; i++) {
        lo = std::min(lo, (long long)yb[i] << i);
        hi = std::max(hi, (long long)ye[i] << i);
    }
    lo = lo & ~(long long)al;
    hi = (hi + al) & ~(long long)al;
    lo = std::max(lo, (long long)fy);
    hi = std::min(hi, (long long)fy + fh);
    if (hi <= lo) return false;
    const int sy0 = (int)lo, sy1 = (int)hi;
    const int reach = 3 * (1 << nb) + (1 << nb);
    const int iy0 = tly - b->ry, iy1 = iy0 + img_h;
    const int qlo = std::max(sy0 - reach, fy), qhi = std::min(sy1 + reach, fy + fh);
    int l = std::max(iy0, qlo), h = std::min(iy1, qhi);
    if (qlo < iy0) h = std::max(h, std::min(iy1, 2 * iy0 - qlo));
    if (qhi > iy1) l = std::min(l, std::max(iy0, 2 * iy1 - qhi));
    if (img_h < 2 * reach) { l = iy0; h = iy1; }
    l = iy0 + ((l - iy0) & ~1);
    h = std::min(iy1, iy0 + ((h - iy0 + 1) & ~1));
    *y0 = l - iy0; *y1 = h - iy0;
    return h > l;
}

STX_EXPORT int stx_view_rect(const stx_blender* b, int img_w, int img_h, int tlx, int tly, int band_x0, int band_x1, int band_y0,
                             int band_y1, int out_x0x1y0y1[4])
{
    if (!b || !out_x0x1y0y1) return stx_fail(STX_ERR_INVALID, "null argument");
    if (b->kind != STX_BLEND_MULTIBAND) return stx_fail(STX_ERR_UNSUPPORTED, "multi-band blender only");
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    if (!mb_strip_range(b, img_w, img_h, tlx, tly, band_x0, band_x1, &x0, &x1) ||
        !mb_strip_range_y(b, img_w, img_h, tlx, tly, band_y0, band_y1, &y0, &y1))
        x0 = x1 = y0 = y1 = 0;
    out_x0x1y0y1[0] = x0; out_x0x1y0y1[1] = x1; out_x0x1y0y1[2] = y0; out_x0x1y0y1[3] = y1;
    return STX_OK;
}

STX_EXPORT int stx_strip_rect(const stx_blender* b, int img_w, int img_h, int tlx, int tly, int band_x0, int band_x1, int out_x0x1[2],
                              size_t* out_bytes)
{
    if (!b || !out_x0x1) return stx_fail(STX_ERR_INVALID, "null argument");
    if (b->kind != STX_BLEND_MULTIBAND) return stx_fail(STX_ERR_UNSUPPORTED, "multi-band blender only");
    int x0 = 0, x1 = 0;
    if (!mb_strip_range(b, img_w, img_h, tlx, tly, band_x0, band_x1, &x0, &x1)) x0 = x1 = 0;
    out_x0x1[0] = x0; out_x0x1[1] = x1;
    if (out_bytes) {
        size_t si, sm, nb = 0;
        if (x1 > x0) strip_layout(x1 - x0, img_h, 0, &si, &sm, &nb);
        *out_bytes = nb;
    }
    return STX_OK;
}

// columns [x0, x1) of u8x3 images and of their u8 masks -> one flat buffer each: the image rows (pitch as an image buffer
// of that width has it), then the mask rows.  All strips of a call are copied by one kernel launch per 16 strips.
static int strip_pack_batch_impl(stx_ctx* ctx, int n, const stx_buf* const* imgs, const stx_buf* const* masks, const int* x0s,
                                 const int* x1s, int flags, stx_buf** out_packed)
{
    if (!ctx || n < 0 || (n > 0 && (!imgs || !masks || !x0s || !x1s || !out_packed))) return stx_fail(STX_ERR_INVALID, "null argument");
    STX_TRY(stx_set_device(ctx));
    std::vector<stx_buf*> flats(n, nullptr);
    std::vector<int> ws(n);
    std::vector<size_t> si(n), sm(n);
    int rc = STX_OK;
    for (int i = 0; i < n && rc == STX_OK; i++) {
        const stx_buf *img = imgs[i], *mask = masks[i];
        if (!img || !mask || img->elem != STX_U8 || img->c != 3 || mask->elem != STX_U8 || mask->c != 1 || mask->w != img->w || mask->h != img->h)
            rc = stx_fail(STX_ERR_INVALID, "strip: u8x3 image with a u8 mask of the same size");
        else if (x0s[i] < 0 || x1s[i] > img->w || x1s[i] <= x0s[i] || (x0s[i] & 7))
            rc = stx_fail(STX_ERR_INVALID, "strip columns [%d,%d) of %d (x0 must be a multiple of 8)", x0s[i], x1s[i], img->w);
        else if (img->parent || mask->parent)
            rc = stx_fail(STX_ERR_INVALID, "strip: whole image buffers only (rows of whole 8-pixel groups)");
        else if ((flags & STX_STRIP_MASK_BITS) && !mask->mask_binary)
            rc = stx_fail(STX_ERR_INVALID, "strip: a mask can travel as bits only when it is known to hold 0 / 255");
        if (rc != STX_OK) break;
        ws[i] = x1s[i] - x0s[i];
        size_t nbytes;
        strip_layout(ws[i], img->h, flags, &si[i], &sm[i], &nbytes);
        if (nbytes > ((size_t)1 << 30)) { rc = stx_fail(STX_ERR_UNSUPPORTED, "strip larger than 1 GiB"); break; }
        rc = stx_buf_new(ctx, (int)nbytes, 1, 1, STX_U8, &flats[i]);
        if (rc == STX_OK) flats[i]->mask_binary = mask->mask_binary;
    }
    if (rc == STX_OK) rc = stx_launch_strip_pack(ctx, n, imgs, masks, x0s, ws.data(), flats.data(), si.data(), sm.data(), (flags & STX_STRIP_MASK_BITS) != 0);
    if (rc != STX_OK) {
        for (stx_buf* f : flats) stx_buf_release(f);
        return rc;
    }
    for (int i = 0; i < n; i++) out_packed[i] = flats[i];
    return STX_OK;
}

STX_EXPORT int stx_strip_pack_batch(stx_ctx* ctx, int n, const stx_buf* const* imgs, const stx_buf* const* masks, const int* x0s,
                                    const int* x1s, stx_buf** out_packed)
{
    return strip_pack_batch_impl(ctx, n, imgs, masks, x0s, x1s, 0, out_packed);
}

STX_EXPORT int stx_strip_pack_batch_ex(stx_ctx* ctx, int n, const stx_buf* const* imgs, const stx_buf* const* masks, const int* x0s,
                                       const int* x1s, int flags, stx_buf** out_packed)
{
    return strip_pack_batch_impl(ctx, n, imgs, masks, x0s, x1s, flags, out_packed);
}

STX_EXPORT int stx_strip_pack(stx_ctx* ctx, const stx_buf* img, const stx_buf* mask, int x0, int x1, stx_buf** out_packed)
{
    return stx_strip_pack_batch(ctx, 1, &img, &mask, &x0, &x1, out_packed);
}

// the image and mask of received strips: views of the flat buffers (which they keep alive); with STX_STRIP_MASK_BITS the masks
// are fresh buffers filled by one expand launch per 16 strips
static int strip_unpack_impl(int n, const stx_buf* const* packed, const int* ws, const int* hs, int flags, stx_buf** out_imgs,
                             stx_buf** out_masks)
{
    const bool bits = (flags & STX_STRIP_MASK_BITS) != 0;
    std::vector<const uint8_t*> bit_rows(n, nullptr);
    std::vector<size_t> sms(n, 0);
    for (int i = 0; i < n; i++) out_imgs[i] = out_masks[i] = nullptr;
    int rc = STX_OK;
    for (int i = 0; i < n && rc == STX_OK; i++) {
        const stx_buf* p = packed[i];
        size_t si, sm, nbytes;
        if (!p || ws[i] <= 0 || hs[i] <= 0) { rc = stx_fail(STX_ERR_INVALID, "strip of %dx%d", ws[i], hs[i]); break; }
        strip_layout(ws[i], hs[i], flags, &si, &sm, &nbytes);
        if (p->elem != STX_U8 || p->c != 1 || p->h != 1 || (size_t)p->w < nbytes) {
            rc = stx_fail(STX_ERR_INVALID, "packed strip of %d bytes, %zu needed for %dx%d", p->w, nbytes, ws[i], hs[i]);
            break;
        }
        stx_buf* root = const_cast<stx_buf*>(p);
        for (int k = 0; k < (bits ? 1 : 2); k++) {
            stx_buf* v = new stx_buf();
            v->ctx = p->ctx; v->base = p->base;
            v->ptr = p->ptr + (k ? si * (size_t)hs[i] : 0);
            v->w = ws[i]; v->h = hs[i]; v->c = k ? 1 : 3; v->elem = STX_U8;
            v->stride = k ? sm : si;
            v->parent = root;
            v->mask_binary = k && (flags & STX_CONTRIB_U8_BINARY) ? 1 : 0;
            stx_buf_retain(root);
            (k ? out_masks : out_imgs)[i] = v;
        }
        if (bits) {
            rc = stx_buf_new(p->ctx, ws[i], hs[i], 1, STX_U8, &out_masks[i]);
            if (rc == STX_OK) out_masks[i]->mask_binary = 1;
            bit_rows[i] = p->ptr + si * (size_t)hs[i];
            sms[i] = sm;
        }
    }
    if (rc == STX_OK && bits && n > 0) {
        rc = stx_set_device(packed[0]->ctx);
        if (rc == STX_OK) rc = stx_launch_strip_bits_expand(packed[0]->ctx, n, bit_rows.data(), sms.data(), out_masks);
    }
    if (rc != STX_OK)
        for (int i = 0; i < n; i++) { stx_buf_release(out_imgs[i]); stx_buf_release(out_masks[i]); out_imgs[i] = out_masks[i] = nullptr; }
    return rc;
}

STX_EXPORT int stx_strip_unpack(const stx_buf* packed, int w, int h, int flags, stx_buf** out_img, stx_buf** out_mask)
{
    if (!packed || !out_img || !out_mask) return stx_fail(STX_ERR_INVALID, "null argument");
    return strip_unpack_impl(1, &packed, &w, &h, flags, out_img, out_mask);
}

STX_EXPORT int stx_blend_feed_strips(stx_blender* b, int n, const stx_buf* const* packed, const int* ws, const int* hs, const int* tlxs,
                                     const int* tlys, const int* orders, int flags)
{
    if (!b || n < 0 || (n > 0 && (!packed || !ws || !hs || !tlxs || !tlys || !orders))) return stx_fail(STX_ERR_INVALID, "null argument");
    if (n == 0) return STX_OK;
    std::vector<stx_buf*> imgs(n, nullptr), masks(n, nullptr);
    STX_TRY(strip_unpack_impl(n, packed, ws, hs, flags, imgs.data(), masks.data()));
    int rc = STX_OK;
    for (int i = 0; i < n; i++) {
        if (rc == STX_OK) rc = stx_blend_feed_ex(b, imgs[i], masks[i], tlxs[i], tlys[i], orders[i]);
        stx_buf_release(imgs[i]);  // the blender holds its own references
        stx_buf_release(masks[i]);
    }
    return rc;
}

STX_EXPORT int stx_blend_export_contrib(stx_blender* b, int order, int band_x0, int band_x1, stx_buf** out_packed,
                                        int out_rect_xywh[4])
{
    return stx_blend_export_contribs(b, 1, &order, &band_x0, &band_x1, out_packed, out_rect_xywh);
}

// All strips a rank owes in one call: the (strip, level) argument blocks are grouped by kernel instantiation and every
// group is ONE launch (blockIdx.z = block), instead of levels x strips small launches.
STX_EXPORT int stx_blend_export_contribs(stx_blender* b, int n, const int* orders, const int* band_x0s, const int* band_x1s,
                                         stx_buf** out_packed, int* out_rects_xywh)
{
    if (!b || n < 0 || (n > 0 && (!orders || !band_x0s || !band_x1s || !out_packed || !out_rects_xywh)))
        return stx_fail(STX_ERR_INVALID, "null argument");
    if (b->kind != STX_BLEND_MULTIBAND) return stx_fail(STX_ERR_UNSUPPORTED, "multi-band blender only");
    if (b->finished) return stx_fail(STX_ERR_STATE, "export after blend()");
    if (!b->ctx) return stx_fail(STX_ERR_STATE, "geometry-only blender (created without a context)");
    if (n == 0) return STX_OK;
    STX_TRY(stx_set_device(b->ctx));
    const int nb = b->num_bands;
    std::vector<StxMbImage> srcs(n);
    std::vector<int> sx0(n), sx1(n);
    for (int i = 0; i < n; i++) {
        const StxMbImage* src = nullptr;
        for (const StxMbImage& im : b->images) if (im.kind == 0 && im.order == orders[i]) src = &im;
        if (!src) return stx_fail(STX_ERR_INVALID, "no fed image with order %d", orders[i]);
        if (!mb_contrib_range(b, src->fx, src->fw, band_x0s[i], band_x1s[i], &sx0[i], &sx1[i]))
            return stx_fail(STX_ERR_INVALID, "image %d does not reach the columns [%d,%d)", orders[i], band_x0s[i], band_x1s[i]);
        srcs[i] = *src;
    }
    STX_TRY(mb_ensure_pyramids(b));
    std::vector<stx_buf*> packed(n, nullptr);
    auto release_all = [&]() { for (stx_buf* p : packed) stx_buf_release(p); };
    std::vector<ContribLayout> Ls(n);
    for (int i = 0; i < n; i++) {
        mb_contrib_layout(nb, sx1[i] - sx0[i], srcs[i].fh, &Ls[i]);
        int rc = stx_buf_new(b->ctx, (int)std::min<size_t>(Ls[i].bytes, 1u << 30), (int)((Ls[i].bytes + (1u << 30) - 1) >> 30), 1, STX_U8,
                             &packed[i]);
        if (rc == STX_OK && packed[i]->stride * (size_t)packed[i]->h < Ls[i].bytes) rc = stx_fail(STX_ERR_OOM, "contribution too large");
        if (rc != STX_OK) { release_all(); return rc; }
    }
    StxMbImage* d_srcs = nullptr;
    int rc = mb_upload(b, srcs.data(), n, &d_srcs);
    if (rc != STX_OK) { release_all(); return rc; }
    // one argument block per (strip, level), sorted by the kernel instantiation it needs
    struct Item { int cls; MbLevelK K; };
    std::vector<Item> items;
    double bytes = 0.0;
    for (int i = 0; i < n; i++) {
        const StxMbImage& one = srcs[i];
        const int sh = one.fh;
        std::vector<StxMbImage> single(1, one);
        for (int lv = 0; lv <= nb; lv++) {
            MbLevelK K;
            mb_fill_common(b, &K, d_srcs + i, 1, lv);
            K.all_u8 = one.img0_is_s16 ? 0 : 1;
            K.x0 = sx0[i] >> lv; K.x1 = sx1[i] >> lv; K.y0 = one.fy >> lv; K.y1 = (one.fy + sh) >> lv;
            K.emit = 1;
            K.out = (short*)(packed[i]->ptr + Ls[i].g_off[lv]); K.out_stride = Ls[i].g_stride[lv];
            K.out_plane = Ls[i].g_stride[lv] * std::max(sh >> lv, 1);
            K.out_x0 = K.x0; K.out_y0 = K.y0;
            K.out_w = (float*)(packed[i]->ptr + Ls[i].w_off[lv]); K.out_w_stride = Ls[i].w_stride[lv];
            if (K.x1 <= K.x0 || K.y1 <= K.y0) continue;
            Item it;
            it.cls = stx_fast_mb_emit_class(K, &it.K);
            if (it.cls < 0) it.cls = lv == 0 ? -1 : -2;  // generic kernel, level 0 / level >= 1 instantiation
            items.push_back(it);
            bytes += mb_level_bytes(b, single, lv, K.x0, K.x1, true, false);
        }
    }
    std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& c) { return a.cls < c.cls; });
    std::vector<MbLevelK> Ks(items.size());
    std::vector<int> classes(items.size());
    for (size_t i = 0; i < items.size(); i++) { Ks[i] = items[i].K; classes[i] = items[i].cls; }
    void* d_Ks = nullptr;
    rc = upload_small(b->ctx, Ks.data(), Ks.size() * sizeof(MbLevelK), &d_Ks);
    if (rc == STX_OK) {
        rc = stx_launch_mb_emit_batch(b->ctx, (const MbLevelK*)d_Ks, Ks.data(), classes.data(), (int)Ks.size(), bytes);
        stx_dev_free(b->ctx, d_Ks);  // stream-ordered reuse
    }
    if (rc != STX_OK) { release_all(); return rc; }
    for (int i = 0; i < n; i++) {
        out_rects_xywh[4 * i] = sx0[i]; out_rects_xywh[4 * i + 1] = srcs[i].fy;
        out_rects_xywh[4 * i + 2] = sx1[i] - sx0[i]; out_rects_xywh[4 * i + 3] = srcs[i].fh;
        packed[i]->mask_binary = (!srcs[i].img0_is_s16 && srcs[i].mask_binary) ? 1 : 0;  // read back with stx_buf_flags
        out_packed[i] = packed[i];
    }
    return STX_OK;
}

STX_EXPORT int stx_blend_build(stx_blender* b)
{
    if (!b) return stx_fail(STX_ERR_INVALID, "null argument");
    if (b->kind != STX_BLEND_MULTIBAND) return STX_OK;
    if (b->finished) return stx_fail(STX_ERR_STATE, "build after blend()");
    if (!b->ctx) return stx_fail(STX_ERR_STATE, "geometry-only blender (created without a context)");
    STX_TRY(stx_set_device(b->ctx));
    return mb_ensure_pyramids(b);
}

STX_EXPORT int stx_blend_feed_contrib(stx_blender* b, int order, const int rect_xywh[4], const stx_buf* packed)
{
    return stx_blend_feed_contrib_ex(b, order, rect_xywh, packed, 0);
}

STX_EXPORT int stx_blend_feed_contrib_ex(stx_blender* b, int order, const int rect_xywh[4], const stx_buf* packed, int flags)
{
    if (!b || !rect_xywh || !packed) return stx_fail(STX_ERR_INVALID, "null argument");
    if (b->kind != STX_BLEND_MULTIBAND) return stx_fail(STX_ERR_UNSUPPORTED, "multi-band blender only");
    if (b->finished) return stx_fail(STX_ERR_STATE, "feed after blend()");
    if (!b->ctx) return stx_fail(STX_ERR_STATE, "geometry-only blender (created without a context)");
    const int nb = b->num_bands, al = (1 << nb) - 1;
    const int x = rect_xywh[0], y = rect_xywh[1], w = rect_xywh[2], h = rect_xywh[3];
    if (w <= 0 || h <= 0 || ((x | y | w | h) & al) || x < 0 || y < 0 || x + w > b->rw || y + h > b->rh)
        return stx_fail(STX_ERR_INVALID, "contribution rect (%d,%d,%d,%d) is not a 2^bands-aligned part of the roi", x, y, w, h);
    ContribLayout L;
    mb_contrib_layout(nb, w, h, &L);
    if (packed->elem != STX_U8 || packed->c != 1 || packed->stride * (size_t)packed->h < L.bytes)
        return stx_fail(STX_ERR_INVALID, "contribution buffer holds %zu bytes, layout needs %zu", packed->stride * (size_t)packed->h, L.bytes);
    StxMbImage im;
    memset(&im, 0, sizeof(im));
    im.kind = 1;
    im.order = order;
    im.mask_binary = (flags & STX_CONTRIB_U8_BINARY) ? 1 : 0;
    im.fx = x; im.fy = y; im.fw = w; im.fh = h;
    for (int i = 0; i <= nb; i++) {
        im.g[i] = (short*)(packed->ptr + L.g_off[i]); im.g_stride[i] = L.g_stride[i];
        im.g_plane[i] = L.g_stride[i] * std::max(h >> i, 1);
        im.wt[i] = (float*)(packed->ptr + L.w_off[i]); im.wt_stride[i] = L.w_stride[i];
    }
    mb_insert_sorted(b, im, true);
    b->next_order = std::max(b->next_order, order + 1);
    stx_buf_retain(const_cast<stx_buf*>(packed));
    b->held.push_back(const_cast<stx_buf*>(packed));
    return STX_OK;
}

// FeatherBlender::blend: weights of all fed images (batched distance transforms), then one gather over the panorama
static int feather_finish(stx_blender* b, stx_buf* pano, stx_buf* pmask, stx_buf* p16)
{
    stx_ctx* ctx = b->ctx;
    const int n = (int)b->feather_images.size();
    void* d_tab = nullptr;
    STX_TRY(upload_small(ctx, b->feather_images.data(), sizeof(FeatherImg) * (size_t)n, &d_tab));
    b->pyr_allocs.push_back(d_tab);
    STX_TRY(stx_launch_feather_weights(ctx, (const FeatherImg*)d_tab, b->feather_images.data(), n));
    double bytes = 4.0 * pano->w * pano->h + (p16 ? 6.0 * pano->w * pano->h : 0.0);
    for (const FeatherImg& im : b->feather_images) bytes += (double)im.w * im.h * ((im.is_s16 ? 6 : 3) + 2);
    FeatherGatherK K;
    K.imgs = (const FeatherImg*)d_tab; K.n = n; K.w = pano->w; K.h = pano->h; K.sharpness = b->sharpness;
    K.pano = pano->ptr; K.pano_stride = (long long)pano->stride; K.pmask = pmask->ptr; K.pmask_stride = (long long)pmask->stride;
    K.pano16 = p16 ? (short*)p16->ptr : nullptr; K.pano16_stride = p16 ? (long long)p16->stride : 0;
    return stx_launch_feather_gather(ctx, K, bytes);
}

// Blender::blend of the "no" blender: one gather over the panorama (stx_blend.hip: no_gather_kernel)
static int no_finish(stx_blender* b, stx_buf* pano, stx_buf* pmask, stx_buf* p16)
{
    stx_ctx* ctx = b->ctx;
    const int n = (int)b->no_images.size();
    void* d_tab = nullptr;
    STX_TRY(stx_dev_alloc(ctx, sizeof(NoImg) * std::max(n, 1), &d_tab));
    b->pyr_allocs.push_back(d_tab);
    bool all_binary = true;
    double bytes = 4.0 * pano->w * pano->h + (p16 ? 6.0 * pano->w * pano->h : 0.0);
    for (const NoImg& im : b->no_images) {
        all_binary = all_binary && im.mask_binary;
        bytes += (double)im.w * im.h;  // every mask once; the image bytes of the winners are counted with the output
    }
    bytes += 3.0 * pano->w * pano->h;
    if (n > 0) STX_TRY(stx_stage_upload(ctx, d_tab, b->no_images.data(), sizeof(NoImg) * (size_t)n));
    NoGatherK K;
    K.imgs = (const NoImg*)d_tab; K.n = n; K.all_binary = all_binary ? 1 : 0;
    K.w = pano->w; K.h = pano->h;
    K.pano = pano->ptr; K.pano_stride = (long long)pano->stride; K.pmask = pmask->ptr; K.pmask_stride = (long long)pmask->stride;
    K.pano16 = p16 ? (short*)p16->ptr : nullptr; K.pano16_stride = p16 ? (long long)p16->stride : 0;
    return stx_launch_no_gather(ctx, K, bytes);
}

STX_EXPORT int stx_blend_finish_ex(stx_blender* b, stx_buf** out_pano_u8, stx_buf** out_mask_u8, stx_buf** out_pano_s16)
{
    if (!b) return stx_fail(STX_ERR_INVALID, "null argument");
    if (b->finished) return stx_fail(STX_ERR_STATE, "blend() was already called on this blender");
    if (!b->ctx) return stx_fail(STX_ERR_STATE, "geometry-only blender (created without a context)");
    STX_TRY(stx_set_device(b->ctx));
    stx_ctx* ctx = b->ctx;
    const int ow = b->kind == STX_BLEND_MULTIBAND ? b->band_x1 - b->band_x0 : b->rw, oh = b->kind == STX_BLEND_MULTIBAND ? b->fh : b->rh;
    stx_buf *pano = nullptr, *pmask = nullptr, *p16 = nullptr;
    int rc = stx_buf_new(ctx, ow, oh, 3, STX_U8, &pano);
    if (rc == STX_OK) rc = stx_buf_new(ctx, ow, oh, 1, STX_U8, &pmask);
    if (rc == STX_OK && out_pano_s16) rc = stx_buf_new(ctx, ow, oh, 3, STX_S16, &p16);
    if (rc == STX_OK) {
        if (b->kind == STX_BLEND_MULTIBAND) rc = mb_finish(b, pano, pmask, p16);
        else if (b->kind == STX_BLEND_NO) rc = no_finish(b, pano, pmask, p16);
        else rc = feather_finish(b, pano, pmask, p16);
    }
    b->finished = true;
    blender_release(b);  // stream-ordered: the kernels above were enqueued before any reuse
    if (rc != STX_OK) {
        stx_buf_release(pano); stx_buf_release(pmask); stx_buf_release(p16);
        return rc;
    }
    if (out_pano_u8) *out_pano_u8 = pano; else stx_buf_release(pano);
    if (out_mask_u8) *out_mask_u8 = pmask; else stx_buf_release(pmask);
    if (out_pano_s16) *out_pano_s16 = p16;
    return STX_OK;
}

STX_EXPORT int stx_blend_finish(stx_blender* b, stx_buf** out_pano_u8, stx_buf** out_mask_u8)
{
    return stx_blend_finish_ex(b, out_pano_u8, out_mask_u8, nullptr);
}

STX_EXPORT int stx_blend_destroy(stx_blender* b)
{
    if (!b) return STX_OK;
    if (b->ctx) {
        hipSetDevice(b->ctx->device);
        blender_release(b);
    }
    delete b;
    return STX_OK;
}
