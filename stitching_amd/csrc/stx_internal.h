// stx_internal.h — host-side structures shared by the C-ABI translation units.
// gfx950 only; no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/stitching_amd.h"
#include "../../include/stitching_amd_debug.h"

#define STX_EXPORT extern "C" __attribute__((visibility("default")))
#define STX_MAX_BANDS 16

// ---------------------------------------------------------------------------------------------
// error reporting
// ---------------------------------------------------------------------------------------------
void stx_set_error(const char* fmt, ...);
int stx_fail(int code, const char* fmt, ...);

#define STX_HIP(call)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return stx_fail(e_ == hipErrorOutOfMemory ? STX_ERR_OOM : STX_ERR_HIP, "%s failed: %s (%s:%d)", \
                            #call, hipGetErrorString(e_), __FILE__, __LINE__);                             \
    } while (0)

#define STX_TRY(call)           \
    do {                        \
        int rc_ = (call);       \
        if (rc_ != STX_OK) return rc_; \
    } while (0)

// ---------------------------------------------------------------------------------------------
// context: stream, caching allocator, profiler
// ---------------------------------------------------------------------------------------------
struct StxProfEntry {
    std::string name;
    int64_t calls = 0;
    double total_ms = 0.0;
    double algo_bytes = 0.0;
};

struct StxPendingEvent {
    hipEvent_t start, stop;
    int entry;
};

struct stx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // side stream + private device scratch of the ROI pass: it depends on nothing queued on `stream`, so a
    // pipeline of panoramas can take its ROIs while the previous panorama is still blending
    hipStream_t aux_stream = nullptr;
    void* aux_scratch = nullptr;
    // caching allocator: bucket size -> free blocks (alloc_mutex: Python finalizers may free buffers from any thread)
    std::mutex alloc_mutex;
    std::map<size_t, std::vector<void*>> free_blocks;
    std::map<void*, size_t> block_size;
    size_t bytes_allocated = 0;
    // pinned scratch for small device->host results (ROI min/max)
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    uint32_t roi_seq = 0;  // sequence number of the last ROI pass (the stamp its blocks write behind their results)
    // pinned ring for small host->device uploads (descriptor tables): truly asynchronous copies.  The ring is cut into
    // STX_STAGE_SEGS segments; leaving a segment records an event behind its last copy, entering one waits for the event of
    // its previous lap (recorded three segments of uploads ago: normally long complete) — not for the whole stream, which
    // with a deep queue of panoramas costs the host tens of milliseconds and the GPU a bubble (stx_stage_upload)
    uint8_t* stage = nullptr;
    size_t stage_bytes = 0, stage_off = 0;
    hipEvent_t stage_ev[4] = {};
    bool stage_ev_set[4] = {};
    int stage_seg = 0;
    // profiler
    bool prof_on = false;
    std::vector<StxProfEntry> prof;
    std::map<std::string, int> prof_index;
    std::vector<StxPendingEvent> prof_pending;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t marks[16] = {};
};

#define STX_STAGE_SEGS 4
int stx_dev_alloc(stx_ctx* ctx, size_t bytes, void** out);
// host -> device copy of a small table through the context's pinned ring: queued on ctx->stream, no host wait in the common case
int stx_stage_upload(stx_ctx* ctx, void* d, const void* h, size_t bytes);
void stx_dev_free(stx_ctx* ctx, void* p);
int stx_set_device(stx_ctx* ctx);

// profiling bracket around one kernel launch
struct StxProfScope {
    stx_ctx* ctx;
    int pending = -1;
    hipStream_t stream;  // the stream the bracketed kernel is launched on (default: the context's main stream)
    bool attached;       // true: the scope holds ONE launch and the caller attaches start() / stop() to it (hipExtLaunchKernelGGL): the
                         // events then carry the dispatch's own begin / end stamps, as rocprofv3 reports them.  false: an event recorded
                         // before and one after whatever the scope launches — 3-5 us more than the kernel (tools/ubench/event_bracket.hip)
    StxProfScope(stx_ctx* c, const char* name, double algo_bytes, hipStream_t on = nullptr, bool attach = false);
    ~StxProfScope();
    hipEvent_t start() const;  // null while the profiler is off: launch plainly
    hipEvent_t stop() const;
};

// ---------------------------------------------------------------------------------------------
// device image
// ---------------------------------------------------------------------------------------------
struct stx_buf {
    stx_ctx* ctx = nullptr;
    void* base = nullptr;  // allocation (owned when parent == nullptr)
    uint8_t* ptr = nullptr;
    int w = 0, h = 0, c = 0, elem = 0;
    size_t stride = 0;  // bytes
    stx_buf* parent = nullptr;
    int mask_binary = 0;  // 1: a u8x1 image known to hold only 0 and 255 (warped masks; scanned host uploads)
    std::atomic<int> refs{1};
};

// readable bytes in front of the first row of every image the library allocates (views inherit them from their parent: whatever
// precedes a view's first pixel is memory of the same allocation)
constexpr size_t STX_BUF_FRONT_PAD = 64;
inline int stx_elem_bytes(int elem) { return elem == STX_U8 ? 1 : (elem == STX_S16 ? 2 : 4); }
int stx_buf_new(stx_ctx* ctx, int w, int h, int c, int elem, stx_buf** out);
void stx_buf_retain(stx_buf* b);
void stx_buf_release(stx_buf* b);

// ---------------------------------------------------------------------------------------------
// projector (host side of ProjectorBase::setCameraParams)
// ---------------------------------------------------------------------------------------------
// projector families behind the 16 warper ids (id -> family, a, b: stx_make_projector)
enum { STX_F_PLANE = 0, STX_F_CYLINDRICAL, STX_F_SPHERICAL, STX_F_FISHEYE, STX_F_STEREOGRAPHIC, STX_F_CRECT,
       STX_F_CRECT_PORTRAIT, STX_F_PANINI, STX_F_PANINI_PORTRAIT, STX_F_MERCATOR, STX_F_TRANSVERSE_MERCATOR };
struct StxProjector {
    int type;
    int family;
    float a, b;
    float scale;
    float k[9], rinv[9], r_kinv[9], k_rinv[9], t[3];
    int trig;  // STX_TRIG_*: the process-wide trig mode at the time the projector was made (stx_set_trig_mode)
    int remap; // STX_REMAP_*: likewise the interpolation model of the image samples (stx_set_remap_mode)
};
int stx_make_projector(int type, float scale, const float* K, const float* R, StxProjector* out);

// kernels' host launchers (defined in the .hip files) --------------------------------------------
struct StxWarpLaunch {
    StxProjector proj;
    int tlx, tly, dw, dh;          // destination roi
    const uint8_t* src; int sw, sh; size_t sstride; int src_channels;  // src may be null (mask only)
    uint8_t* dimg; size_t dimg_stride;    // u8x3 or null
    uint8_t* dmask; size_t dmask_stride;  // u8x1 or null
    int nearest_src;               // 1: out image = nearest sample of a u8x1 source (generic mask warp)
    int debug_maps = 0;            // stx_debug_warp_maps (test hook): dimg / dmask are f32 maps of x / y; 1: the kernel a warp would take, 2: the generic one
    // fused exposure gain (stx_warp_batch_gain): device tables made by stx_launch_gain_rows for this destination rectangle; null: none
    const float* gain_H = nullptr; long long gain_hstride = 0; const void* gain_yt = nullptr; int gain_gh = 0;
};
// typed projectors only (plane / affine / cylindrical / spherical / mercator), Q15 or float remap: what a fused gain needs
bool stx_warp_fast_eligible(const StxWarpLaunch& L);
int stx_launch_warp(stx_ctx* ctx, const StxWarpLaunch& L);
int stx_launch_warp_batch(stx_ctx* ctx, const StxWarpLaunch* Ls, int n);
int stx_launch_roi_minmax(stx_ctx* ctx, int n, const StxProjector* projs, const int* sizes_wh, float* out_minmax4);

// multi-band -------------------------------------------------------------------------------------
struct StxMbImage {  // device-visible descriptor of one fed image (all levels)
    // kind 0: an image fed on this rank.  kind 1: a contribution strip received from another rank:
    // per level i, g[i] holds (short)(L_i * W_i) and wt[i] holds W_i over the rect (fx,fy,fw,fh) >> i.
    int kind; int order;
    const uint8_t* img0; long long img0_stride; int img0_is_s16;
    const uint8_t* mask0; long long mask0_stride; int mask_binary;
    int iw, ih;            // image size
    int ix, iy;            // image corner relative to the (padded) panorama roi
    int fx, fy, fw, fh;    // feed rect (tl_new .. br_new) relative to the panorama roi, level 0
    int left, top;         // copyMakeBorder offsets: bordered(x,y) = img(reflect(x-left), reflect(y-top))
    // levels 1..B: planar Gaussian pyramid (3 planes) and fp32 weight pyramid.  g_u8 = 1 (every image fed as u8: all values are
    // 0..255): the planes hold one BYTE per sample — g[] then points to bytes, g_stride / g_plane count samples either way.  int16
    // images (and received contribution strips, kind 1) keep int16 planes.  64 readable bytes in front of every allocation.
    int g_u8;
    short* g[STX_MAX_BANDS + 1]; long long g_stride[STX_MAX_BANDS + 1]; long long g_plane[STX_MAX_BANDS + 1];
    float* wt[STX_MAX_BANDS + 1]; long long wt_stride[STX_MAX_BANDS + 1];
    // w1_f16 = 1 (round 6: an image fed on this rank whose mask holds only 0 / 255): wt[1] points to IEEE HALF values (wt_stride[1] still counts
    // samples).  W_0 is then exactly 0.f / 1.f, pyrDown's 25-tap sum a small integer k <= 256 in whatever order it is taken, and
    // W_1 = k / 256 has a 9-bit significand: the half holds it exactly, the conversion back is exact, every consumer sees the same
    // fp32 value as before — at 2 instead of 4 of the 7 bytes a level-1 sample costs (written once, read by the level-1 pyrDown and by
    // the level-1 gather).  Levels >= 2 need 17 and more bits and stay fp32.
    int w1_f16;
    // occupancy of the weight pyramid (null: not recorded): occ[i][p * nt + t] != 0 iff W_i has a non-zero value in the rows
    // 2 p, 2 p + 1 and the columns 64 t .. 64 t + 63 of level i (frame coordinates), nt = ((fw >> i) + 63) / 64 rounded up to 4.  Every entry is
    // written by the pyramid kernel that produces the level (one byte store per half-wavefront: no atomics, no clearing);
    // the gather kernels read it to pass over the empty parts of a feed rectangle (bounding boxes of pitched / rolled frames,
    // seam masks, exchange strips) a wavefront at a time.
    uint8_t* occ[STX_MAX_BANDS + 1];
};
// pyr_mode / pyr_lanes: STX_PYRDOWN_* (include/stitching_amd.h); anything but SCALAR builds every level with the generic kernels
int stx_launch_mb_pyramids(stx_ctx* ctx, const StxMbImage* d_images, const StxMbImage* h_images, int n, int num_bands, int pyr_mode,
                           int pyr_lanes);
struct MbLevelK;
int stx_launch_mb_level(stx_ctx* ctx, const MbLevelK& K, double algo_bytes);
int stx_launch_mb_coarse(stx_ctx* ctx, const MbLevelK& K_level_Bm2, double algo_bytes);  // levels B, B-1, B-2 in one launch

// pointwise exposure gain (next row N1) --------------------------------------------------------------
int stx_launch_gain_apply(stx_ctx* ctx, stx_buf* img, const float g[3]);
int stx_launch_block_gain(stx_ctx* ctx, stx_buf* img, const stx_buf* gmap, const int* d_xt, const int* d_yt);
int stx_launch_block_gain_batch(stx_ctx* ctx, int n, stx_buf* const* imgs, const stx_buf* const* gmaps, const int* full_wh_xy0,
                                float* const* Hs, void* const* yts, const int* fast);
// only the first half of it — H rows and row tables of n rectangles (w, h at (x0, y0) of a full_w x full_h image each) — for a consumer
// that multiplies the gain in itself (the warp kernel's epilogue); wh = {w, h} per rectangle
int stx_launch_gain_rows(stx_ctx* ctx, int n, const int* wh, const stx_buf* const* gmaps, const int* full_wh_xy0, float* const* Hs,
                         void* const* yts);
// cv::resize(INTER_LINEAR_EXACT) u8 (next rows N2 / N3); d_xt / d_yt: device tables of (offset, coeff1 | interior << 16)
int stx_launch_resize_exact(stx_ctx* ctx, const stx_buf* src, stx_buf* dst, const int* d_xt, const int* d_yt, bool dilate,
                            const stx_buf* andmask);

int stx_launch_seam_resize_batch(stx_ctx* ctx, int n, const stx_buf* const* seams, const stx_buf* const* masks, stx_buf* const* dsts,
                                 const int* const* d_xt, const int* const* d_yt, uint8_t* const* tmp, const size_t* tstride);
// SeamFinder.resize of n images as ONE launch (coefficients made in the kernel, dilation in LDS); full_wh_xy0 as stx_seam_mask_resize_batch_sub
int stx_launch_seam_resize_lds(stx_ctx* ctx, int n, const stx_buf* const* seams, const stx_buf* const* masks, stx_buf* const* dsts,
                               const int* full_wh_xy0, bool* done);
// image-strip sharding: pack the columns [x0, x0 + w) of n images + masks into n flat buffers (one launch per 16 strips)
int stx_launch_strip_pack(stx_ctx* ctx, int n, const stx_buf* const* imgs, const stx_buf* const* masks, const int* x0, const int* w,
                          stx_buf* const* dsts, const size_t* si, const size_t* sm, bool mask_bits);
int stx_launch_strip_bits_expand(stx_ctx* ctx, int n, const uint8_t* const* bits, const size_t* sm, stx_buf* const* masks);

// saturation of the L1 distance transform (OpenCV's 16.16 fixed point clamps at INT_MAX >> 2 = 8192.0f): the kernels clamp to it, the
// sharded feather blender sizes its halo by it (stx_debug_feather_dist_cap -> distributed.FEATHER_DIST_CAP, checked by a host test)
constexpr int STX_FEATHER_DIST_CAP = 8192;
constexpr int STX_DT_RC = 64;  // rows per chunk of the distance transform's column pass (stx_blend.hip: DT_RC)
// feather blender as a deferred gather: device table of the fed images, in feed order
struct FeatherImg {
    const uint8_t* img; long long istride; int is_s16;
    const uint8_t* mask; long long mstride;
    int x, y, w, h;                 // rectangle inside the panorama roi
    uint16_t* dist; long long dstride; // L1 distance to the nearest zero of the mask, saturated at 8192; elements per row (multiple of 16)
    int* first; int* last; unsigned long long* zbits; int n_chunks;  // column-pass summaries (DT_RC rows per chunk): first / last zero row, zero rows as a bit set
};
struct FeatherGatherK {
    const FeatherImg* imgs; int n; int w, h; float sharpness;
    uint8_t* pano; long long pano_stride; uint8_t* pmask; long long pmask_stride; short* pano16; long long pano16_stride;
};
int stx_launch_feather_weights(stx_ctx* ctx, const FeatherImg* d_imgs, const FeatherImg* h_imgs, int n);
int stx_launch_feather_gather(stx_ctx* ctx, const FeatherGatherK& K, double algo_bytes);

// "no" blender as a deferred gather: device table of the fed images, in feed order
struct NoImg { const uint8_t* img; long long istride; const uint8_t* mask; long long mstride; int is_s16; int x, y, w, h; int mask_binary; };
struct NoGatherK {
    const NoImg* imgs; int n; int all_binary;  // all_binary: every mask holds only 0 / 255
    int w, h;
    uint8_t* pano; long long pano_stride; uint8_t* pmask; long long pmask_stride; short* pano16; long long pano16_stride;
};
int stx_launch_no_gather(stx_ctx* ctx, const NoGatherK& K, double algo_bytes);

// simple blenders --------------------------------------------------------------------------------

