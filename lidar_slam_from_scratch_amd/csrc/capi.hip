// capi.hip -- host side of libicp_mi355x.so: the C ABI declared in include/icp_mi355x.h.
//
// Orchestrates the kernels of kernels.h on one HIP stream per context.  The ICP loop
// (icp.hpp:181-232) runs device-side: the error, both convergence tests, the 6x6 solve
// and the pose accumulation are done by k_finish_step, and once the loop has ended the
// remaining queued kernels return at their first instruction.  The host only looks at a
// 4-byte flag, two iterations behind the launch front, to stop queueing.
//
// There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>

#include <dirent.h>
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/icp_mi355x.h"
#include "kernels.h"
#include "nn_mfma.h"
#include "icp_small.h"
#include "knn_lists.h"
#include "nn_bounded.h"
#include "nn_culled.h"
#include "voxel.h"
#include "scan_context.h"
#include "occupancy.h"

using namespace icpmi;

namespace icpmi {
hipError_t sort_pairs_u32(void *temp, size_t *temp_bytes, const unsigned *keys_in, unsigned *keys_out,
                          const unsigned *vals_in, unsigned *vals_out, unsigned n, hipStream_t stream);
hipError_t sort_pairs_u64(void *temp, size_t *temp_bytes, const unsigned long long *keys_in,
                          unsigned long long *keys_out, const unsigned *vals_in, unsigned *vals_out,
                          unsigned n, hipStream_t stream);
hipError_t run_lengths_u64(void *temp, size_t *temp_bytes, const unsigned long long *keys, unsigned n,
                           unsigned long long *unique_out, unsigned *counts_out, unsigned *runs_out,
                           hipStream_t stream);
hipError_t exclusive_sum_u32(void *temp, size_t *temp_bytes, const unsigned *in, unsigned *out, unsigned n,
                             hipStream_t stream);
hipError_t sort_keys_u64(void *temp, size_t *temp_bytes, const unsigned long long *keys_in,
                         unsigned long long *keys_out, unsigned n, hipStream_t stream);
hipError_t merge_keys_u64(void *temp, size_t *temp_bytes, const unsigned long long *a, const unsigned long long *b,
                          unsigned long long *out, unsigned na, unsigned nb, hipStream_t stream);
}

namespace {

// last error of calls that have no context (icpmi_create, icpmi_load_cloud): per thread, so that
// contexts created on different threads (slam_icp_adapter.hpp holds one per thread) do not race
thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct EventPair {
    hipEvent_t a, b;
    int stage;
};

enum Stage { ST_NN = 0, ST_REDUCE = 1, ST_TRANSFORM = 2, ST_NORMALS = 3, ST_TOTAL = 4, ST_LOOP = 5, ST_SETUP = 6, ST_COARSE = 7, ST_EXCHANGE = 8 };

// roctx ranges around the stages of a call (SURVEY section 5 "Tracing"): visible to
// `rocprofv3 --marker-trace`.  The library is looked up at run time (the rocprofiler-sdk's roctx
// first, the older libroctx64 otherwise) and a missing one only means no ranges.  Costs two indirect calls per stage when nothing listens.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        // rocprofv3 traces the rocprofiler-sdk's roctx; the older libroctx64 is what roctracer-era tools see
        const char *names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1",
                               "libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so.4"};
        for (const char *nm : names) {
            void *lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (!lib) continue;
            push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
const Roctx &roctx()
{
    static Roctx r;
    return r;
}
struct Range { // RAII: one roctx range
    explicit Range(const char *name) { if (roctx().push) roctx().push(name); }
    ~Range() { if (roctx().pop) roctx().pop(); }
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
};

} // namespace

// A KITTI .bin on its way from disk to the device while the caller's thread runs the previous frame
// (icpmi_stream_prefetch_file): a worker thread reads it into pinned memory (~140 us for a 1.8 MB scan
// out of the page cache, nothing of the GPU's), copies it over on a stream of its own (the copy engine
// runs beside the compute of the frame in flight; on the context's stream it would queue behind it) and
// widens it there (k_widen_f32).  Two device slots: the push of a frame works out of one while the next
// frame lands in the other.
struct FilePrefetch {
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::string want;         // path being read (valid while `busy`)
    std::string ready[2];     // per device slot: the path whose points wait there for the push of that path
    void *h_buf[2] = {nullptr, nullptr};   // pinned staging, one per device slot (the worker's alone)
    size_t h_cap[2] = {0, 0};
    // The worker does not wait for its own GPU work: it records `filled[slot]` behind the copies, the widening and the
    // filter and announces the file; the push of that file waits for the event and reads the filter's outcome (number
    // of voxels, range flag) from pinned memory.  The worker's host time per file is then the read alone.
    hipEvent_t filled[2] = {nullptr, nullptr};
    unsigned *h_runs = nullptr;            // pinned: per slot {voxels, flag}
    bool pending[2] = {false, false};      // announced, event not yet waited for
    bool pend_filtered[2] = {false, false};
    double pend_voxel[2] = {0.0, 0.0};
    void *d_f32[2] = {nullptr, nullptr};   // device: the float32 records as read
    double *d_raw[2] = {nullptr, nullptr}; // device: N x 3 fp64 raw points
    size_t d_cap[2] = {0, 0};              // records each slot holds
    int64_t ready_n[2] = {0, 0};
    unsigned long long ready_seq[2] = {0, 0}, seq = 0; // which pending file is the older one
    hipEvent_t slot_done[2] = {nullptr, nullptr};      // recorded on the context's stream behind the push that read a slot
    bool slot_used[2] = {false, false};                // ... if any push has read it yet
    int taking = -1;                                   // the slot a push is reading right now: never the worker's choice
    hipStream_t stream = nullptr;
    bool busy = false, quit = false;
    int device = 0;
    // ... and filtered there too (voxel_filter_core on the worker's stream, with working memory of its own), with the
    // voxel size of the stream's last push: the push of the file then starts at the registration
    double voxel_hint = 0.0;               // 0: no push yet, the worker stops after the widening
    DevBuf filtered[2];                    // per slot: the filtered scan ...
    int64_t filtered_n[2] = {0, 0};        // ... its rows ...
    double filtered_voxel[2] = {0.0, 0.0}; // ... and the voxel size it was made with (0: not filtered)
    DevBuf sc_bbox, sc_box, sc_keys, sc_vals, sc_tmp;
    // where the worker's time goes (ICPMI_PREFETCH_STATS=1: printed to stderr when the context is destroyed)
    double t_wait = 0, t_read = 0, t_upload = 0, t_filter = 0;
    long files = 0;
};

// One more registration beside the caller's (icpmi_align_batch): a helper context -- stream, workspace -- and the
// host thread that drives it (align_device spins on the device's progress words, so every concurrent
// registration needs a host thread); the thread is kept between calls (starting one costs tens of microseconds,
// a tenth of a small registration).
struct BatchWorker {
    icpmi_ctx *helper = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = false, quit = false;
};

struct icpmi_ctx {
    icpmi_options opt;
    hipStream_t stream = nullptr;
    std::string err;
    int cu_count = 256;

    DevBuf cur, nrm, idx, part_d2, part_idx, partials, history, stage_a, stage_b, stage_c, d2out;
    DevBuf knn_idx, slotmin, fb_list;         // k-NN lists, slot minima, rows for the exact fallback
    DevBuf nn_lists;                          // bounded 1-NN pass: per row bound, threshold, count, listed columns (nn_bounded.h)
    DevBuf nrm_sorted;                        // small-cloud kernel: the target normals in Morton order (SoA, stride nn_ms)
    DevBuf vox_keys, vox_vals, vox_out;             // voxel filter: 64-bit keys (in/out/unique), values + run data, result
    DevBuf stream_prev, stream_cur, f32_stage;      // odometry stream: previous / current filtered scan; float32 upload staging
    int64_t stream_prev_n = -1;                     // rows of stream_prev (-1: no frame yet)
    int64_t stream_cur_n = 0;                       // rows of stream_cur while a push is registering it
    DevBuf grid_set, grid_in, grid_out, grid_cnt, world; // occupancy grid: the set (sorted unique keys), {set, new keys}, sorted, run data; world points
    int64_t grid_n = 0;                             // cells in grid_set
    unsigned *h_grid = nullptr;                     // pinned: the set's size on its way back
    unsigned long long *h_cnt = nullptr;            // pinned: the resolve's counters on their way back (profiling)
    FilePrefetch *prefetch = nullptr;               // worker reading the next frame file (icpmi_stream_prefetch_file)
    // the target whose search structure and normals the context's buffers currently hold (prepare_target):
    // icpmi_stream_push prepares the NEXT frame's target while the caller is still busy with this frame's result
    const double *prep_tgt = nullptr;
    int prep_m = 0, prep_engine = 0;
    bool prep_valid = false;
    hipEvent_t result_ready = nullptr;              // recorded behind a call's result copies (stream path)
    bool trailing_work = false;                     // a push returned with the next target's preparation still queued
    // odometry stream: the NEXT target's search structure and normals are built on a stream and workspace of their
    // own (a helper context) BESIDE the registration of the current frame, and adopted by swapping buffers
    icpmi_ctx *prep_helper = nullptr;
    hipEvent_t prep_done = nullptr;                 // recorded on the helper's stream behind a preparation
    hipEvent_t scan_ready = nullptr;                // recorded on this context's stream behind the filter of the scan to prepare
    bool helper_busy = false;                       // a preparation is queued on the helper and nothing has waited for it yet
    // where the calling thread's time goes in the frame stream (ICPMI_STREAM_STATS=1: printed when the context is destroyed)
    double t_push = 0, t_wait_file = 0, t_prep_queue = 0, t_align = 0, t_gpu_span = 0;
    long pushes = 0;
    DevBuf sort_keys, sort_tmp, tgt_sorted, frames; // Morton pre-pass: keys/values, sorted copy, split frames
    DevBuf bpack, coarse, bbox_part, nn_misc; // MFMA engine: operands, coarse minima, frame + counters
    DevBuf src_sort;                          // Morton order of the source rows (the loop's internal order)

    DevBuf grp_cnt, grp_items;                // culled engine: per target split the list of 64-row groups within reach (nn_culled.h)
    bool nn_pruned = false;                   // the culled engine (ICPMI_SEARCH_MFMA_PRUNED; AUTO on large targets)
    int nn_engine = ICPMI_SEARCH_EXACT_F64;   // engine prepared for the current target
    int nn_splits = 0;
    int nn_ms = 0;                            // component stride of the SoA sorted target
    int grp_cap = 0;                          // groups a split's list in grp_items can hold
    IcpState *d_state = nullptr;   // two of them (align_device alternates in the sharded loop)
    IcpState *h_state = nullptr;   // pinned
    double *h_hist = nullptr;      // pinned: the error history of the last call
    size_t h_hist_cap = 0;
    void *h_file = nullptr;        // pinned: a frame file on its way to the device (icpmi_stream_push_file)
    size_t h_file_cap = 0;
    int32_t *h_flags = nullptr;    // host-mapped ring: (iteration + 1) * 2 + done, written by the device
    int32_t *d_flags = nullptr;    // the same words through the device's address space

    // profiling
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    unsigned coarse_seen = 0;      // k_nn_coarse launches since the context was made (profile 1 samples them)
    unsigned exchange_seen = 0;    // ... and the per-pass all-reduces of a sharded run
    icpmi_profile prof;

    // multi-GPU
    Rccl rccl;
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
    icpmi_allreduce_fn cb_allreduce = nullptr;
    icpmi_allgather_fn cb_allgather = nullptr;
    void *cb_user = nullptr;
    std::vector<double> cb_host;
    std::vector<BatchWorker *> helpers; // icpmi_align_batch: one more stream + workspace + host thread per concurrent registration
};

namespace {

constexpr int kFlagRing = 8;
constexpr int kLag = 2;

int fail(icpmi_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                    \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ctx, ICPMI_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                   \
                        hipGetErrorString(e_), __FILE__, __LINE__);                           \
    } while (0)

// grow-only with slack: odometry clouds vary frame to frame.  No context: also what the prefetch worker calls.
hipError_t reserve_raw(DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return hipSuccess;
    hipError_t e;
    if (b.p && (e = hipFree(b.p)) != hipSuccess) return e;
    b.p = nullptr;
    b.cap = 0;
    size_t want = std::max(bytes, (size_t)4096);
    want = want + want / 4;
    if ((e = hipMalloc(&b.p, want)) != hipSuccess) return e;
    b.cap = want;
    return hipSuccess;
}
int reserve(icpmi_ctx *ctx, DevBuf &b, size_t bytes)
{
    HIP_TRY(ctx, reserve_raw(b, bytes));
    return ICPMI_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// ---- profiling helpers ----------------------------------------------------------------
struct StageTimer {
    icpmi_ctx *ctx;
    long slot = -1; // index, not pointer: the pool may reallocate while an outer timer is open
    StageTimer(icpmi_ctx *c, int stage) : ctx(c)
    {
        // profile 1: only the dominant kernel and the call/loop brackets (3 event pairs per
        // iteration would already cost ~10 us of stream time); profile >= 2: every stage
        if (!ctx->opt.profile) return;
        if (ctx->opt.profile == 1 && stage != ST_COARSE && stage != ST_TOTAL && stage != ST_LOOP && stage != ST_EXCHANGE) return;
        // profile 1 brackets the dominant kernel on every 4th launch only: an event pair costs the
        // stream ~5 us, and on every launch that was 2.2 % of a C3 call (scripts/event_overhead.py);
        // the average over the sampled launches is the same number
        if (ctx->opt.profile == 1 && stage == ST_COARSE && (ctx->coarse_seen++ & 3) != 0) return;
        if (ctx->opt.profile == 1 && stage == ST_EXCHANGE && (ctx->exchange_seen++ & 3) != 0) return; // (the same sampling)
        if (ctx->ev_used == ctx->ev_pool.size()) {
            EventPair p;
            if (hipEventCreate(&p.a) != hipSuccess) return;
            if (hipEventCreate(&p.b) != hipSuccess) {
                (void)hipEventDestroy(p.a);
                return;
            }
            p.stage = stage;
            ctx->ev_pool.push_back(p);
        }
        slot = (long)ctx->ev_used++;
        ctx->ev_pool[slot].stage = stage;
        (void)hipEventRecord(ctx->ev_pool[slot].a, ctx->stream);
    }
    ~StageTimer()
    {
        if (slot >= 0) (void)hipEventRecord(ctx->ev_pool[slot].b, ctx->stream);
    }
    StageTimer(const StageTimer &) = delete;
    StageTimer &operator=(const StageTimer &) = delete;
};

// call after the stream has been synchronised
void harvest_profile(icpmi_ctx *ctx)
{
    if (ctx->opt.profile && ctx->nn_engine == ICPMI_SEARCH_MFMA_BF16 && ctx->nn_misc.p) {
        unsigned long long cnt[3] = {0, 0, 0};
        if (hipMemcpy(cnt, (char *)ctx->nn_misc.p + 128, sizeof(cnt), hipMemcpyDeviceToHost) == hipSuccess) {
            ctx->prof.nn_recheck_queries += (int64_t)cnt[0];
            ctx->prof.nn_fallback_queries += (int64_t)cnt[1];
            ctx->prof.nn_pruned_blocks += (int64_t)cnt[2];
            (void)hipMemset((char *)ctx->nn_misc.p + 128, 0, sizeof(cnt));
        }
    }
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        float ms = 0.f;
        EventPair &p = ctx->ev_pool[i];
        if (hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) continue;
        switch (p.stage) {
        case ST_NN: ctx->prof.nn_ms += ms; ctx->prof.nn_launches++; break;
        case ST_REDUCE: ctx->prof.reduce_ms += ms; ctx->prof.reduce_launches++; break;
        case ST_TRANSFORM: ctx->prof.transform_ms += ms; ctx->prof.transform_launches++; break;
        case ST_NORMALS: ctx->prof.normals_ms += ms; ctx->prof.normals_launches++; break;
        case ST_TOTAL: ctx->prof.total_ms += ms; ctx->prof.calls++; break;
        case ST_LOOP: ctx->prof.loop_ms += ms; break;
        case ST_SETUP: ctx->prof.setup_ms += ms; break;
        case ST_COARSE: ctx->prof.coarse_ms += ms; ctx->prof.coarse_launches++; break;
        case ST_EXCHANGE: ctx->prof.exchange_ms += ms; ctx->prof.exchange_launches++; break;
        default: break;
        }
    }
    ctx->ev_used = 0;
}

// ---- kernel launch wrappers ------------------------------------------------------------

// Smallest target for which AUTO takes the MFMA engine (and its k-NN path).  The environment
// variable ICPMI_MFMA_MIN_TARGETS overrides it (tuning runs); both engines return the same answers.
// Round 1 had 8192 here, chosen on the 1-NN pass alone; the exact k-NN kernel (one thread per row,
// serial over the targets) costs 2.3 us per target point, so a 7.4k-point frame spent 17.6 ms in
// normal estimation where the MFMA path takes 0.2 ms (scripts/threshold_sweep.py: 1.5k points
// 4.15 -> 0.39 ms per registration, 8k points 17.7 -> 0.44 ms).
// The MFMA engine still wins at 300 points (1.5 vs 2.6 ms for 50 iterations, 0.2 vs 0.8 ms of
// normals); below 256 targets a split is mostly padding and the exact kernels are kept.
#ifndef ICPMI_MFMA_MIN_TARGETS
#define ICPMI_MFMA_MIN_TARGETS 256
#endif
constexpr int kMfmaMinQueries = 64;
int mfma_min_targets()
{
    static const int v = [] {
        if (const char *e = getenv("ICPMI_MFMA_MIN_TARGETS")) {
            char *end = nullptr;
            const long x = strtol(e, &end, 10);
            if (end != e && *end == '\0' && x >= 64 && x <= 100000000) return (int)x;
        }
        return (int)(ICPMI_MFMA_MIN_TARGETS);
    }();
    return v;
}

// Single-GPU loop: final sum, step and pose update in one launch (k_finish_step_transform).
// ICPMI_FUSE_FINISH=0 keeps the two separate kernels (the A/B knob of scripts/ab_fuse_finish.sh);
// ICPMI_FUSE_BLOCKS sets the number of workgroups (each repeats the final sum).
bool fuse_finish_enabled()
{
    static const bool v = [] {
        const char *e = getenv("ICPMI_FUSE_FINISH");
        return !(e && e[0] == '0' && e[1] == '\0');
    }();
    return v;
}
int finish_transform_blocks(int n)
{
    static const int cap = [] {
        if (const char *e = getenv("ICPMI_FUSE_BLOCKS")) {
            const long x = strtol(e, nullptr, 10);
            if (x >= 1 && x <= 2048) return (int)x;
        }
        // (round 3: 128.  With the rows' bounds formed here too (RowBounds) the row part is worth spreading over more
        // CUs than the repeated sums cost: 32 / 64 / 128 / 256 workgroups 22.1 / 18.2 / 16.5 / 16.5 us at 100k rows)
        return 128;
    }();
    return std::max(1, std::min(cap, (n + kFinishThreads - 1) / kFinishThreads));
}

// Small clouds (the reference's real callers: filtered scans of 5-20k points): search, residuals and
// pose update of the rows in ONE kernel per iteration (icp_small.h) when the target is at most
// kSmallMaxSplits splits and the source at most this many rows (beyond, the general path's 512-query
// units amortise the operand reads better).  ICPMI_SMALL=0 switches the path off (A/B runs, parity
// test); ICPMI_SMALL_MAX_QUERIES moves the row limit.
bool small_enabled()
{
    static const bool v = [] {
        const char *e = getenv("ICPMI_SMALL");
        return !(e && e[0] == '0' && e[1] == '\0');
    }();
    return v;
}
int small_max_queries()
{
    static const int v = [] {
        if (const char *e = getenv("ICPMI_SMALL_MAX_QUERIES")) {
            const long x = strtol(e, nullptr, 10);
            if (x >= 0 && x <= 100000000) return (int)x;
        }
        return 32768;
    }();
    return v;
}
// The small-cloud kernel's workgroup takes 32 rows through ALL splits, so its time grows with rows x splits where the
// general path's grows with the units on the chip: measured (scripts/engine_threshold.py, profiles/r4_final/
// engine_threshold.json: registrations of n -> n points by the three paths) it wins or ties up to 7 splits (14k points:
// 0.385 against 0.416 ms on a LiDAR-like pair), ties at 10 and loses from 14 (uniform 28k: 1.03 against 0.93 ms).
// ICPMI_SMALL_MAX_SPLITS moves the limit (at most kSmallMaxSplits, what the kernel's LDS records hold).
constexpr int kSmallSplitsDefault = 8;
int small_max_splits()
{
    static const int v = [] {
        if (const char *e = getenv("ICPMI_SMALL_MAX_SPLITS")) {
            const long x = strtol(e, nullptr, 10);
            if (x >= 0 && x <= kSmallMaxSplits) return (int)x;
        }
        return kSmallSplitsDefault;
    }();
    return v;
}
// (by the target alone: whether its Morton-ordered normals are worth keeping)
bool small_target(const icpmi_ctx *ctx)
{
    return small_enabled() && ctx->nn_engine == ICPMI_SEARCH_MFMA_BF16 && !ctx->nn_pruned && ctx->nn_splits <= small_max_splits();
}

// Choose and prepare the search engine for a target cloud (once per call: the target does
// not move).  Both engines return the same indices; AUTO takes the MFMA engine once the
// pair count makes its fixed costs (Morton sort, operand packing, resolve) worthwhile.
// AUTO: the exact fp64 kernels for tiny clouds; the MFMA engine from 256 targets -- over all pairs while the target is
// a dozen splits or fewer (the small-cloud kernel of icp_small.h up to 8, the general kernels to 12), culled
// (nn_culled.h) beyond: the same correspondences bit for bit, 90-98 % of the (row tile, split) pairs never evaluated (round 4; ICPMI_AUTO_CULLED=0 keeps AUTO on the
// all-pairs engine everywhere, the A/B knob).
bool auto_culled_enabled()
{
    static const bool v = [] {
        const char *e = getenv("ICPMI_AUTO_CULLED");
        return !(e && e[0] == '0' && e[1] == '\0');
    }();
    return v;
}
// (targets of more than this many splits; ICPMI_AUTO_CULLED_FROM moves it.  scripts/engine_threshold.py: at 10 splits the
// culled engine's own fixed costs -- the rows' Morton sort, the lists -- still lose to the all-pairs kernels by 5-7 %, at 14
// it wins by 7-13 %, at 49 by 2.3-3.1x)
int auto_culled_from_splits()
{
    static const int v = [] {
        if (const char *e = getenv("ICPMI_AUTO_CULLED_FROM")) {
            const long x = strtol(e, nullptr, 10);
            if (x >= 0 && x <= 1000000) return (int)x;
        }
        return 12;
    }();
    return v;
}
int engine_for(const icpmi_ctx *ctx, int m, int n_hint)
{
    int engine = ctx->opt.search;
    const int splits = (m + kSplitTargets - 1) / kSplitTargets;
    if (engine == ICPMI_SEARCH_AUTO) {
        engine = (m >= mfma_min_targets() && n_hint >= kMfmaMinQueries) ? ICPMI_SEARCH_MFMA_BF16 : ICPMI_SEARCH_EXACT_F64;
        if (engine == ICPMI_SEARCH_MFMA_BF16 && splits > auto_culled_from_splits() && auto_culled_enabled()) engine = ICPMI_SEARCH_MFMA_PRUNED;
    }
    // (the culled coarse kernel keeps running sums over the splits in LDS: beyond kCullMaxSplits -- 6.3M targets -- all pairs)
    // (and the length of a split's list in 20 bits: fewer than 2^20 tiles of 32 rows)
    if (engine == ICPMI_SEARCH_MFMA_PRUNED && (splits > kCullMaxSplits || n_hint >= kCullMaxRows)) engine = ICPMI_SEARCH_MFMA_BF16;
    return engine;
}

int prepare_nn(icpmi_ctx *ctx, const double *d_tgt, int m, int n_hint)
{
    ctx->prep_valid = false; // whatever target the buffers held: it is being replaced
    int engine = engine_for(ctx, m, n_hint);
    // the culled engine is the MFMA engine plus the box test of (row group, split) pairs in the ICP loop and in normal
    // estimation; the stand-alone 1-NN search (nearest_batch) runs over all pairs
    ctx->nn_pruned = engine == ICPMI_SEARCH_MFMA_PRUNED;
    if (ctx->nn_pruned) engine = ICPMI_SEARCH_MFMA_BF16;
    ctx->nn_engine = engine;
    if (engine != ICPMI_SEARCH_MFMA_BF16) return ICPMI_OK;
    if (m >= (1 << 27)) return fail(ctx, ICPMI_ERR_ARG, "the MFMA engines take targets of fewer than 2^27 points (32-bit byte offsets into the sorted records)");
    const int splits = (m + kSplitTargets - 1) / kSplitTargets;
    ctx->nn_splits = splits;
    int rc;
    const int bblocks = std::max(1, std::min(256, (m + 255) / 256));
    size_t sort_bytes = 0;
    HIP_TRY(ctx, sort_pairs_u32(nullptr, &sort_bytes, nullptr, nullptr, nullptr, nullptr, (unsigned)m, ctx->stream));
    if ((rc = reserve(ctx, ctx->bpack, sizeof(uint4) * kSplitTiles * 64 * (size_t)splits))) return rc;
    if ((rc = reserve(ctx, ctx->bbox_part, sizeof(double) * 6 * (size_t)bblocks))) return rc;
    if ((rc = reserve(ctx, ctx->nn_misc, 256))) return rc;
    if ((rc = reserve(ctx, ctx->sort_keys, sizeof(unsigned) * 4 * (size_t)m))) return rc;
    if ((rc = reserve(ctx, ctx->sort_tmp, sort_bytes))) return rc;
    const int ms = (m + 63) / 64 * 64; // component stride of the SoA sorted copy
    ctx->nn_ms = ms;
    if ((rc = reserve(ctx, ctx->tgt_sorted, sizeof(double) * sorted_doubles(ms)))) return rc; // three planes + the records (nn_mfma.h)
    if ((rc = reserve(ctx, ctx->frames, frames_bytes(splits)))) return rc; // split frames, then the slots' boxes
    NnFrame *frame = (NnFrame *)ctx->nn_misc.p;
    unsigned *keys_in = (unsigned *)ctx->sort_keys.p, *keys_out = keys_in + m, *vals_in = keys_in + 2 * (size_t)m,
             *perm = keys_in + 3 * (size_t)m;
    hipStream_t s = ctx->stream;
    Range range("icpmi:target_prepass");
    StageTimer t(ctx, ST_SETUP);
    unsigned long long *stat3 = (unsigned long long *)((char *)ctx->nn_misc.p + 128); // the resolve's statistics words of the coming call
    if (m <= kBboxSingleMax)
        hipLaunchKernelGGL(k_bbox_single, dim3(1), dim3(1024), 0, s, d_tgt, m, frame, 1, stat3);
    else {
        hipLaunchKernelGGL(k_bbox_partial, dim3(bblocks), dim3(256), 0, s, d_tgt, m, (double *)ctx->bbox_part.p, 1);
        hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(64), 0, s, (const double *)ctx->bbox_part.p, bblocks, frame);
        HIP_TRY(ctx, hipMemsetAsync(stat3, 0, 24, s));
    }
    hipLaunchKernelGGL(k_morton_keys, dim3((m + 255) / 256), dim3(256), 0, s, d_tgt, m, (const NnFrame *)frame,
                       keys_in, vals_in);
    HIP_TRY(ctx, sort_pairs_u32(ctx->sort_tmp.p, &sort_bytes, keys_in, keys_out, vals_in, perm, (unsigned)m, s));
    hipLaunchKernelGGL(k_gather_points_rec, dim3((m + 255) / 256), dim3(256), 0, s, d_tgt, (const unsigned *)perm, m, ms,
                       (double *)ctx->tgt_sorted.p);
    hipLaunchKernelGGL(k_split_frames, dim3(splits), dim3(256), 0, s, (const double *)ctx->tgt_sorted.p, m, ms,
                       (SplitFrame *)ctx->frames.p);
    hipLaunchKernelGGL(k_pack_targets, dim3((splits * kSplitTiles * 64 + 255) / 256), dim3(256), 0, s,
                       (const double *)ctx->tgt_sorted.p, m, ms, (const SplitFrame *)ctx->frames.p,
                       (uint4 *)ctx->bpack.p, splits);
    HIP_TRY(ctx, hipGetLastError());
    return ICPMI_OK;
}

// when d_nrm/d_partials are given the resolve kernel also does k_reduce's work (one partial
// row per resolve workgroup: resolve_blocks(n) rows)
// Build-time overrides for A/B runs of the resolve kernels (see resolve_waves).
#ifndef ICPMI_RESOLVE_WAVES
#define ICPMI_RESOLVE_WAVES -1
#endif
#ifndef ICPMI_RESOLVE_Q32_FROM
#define ICPMI_RESOLVE_Q32_FROM 2000000000 /* queries; above: k_nn_resolve<32>.  Off: at C3 (set to 81920) it
                                             measured 55.7 us against 48.7 us (k_finish_step 11.7 against 13.3) */
#endif
// Which resolve kernel: 0 = k_nn_resolve<16> (16 queries per wave, 64 per workgroup); -32 = k_nn_resolve<32>
// (32 per wave, 128 per workgroup); W > 0 = k_nn_resolve4<W> (4 queries per wave, 4 W per workgroup).
int resolve_waves(int n)
{
    if (ICPMI_RESOLVE_WAVES >= 0) return ICPMI_RESOLVE_WAVES;
    // measured (scripts/sweep_resolve*.sh, resolve + finish_step per pass, 100k targets): 8.8k / 12.5k /
    // 25k queries 34 -> 25 / 37 -> 27 / 41 -> 36 us with the quarter-wave kernel, 50k / 100k queries
    // 46 -> 52 / 63 -> 80 us (its 2-4x more partial rows and waves cost more than the shorter chains save)
    if (n <= 32768) return 8;
    // 32 queries per wave would put all the waves of a C3 pass on the chip at once (3,125; the 6,250 of 16
    // per wave need a second round at 5 per SIMD) -- and lost: the longer chain per wave costs more
    return n <= ICPMI_RESOLVE_Q32_FROM ? 0 : -32;
}
// All-pairs coarse pass on few units (filtered scans: 14 query blocks x 4 splits): 256-query units -- one
// 32-query tile per wave instead of two -- when even those fit the chip in one round.  The units' time is
// their waves' time there, so half the work per wave is a shorter pass; once the chip is full the 512-query
// unit amortises its staging and operand build over twice the results.  ICPMI_COARSE_HALF_UNITS=<units> moves
// the switch (0: never).
bool coarse_half_units(const icpmi_ctx *ctx, int n, int splits)
{
    static const long limit = [] {
        const char *e = getenv("ICPMI_COARSE_HALF_UNITS");
        return e ? atol(e) : -1l;
    }();
    const long units = (long)((n + kCoarseQueries - 1) / kCoarseQueries) * splits;
    return units <= (limit >= 0 ? limit : (long)ctx->cu_count);
}
int resolve_blocks(int n)
{
    const int w = resolve_waves(n);
    const int per = w > 0 ? 4 * w : (w == -32 ? kResolveWW * 32 : kResolveWW * kResolveQ);
    return (n + per - 1) / per;
}

// Diagnostic build (-DICPMI_COARSE_CLOCKS, scripts/coarse_clock.py): the all-pairs 1-NN pass stamps its clocks
// into the (idle) slot-minimum buffer; icpmi_debug_coarse_clocks copies them out.  Null in the product build.
float *coarse_clock_buffer(icpmi_ctx *ctx, int n, int splits)
{
#ifdef ICPMI_COARSE_CLOCKS
    const size_t need = sizeof(unsigned long long) * 4 * (size_t)((n + kCoarseQueries - 1) / kCoarseQueries) * (size_t)splits;
    if (reserve(ctx, ctx->slotmin, need) != ICPMI_OK) return nullptr;
    return (float *)ctx->slotmin.p;
#else
    (void)ctx, (void)n, (void)splits;
    return nullptr;
#endif
}

// ICPMI_NN_BOUNDED=0: every pass of the ICP loop keeps its coarse minima and certifies afterwards (round 2's form)
// (read at every call, so that one process can run both forms on the same input: scripts/fuzz_bounded.py)
bool nn_bounded_enabled()
{
    const char *e = getenv("ICPMI_NN_BOUNDED");
    return !(e && e[0] == '0');
}

// The per-row arrays of the bounded pass inside ctx->nn_lists (n rows)
struct NnListRows {
    double *ub;
    unsigned *ent;
    float *ubf, *sqf;
    int *cnt;
};
constexpr size_t kNnListRowBytes = sizeof(double) + 2 * sizeof(float) + sizeof(int) + sizeof(unsigned) * kNnEntCap;
NnListRows nn_list_rows(const icpmi_ctx *ctx, int n)
{
    NnListRows r;
    r.ub = (double *)ctx->nn_lists.p;
    r.ent = (unsigned *)(r.ub + n); // (8-byte aligned: read two words at a time)
    r.ubf = (float *)(r.ent + (size_t)kNnEntCap * n);
    r.sqf = r.ubf + n;
    r.cnt = (int *)(r.sqf + n);
    return r;
}

// The culled engine's group lists (nn_culled.h) inside ctx->grp_cnt / grp_items: two alternating sets of per-split counters
// (a pass reads one and clears the other), one array of lists, and two statistics words behind the counters.
constexpr int kGrpCntPad = 64;
int reserve_group_lists(icpmi_ctx *ctx, int splits, long groups)
{
    int rc;
    const size_t per = ((size_t)splits + kGrpCntPad - 1) / kGrpCntPad * kGrpCntPad;
    if ((rc = reserve(ctx, ctx->grp_cnt, sizeof(unsigned) * 2 * per + 2 * sizeof(unsigned long long) + 4 * sizeof(unsigned)))) return rc;
    const long cap = (std::max<long>(groups, 1) + 1) / 2 * 2; // (even: the coarse kernel reads a wave's two entries as one 8-byte word)
    if ((rc = reserve(ctx, ctx->grp_items, sizeof(unsigned) * (size_t)splits * (size_t)cap + 16))) return rc;
    ctx->grp_cap = (int)cap;
    return ICPMI_OK;
}
GroupLists group_lists(const icpmi_ctx *ctx, int set)
{
    const size_t per = ((size_t)ctx->nn_splits + kGrpCntPad - 1) / kGrpCntPad * kGrpCntPad;
    return GroupLists{(unsigned *)ctx->grp_cnt.p + (size_t)set * per, (unsigned *)ctx->grp_items.p, ctx->grp_cap};
}
unsigned long long *group_stats(const icpmi_ctx *ctx)
{
    const size_t per = ((size_t)ctx->nn_splits + kGrpCntPad - 1) / kGrpCntPad * kGrpCntPad;
    return (unsigned long long *)((unsigned *)ctx->grp_cnt.p + 2 * per);
}
// the coarse kernel's chunk counters (nn_culled.h), one per parity of the pass: a pass hands its chunks out through its own
// and clears the other; behind the statistics
unsigned *group_work(const icpmi_ctx *ctx, int parity) { return (unsigned *)(group_stats(ctx) + 2) + (parity & 1); }
size_t group_cnt_bytes(const icpmi_ctx *ctx)
{
    const size_t per = ((size_t)ctx->nn_splits + kGrpCntPad - 1) / kGrpCntPad * kGrpCntPad;
    return sizeof(unsigned) * 2 * per + 2 * sizeof(unsigned long long) + 4 * sizeof(unsigned);
}
// workgroups of k_nn_coarse_groups: the chip's resident set (two 8-wave workgroups per CU at 4 waves per SIMD); a
// workgroup walks over the chunks blockIdx, blockIdx + grid, ...  ICPMI_GROUPS_GRID=<per CU> for tuning runs.
// ICPMI_GROUPS_WAVES=4|8: pairs per workgroup (8: two workgroups per CU; 4: four, half the operand reuse, finer rounds)
int coarse_groups_waves()
{
    static const int w = [] {
        const char *e = getenv("ICPMI_GROUPS_WAVES");
        return e && e[0] == '4' ? 4 : 8;
    }();
    return w;
}
int coarse_groups_grid(const icpmi_ctx *ctx, int waves)
{
    static const int per_cu = [] {
        if (const char *e = getenv("ICPMI_GROUPS_GRID")) {
            const long x = strtol(e, nullptr, 10);
            if (x >= 1 && x <= 64) return (int)x;
        }
        return 0;
    }();
    return (per_cu ? per_cu : 16 / waves) * ctx->cu_count;
}
size_t coarse_groups_lds(int splits) { return sizeof(unsigned) * (2 * (size_t)splits + 1); }
// Diagnostic build (-DICPMI_GROUPS_CLOCKS, scripts/groups_clock.py): the culled coarse pass of the ICP loop stamps its
// workgroups' phases into the (idle) slot-minimum buffer; icpmi_debug_coarse_clocks copies them out.  Null in the product build.
unsigned long long *groups_clock_buffer(icpmi_ctx *ctx)
{
#ifdef ICPMI_GROUPS_CLOCKS
    if (reserve(ctx, ctx->slotmin, sizeof(unsigned long long) * kGroupStamps * 64 * (size_t)ctx->cu_count) != ICPMI_OK) return nullptr;
    return (unsigned long long *)ctx->slotmin.p;
#else
    (void)ctx;
    return nullptr;
#endif
}

// `bounded`: d_idx holds the rows' matches of the previous pass and the kernel that moved the rows has left their bounds
// in ctx->nn_lists (RowBounds, kernels.h; nn_bounded.h)
int launch_nn_mfma(icpmi_ctx *ctx, const double *d_qry, int n, int m, int *d_idx, double *d_d2,
                   const IcpState *st, const double *d_tgt = nullptr, const double *d_nrm = nullptr,
                   double *d_partials = nullptr, int pruned_pass = -1, bool bounded = false)
{
    const int splits = ctx->nn_splits;
    int rc;
    const SplitFrame *frames = (const SplitFrame *)ctx->frames.p;
    const unsigned *perm = (const unsigned *)ctx->sort_keys.p + 3 * (size_t)m;
    unsigned long long *counters = (unsigned long long *)((char *)ctx->nn_misc.p + 128);
    Range range("icpmi:nn_search");
    StageTimer t(ctx, ST_NN);
    if (bounded) {
        const NnListRows lr = nn_list_rows(ctx, n);
        double *ub_row = lr.ub;
        unsigned *ent_row = lr.ent;
        float *ubf_row = lr.ubf, *sqf_row = lr.sqf;
        int *cnt_row = lr.cnt;
        {
            StageTimer tc(ctx, ST_COARSE);
            const KnnLists kl{ubf_row, sqf_row, cnt_row, ent_row, kNnEntCap};
            if (pruned_pass >= 0) {
                // the culled engine's surviving (row group, split) pairs, list epilogue: a row's list then holds slots of
                // evaluated splits only, and every split that can hold a target within the row's bound IS evaluated (the
                // group's bound is the largest of its rows'); pass p reads the lists counted in set p & 1 and clears the other
                const GroupLists cur = group_lists(ctx, pruned_pass & 1), nxt = group_lists(ctx, (pruned_pass + 1) & 1);
                const long groups = (n + kGroupRows - 1) / kGroupRows;
#define ICPMI_GROUPS_ARGS                                                                                                              \
    d_qry, n, (size_t)0, (const uint4 *)ctx->bpack.p, frames, splits, (const unsigned *)cur.items, cur.cap, (const unsigned *)cur.cnt, \
        nxt.cnt, ctx->opt.profile ? group_stats(ctx) : (unsigned long long *)nullptr, (unsigned long long)(groups * splits), st, kl, \
        group_work(ctx, pruned_pass & 1), group_work(ctx, (pruned_pass + 1) & 1), groups_clock_buffer(ctx)
                if (coarse_groups_waves() == 4)
                    hipLaunchKernelGGL((k_nn_coarse_groups<false, 4>), dim3(coarse_groups_grid(ctx, 4)), dim3(256), coarse_groups_lds(splits), ctx->stream,
                                       ICPMI_GROUPS_ARGS);
                else
                    hipLaunchKernelGGL((k_nn_coarse_groups<false, 8>), dim3(coarse_groups_grid(ctx, 8)), dim3(512), coarse_groups_lds(splits), ctx->stream,
                                       ICPMI_GROUPS_ARGS);
#undef ICPMI_GROUPS_ARGS
            } else if (coarse_half_units(ctx, n, splits)) {
                constexpr int per = kCoarseQueries / kCoarseQT; // queries per workgroup with one tile per wave
                hipLaunchKernelGGL((k_nn_coarse_bounded<1, kCoarseWaves>), dim3((n + per - 1) / per, splits), dim3(kCoarseThreads), 0,
                                   ctx->stream, d_qry, n, (const uint4 *)ctx->bpack.p, frames, kl, st);
            } else
                hipLaunchKernelGGL((k_nn_coarse_bounded<kCoarseQT, kCoarseWaves>), dim3((n + kCoarseQueries - 1) / kCoarseQueries, splits),
                                   dim3(kCoarseThreads), 0, ctx->stream, d_qry, n, (const uint4 *)ctx->bpack.p, frames, kl, st);
            ctx->prof.nn_coarse_blocks += (int64_t)((n + kCoarseQueries - 1) / kCoarseQueries) * splits;
        }
        // (the statistics: with one row in a hundred listing a second slot, most waves of a pass have something to add to the
        // same two words, and per wave those atomics were 25 of the kernel's 55 us -- they leave per workgroup now, and
        // only when the stage profile (level 2) is asked for)
#define ICPMI_BOUNDED_ARGS                                                                                                        \
    d_qry, n, (const double *)ctx->tgt_sorted.p, perm, m, ctx->nn_ms, splits, frames, (const double *)ub_row, (const int *)cnt_row, \
        (const unsigned *)ent_row, d_idx, ctx->opt.profile >= 2 ? counters : (unsigned long long *)nullptr, d_tgt, d_nrm, d_partials, st
        switch (resolve_waves(n)) {
        case 0: hipLaunchKernelGGL(k_nn_resolve_bounded, dim3(resolve_blocks(n)), dim3(64 * kResolveWW), 0, ctx->stream, ICPMI_BOUNDED_ARGS); break;
        case 4: hipLaunchKernelGGL(k_nn_resolve4_bounded<4>, dim3(resolve_blocks(n)), dim3(256), 0, ctx->stream, ICPMI_BOUNDED_ARGS); break;
        case 8: hipLaunchKernelGGL(k_nn_resolve4_bounded<8>, dim3(resolve_blocks(n)), dim3(512), 0, ctx->stream, ICPMI_BOUNDED_ARGS); break;
        default: hipLaunchKernelGGL(k_nn_resolve4_bounded<16>, dim3(resolve_blocks(n)), dim3(1024), 0, ctx->stream, ICPMI_BOUNDED_ARGS); break;
        }
#undef ICPMI_BOUNDED_ARGS
        ctx->prof.nn_pairs += (double)n * (double)m;
        ctx->prof.bounded_launches += 1;
        HIP_TRY(ctx, hipGetLastError());
        return ICPMI_OK;
    }
    if (pruned_pass >= 0) return fail(ctx, ICPMI_ERR_ARG, "internal: the culled engine has bounded passes only");
    if ((rc = reserve(ctx, ctx->coarse, coarse_bytes(splits, n)))) return rc;
    {
        StageTimer tc(ctx, ST_COARSE); // the dominant kernel alone (matches rocprofv3's per-kernel average)
        if (coarse_half_units(ctx, n, splits)) {
            constexpr int per = kCoarseQueries / kCoarseQT; // queries per workgroup with one tile per wave
            hipLaunchKernelGGL((k_nn_coarse<0, 1, kCoarseWaves>), dim3((n + per - 1) / per, splits), dim3(kCoarseThreads), 0,
                               ctx->stream, d_qry, n, (const uint4 *)ctx->bpack.p, frames, (float2 *)ctx->coarse.p,
                               (float *)nullptr, st);
        } else {
            hipLaunchKernelGGL((k_nn_coarse<0, kCoarseQT, kCoarseWaves>), dim3((n + kCoarseQueries - 1) / kCoarseQueries, splits),
                               dim3(kCoarseThreads), 0, ctx->stream, d_qry, n, (const uint4 *)ctx->bpack.p, frames,
                               (float2 *)ctx->coarse.p, coarse_clock_buffer(ctx, n, splits), st);
        }
        ctx->prof.nn_coarse_blocks += (int64_t)((n + kCoarseQueries - 1) / kCoarseQueries) * splits;
    }
    const int *blk_cnt = nullptr, *blk_list = nullptr; // (round 3's per-block split lists: the culled engine no longer has unbounded passes)
#define ICPMI_RESOLVE_ARGS                                                                                            \
    d_qry, n, (const double *)ctx->tgt_sorted.p, perm, m, ctx->nn_ms, (const float2 *)ctx->coarse.p, splits, frames,  \
        (const NnFrame *)ctx->nn_misc.p, d_idx, d_d2, counters, d_tgt, d_nrm, d_partials, blk_cnt, blk_list, st
    switch (resolve_waves(n)) {
    case 0: hipLaunchKernelGGL(k_nn_resolve<16>, dim3(resolve_blocks(n)), dim3(64 * kResolveWW), 0, ctx->stream, ICPMI_RESOLVE_ARGS); break;
    case -32: hipLaunchKernelGGL(k_nn_resolve<32>, dim3(resolve_blocks(n)), dim3(64 * kResolveWW), 0, ctx->stream, ICPMI_RESOLVE_ARGS); break;
    case 4: hipLaunchKernelGGL(k_nn_resolve4<4>, dim3(resolve_blocks(n)), dim3(256), 0, ctx->stream, ICPMI_RESOLVE_ARGS); break;
    case 8: hipLaunchKernelGGL(k_nn_resolve4<8>, dim3(resolve_blocks(n)), dim3(512), 0, ctx->stream, ICPMI_RESOLVE_ARGS); break;
    default: hipLaunchKernelGGL(k_nn_resolve4<16>, dim3(resolve_blocks(n)), dim3(1024), 0, ctx->stream, ICPMI_RESOLVE_ARGS); break;
    }
#undef ICPMI_RESOLVE_ARGS
    ctx->prof.nn_pairs += (double)n * (double)m;
    HIP_TRY(ctx, hipGetLastError());
    return ICPMI_OK;
}

// nearest neighbour of every row of d_qry among d_tgt -> d_idx (+ optional d_d2)
int launch_nn(icpmi_ctx *ctx, const double *d_qry, int n, const double *d_tgt, int m, int *d_idx,
              double *d_d2, const IcpState *st)
{
    if (ctx->nn_engine == ICPMI_SEARCH_MFMA_BF16) return launch_nn_mfma(ctx, d_qry, n, m, d_idx, d_d2, st);
    constexpr int QPT = 2;
    const int qblocks = (n + 256 * QPT - 1) / (256 * QPT);
    // enough workgroups to fill 256 CUs several times over; every split keeps >= 256 targets
    int splits = (8 * ctx->cu_count + qblocks - 1) / qblocks;
    splits = std::max(1, std::min(splits, std::min(64, (m + 255) / 256)));
    const int per = (m + splits - 1) / splits;
    splits = (m + per - 1) / per;
    int rc;
    if ((rc = reserve(ctx, ctx->part_d2, sizeof(double) * (size_t)splits * n))) return rc;
    if ((rc = reserve(ctx, ctx->part_idx, sizeof(int) * (size_t)splits * n))) return rc;
    Range range("icpmi:nn_search");
    StageTimer t(ctx, ST_NN);
    hipLaunchKernelGGL(k_nn_f64<QPT>, dim3(qblocks, splits), dim3(256), 0, ctx->stream, d_qry, n,
                       d_tgt, m, per, (double *)ctx->part_d2.p, (int *)ctx->part_idx.p, st);
    hipLaunchKernelGGL(k_nn_merge, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                       (const double *)ctx->part_d2.p, (const int *)ctx->part_idx.p, n, splits,
                       d_idx, d_d2, st);
    ctx->prof.nn_pairs += (double)n * (double)m;
    HIP_TRY(ctx, hipGetLastError());
    return ICPMI_OK;
}

int reduce_blocks(const icpmi_ctx *ctx, int n)
{
    return std::max(1, std::min(ctx->cu_count, (n + 255) / 256));
}

// Normal estimation takes the target's rows in Morton order (a rank's slice is then a range of sorted positions) with
// the pruned engine, whose blocks must be compact, and with the all-pairs engine's list form (knn_lists.h;
// ICPMI_KNN_LISTS=0: round 2's slot-minimum form, rows in point order)
bool knn_lists_enabled()
{
    static const bool on = [] {
        const char *e = getenv("ICPMI_KNN_LISTS");
        return !(e && e[0] == '0');
    }();
    return on;
}
bool sorted_normal_rows(const icpmi_ctx *ctx, int k, int m)
{
    return ctx->nn_engine == ICPMI_SEARCH_MFMA_BF16 && (ctx->nn_pruned || knn_lists_enabled()) && k <= 32 && m >= mfma_min_targets();
}

#ifndef ICPMI_KNN_CHUNK_DEFAULT_MB
#define ICPMI_KNN_CHUNK_DEFAULT_MB 1024
#endif
// k-NN lists of rows [row0,row1) of d_qry among the m points d_pts (kdtree.hpp:65-78): MFMA
// coarse pass + exact resolve, or the exact fp64 kernel.  List of row i -> ctx->knn_idx[i * k ..],
// closest first.  prepare_nn() must have run for d_pts.  d_qry == d_pts for normal estimation.
// With the pruned engine (`by_sorted_row`, d_qry == d_pts only), [row0, row1) are positions in the
// Morton-sorted target (row0 a multiple of 64).
int launch_knn(icpmi_ctx *ctx, const double *d_qry, int nq_total, const double *d_pts, int m, int k, int row0, int row1,
               bool by_sorted_row = false)
{
    const int rows = row1 - row0;
    if (rows <= 0) return ICPMI_OK;
    int rc;
    if ((rc = reserve(ctx, ctx->knn_idx, sizeof(int) * (size_t)std::max(m, nq_total) * k))) return rc;
    int *knn = (int *)ctx->knn_idx.p;
    hipStream_t s = ctx->stream;
    constexpr int BLOCK = 128;
    const size_t smem = (size_t)k * BLOCK * (sizeof(double) + sizeof(int));
    const bool mfma = ctx->nn_engine == ICPMI_SEARCH_MFMA_BF16 && k <= 32 && m >= mfma_min_targets();
    const unsigned *perm = mfma ? (const unsigned *)ctx->sort_keys.p + 3 * (size_t)m : nullptr;
    const double *qsep = d_qry == d_pts ? nullptr : d_qry; // the exact kernels read rows from d_pts unless told otherwise
    if (by_sorted_row && (!mfma || qsep)) return fail(ctx, ICPMI_ERR_ARG, "sorted-row lists need the MFMA engine's sorted target");
    if (mfma) {
        const int splits = ctx->nn_splits, nslots = splits * kCols;
        // bound the slot-minimum buffer (2 B x nslots per row) by chunking the rows: ~1 GiB, or
        // ~4 GiB when culling leaves most of it untouched (the pruned engine writes only the
        // listed splits of a row, the layout stays dense)
        // (ICPMI_KNN_CHUNK_MB: the all-pairs budget in MiB, for tuning runs.  The buffer is written by the MODE-1 coarse
        // pass and read once by k_knn_resolve: a chunk that fits the 256 MB Infinity Cache beside the target never
        // goes to HBM between the two)
        static const long knn_budget = [] {
            if (const char *e = getenv("ICPMI_KNN_CHUNK_MB")) {
                const long x = strtol(e, nullptr, 10);
                if (x >= 1 && x <= 16384) return x << 20;
            }
            return (long)ICPMI_KNN_CHUNK_DEFAULT_MB << 20;
        }();
        // all-pairs engine, rows in the target's Morton order: bound first, lists instead of minima (knn_lists.h)
        // (pruned engine: the same lists on the units that survive the box test against the block's largest row bound)
        const bool lists = by_sorted_row; // (round 3's block lists over slot minima for the culled engine are gone)
        const bool lists_culled = lists && ctx->nn_pruned;
        // rows that are not a range of sorted positions (arbitrary queries: icpmi_k_nearest; rows by point index:
        // icpmi_estimate_normals_rows) get their place in the sorted order from their Morton key (k_knn_prebound_q)
        const bool lists_q = !by_sorted_row && knn_lists_enabled();
        constexpr long kListRowBytes = sizeof(double) + 2 * sizeof(float) + sizeof(int) + sizeof(unsigned) * kKnnEntCap;
        const long budget = (lists || lists_q) ? (1l << 30) : by_sorted_row ? (4l << 30) : knn_budget;
        long chunk = (budget / ((lists || lists_q) ? kListRowBytes : (long)nslots * 2)) / kCoarseQueries * kCoarseQueries; // 2 bytes per slot minimum (bf16)
        chunk = std::max<long>(kCoarseQueries, std::min<long>(chunk, ((long)rows + kCoarseQueries - 1) / kCoarseQueries * kCoarseQueries));
        if ((rc = reserve(ctx, ctx->slotmin, (lists || lists_q) ? (size_t)kListRowBytes * chunk : sizeof(unsigned short) * (size_t)chunk * nslots))) return rc;
        if ((rc = reserve(ctx, ctx->fb_list, sizeof(int) * ((size_t)rows + 16)))) return rc;
        int *fb_count = (int *)ctx->fb_list.p, *fb_list = fb_count + 16;
        HIP_TRY(ctx, hipMemsetAsync(fb_count, 0, sizeof(int), s));
        const SplitFrame *frames = (const SplitFrame *)ctx->frames.p;
        const double *sorted = (const double *)ctx->tgt_sorted.p;
        if (lists_culled && (rc = reserve_group_lists(ctx, splits, chunk / kGroupRows))) return rc;
        for (long c0 = row0; c0 < row1; c0 += chunk) {
            const int nq = (int)std::min<long>(chunk, row1 - c0);
            const int nblk = (nq + kCoarseQueries - 1) / kCoarseQueries;
            if (lists) {
                double *t_row = (double *)ctx->slotmin.p;
                float *tf_row = (float *)(t_row + chunk), *sqf_row = tf_row + chunk;
                int *cnt_row = (int *)(sqf_row + chunk);
                unsigned *ent_row = (unsigned *)(cnt_row + chunk);
                const KnnLists kl{tf_row, sqf_row, cnt_row, ent_row, kKnnEntCap};
                hipLaunchKernelGGL(k_knn_prebound, dim3((nq + kPreRows - 1) / kPreRows), dim3(256), 0, s, sorted, m, ctx->nn_ms, k, (int)c0, nq,
                                   t_row, tf_row, sqf_row, cnt_row);
                if (lists_culled) {
                    // the groups are runs of 32 sorted rows, their bound the largest of their rows' (nn_culled.h); a wave takes two
                    const GroupLists gl = group_lists(ctx, 0);
                    const int waves = (nq + 63) / 64;
                    HIP_TRY(ctx, hipMemsetAsync(ctx->grp_cnt.p, 0, group_cnt_bytes(ctx), s));
                    hipLaunchKernelGGL(k_knn_group_cull, dim3((waves + 15) / 16), dim3(1024), 0, s, sorted, m, ctx->nn_ms, (int)c0, nq,
                                       (const double *)t_row, frames, splits, gl);
#define ICPMI_GROUPS_ARGS                                                                                                                   \
    sorted + c0, nq, (size_t)ctx->nn_ms, (const uint4 *)ctx->bpack.p, frames, splits, (const unsigned *)gl.items, gl.cap,                  \
        (const unsigned *)gl.cnt, (unsigned *)nullptr, (unsigned long long *)nullptr, 0ull, (const IcpState *)nullptr, kl,               \
        group_work(ctx, 0), (unsigned *)nullptr
                    if (coarse_groups_waves() == 4)
                        hipLaunchKernelGGL((k_nn_coarse_groups<true, 4>), dim3(coarse_groups_grid(ctx, 4)), dim3(256), coarse_groups_lds(splits), s,
                                           ICPMI_GROUPS_ARGS);
                    else
                        hipLaunchKernelGGL((k_nn_coarse_groups<true, 8>), dim3(coarse_groups_grid(ctx, 8)), dim3(512), coarse_groups_lds(splits), s,
                                           ICPMI_GROUPS_ARGS);
#undef ICPMI_GROUPS_ARGS
                } else if (coarse_half_units(ctx, nq, splits)) {
                    constexpr int per = kCoarseQueries / kCoarseQT;
                    hipLaunchKernelGGL((k_nn_coarse_rows<1, kCoarseWaves>), dim3((nq + per - 1) / per, splits), dim3(kCoarseThreads), 0,
                                       s, sorted + c0, nq, (size_t)ctx->nn_ms, (const uint4 *)ctx->bpack.p, frames, kl);
                } else
                    hipLaunchKernelGGL((k_nn_coarse_rows<kCoarseQT, kCoarseWaves>), dim3(nblk, splits), dim3(kCoarseThreads), 0, s,
                                       sorted + c0, nq, (size_t)ctx->nn_ms, (const uint4 *)ctx->bpack.p, frames, kl);
                hipLaunchKernelGGL(k_knn_resolve_lists<false>, dim3((nq + 3) / 4), dim3(256), 0, s, (const double *)nullptr, sorted, perm, m,
                                   ctx->nn_ms, k, (int)c0, nq, (const double *)t_row, (const int *)cnt_row, (const unsigned *)ent_row, knn,
                                   fb_list, fb_count);
            } else if (lists_q) {
                double *t_row = (double *)ctx->slotmin.p;
                float *tf_row = (float *)(t_row + chunk), *sqf_row = tf_row + chunk;
                int *cnt_row = (int *)(sqf_row + chunk);
                unsigned *ent_row = (unsigned *)(cnt_row + chunk);
                const KnnLists kl{tf_row, sqf_row, cnt_row, ent_row, kKnnEntCap};
                hipLaunchKernelGGL(k_knn_prebound_q, dim3((nq + 7) / 8), dim3(256), 0, s, d_qry, (int)c0, nq, sorted,
                                   (const unsigned *)ctx->sort_keys.p + (size_t)m /* the sorted keys */, m, ctx->nn_ms, k,
                                   (const NnFrame *)ctx->nn_misc.p, t_row, tf_row, sqf_row, cnt_row);
                if (coarse_half_units(ctx, nq, splits)) {
                    constexpr int per = kCoarseQueries / kCoarseQT;
                    hipLaunchKernelGGL((k_nn_coarse_bounded<1, kCoarseWaves>), dim3((nq + per - 1) / per, splits), dim3(kCoarseThreads), 0,
                                       s, d_qry + 3 * c0, nq, (const uint4 *)ctx->bpack.p, frames, kl, (const IcpState *)nullptr);
                } else
                    hipLaunchKernelGGL((k_nn_coarse_bounded<kCoarseQT, kCoarseWaves>), dim3(nblk, splits), dim3(kCoarseThreads), 0, s,
                                       d_qry + 3 * c0, nq, (const uint4 *)ctx->bpack.p, frames, kl, (const IcpState *)nullptr);
                hipLaunchKernelGGL(k_knn_resolve_lists<true>, dim3((nq + 3) / 4), dim3(256), 0, s, d_qry, sorted, perm, m, ctx->nn_ms, k,
                                   (int)c0, nq, (const double *)t_row, (const int *)cnt_row, (const unsigned *)ent_row, knn, fb_list,
                                   fb_count);
            } else {
                if (coarse_half_units(ctx, nq, splits)) {
                    constexpr int per = kCoarseQueries / kCoarseQT;
                    hipLaunchKernelGGL((k_nn_coarse<1, 1, kCoarseWaves>), dim3((nq + per - 1) / per, splits),
                                       dim3(kCoarseThreads), 0, s, d_qry + 3 * c0, nq, (const uint4 *)ctx->bpack.p,
                                       frames, (float2 *)nullptr, (float *)ctx->slotmin.p, (const IcpState *)nullptr);
                } else
                hipLaunchKernelGGL((k_nn_coarse<1, kCoarseQT, kCoarseWaves>), dim3(nblk, splits),
                                   dim3(kCoarseThreads), 0, s, d_qry + 3 * c0, nq, (const uint4 *)ctx->bpack.p,
                                   frames, (float2 *)nullptr, (float *)ctx->slotmin.p, (const IcpState *)nullptr);
                hipLaunchKernelGGL(k_knn_resolve<false>, dim3((nq + 3) / 4), dim3(256), 0, s, d_qry, (int)c0, nq, sorted, perm, m,
                                   ctx->nn_ms, k, (const float *)ctx->slotmin.p, nslots, frames, (const NnFrame *)ctx->nn_misc.p, knn,
                                   fb_list, fb_count, (const int *)nullptr, (const int *)nullptr);
            }
        }
        hipLaunchKernelGGL(k_knn_exact_rows, dim3((lists || lists_q) ? ctx->cu_count : 1024), dim3(256), (size_t)k * 256 * (sizeof(double) + sizeof(int)),
                           s, d_pts, m, k, (const int *)fb_list, (const int *)fb_count, knn,
                           by_sorted_row ? perm : (const unsigned *)nullptr, qsep);
        if (ctx->opt.profile) { // visibility only: how many rows took the exact path
            int cnt = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&cnt, fb_count, sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            ctx->prof.knn_fallback_rows += cnt;
        }
    } else {
        hipLaunchKernelGGL(k_knn_exact_list<BLOCK>, dim3((rows + BLOCK - 1) / BLOCK), dim3(BLOCK), smem, s,
                           d_pts, m, k, row0, row1, (const int *)nullptr, (const int *)nullptr, knn, qsep);
    }
    HIP_TRY(ctx, hipGetLastError());
    return ICPMI_OK;
}

// normals of rows [row0,row1) of d_pts against all m points (icp.hpp:23-67): k-NN lists then PCA.
// With the pruned engine (`by_sorted_row`) the rows are sorted positions and `scatter` selects
// where a normal goes: to its point's row of d_normals (m rows), or to its sorted row (what a
// rank contributes to the all-gather).
int launch_normals(icpmi_ctx *ctx, const double *d_pts, int m, int k, int row0, int row1,
                   double *d_normals, bool by_sorted_row = false, bool scatter = true)
{
    const int rows = row1 - row0;
    if (rows <= 0) return ICPMI_OK;
    int rc;
    Range range("icpmi:normals");
    StageTimer t(ctx, ST_NORMALS);
    if ((rc = launch_knn(ctx, d_pts, m, d_pts, m, k, row0, row1, by_sorted_row))) return rc;
    const bool mfma = ctx->nn_engine == ICPMI_SEARCH_MFMA_BF16 && k <= 32 && m >= mfma_min_targets();
    const unsigned *perm = mfma ? (const unsigned *)ctx->sort_keys.p + 3 * (size_t)m : nullptr;
    hipLaunchKernelGGL(k_normals_from_knn, dim3((rows + 255) / 256), dim3(256), 0, ctx->stream, d_pts, m, k, row0, row1,
                       (const int *)ctx->knn_idx.p, d_normals, by_sorted_row ? perm : (const unsigned *)nullptr,
                       (int)(by_sorted_row && scatter));
    HIP_TRY(ctx, hipGetLastError());
    return ICPMI_OK;
}

int check_common(icpmi_ctx *ctx)
{
    if (!ctx) return ICPMI_ERR_NULL;
    HIP_TRY(ctx, hipSetDevice(ctx->opt.device));
    if (ctx->trailing_work) {
        // The previous icpmi_stream_push* returned with the preparation of its NEXT target (Morton sort, operand
        // packing, 20-NN, normals) still queued.  A fault in those kernels belongs to that step, not to whatever
        // this call is about to do: look before queueing anything behind it.
        ctx->trailing_work = false;
        const hipError_t e = hipStreamQuery(ctx->stream);
        if (e != hipSuccess && e != hipErrorNotReady) {
            (void)hipGetLastError();
            ctx->prep_valid = false;
            return fail(ctx, ICPMI_ERR_HIP, "the target preparation queued behind the previous icpmi_stream_push failed: %s",
                        hipGetErrorString(e));
        }
    }
    return ICPMI_OK;
}

int load_rccl(icpmi_ctx *ctx)
{
    Rccl &r = ctx->rccl;
    if (r.lib) return ICPMI_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) return fail(ctx, ICPMI_ERR_RCCL, "cannot dlopen librccl: %s", dlerror());
#define SYM(field, name)                                                                      \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));                        \
    if (!r.field) return fail(ctx, ICPMI_ERR_RCCL, "librccl lacks %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather");
    SYM(GetErrorString, "ncclGetErrorString");
    SYM(CommCount, "ncclCommCount");
    SYM(CommUserRank, "ncclCommUserRank");
    SYM(CommCuDevice, "ncclCommCuDevice");
#undef SYM
    return ICPMI_OK;
}

#define RCCL_TRY(ctx, expr)                                                                   \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess)                                                                \
            return fail(ctx, ICPMI_ERR_RCCL, "%s failed: %s", #expr,                          \
                        ctx->rccl.GetErrorString(r_));                                        \
    } while (0)

// ---- exchanges of the sharded path --------------------------------------------------------
// in-place sum of `count` doubles at device address d_buf over all ranks
int exchange_allreduce(icpmi_ctx *ctx, double *d_buf, int count)
{
    if (ctx->comm) {
        RCCL_TRY(ctx, ctx->rccl.AllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, ctx->comm, ctx->stream));
        return ICPMI_OK;
    }
    ctx->cb_host.resize(std::max<size_t>(ctx->cb_host.size(), count));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->cb_host.data(), d_buf, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->cb_allreduce(ctx->cb_user, ctx->cb_host.data(), count) != 0)
        return fail(ctx, ICPMI_ERR_RCCL, "all-reduce callback failed");
    HIP_TRY(ctx, hipMemcpyAsync(d_buf, ctx->cb_host.data(), sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ICPMI_OK;
}

// in-place all-gather: rank r's `per` doubles sit at d_buf + r * per
int exchange_allgather(icpmi_ctx *ctx, double *d_buf, size_t per)
{
    if (ctx->comm) {
        RCCL_TRY(ctx, ctx->rccl.AllGather(d_buf + (size_t)ctx->rank * per, d_buf, per, ncclDouble, ctx->comm, ctx->stream));
        return ICPMI_OK;
    }
    const size_t total = per * ctx->n_ranks;
    ctx->cb_host.resize(std::max(ctx->cb_host.size(), total));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->cb_host.data(), d_buf, sizeof(double) * total, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->cb_allgather(ctx->cb_user, ctx->cb_host.data(), (int32_t)per) != 0)
        return fail(ctx, ICPMI_ERR_RCCL, "all-gather callback failed");
    HIP_TRY(ctx, hipMemcpyAsync(d_buf, ctx->cb_host.data(), sizeof(double) * total, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ICPMI_OK;
}

// The target normals (ctx->nrm, by point) once more in the target's Morton order, for the small-cloud
// kernel: its slot scans then read a candidate's normal next to its coordinates.
int sort_normals(icpmi_ctx *ctx, int m)
{
    if (!small_target(ctx)) return ICPMI_OK;
    int rc;
    if ((rc = reserve(ctx, ctx->nrm_sorted, sizeof(double) * 3 * (size_t)ctx->nn_ms))) return rc;
    hipLaunchKernelGGL(k_gather_points, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, (const double *)ctx->nrm.p,
                       (const unsigned *)ctx->sort_keys.p + 3 * (size_t)m, m, ctx->nn_ms, (double *)ctx->nrm_sorted.p);
    HIP_TRY(ctx, hipGetLastError());
    return ICPMI_OK;
}

// A target prepared on a helper context (prepare_target on its stream and workspace) becomes this context's:
// the buffers that DEFINE a prepared target change hands (the helper gets this context's old ones to build the
// next target in), scratch stays where it is.  The caller orders this context's stream behind the helper's work.
void adopt_target(icpmi_ctx *ctx, icpmi_ctx *h)
{
    std::swap(ctx->bpack, h->bpack);
    std::swap(ctx->nn_misc, h->nn_misc);       // the target's frame (+ counters, + the voxel filter's box: scratch)
    std::swap(ctx->sort_keys, h->sort_keys);   // holds the Morton permutation
    std::swap(ctx->tgt_sorted, h->tgt_sorted);
    std::swap(ctx->frames, h->frames);
    std::swap(ctx->nrm, h->nrm);
    std::swap(ctx->nrm_sorted, h->nrm_sorted);
    ctx->nn_splits = h->nn_splits;
    ctx->nn_ms = h->nn_ms;
    ctx->nn_engine = h->nn_engine;
    ctx->nn_pruned = h->nn_pruned;
    ctx->prep_tgt = h->prep_tgt;
    ctx->prep_m = h->prep_m;
    ctx->prep_engine = h->prep_engine;
    ctx->prep_valid = true;
    h->prep_valid = false;
}

// ---- the ICP call, device pointers -----------------------------------------------------
// Search structure + normals of a target for a single-GPU registration (icp.hpp:169-171), queued on the
// context's stream; remembered, so that an align against the same device pointer and size with the
// same engine finds them in place.  Everything else that prepares a search (prepare_nn) forgets it.
int prepare_target(icpmi_ctx *ctx, const double *d_tgt, int m, int n_hint)
{
    int rc;
    if ((rc = reserve(ctx, ctx->nrm, sizeof(double) * 3 * (size_t)m))) return rc;
    if ((rc = prepare_nn(ctx, d_tgt, m, n_hint))) return rc;
    const bool sorted_rows = sorted_normal_rows(ctx, ctx->opt.normal_k, m);
    if ((rc = launch_normals(ctx, d_tgt, m, ctx->opt.normal_k, 0, m, (double *)ctx->nrm.p, sorted_rows, true))) return rc;
    if ((rc = sort_normals(ctx, m))) return rc;
    ctx->prep_tgt = d_tgt;
    ctx->prep_m = m;
    ctx->prep_engine = engine_for(ctx, m, n_hint);
    ctx->prep_valid = true;
    return ICPMI_OK;
}

// `before_wait` (may be null): called once everything of the registration, the copies of its results
// included, is queued and an event behind them recorded -- what it queues runs while the host waits
// for that event only (icpmi_stream_push prepares the next frame's target there).
int align_device(icpmi_ctx *ctx, const double *d_src, int64_t n_src64, const double *d_tgt,
                 int64_t n_tgt64, const icpmi_config *cfg, icpmi_result *result,
                 double *error_history, int32_t history_cap, int (*before_wait)(icpmi_ctx *) = nullptr,
                 int (*after_first)(icpmi_ctx *) = nullptr)
{
    const int n = (int)n_src64, m = (int)n_tgt64;
    const int max_it = cfg->max_iterations;
    const int max_hist = max_it + 1;
    hipStream_t s = ctx->stream;
    int rc;

    if ((rc = reserve(ctx, ctx->cur, sizeof(double) * 3 * (size_t)n))) return rc;
    if ((rc = reserve(ctx, ctx->nrm, sizeof(double) * 3 * (size_t)m))) return rc;
    if ((rc = reserve(ctx, ctx->idx, sizeof(int) * (size_t)n))) return rc;
    if ((rc = reserve(ctx, ctx->history, sizeof(double) * (size_t)(max_hist + 1)))) return rc;

    double *cur = (double *)ctx->cur.p;
    double *nrm = (double *)ctx->nrm.p;
    int *idx = (int *)ctx->idx.p;
    double *hist = (double *)ctx->history.p;

    Range range_call("icpmi:align");
    StageTimer *t_total = new StageTimer(ctx, ST_TOTAL);
    struct Closer { StageTimer *&p; ~Closer() { delete p; p = nullptr; } } close_total{t_total};

    // state: total = initial_transform (icp.hpp:163,178), prev_error = DBL_MAX (icp.hpp:179)
    IcpState *hs = ctx->h_state;
    memset(hs, 0, sizeof(*hs));
    memcpy(hs->total, cfg->initial_transform, sizeof(double) * 16);
    hs->prev_error = 1.7976931348623157e308;
    hs->tolerance = cfg->tolerance;
    hs->min_error = cfg->min_error;
    hs->max_hist = max_hist;
    // two state buffers: the sharded loop's fused step + transform kernel reads one and writes the
    // other (k_step_transform); `st` is the one the coming kernels read
    IcpState *st = ctx->d_state;
    HIP_TRY(ctx, hipMemcpyAsync(st, hs, sizeof(IcpState), hipMemcpyHostToDevice, s));

    const bool sharded_run = ctx->comm != nullptr || ctx->cb_allreduce != nullptr;
    const bool prepared = !sharded_run && ctx->prep_valid && ctx->prep_tgt == d_tgt && ctx->prep_m == m &&
                          ctx->prep_engine == engine_for(ctx, m, n);
    // (the ranks of a sharded run must agree on the engine -- it decides the order of the normal rows they gather -- so
    // there AUTO looks at the target alone, whatever this rank's share of the source)
    if (!prepared && (rc = prepare_nn(ctx, d_tgt, m, sharded_run ? std::max(n, kMfmaMinQueries) : n))) return rc;
    // with the MFMA engine the resolve kernel also forms the normal-equation partial sums
    const bool fused = ctx->nn_engine == ICPMI_SEARCH_MFMA_BF16;
    // small clouds: one kernel per iteration for the rows' work (icp_small.h), then k_finish_step
    const bool small = !sharded_run && n > 0 && n <= small_max_queries() && small_target(ctx);
    const int rblocks = small ? (n + kSmallQ - 1) / kSmallQ : (fused ? resolve_blocks(n) : reduce_blocks(ctx, n));
    // very many partial rows (a resolve workgroup leaves one per 64 queries) are first added in groups of kSumGroup by a
    // kernel of their own (k_sum_groups, kernels.h): the step kernels then sum rblocks2 rows
    static const int sum_tree_from = [] {
        if (const char *e = getenv("ICPMI_SUM_TREE_FROM")) {
            const long x = strtol(e, nullptr, 10);
            if (x >= 1) return (int)std::min<long>(x, 2000000000l);
        }
        return kSumTreeFrom;
    }();
    const bool sum_tree = rblocks >= sum_tree_from;
    const int rblocks2 = sum_tree ? (rblocks + kSumGroup - 1) / kSumGroup : 0;
    if ((rc = reserve(ctx, ctx->partials, sizeof(double) * kSumsStride * ((size_t)rblocks + (size_t)rblocks2)))) return rc;
    double *partials = (double *)ctx->partials.p;
    // what the step kernels sum
    const double *fin_rows = sum_tree ? partials + kSumsStride * (size_t)rblocks : partials;
    const int fin_blocks = sum_tree ? rblocks2 : rblocks;

    // normals of the target (icp.hpp:169-171).  With several ranks each computes a slice of
    // rows against the full target and the slices are all-gathered.
    const bool sharded = ctx->comm != nullptr || ctx->cb_allreduce != nullptr; // exchanges on, even for 1 rank
    // pruned engine: rows are taken in the target's Morton order, so that a block's neighbours
    // lie in few splits (launch_normals); a rank's slice is then a range of sorted positions
    const bool sorted_rows = sorted_normal_rows(ctx, ctx->opt.normal_k, m);
    if (sharded) {
        int per = (m + ctx->n_ranks - 1) / ctx->n_ranks;
        if (sorted_rows) per = (per + kCoarseQueries - 1) / kCoarseQueries * kCoarseQueries; // whole blocks (and slots) per rank
        if ((rc = reserve(ctx, ctx->stage_a, sizeof(double) * 3 * (size_t)per * ctx->n_ranks))) return rc;
        double *gathered = (double *)ctx->stage_a.p;
        const int row0 = (int)std::min<long>(m, (long)ctx->rank * per), row1 = (int)std::min<long>(m, (long)row0 + per);
        // rows land at their global offset inside `gathered`, so the gather is in place
        if ((rc = launch_normals(ctx, d_tgt, m, ctx->opt.normal_k, row0, row1, gathered, sorted_rows, false))) return rc;
        if ((rc = exchange_allgather(ctx, gathered, 3 * (size_t)per))) return rc;
        if (sorted_rows)
            hipLaunchKernelGGL(k_scatter_rows, dim3((m + 255) / 256), dim3(256), 0, s, (const double *)gathered,
                               (const unsigned *)ctx->sort_keys.p + 3 * (size_t)m, m, nrm);
        else
            HIP_TRY(ctx, hipMemcpyAsync(nrm, gathered, sizeof(double) * 3 * (size_t)m,
                                        hipMemcpyDeviceToDevice, s));
    } else if (!prepared) {
        if ((rc = launch_normals(ctx, d_tgt, m, ctx->opt.normal_k, 0, m, nrm, sorted_rows, true))) return rc;
        if ((rc = sort_normals(ctx, m))) return rc;
    }
    ctx->prep_valid = false; // (the stream path prepares the next target below; nothing else relies on it)

    // Pruned engine: the source is put in Morton order once (a rigid motion keeps neighbours
    // together), so that every block of kCoarseQueries consecutive rows of `cur` is a compact
    // blob whose bounding box can rule whole target splits out.  The order of the rows of
    // `cur` is internal: only sums over all rows leave this function.
    const bool pruned = fused && ctx->nn_pruned && n > 0;
    const SplitFrame *frames = (const SplitFrame *)ctx->frames.p;
    const int splits = ctx->nn_splits;
    int pass_no = 0; // coarse passes queued so far: selects the work counter (see k_nn_coarse_list)
    const unsigned *src_perm = nullptr;
    // All-pairs engine, general kernels: the rows are taken in Morton order too (the order is internal here as well).  A
    // tile of 32 neighbouring rows has its matches in one or two splits, so the bounded pass's epilogue (nn_bounded.h)
    // finds nothing to list in the other 47 and leaves by its short path: with rows in the caller's order half of all
    // (tile, split) pairs listed something (k_nn_coarse_bounded 305 -> 295 us on C3, k_nn_resolve_bounded 29 -> 28).
    // (from 4,096 rows: below, the Morton sort of the rows costs a call more than its passes gain)
    // (not where ICPMI_SMALL=0 puts the general kernels in the small-cloud kernel's place: there they keep its order of rows,
    // hence its bits)
    const bool small_regime = !sharded_run && n <= small_max_queries() && ctx->nn_splits <= small_max_splits();
    // (and only against targets of more than a dozen splits: at 10 splits -- 18k x 20k points, six iterations -- the rows' sort is
    // 55 us of a 0.45 ms registration whose coarse pass is 5 us; profiles/r4_final/c2_20k/timeline.txt)
    constexpr int kSortRowsFromSplits = 12;
    const bool sorted_rows_loop = fused && !pruned && !small && !small_regime && n >= 4096 && ctx->nn_splits > kSortRowsFromSplits &&
                                  resolve_waves(n) != -32;
    size_t sort_bytes = 0;
    if (pruned || sorted_rows_loop) {
        HIP_TRY(ctx, sort_pairs_u32(nullptr, &sort_bytes, nullptr, nullptr, nullptr, nullptr, (unsigned)n, s));
        const int bblocks = std::max(1, std::min(256, (n + 255) / 256));
        if ((rc = reserve(ctx, ctx->src_sort, sizeof(unsigned) * 4 * (size_t)n))) return rc;
        if ((rc = reserve(ctx, ctx->sort_tmp, sort_bytes))) return rc; // the target's sort is done (stream order)
        if ((rc = reserve(ctx, ctx->bbox_part, sizeof(double) * 6 * (size_t)bblocks))) return rc;
    }
    if (pruned) { // the group lists of the coming passes (nn_culled.h); both sets of counters start empty
        if ((rc = reserve_group_lists(ctx, splits, (n + kGroupRows - 1) / kGroupRows))) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->grp_cnt.p, 0, group_cnt_bytes(ctx), s));
    }
    if (pruned || sorted_rows_loop) {
        // (Measured and not kept: this dozen of small launches on a stream of its own beside the target's pre-pass and normals
        // -- 180 us off the call under rocprofv3, whose launch overhead starves the device at the head of a call, and nothing
        // without it: 8,567 against 8,595 iterations/s at C3.)
        const int bblocks = std::max(1, std::min(256, (n + 255) / 256));
        NnFrame *sframe = (NnFrame *)((char *)ctx->nn_misc.p + 64);
        unsigned *keys_in = (unsigned *)ctx->src_sort.p, *keys_out = keys_in + n, *vals_in = keys_in + 2 * (size_t)n,
                 *perm = keys_in + 3 * (size_t)n;
        StageTimer t(ctx, ST_SETUP);
        if (n <= kBboxSingleMax)
            hipLaunchKernelGGL(k_bbox_single, dim3(1), dim3(1024), 0, s, d_src, n, sframe, 1, (unsigned long long *)nullptr);
        else {
            hipLaunchKernelGGL(k_bbox_partial, dim3(bblocks), dim3(256), 0, s, d_src, n, (double *)ctx->bbox_part.p, 1);
            hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(64), 0, s, (const double *)ctx->bbox_part.p, bblocks, sframe);
        }
        hipLaunchKernelGGL(k_morton_keys, dim3((n + 255) / 256), dim3(256), 0, s, d_src, n, (const NnFrame *)sframe,
                           keys_in, vals_in);
        HIP_TRY(ctx, sort_pairs_u32(ctx->sort_tmp.p, &sort_bytes, keys_in, keys_out, vals_in, perm, (unsigned)n, s));
        src_perm = perm;
    }

    // Bounded passes (nn_bounded.h): every kernel that moves the rows after a pass also leaves, per row, the exact distance
    // to the target it was just matched with -- the bound the next pass searches behind.
    // Round 4: the FIRST pass has a bound too -- the nearest of the sorted targets around each row's place in the target's
    // Morton order (k_nn_prebound1) stands in for the previous match -- so no pass of a registration keeps coarse minima:
    // the (row, split) buffer (6 B each: 2.9 GB at 1M x 1M) is not reserved by this function at all.  Every general-path
    // loop of the MFMA engines is bounded now, whatever its row order (ICPMI_NN_BOUNDED=0: round 2's form for the
    // all-pairs engine, the A/B and fuzz reference; the culled engine has no other form).
    const bool bounded_loop = fused && !small && n > 0 && resolve_waves(n) != -32 && (pruned || nn_bounded_enabled());
    if (pruned && !bounded_loop) return fail(ctx, ICPMI_ERR_ARG, "internal: the culled engine needs the bounded resolve kernels");
    RowBounds rb{nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (bounded_loop) {
        if ((rc = reserve(ctx, ctx->nn_lists, kNnListRowBytes * (size_t)n + 64))) return rc;
        const NnListRows lr = nn_list_rows(ctx, n);
        rb = RowBounds{d_tgt, idx, m, lr.ub, lr.ubf, lr.sqf, lr.cnt};
    }

    // current_source = source * R0^T + t0^T (icp.hpp:174-176)
    if (n > 0 && !small) { // (an empty shard of a sharded run launches nothing over its rows; the small-cloud kernel moves them itself)
        StageTimer t(ctx, ST_TRANSFORM);
        hipLaunchKernelGGL(k_transform, dim3(std::min(2048, (n + 255) / 256)), dim3(256), 0, s, d_src, cur, n, st, 1, 0, src_perm);
        if (bounded_loop) // the first pass's incumbents and bounds (and, culled engine, its group lists: set 0)
            hipLaunchKernelGGL(k_nn_prebound1, dim3((n + kPreThreads - 1) / kPreThreads), dim3(kPreThreads), 0, s, (const double *)cur, n, (const double *)ctx->tgt_sorted.p,
                               (const unsigned *)ctx->sort_keys.p + (size_t)m /* the sorted keys */,
                               (const unsigned *)ctx->sort_keys.p + 3 * (size_t)m, m, ctx->nn_ms, (const NnFrame *)ctx->nn_misc.p, idx, rb,
                               frames, splits, pruned ? group_lists(ctx, 0) : GroupLists{nullptr, nullptr, 0});
    }

    unsigned long long *small_clocks = nullptr;
#ifdef ICPMI_SMALL_CLOCKS /* diagnostic build: the stamps of the last pass land in the (idle) slot-minimum buffer */
    if (small) {
        if ((rc = reserve(ctx, ctx->slotmin, sizeof(unsigned long long) * 2 * kSmallStamps * (size_t)rblocks))) return rc;
        small_clocks = (unsigned long long *)ctx->slotmin.p;
    }
#endif
    bool small_first = true; // the small-cloud kernel's first pass applies the initial transform to the caller's rows
    auto iteration = [&](int final_pass, int *progress, int ticket) -> int {
        int r2;
        if (small) {
            {
                Range range("icpmi:nn_search");
                StageTimer t(ctx, ST_NN);
#define ICPMI_SMALL_ARGS                                                                                                        \
    small_first ? d_src : (const double *)cur, cur, n, (const IcpState *)st, small_first ? 1 : 0, (const uint4 *)ctx->bpack.p,    \
        frames, splits, (const double *)ctx->tgt_sorted.p, (const double *)ctx->nrm_sorted.p,                                       \
        (const unsigned *)ctx->sort_keys.p + 3 * (size_t)m, m, ctx->nn_ms, d_tgt, (const double *)nrm, partials,                    \
        (unsigned long long *)((char *)ctx->nn_misc.p + 128), small_clocks
                hipLaunchKernelGGL(k_icp_small, dim3(rblocks), dim3(kSmallThreads), 0, s, ICPMI_SMALL_ARGS);
#undef ICPMI_SMALL_ARGS
                small_first = false;
                ctx->prof.nn_pairs += (double)n * (double)m;
                ctx->prof.small_launches += 1;
                ctx->prof.nn_coarse_blocks += (int64_t)((n + kCoarseQueries - 1) / kCoarseQueries) * splits; // the units the general path would run
            }
            Range range("icpmi:reduce_solve");
            StageTimer t(ctx, ST_REDUCE);
            hipLaunchKernelGGL(k_finish_step, dim3(1), dim3(kFinishThreads), 0, s, partials, rblocks, n, st, hist, final_pass, progress, ticket);
            HIP_TRY(ctx, hipGetLastError());
            return ICPMI_OK;
        }
        const bool fuse_step = sharded && n > 0 && !final_pass;
        const bool fuse_finish = !sharded && n > 0 && !final_pass && fuse_finish_enabled();
        if (n > 0 && fused) {
            if ((r2 = launch_nn_mfma(ctx, cur, n, m, idx, nullptr, st, d_tgt, nrm, partials,
                                     pruned ? pass_no : -1, bounded_loop /* incumbents in idx, bounds in place: from k_nn_prebound1 or the kernel that moved the rows */))) return r2;
            ++pass_no;
        } else if (n > 0) {
            if ((r2 = launch_nn(ctx, cur, n, d_tgt, m, idx, nullptr, st))) return r2;
        }
        {
            Range range("icpmi:reduce_solve");
            StageTimer t(ctx, ST_REDUCE);
            if (!fused)
                hipLaunchKernelGGL(k_reduce, dim3(rblocks), dim3(256), 0, s, cur, n, d_tgt, m, nrm, idx, partials,
                                   st);
            if (sum_tree)
                hipLaunchKernelGGL(k_sum_groups, dim3((rblocks2 + 7) / 8), dim3(256), 0, s, (const double *)partials, rblocks,
                                   partials + kSumsStride * (size_t)rblocks, (const IcpState *)st);
            if (sharded) {
                hipLaunchKernelGGL(k_finish, dim3(1), dim3(kFinishThreads), 0, s, fin_rows, fin_blocks, n,
                                   st);
                {
                    StageTimer tx(ctx, ST_EXCHANGE); // the per-pass all-reduce alone (30 doubles; RCCL on the library's stream)
                    if ((r2 = exchange_allreduce(ctx, st->sums, kNumExchanged))) return r2;
                }
                if (fuse_step) { // step + pose update of this rank's rows in one launch, into the other state buffer
                    IcpState *other = st == ctx->d_state ? ctx->d_state + 1 : ctx->d_state;
                    if (pruned) // (+ the group lists of the coming pass: set pass_no & 1)
                        hipLaunchKernelGGL(k_step_transform_cull, dim3(std::min(2048, (n + 255) / 256)), dim3(256), 0, s,
                                           (const double *)cur, cur, n, (const IcpState *)st, other, hist, progress, ticket,
                                           ctx->n_ranks, rb, frames, splits, group_lists(ctx, pass_no & 1));
                    else
                        hipLaunchKernelGGL(k_step_transform, dim3(std::min(2048, (n + 255) / 256)), dim3(256), 0, s,
                                           (const double *)cur, cur, n, (const IcpState *)st, other, hist, progress, ticket,
                                           ctx->n_ranks, rb);
                    st = other;
                } else {
                    hipLaunchKernelGGL(k_step, dim3(1), dim3(64), 0, s, st, hist, final_pass, progress, ticket,
                                       ctx->n_ranks);
                }
            } else if (fuse_finish) { // final sum + step + pose update of the rows in one launch
                IcpState *other = st == ctx->d_state ? ctx->d_state + 1 : ctx->d_state;
                if (pruned) // (+ the group lists of the coming pass: set pass_no & 1)
                    hipLaunchKernelGGL(k_finish_step_transform_cull, dim3(finish_transform_blocks(n)), dim3(kFinishThreads), 0, s,
                                       fin_rows, fin_blocks, n, (const double *)cur, cur, n, (const IcpState *)st, other,
                                       hist, progress, ticket, rb, frames, splits, group_lists(ctx, pass_no & 1));
                else
                    hipLaunchKernelGGL(k_finish_step_transform, dim3(finish_transform_blocks(n)), dim3(kFinishThreads), 0, s,
                                       fin_rows, fin_blocks, n, (const double *)cur, cur, n,
                                       (const IcpState *)st, other, hist, progress, ticket, rb);
                st = other;
            } else {
                hipLaunchKernelGGL(k_finish_step, dim3(1), dim3(kFinishThreads), 0, s, fin_rows, fin_blocks, n,
                                   st, hist, final_pass, progress, ticket);
            }
        }
        if (!final_pass && n > 0 && !fuse_step && !fuse_finish) {
            Range range("icpmi:transform");
            StageTimer t(ctx, ST_TRANSFORM);
            if (pruned) // (ICPMI_FUSE_FINISH=0 only: + the rows' bounds and the group lists of the coming pass, set pass_no & 1)
                hipLaunchKernelGGL(k_transform_cull, dim3((n + 255) / 256), dim3(256), 0, s, (const double *)cur, (const unsigned *)nullptr,
                                   cur, n, (const IcpState *)st, 0, 1, rb, frames, splits, group_lists(ctx, pass_no & 1));
            else
                hipLaunchKernelGGL(k_transform, dim3(std::min(2048, (n + 255) / 256)), dim3(256), 0, s,
                                   cur, cur, n, st, 0, 1, (const unsigned *)nullptr, rb);
        }
        return ICPMI_OK;
    };

    Range range_loop("icpmi:icp_loop");
    StageTimer *t_loop = new StageTimer(ctx, ST_LOOP);
    struct Closer2 { StageTimer *&p; ~Closer2() { delete p; p = nullptr; } } close_loop{t_loop};
    for (int i = 0; i < kFlagRing; ++i) ctx->h_flags[i] = 0;
    bool ended_early = false; // the host has SEEN the device leave the loop on a convergence test
    // `after_first` (may be null): called once the first iterations are queued and before the host first waits for
    // the device -- what it queues (on another stream) costs the host its launch time while the device is busy
    bool after_first_done = after_first == nullptr;
    for (int it = 0; it < max_it; ++it) {
        if (it >= kLag && !after_first_done) {
            after_first_done = true;
            (void)after_first(ctx); // (a failure there only means the next call prepares its target itself)
        }
        if (it >= kLag) {
            // wait until iteration it - kLag has reported; stop queueing once the loop has ended
            const int slot = (it - kLag) % kFlagRing, want = it - kLag + 1;
            volatile int32_t *flag = &ctx->h_flags[slot];
            int32_t v = *flag;
            for (long spin = 0; (v >> 1) != want; ++spin) {
                if (spin > 2000000) { // ~seconds: something is wrong with the stream, find out what
                    HIP_TRY(ctx, hipStreamSynchronize(s));
                    v = *flag;
                    if ((v >> 1) != want) return fail(ctx, ICPMI_ERR_HIP, "device progress word never arrived");
                    break;
                }
                if ((spin & 63) == 63) sched_yield();
                v = *flag;
            }
            if (v & 1) { // loop already left on the device
                ended_early = true;
                break;
            }
        }
        if ((rc = iteration(0, ctx->d_flags + it % kFlagRing, it + 1))) return rc;
    }
    // post-loop evaluation (icp.hpp:235-252): a full pass after exhaustion; after a convergence break it restates
    // the last error, which the step that broke the loop has already entered (step_update): nothing to queue then.
    // (Single-GPU loops only: in a sharded run `done` is agreed one exchange later, k_step.)
    if (!(ended_early && !sharded) && (rc = iteration(1, nullptr, 0))) return rc;
    if (!after_first_done) (void)after_first(ctx);
    delete t_loop;
    t_loop = nullptr;
    delete t_total;
    t_total = nullptr;

    // state and history come back behind ONE wait (the history through the pinned staging area: a
    // second, blocking copy after the wait was one more host round trip per call)
    const int hcopy = std::min(max_hist, history_cap);
    if ((int)ctx->h_hist_cap < hcopy) {
        if (ctx->h_hist) (void)hipHostFree(ctx->h_hist);
        ctx->h_hist = nullptr;
        ctx->h_hist_cap = 0;
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_hist, sizeof(double) * (size_t)(hcopy + 64), hipHostMallocDefault));
        ctx->h_hist_cap = (size_t)hcopy + 64;
    }
    HIP_TRY(ctx, hipMemcpyAsync(hs, st, sizeof(IcpState), hipMemcpyDeviceToHost, s));
    if (hcopy > 0) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_hist, hist, sizeof(double) * (size_t)hcopy, hipMemcpyDeviceToHost, s));
    // profiling: the resolve's counters come back behind the same wait, and the events are read when somebody asks
    // for the profile (icpmi_get_profile) -- a blocking copy and a dozen hipEventElapsedTime calls inside every
    // call were ~0.1 ms of host time in a 9 ms C3 call, charged to the very thing they measure
    const bool counters_back = ctx->opt.profile && ctx->nn_engine == ICPMI_SEARCH_MFMA_BF16 && ctx->nn_misc.p;
    if (counters_back) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_cnt, (char *)ctx->nn_misc.p + 128, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipMemsetAsync((char *)ctx->nn_misc.p + 128, 0, 3 * sizeof(unsigned long long), s));
        ctx->h_cnt[3] = ctx->h_cnt[4] = 0;
        if (pruned) // (row group, split) pairs run / of all passes: cleared with the counters at the head of the next call
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_cnt + 3, group_stats(ctx), 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    }
    if (before_wait && ctx->opt.profile == 0) {
        // the results are waited for through an event; what before_wait queues behind it keeps the device busy
        // while the caller digests them (with profiling on, the stage timers want the whole stream drained)
        if (!ctx->result_ready) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->result_ready, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(ctx->result_ready, s));
        {   // (a failure there only means the next call prepares its target itself: it must not leave its text in
            // the error string of a call that succeeded)
            const std::string keep = ctx->err;
            if (before_wait(ctx) != ICPMI_OK) ctx->err = keep;
            ctx->trailing_work = true; // kernels of the NEXT call's target are queued behind this call's results
        }
        HIP_TRY(ctx, hipEventSynchronize(ctx->result_ready));
    } else {
        HIP_TRY(ctx, hipStreamSynchronize(s));
    }
    HIP_TRY(ctx, hipGetLastError());
    const int hl = std::min(hs->hist_len, max_hist);
    if (hl > 0) memcpy(error_history, ctx->h_hist, sizeof(double) * (size_t)std::min(hl, history_cap));
    if (counters_back) {
        ctx->prof.nn_recheck_queries += (int64_t)ctx->h_cnt[0];
        ctx->prof.nn_fallback_queries += (int64_t)ctx->h_cnt[1];
        ctx->prof.nn_pruned_blocks += (int64_t)ctx->h_cnt[2];
        ctx->prof.nn_group_pairs_run += (int64_t)ctx->h_cnt[3];
        ctx->prof.nn_group_pairs += (int64_t)ctx->h_cnt[4];
        // (the older pair of fields, in 512-row units like the all-pairs engine's)
        ctx->prof.nn_pruned_blocks += (int64_t)(ctx->h_cnt[4] - ctx->h_cnt[3]) / (kCoarseQueries / kGroupRows); // (16 tiles to a 512-row block)
    }
    if (ctx->ev_used > 4096) harvest_profile(ctx); // (otherwise when the profile is asked for)
    if (hs->error)
        return fail(ctx, ICPMI_ERR_RCCL, "the ranks of this sharded run disagreed on the end of the loop "
                                         "(different icpmi_config per rank, or an exchange that is not bit-identical on every rank)");

    memcpy(result->transformation, hs->total, sizeof(double) * 16); // icp.hpp:254
    result->converged = hs->converged;
    result->num_iterations = hs->hist_len - 1;                      // icp.hpp:255
    result->final_error = hs->final_error;
    result->history_len = hl;
    result->loop_iterations = hs->loops;
    return ICPMI_OK;
}

// voxel filter on device memory -> d_out (n_out rows); see voxel.h
// `finish`: wait for the centroids before returning (the public entry points promise complete
// outputs; icpmi_stream_push queues the registration behind it on the same stream instead)
// The voxel filter's working memory: the context has one set (shared with its other stages), the prefetch
// worker another, so that it can filter the next scan on its own stream while the context registers this one.
struct VoxelScratch {
    DevBuf *bbox_part, *box, *keys, *vals, *tmp;
    size_t box_offset; // of the VoxelBox inside *box
};

// No context in here (the worker thread runs it too): a code and a static message come back.
// `deferred` (the prefetch worker; implies `finish`): nothing waits for the number of voxels in the middle -- the
// prefix sum runs over all n counts, the centroid kernel is launched for min(n, out_cap) voxels and reads the
// number itself -- and count and range flag come back behind the centroids, with the call's only wait.
// `async_out` (pinned, two words; with `deferred` only): nothing is waited for at all -- run count and range flag are
// copied there behind the centroids and the CALLER looks at them once it has waited for the stream (the prefetch
// worker: the push of the file does); *n_out is then left alone.
int voxel_filter_core(hipStream_t s, VoxelScratch vs, const double *d_pts, int n, double voxel, double *d_out, int64_t out_cap,
                      int64_t *n_out, bool finish, const char **msg, bool deferred = false, unsigned *async_out = nullptr)
{
#define VOX_TRY(call)                                   \
    do {                                                \
        const hipError_t e_ = (call);                   \
        if (e_ != hipSuccess) {                         \
            *msg = hipGetErrorString(e_);               \
            return ICPMI_ERR_HIP;                       \
        }                                               \
    } while (0)
    // key origin from the bounding box, formed on the device (k_voxel_keys)
    const int bblocks = std::max(1, std::min(256, (n + 255) / 256));
    VOX_TRY(reserve_raw(*vs.bbox_part, sizeof(double) * 6 * (size_t)bblocks));
    VOX_TRY(reserve_raw(*vs.box, vs.box_offset + 64));
    VoxelBox *box = (VoxelBox *)((char *)vs.box->p + vs.box_offset);
    static_assert(sizeof(VoxelBox) == sizeof(NnFrame) && sizeof(VoxelBox) <= 64, "k_bbox_final writes an NnFrame here");
    hipLaunchKernelGGL(k_bbox_partial, dim3(bblocks), dim3(256), 0, s, d_pts, n, (double *)vs.bbox_part->p, 0);
    hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(64), 0, s, (const double *)vs.bbox_part->p, bblocks, (NnFrame *)box);
    // keys_in | keys_out | unique ; vals_in | order | counts | offsets | runs, flag
    const size_t un = (size_t)n;
    VOX_TRY(reserve_raw(*vs.keys, sizeof(unsigned long long) * 3 * un));
    VOX_TRY(reserve_raw(*vs.vals, sizeof(unsigned) * (4 * un + 16)));
    unsigned long long *keys_in = (unsigned long long *)vs.keys->p, *keys_out = keys_in + un, *uniq = keys_in + 2 * un;
    unsigned *vals_in = (unsigned *)vs.vals->p, *order = vals_in + un, *counts = vals_in + 2 * un,
             *offsets = vals_in + 3 * un, *runs_d = vals_in + 4 * un;
    size_t b1 = 0, b2 = 0, b3 = 0;
    VOX_TRY(sort_pairs_u64(nullptr, &b1, keys_in, keys_out, vals_in, order, (unsigned)n, s));
    VOX_TRY(run_lengths_u64(nullptr, &b2, keys_out, (unsigned)n, uniq, counts, runs_d, s));
    VOX_TRY(exclusive_sum_u32(nullptr, &b3, counts, offsets, (unsigned)n, s));
    VOX_TRY(reserve_raw(*vs.tmp, std::max(b1, std::max(b2, b3))));
    VOX_TRY(hipMemsetAsync(runs_d + 1, 0, sizeof(unsigned), s)); // the flag word: k_voxel_keys ORs into it
    hipLaunchKernelGGL(k_voxel_keys, dim3((n + 255) / 256), dim3(256), 0, s, d_pts, n, voxel, (const VoxelBox *)box, runs_d + 1,
                       keys_in, vals_in);
    VOX_TRY(sort_pairs_u64(vs.tmp->p, &b1, keys_in, keys_out, vals_in, order, (unsigned)n, s));
    VOX_TRY(run_lengths_u64(vs.tmp->p, &b2, keys_out, (unsigned)n, uniq, counts, runs_d, s));
    unsigned runs_flag[2] = {0, 0};
    if (deferred) {
        const int bound = (int)std::min<int64_t>(n, out_cap);
        VOX_TRY(exclusive_sum_u32(vs.tmp->p, &b3, counts, offsets, (unsigned)n, s));
        hipLaunchKernelGGL(k_voxel_centroids, dim3((bound + 3) / 4), dim3(256), 0, s, d_pts, (const unsigned *)order,
                           (const unsigned *)offsets, (const unsigned *)counts, bound, d_out, (const unsigned *)runs_d);
        VOX_TRY(hipGetLastError());
    }
    if (deferred && async_out) {
        VOX_TRY(hipMemcpyAsync(async_out, runs_d, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
        return ICPMI_OK;
    }
    VOX_TRY(hipMemcpyAsync(runs_flag, runs_d, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s)); // the call's one round trip
    VOX_TRY(hipStreamSynchronize(s));
    if (runs_flag[1]) {
        *msg = "voxel grid spans more than 2^21 cells on an axis, or a point has a non-finite coordinate";
        return ICPMI_ERR_ARG;
    }
    const unsigned runs = runs_flag[0];
    if ((int64_t)runs > out_cap) {
        *msg = "output holds fewer rows than the filter yields";
        return ICPMI_ERR_CAPACITY;
    }
    if (!deferred) {
        VOX_TRY(exclusive_sum_u32(vs.tmp->p, &b3, counts, offsets, runs, s));
        hipLaunchKernelGGL(k_voxel_centroids, dim3((runs + 3) / 4), dim3(256), 0, s, d_pts, (const unsigned *)order,
                           (const unsigned *)offsets, (const unsigned *)counts, (int)runs, d_out);
        VOX_TRY(hipGetLastError());
        if (finish) VOX_TRY(hipStreamSynchronize(s));
    }
    *n_out = runs;
    return ICPMI_OK;
#undef VOX_TRY
}

int voxel_downsample_device(icpmi_ctx *ctx, const double *d_pts, int n, double voxel, double *d_out,
                            int64_t out_cap, int64_t *n_out, bool finish = true)
{
    hipStream_t s = ctx->stream;
    if (!(voxel > 0.0)) { // file_utils.cpp:152: returns the input unchanged
        if (out_cap < n) return fail(ctx, ICPMI_ERR_CAPACITY, "output holds %lld rows, needs %d", (long long)out_cap, n);
        HIP_TRY(ctx, hipMemcpyAsync(d_out, d_pts, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToDevice, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        *n_out = n;
        return ICPMI_OK;
    }
    // (the box sits at offset 192 of nn_misc: offset 0 is the search's own frame of the target)
    const VoxelScratch vs{&ctx->bbox_part, &ctx->nn_misc, &ctx->vox_keys, &ctx->vox_vals, &ctx->sort_tmp, 192};
    const char *msg = "";
    const int rc = voxel_filter_core(s, vs, d_pts, n, voxel, d_out, out_cap, n_out, finish, &msg);
    return rc == ICPMI_OK ? rc : fail(ctx, rc, "%s", msg);
}

// update_occupancy_grid (slam_node.cpp:211-221) on device memory: the frame's keys, less those the set already
// holds, sorted, made unique and merged into the set (occupancy.h); the set's new size is queued for the host
// (ctx->h_grid[0]) and picked up by grid_finish after the caller's wait.
int grid_update_queue(icpmi_ctx *ctx, const double *d_world, int n, const double sensor[3], const icpmi_grid_config *grid)
{
    if (!(grid->resolution > 0.0)) return fail(ctx, ICPMI_ERR_ARG, "grid resolution must be positive");
    hipStream_t s = ctx->stream;
    int rc;
    const size_t set_n = (size_t)ctx->grid_n, total = set_n + (size_t)n;
    if (total > (size_t)2000000000) return fail(ctx, ICPMI_ERR_ARG, "occupancy set too large");
    if (n <= 0) { // nothing to insert
        ctx->h_grid[0] = (unsigned)set_n;
        return ICPMI_OK;
    }
    const size_t un = (size_t)n;
    // grid_in: the frame's keys | sorted | unique ; grid_cnt: run lengths | runs, count ; grid_out: the merged set
    if ((rc = reserve(ctx, ctx->grid_in, sizeof(unsigned long long) * 3 * un))) return rc;
    if ((rc = reserve(ctx, ctx->grid_cnt, sizeof(unsigned) * (un + 16)))) return rc;
    if ((rc = reserve(ctx, ctx->grid_out, sizeof(unsigned long long) * total))) return rc;
    unsigned long long *keys = (unsigned long long *)ctx->grid_in.p, *sorted = keys + un, *uniq = keys + 2 * un;
    unsigned *counts = (unsigned *)ctx->grid_cnt.p, *runs_d = counts + un, *count_d = runs_d + 1;
    size_t b1 = 0, b2 = 0, b3 = 0;
    HIP_TRY(ctx, sort_keys_u64(nullptr, &b1, keys, sorted, (unsigned)n, s));
    HIP_TRY(ctx, run_lengths_u64(nullptr, &b2, sorted, (unsigned)n, uniq, counts, runs_d, s));
    HIP_TRY(ctx, merge_keys_u64(nullptr, &b3, (const unsigned long long *)ctx->grid_set.p, uniq, (unsigned long long *)ctx->grid_out.p,
                                (unsigned)set_n, (unsigned)n, s));
    if ((rc = reserve(ctx, ctx->sort_tmp, std::max(b1, std::max(b2, b3))))) return rc;
    GridParams g{sensor[0], sensor[1], grid->resolution, grid->height_min, grid->height_max, grid->max_range};
    hipLaunchKernelGGL(k_grid_keys, dim3((n + 255) / 256), dim3(256), 0, s, d_world, n, g, keys);
    if (set_n > 0)
        hipLaunchKernelGGL(k_grid_drop_known, dim3((n + 255) / 256), dim3(256), 0, s, keys, n,
                           (const unsigned long long *)ctx->grid_set.p, (int)set_n);
    HIP_TRY(ctx, sort_keys_u64(ctx->sort_tmp.p, &b1, keys, sorted, (unsigned)n, s));
    HIP_TRY(ctx, run_lengths_u64(ctx->sort_tmp.p, &b2, sorted, (unsigned)n, uniq, counts, runs_d, s));
    hipLaunchKernelGGL(k_grid_pad, dim3((n + 255) / 256), dim3(256), 0, s, uniq, n, (const unsigned *)runs_d, (unsigned)set_n, count_d);
    HIP_TRY(ctx, merge_keys_u64(ctx->sort_tmp.p, &b3, set_n > 0 ? (const unsigned long long *)ctx->grid_set.p : uniq, uniq,
                                (unsigned long long *)ctx->grid_out.p, (unsigned)set_n, (unsigned)n, s));
    std::swap(ctx->grid_set, ctx->grid_out); // the merged array (cells, then n - new entries of kGridNone) is the set now
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_grid, count_d, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipGetLastError());
    return ICPMI_OK;
}
// after the stream has been waited for
void grid_finish(icpmi_ctx *ctx) { ctx->grid_n = (int64_t)ctx->h_grid[0]; }

int validate_align(icpmi_ctx *ctx, const void *src, int64_t n_src, const void *tgt, int64_t n_tgt,
                   const icpmi_config *cfg, icpmi_result *result, double *hist, int32_t cap)
{
    if (!ctx) return ICPMI_ERR_NULL;
    if (!tgt || !cfg || !result || !hist || (!src && n_src != 0)) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    // a rank of a sharded run may hold an empty shard (fewer source points than ranks, uneven
    // sharding): it contributes zero sums and a zero count and takes part in every exchange
    const bool may_be_empty = ctx->n_ranks > 1 && (ctx->comm != nullptr || ctx->cb_allreduce != nullptr);
    if (n_src < 0 || (n_src == 0 && !may_be_empty)) return fail(ctx, ICPMI_ERR_EMPTY_SOURCE, "empty source cloud");
    if (n_tgt <= 0) return fail(ctx, ICPMI_ERR_EMPTY_TARGET, "empty target cloud");
    if (n_src > (int64_t)700000000 || n_tgt > (int64_t)700000000)
        return fail(ctx, ICPMI_ERR_ARG, "cloud larger than 7e8 points");
    if (cfg->max_iterations < 0) return fail(ctx, ICPMI_ERR_ARG, "max_iterations < 0");
    if (cap < cfg->max_iterations + 1)
        return fail(ctx, ICPMI_ERR_CAPACITY, "error_history holds %d entries, needs %d", cap,
                    cfg->max_iterations + 1);
    return ICPMI_OK;
}

} // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

const char *icpmi_version(void) { return "icp_mi355x 0.1 (gfx950)"; }

void icpmi_options_default(icpmi_options *opt)
{
    if (!opt) return;
    opt->device = 0;
    opt->normal_k = 20; // icp.hpp:170
    opt->search = ICPMI_SEARCH_AUTO;
    opt->profile = 0;
    // a drop-in caller (slam_icp_adapter.hpp) never sees the options: let the environment pick
    // the engine.  Values outside 0..3 are ignored.
    if (const char *e = getenv("ICPMI_SEARCH")) {
        char *end = nullptr;
        const long v = strtol(e, &end, 10);
        if (end != e && *end == '\0' && v >= ICPMI_SEARCH_AUTO && v <= ICPMI_SEARCH_MFMA_PRUNED) opt->search = (int32_t)v;
    }
}

void icpmi_config_default(icpmi_config *cfg)
{
    if (!cfg) return;
    cfg->max_iterations = 50; // types.hpp:144
    cfg->reserved = 0;
    cfg->tolerance = 1e-6;    // types.hpp:145
    cfg->min_error = 1e-9;    // types.hpp:146
    for (int i = 0; i < 16; ++i) cfg->initial_transform[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

int icpmi_create(const icpmi_options *opt, icpmi_ctx **out)
{
    if (!out) return fail(nullptr, ICPMI_ERR_NULL, "out is NULL");
    *out = nullptr;
    icpmi_options o;
    if (opt) o = *opt;
    else icpmi_options_default(&o);
    if (o.normal_k < 1 || o.normal_k > 64)
        return fail(nullptr, ICPMI_ERR_ARG, "normal_k %d outside [1,64]", o.normal_k);
    if (o.search < ICPMI_SEARCH_AUTO || o.search > ICPMI_SEARCH_MFMA_PRUNED)
        return fail(nullptr, ICPMI_ERR_ARG, "search %d is not an ICPMI_SEARCH_* value", o.search);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(nullptr, ICPMI_ERR_NO_DEVICE, "no HIP device visible");
    if (o.device < 0 || o.device >= count)
        return fail(nullptr, ICPMI_ERR_ARG, "device %d outside [0,%d)", o.device, count);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, o.device) != hipSuccess)
        return fail(nullptr, ICPMI_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, ICPMI_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only",
                    o.device, prop.gcnArchName);
    icpmi_ctx *ctx = new icpmi_ctx();
    ctx->opt = o;
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    memset(&ctx->prof, 0, sizeof(ctx->prof));
    auto bail = [&](const char *what) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(hipGetLastError());
        icpmi_destroy(ctx);
        return ICPMI_ERR_HIP;
    };
    if (hipSetDevice(o.device) != hipSuccess) return bail("hipSetDevice");
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate");
    if (hipMalloc((void **)&ctx->d_state, 2 * sizeof(IcpState)) != hipSuccess) return bail("hipMalloc state");
    if (hipHostMalloc((void **)&ctx->h_state, sizeof(IcpState), hipHostMallocDefault) != hipSuccess) return bail("hipHostMalloc");
    if (hipHostMalloc((void **)&ctx->h_flags, sizeof(int32_t) * kFlagRing, hipHostMallocMapped) != hipSuccess) return bail("hipHostMalloc");
    if (hipHostMalloc((void **)&ctx->h_grid, 4 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) return bail("hipHostMalloc");
    if (hipHostMalloc((void **)&ctx->h_cnt, 8 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) return bail("hipHostMalloc");
    memset(ctx->h_flags, 0, sizeof(int32_t) * kFlagRing);
    if (hipHostGetDevicePointer((void **)&ctx->d_flags, ctx->h_flags, 0) != hipSuccess) return bail("hipHostGetDevicePointer");
    *out = ctx;
    return ICPMI_OK;
}

void icpmi_destroy(icpmi_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->opt.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (FilePrefetch *pf = ctx->prefetch) {
        {
            std::lock_guard<std::mutex> lk(pf->mu);
            pf->quit = true;
        }
        pf->cv.notify_all();
        if (pf->worker.joinable()) pf->worker.join();
        if (getenv("ICPMI_PREFETCH_STATS") && pf->files > 0)
            fprintf(stderr, "prefetch worker, ms per file over %ld files: wait for slot %.4f, open + read (first half on its way) %.4f, queueing %.4f, filter + wait %.4f\n",
                    pf->files, pf->t_wait / pf->files, pf->t_read / pf->files, pf->t_upload / pf->files, pf->t_filter / pf->files);
        if (pf->stream) (void)hipStreamSynchronize(pf->stream);
        if (pf->h_runs) (void)hipHostFree(pf->h_runs);
        for (int k = 0; k < 2; ++k) {
            if (pf->h_buf[k]) (void)hipHostFree(pf->h_buf[k]);
            if (pf->filled[k]) (void)hipEventDestroy(pf->filled[k]);
            if (pf->d_f32[k]) (void)hipFree(pf->d_f32[k]);
            if (pf->d_raw[k]) (void)hipFree(pf->d_raw[k]);
        }
        if (pf->stream) (void)hipStreamDestroy(pf->stream);
        for (int k = 0; k < 2; ++k)
            if (pf->slot_done[k]) (void)hipEventDestroy(pf->slot_done[k]);
        for (DevBuf *b : {&pf->filtered[0], &pf->filtered[1], &pf->sc_bbox, &pf->sc_box, &pf->sc_keys, &pf->sc_vals, &pf->sc_tmp})
            release(*b);
        delete pf;
        ctx->prefetch = nullptr;
    }
    if (getenv("ICPMI_STREAM_STATS") && ctx->pushes > 0)
        fprintf(stderr, "frame stream, calling thread, ms per push over %ld pushes (the first ones carry the allocations): whole push %.4f, of it "
                        "waiting for the prefetched file %.4f, registration call %.4f (of it queueing the next target's preparation %.4f)\n",
                ctx->pushes, ctx->t_push / ctx->pushes, ctx->t_wait_file / ctx->pushes, ctx->t_align / ctx->pushes, ctx->t_prep_queue / ctx->pushes);
    if (getenv("ICPMI_STREAM_STATS") && ctx->pushes > 0) fprintf(stderr, "  device span of the registration (events on the context's stream): %.4f ms per push\n", ctx->t_gpu_span / ctx->pushes);
    if (ctx->prep_helper) icpmi_destroy(ctx->prep_helper);
    ctx->prep_helper = nullptr;
    if (ctx->prep_done) (void)hipEventDestroy(ctx->prep_done);
    if (ctx->scan_ready) (void)hipEventDestroy(ctx->scan_ready);
    for (BatchWorker *w : ctx->helpers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        icpmi_destroy(w->helper);
        delete w;
    }
    ctx->helpers.clear();
    if (ctx->comm && ctx->rccl.CommDestroy) ctx->rccl.CommDestroy(ctx->comm);
    for (DevBuf *b : {&ctx->cur, &ctx->nrm, &ctx->idx, &ctx->part_d2, &ctx->part_idx, &ctx->partials,
                      &ctx->history, &ctx->stage_a, &ctx->stage_b, &ctx->stage_c, &ctx->d2out, &ctx->src_sort, &ctx->grp_cnt, &ctx->grp_items,
                      &ctx->bpack, &ctx->coarse, &ctx->bbox_part, &ctx->nn_misc, &ctx->knn_idx, &ctx->slotmin, &ctx->nn_lists, &ctx->nrm_sorted,
                      &ctx->fb_list, &ctx->sort_keys, &ctx->sort_tmp, &ctx->tgt_sorted, &ctx->frames, &ctx->vox_keys,
                      &ctx->vox_vals, &ctx->vox_out, &ctx->stream_prev, &ctx->stream_cur, &ctx->f32_stage, &ctx->grid_set,
                      &ctx->grid_in, &ctx->grid_out, &ctx->grid_cnt, &ctx->world})
        release(*b);
    if (ctx->h_grid) (void)hipHostFree(ctx->h_grid);
    if (ctx->h_cnt) (void)hipHostFree(ctx->h_cnt);
    if (ctx->result_ready) (void)hipEventDestroy(ctx->result_ready);
    if (ctx->d_state) (void)hipFree(ctx->d_state);
    if (ctx->h_state) (void)hipHostFree(ctx->h_state);
    if (ctx->h_hist) (void)hipHostFree(ctx->h_hist);
    if (ctx->h_file) (void)hipHostFree(ctx->h_file);
    if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
    for (EventPair &p : ctx->ev_pool) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *icpmi_last_error(const icpmi_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int icpmi_align_device(icpmi_ctx *ctx, const double *d_source_xyz, int64_t n_src,
                       const double *d_target_xyz, int64_t n_tgt, const icpmi_config *cfg,
                       icpmi_result *result, double *error_history, int32_t history_cap)
{
    int rc = validate_align(ctx, d_source_xyz, n_src, d_target_xyz, n_tgt, cfg, result,
                            error_history, history_cap);
    if (rc) return rc;
    if ((rc = check_common(ctx))) return rc;
    return align_device(ctx, d_source_xyz, n_src, d_target_xyz, n_tgt, cfg, result, error_history,
                        history_cap);
}

int icpmi_align(icpmi_ctx *ctx, const double *source_xyz, int64_t n_src, const double *target_xyz,
                int64_t n_tgt, const icpmi_config *cfg, icpmi_result *result,
                double *error_history, int32_t history_cap)
{
    int rc = validate_align(ctx, source_xyz, n_src, target_xyz, n_tgt, cfg, result, error_history,
                            history_cap);
    if (rc) return rc;
    if ((rc = check_common(ctx))) return rc;
    if ((rc = reserve(ctx, ctx->stage_b, sizeof(double) * 3 * (size_t)n_src))) return rc;
    if ((rc = reserve(ctx, ctx->stage_c, sizeof(double) * 3 * (size_t)n_tgt))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_b.p, source_xyz, sizeof(double) * 3 * (size_t)n_src,
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_c.p, target_xyz, sizeof(double) * 3 * (size_t)n_tgt,
                                hipMemcpyHostToDevice, ctx->stream));
    return align_device(ctx, (const double *)ctx->stage_b.p, n_src, (const double *)ctx->stage_c.p,
                        n_tgt, cfg, result, error_history, history_cap);
}

int icpmi_align_batch(icpmi_ctx *ctx, int32_t count, const double *const *sources_xyz, const int64_t *n_src,
                      const double *const *targets_xyz, const int64_t *n_tgt, const icpmi_config *cfgs,
                      icpmi_result *results, double *error_history, int32_t history_stride, int32_t *status)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!sources_xyz || !n_src || !targets_xyz || !n_tgt || !cfgs || !results || !error_history || !status)
        return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (count < 1 || count > ICPMI_MAX_BATCH) return fail(ctx, ICPMI_ERR_ARG, "count %d outside [1,%d]", count, ICPMI_MAX_BATCH);
    if (ctx->comm || ctx->cb_allreduce) return fail(ctx, ICPMI_ERR_ARG, "a context with a communicator registers one (sharded) problem at a time");
    while ((int)ctx->helpers.size() < count - 1) { // grown once, kept: their workspaces are grow-only like ctx's own
        icpmi_options o = ctx->opt;
        o.profile = 0;
        icpmi_ctx *h = nullptr;
        if ((rc = icpmi_create(&o, &h))) return fail(ctx, rc, "helper context: %s", icpmi_last_error(nullptr));
        BatchWorker *w = new BatchWorker();
        w->helper = h;
        w->th = std::thread([w] {
            std::unique_lock<std::mutex> lk(w->mu);
            for (;;) {
                w->cv.wait(lk, [w] { return w->quit || w->has_job; });
                if (w->quit) return;
                lk.unlock();
                w->job();
                lk.lock();
                w->has_job = false;
                w->done = true;
                w->cv.notify_all();
            }
        });
        ctx->helpers.push_back(w);
    }
    auto run = [&](int k) {
        icpmi_ctx *c = k == 0 ? ctx : ctx->helpers[(size_t)k - 1]->helper;
        status[k] = icpmi_align(c, sources_xyz[k], n_src[k], targets_xyz[k], n_tgt[k], &cfgs[k], &results[k],
                                error_history + (size_t)k * (size_t)history_stride, history_stride);
    };
    for (int k = 1; k < count; ++k) {
        BatchWorker *w = ctx->helpers[(size_t)k - 1];
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->job = [&run, k] { run(k); };
            w->done = false;
            w->has_job = true;
        }
        w->cv.notify_all();
    }
    run(0);
    for (int k = 1; k < count; ++k) {
        BatchWorker *w = ctx->helpers[(size_t)k - 1];
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [w] { return w->done; });
    }
    for (int k = 0; k < count; ++k)
        if (status[k] != ICPMI_OK) {
            if (k > 0) ctx->err = ctx->helpers[(size_t)k - 1]->helper->err;
            return status[k];
        }
    return ICPMI_OK;
}

int icpmi_nearest_batch(icpmi_ctx *ctx, const double *targets_xyz, int64_t n_tgt,
                        const double *queries_xyz, int64_t n_qry, int32_t *indices, double *dist_sq)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!targets_xyz || !queries_xyz || !indices) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n_tgt <= 0) return fail(ctx, ICPMI_ERR_EMPTY_TARGET, "empty target cloud");
    if (n_qry < 0) return fail(ctx, ICPMI_ERR_ARG, "n_qry < 0");
    if (n_qry > (int64_t)700000000 || n_tgt > (int64_t)700000000) return fail(ctx, ICPMI_ERR_ARG, "cloud larger than 7e8 points");
    if (n_qry == 0) return ICPMI_OK;
    const int n = (int)n_qry, m = (int)n_tgt;
    if ((rc = reserve(ctx, ctx->stage_b, sizeof(double) * 3 * (size_t)n))) return rc;
    if ((rc = reserve(ctx, ctx->stage_c, sizeof(double) * 3 * (size_t)m))) return rc;
    if ((rc = reserve(ctx, ctx->idx, sizeof(int) * (size_t)n))) return rc;
    if ((rc = reserve(ctx, ctx->d2out, sizeof(double) * (size_t)n))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_b.p, queries_xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_c.p, targets_xyz, sizeof(double) * 3 * (size_t)m, hipMemcpyHostToDevice, s));
    if ((rc = prepare_nn(ctx, (const double *)ctx->stage_c.p, m, n))) return rc;
    if ((rc = launch_nn(ctx, (const double *)ctx->stage_b.p, n, (const double *)ctx->stage_c.p, m,
                        (int *)ctx->idx.p, (double *)ctx->d2out.p, nullptr)))
        return rc;
    HIP_TRY(ctx, hipMemcpyAsync(indices, ctx->idx.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
    if (dist_sq)
        HIP_TRY(ctx, hipMemcpyAsync(dist_sq, ctx->d2out.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    harvest_profile(ctx);
    return ICPMI_OK;
}

int icpmi_estimate_normals(icpmi_ctx *ctx, const double *points_xyz, int64_t n, int32_t k,
                           double *normals_xyz)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!points_xyz || !normals_xyz) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n <= 0) return fail(ctx, ICPMI_ERR_EMPTY_TARGET, "empty cloud");
    if (n > (int64_t)700000000) return fail(ctx, ICPMI_ERR_ARG, "cloud larger than 7e8 points");
    if (k < 1 || k > 64) return fail(ctx, ICPMI_ERR_ARG, "k %d outside [1,64]", k);
    const int m = (int)n;
    if ((rc = reserve(ctx, ctx->stage_c, sizeof(double) * 3 * (size_t)m))) return rc;
    if ((rc = reserve(ctx, ctx->nrm, sizeof(double) * 3 * (size_t)m))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_c.p, points_xyz, sizeof(double) * 3 * (size_t)m, hipMemcpyHostToDevice, s));
    if ((rc = prepare_nn(ctx, (const double *)ctx->stage_c.p, m, m))) return rc;
    const bool sorted_rows = sorted_normal_rows(ctx, k, m);
    if ((rc = launch_normals(ctx, (const double *)ctx->stage_c.p, m, k, 0, m, (double *)ctx->nrm.p, sorted_rows, true))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(normals_xyz, ctx->nrm.p, sizeof(double) * 3 * (size_t)m, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    harvest_profile(ctx);
    return ICPMI_OK;
}

int icpmi_k_nearest(icpmi_ctx *ctx, const double *targets_xyz, int64_t n_tgt, const double *queries_xyz,
                    int64_t n_qry, int32_t k, int32_t *indices, double *dist_sq)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!targets_xyz || !queries_xyz || !indices) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n_tgt <= 0) return fail(ctx, ICPMI_ERR_EMPTY_TARGET, "empty target cloud");
    if (n_qry < 0) return fail(ctx, ICPMI_ERR_ARG, "n_qry < 0");
    if (k < 1 || k > 64) return fail(ctx, ICPMI_ERR_ARG, "k %d outside [1,64]", k);
    if (n_qry > (int64_t)30000000 || n_tgt > (int64_t)700000000) return fail(ctx, ICPMI_ERR_ARG, "cloud too large");
    if (n_qry == 0) return ICPMI_OK;
    const int n = (int)n_qry, m = (int)n_tgt;
    if ((rc = reserve(ctx, ctx->stage_b, sizeof(double) * 3 * (size_t)n))) return rc;
    if ((rc = reserve(ctx, ctx->stage_c, sizeof(double) * 3 * (size_t)m))) return rc;
    if ((rc = reserve(ctx, ctx->knn_idx, sizeof(int) * (size_t)std::max(m, n) * k))) return rc;
    if (dist_sq && (rc = reserve(ctx, ctx->d2out, sizeof(double) * (size_t)n * k))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_b.p, queries_xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_c.p, targets_xyz, sizeof(double) * 3 * (size_t)m, hipMemcpyHostToDevice, s));
    // entries a list does not reach (k > n_tgt, a query with a NaN coordinate) stay -1 / +infinity
    HIP_TRY(ctx, hipMemsetAsync(ctx->knn_idx.p, 0xFF, sizeof(int) * (size_t)n * k, s));
    if ((rc = prepare_nn(ctx, (const double *)ctx->stage_c.p, m, n))) return rc;
    if ((rc = launch_knn(ctx, (const double *)ctx->stage_b.p, n, (const double *)ctx->stage_c.p, m, k, 0, n))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(indices, ctx->knn_idx.p, sizeof(int) * (size_t)n * k, hipMemcpyDeviceToHost, s));
    if (dist_sq) {
        for (size_t e = 0; e < (size_t)n * k; ++e) dist_sq[e] = __builtin_inf();
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d2out.p, dist_sq, sizeof(double) * (size_t)n * k, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_knn_distances, dim3((unsigned)(((size_t)n * k + 255) / 256)), dim3(256), 0, s,
                           (const double *)ctx->stage_b.p, n, (const double *)ctx->stage_c.p, m, k,
                           (const int *)ctx->knn_idx.p, (double *)ctx->d2out.p);
        HIP_TRY(ctx, hipMemcpyAsync(dist_sq, ctx->d2out.p, sizeof(double) * (size_t)n * k, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    HIP_TRY(ctx, hipGetLastError());
    harvest_profile(ctx);
    return ICPMI_OK;
}

int icpmi_solve_point_to_plane(icpmi_ctx *ctx, const double *source_xyz, const double *target_xyz,
                               const double *normals_xyz, int64_t n64, double transform_out[16])
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!source_xyz || !target_xyz || !normals_xyz || !transform_out)
        return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n64 <= 0) return fail(ctx, ICPMI_ERR_EMPTY_SOURCE, "empty correspondence set");
    if (n64 > (int64_t)700000000) return fail(ctx, ICPMI_ERR_ARG, "more than 7e8 correspondences");
    const int n = (int)n64;
    const size_t bytes = sizeof(double) * 3 * (size_t)n;
    if ((rc = reserve(ctx, ctx->stage_a, bytes))) return rc;
    if ((rc = reserve(ctx, ctx->stage_b, bytes))) return rc;
    if ((rc = reserve(ctx, ctx->stage_c, bytes))) return rc;
    const int rblocks = reduce_blocks(ctx, n);
    if ((rc = reserve(ctx, ctx->partials, sizeof(double) * kSumsStride * (size_t)rblocks))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_a.p, source_xyz, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_b.p, target_xyz, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_c.p, normals_xyz, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_state, 0, sizeof(IcpState), s));
    {
        StageTimer t(ctx, ST_REDUCE);
        hipLaunchKernelGGL(k_reduce, dim3(rblocks), dim3(256), 0, s, (const double *)ctx->stage_a.p, n,
                           (const double *)ctx->stage_b.p, n, (const double *)ctx->stage_c.p,
                           (const int *)nullptr, (double *)ctx->partials.p, (const IcpState *)nullptr);
        hipLaunchKernelGGL(k_finish_solve, dim3(1), dim3(kFinishThreads), 0, s, (const double *)ctx->partials.p,
                           rblocks, n, ctx->d_state);
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(IcpState), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    HIP_TRY(ctx, hipGetLastError());
    harvest_profile(ctx);
    memcpy(transform_out, ctx->h_state->delta, sizeof(double) * 16);
    return ICPMI_OK;
}

int icpmi_transform_points(icpmi_ctx *ctx, const double transform[16], const double *in_xyz,
                           int64_t n64, double *out_xyz)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!transform || !in_xyz || !out_xyz) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n64 < 0 || n64 > (int64_t)700000000) return fail(ctx, ICPMI_ERR_ARG, "n out of range");
    if (n64 == 0) return ICPMI_OK;
    const int n = (int)n64;
    const size_t bytes = sizeof(double) * 3 * (size_t)n;
    if ((rc = reserve(ctx, ctx->stage_a, bytes))) return rc;
    hipStream_t s = ctx->stream;
    memset(ctx->h_state, 0, sizeof(IcpState));
    memcpy(ctx->h_state->total, transform, sizeof(double) * 16);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(IcpState), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_a.p, in_xyz, bytes, hipMemcpyHostToDevice, s));
    {
        StageTimer t(ctx, ST_TRANSFORM);
        hipLaunchKernelGGL(k_transform, dim3(std::min(2048, (n + 255) / 256)), dim3(256), 0, s,
                           (const double *)ctx->stage_a.p, (double *)ctx->stage_a.p, n, ctx->d_state, 1, 0);
    }
    HIP_TRY(ctx, hipMemcpyAsync(out_xyz, ctx->stage_a.p, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    HIP_TRY(ctx, hipGetLastError());
    harvest_profile(ctx);
    return ICPMI_OK;
}

int icpmi_voxel_downsample_device(icpmi_ctx *ctx, const double *d_points_xyz, int64_t n, double voxel_size,
                                  double *d_out_xyz, int64_t out_cap, int64_t *n_out)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!d_points_xyz || !d_out_xyz || !n_out) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n < 0 || n > 700000000) return fail(ctx, ICPMI_ERR_ARG, "n out of range");
    *n_out = 0;
    if (n == 0) return ICPMI_OK;
    return voxel_downsample_device(ctx, d_points_xyz, (int)n, voxel_size, d_out_xyz, out_cap, n_out);
}

int icpmi_voxel_downsample(icpmi_ctx *ctx, const double *points_xyz, int64_t n, double voxel_size,
                           double *out_xyz, int64_t out_cap, int64_t *n_out)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!points_xyz || !out_xyz || !n_out) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n < 0 || n > 700000000) return fail(ctx, ICPMI_ERR_ARG, "n out of range");
    *n_out = 0;
    if (n == 0) return ICPMI_OK;
    const size_t bytes = sizeof(double) * 3 * (size_t)n;
    if ((rc = reserve(ctx, ctx->stage_a, bytes))) return rc;
    if ((rc = reserve(ctx, ctx->vox_out, bytes))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_a.p, points_xyz, bytes, hipMemcpyHostToDevice, ctx->stream));
    int64_t rows = 0;
    if ((rc = voxel_downsample_device(ctx, (const double *)ctx->stage_a.p, (int)n, voxel_size,
                                      (double *)ctx->vox_out.p, n, &rows)))
        return rc;
    if (rows > out_cap) return fail(ctx, ICPMI_ERR_CAPACITY, "output holds %lld rows, needs %lld", (long long)out_cap, (long long)rows);
    HIP_TRY(ctx, hipMemcpy(out_xyz, ctx->vox_out.p, sizeof(double) * 3 * (size_t)rows, hipMemcpyDeviceToHost));
    *n_out = rows;
    return ICPMI_OK;
}

// ---- on-disk formats: host-side, no device work ------------------
// KITTI .bin: x, y, z, intensity as float32, intensity dropped (file_utils.cpp:115-141).
// PLY: header parse as file_utils.cpp:31-60 (every `property` line counts towards the vertex
// stride, whatever element it belongs to -- like the reference); binary payload read as
// little-endian float32 at the x/y/z offsets whatever the declared type or endianness
// (file_utils.cpp:87-98); ASCII payload: first three numbers of each line (file_utils.cpp:100-104).
int icpmi_load_cloud(const char *path, double *out_xyz, int64_t cap, int64_t *n_out)
{
    if (!path || !n_out) return fail(nullptr, ICPMI_ERR_NULL, "null argument");
    *n_out = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(nullptr, ICPMI_ERR_ARG, "Cannot open file: %s", path); // file_utils.cpp:22-24
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
    const size_t len = strlen(path);
    if (len >= 4 && strcmp(path + len - 4, ".bin") == 0) {
        fseek(f, 0, SEEK_END);
        const long size = ftell(f);
        fseek(f, 0, SEEK_SET);
        const int64_t n = size / (4 * (long)sizeof(float)); // file_utils.cpp:127
        *n_out = n;
        if (!out_xyz) return ICPMI_OK;
        if (cap < n) return fail(nullptr, ICPMI_ERR_CAPACITY, "output holds %lld rows, needs %lld", (long long)cap, (long long)n);
        std::vector<float> buf(4 * 4096);
        for (int64_t i = 0; i < n;) {
            const size_t want = (size_t)std::min<int64_t>(4096, n - i);
            const size_t got = fread(buf.data(), 4 * sizeof(float), want, f);
            for (size_t k = 0; k < got; ++k)
                for (int a = 0; a < 3; ++a) out_xyz[3 * (i + k) + a] = (double)buf[4 * k + a];
            for (size_t k = got; k < want; ++k)
                for (int a = 0; a < 3; ++a) out_xyz[3 * (i + k) + a] = 0.0;
            i += want;
        }
        return ICPMI_OK;
    }
    // PLY header
    std::string line;
    auto getline = [&](std::string &s) -> bool {
        s.clear();
        int ch;
        while ((ch = fgetc(f)) != EOF) {
            if (ch == '\n') return true;
            s.push_back((char)ch);
        }
        return !s.empty();
    };
    long num_vertices = 0;
    bool is_binary = false;
    std::vector<std::pair<std::string, std::string>> props;
    while (getline(line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back(); // file_utils.cpp:34-36
        char tok[64] = "", a[64] = "", b[64] = "";
        const int nt = sscanf(line.c_str(), "%63s %63s %63s", tok, a, b);
        if (nt < 1) continue;
        if (!strcmp(tok, "format")) {
            if (!strcmp(a, "binary_little_endian") || !strcmp(a, "binary_big_endian")) is_binary = true;
        } else if (!strcmp(tok, "element")) {
            if (!strcmp(a, "vertex")) num_vertices = nt >= 3 ? atol(b) : 0;
        } else if (!strcmp(tok, "property")) {
            props.emplace_back(std::string(b), std::string(a)); // (name, dtype), file_utils.cpp:54-56
        } else if (!strcmp(tok, "end_header")) {
            break;
        }
    }
    auto type_size = [](const std::string &t) -> size_t { // file_utils.cpp:63-70
        if (t == "float" || t == "float32") return 4;
        if (t == "double" || t == "float64") return 8;
        if (t == "uchar" || t == "uint8" || t == "char" || t == "int8") return 1;
        if (t == "ushort" || t == "uint16" || t == "short" || t == "int16") return 2;
        if (t == "uint" || t == "uint32" || t == "int" || t == "int32") return 4;
        return 4;
    };
    size_t stride = 0, xo = 0, yo = 0, zo = 0;
    for (auto &pr : props) {
        if (pr.first == "x") xo = stride;
        else if (pr.first == "y") yo = stride;
        else if (pr.first == "z") zo = stride;
        stride += type_size(pr.second);
    }
    const int64_t n = num_vertices > 0 ? num_vertices : 0;
    *n_out = n;
    if (!out_xyz) return ICPMI_OK;
    if (cap < n) return fail(nullptr, ICPMI_ERR_CAPACITY, "output holds %lld rows, needs %lld", (long long)cap, (long long)n);
    if (is_binary) {
        if (stride < 4 || xo + 4 > stride || yo + 4 > stride || zo + 4 > stride)
            return fail(nullptr, ICPMI_ERR_ARG, "PLY vertex layout has no float x/y/z");
        std::vector<char> buf(stride);
        for (int64_t i = 0; i < n; ++i) {
            if (fread(buf.data(), 1, stride, f) != stride) memset(buf.data(), 0, stride);
            float v[3];
            memcpy(&v[0], buf.data() + xo, 4);
            memcpy(&v[1], buf.data() + yo, 4);
            memcpy(&v[2], buf.data() + zo, 4);
            for (int a2 = 0; a2 < 3; ++a2) out_xyz[3 * i + a2] = (double)v[a2];
        }
    } else {
        for (int64_t i = 0; i < n; ++i) {
            double v[3] = {0, 0, 0};
            if (getline(line)) sscanf(line.c_str(), "%lf %lf %lf", &v[0], &v[1], &v[2]);
            for (int a2 = 0; a2 < 3; ++a2) out_xyz[3 * i + a2] = v[a2];
        }
    }
    return ICPMI_OK;
}

// ---- frame files of a sequence directory (file_utils.cpp:203-247) ---------------------------------
// The reference's std::regex_search(filename, "(\\d+)\\.ply") finds the leftmost run of digits that is
// followed by the extension: a run followed by anything else can only fail (the greedy \d+ has
// nothing to give back that "\\." would match), so no regex engine is needed.
static bool frame_number(const char *name, const char *ext, long long *out)
{
    const size_t el = strlen(ext);
    for (size_t i = 0; name[i];) {
        if (name[i] < '0' || name[i] > '9') {
            ++i;
            continue;
        }
        size_t j = i;
        while (name[j] >= '0' && name[j] <= '9') ++j;
        if (strncmp(name + j, ext, el) == 0) {
            if (j - i > 18) return false; // std::stoll would throw std::out_of_range: not a frame
            long long v = 0;
            for (size_t k = i; k < j; ++k) v = v * 10 + (name[k] - '0');
            *out = v;
            return true;
        }
        i = j;
    }
    return false;
}

int icpmi_discover_frames(const char *data_dir, int64_t *stamps, int64_t frames_cap, char *paths, int64_t paths_cap,
                          int64_t *n_frames, int64_t *paths_bytes)
{
    if (!data_dir || !n_frames || !paths_bytes) return fail(nullptr, ICPMI_ERR_NULL, "null argument");
    *n_frames = 0;
    *paths_bytes = 0;
    if (!*data_dir) return fail(nullptr, ICPMI_ERR_ARG, "empty directory name");
    DIR *d = opendir(data_dir);
    if (!d) return fail(nullptr, ICPMI_ERR_ARG, "Cannot open directory: %s", data_dir); // fs::directory_iterator throws
    std::vector<std::pair<long long, std::string>> frames;
    while (struct dirent *e = readdir(d)) {
        const char *name = e->d_name;
        const char *dot = strrchr(name, '.');
        if (!dot || dot == name) continue; // path::extension() of ".ply" / "x" is empty
        long long stamp = -1;
        // file_utils.cpp:224-241: extension first, then the number in front of that extension's first occurrence
        if ((strcmp(dot, ".ply") == 0 && frame_number(name, ".ply", &stamp)) ||
            (strcmp(dot, ".bin") == 0 && frame_number(name, ".bin", &stamp)))
            frames.emplace_back(stamp, std::string(data_dir) + (data_dir[strlen(data_dir) - 1] == '/' ? "" : "/") + name);
    }
    closedir(d);
    // file_utils.cpp:244-245 sorts by number only (std::sort: equal numbers in unspecified order);
    // here equal numbers come out by path so that the result does not depend on the file system
    std::sort(frames.begin(), frames.end());
    int64_t bytes = 0;
    for (auto &f : frames) bytes += (int64_t)f.second.size() + 1;
    *n_frames = (int64_t)frames.size();
    *paths_bytes = bytes;
    if (!stamps && !paths) return ICPMI_OK; // sizing call
    if (frames_cap < *n_frames || paths_cap < bytes)
        return fail(nullptr, ICPMI_ERR_CAPACITY, "%lld frames / %lld path bytes do not fit", (long long)*n_frames, (long long)bytes);
    char *w = paths;
    for (size_t i = 0; i < frames.size(); ++i) {
        if (stamps) stamps[i] = frames[i].first;
        if (paths) {
            memcpy(w, frames[i].second.c_str(), frames[i].second.size() + 1);
            w += frames[i].second.size() + 1;
        }
    }
    return ICPMI_OK;
}

int icpmi_upload_points_f32(icpmi_ctx *ctx, const float *records, int64_t n, int32_t stride_floats, double *d_out_xyz)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!records || !d_out_xyz) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n < 0 || n > 700000000) return fail(ctx, ICPMI_ERR_ARG, "n out of range");
    if (stride_floats < 3 || stride_floats > 64) return fail(ctx, ICPMI_ERR_ARG, "stride %d outside [3,64]", stride_floats);
    if (n == 0) return ICPMI_OK;
    const size_t bytes = sizeof(float) * (size_t)stride_floats * (size_t)n;
    if ((rc = reserve(ctx, ctx->f32_stage, bytes))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->f32_stage.p, records, bytes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)ctx->f32_stage.p, (int)n,
                       (int)stride_floats, d_out_xyz);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return ICPMI_OK;
}

int icpmi_load_cloud_device(icpmi_ctx *ctx, const char *path, double *d_out_xyz, int64_t cap, int64_t *n_out)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!path || !n_out) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    *n_out = 0;
    const size_t len = strlen(path);
    if (len >= 4 && strcmp(path + len - 4, ".bin") == 0) {
        FILE *f = fopen(path, "rb");
        if (!f) return fail(ctx, ICPMI_ERR_ARG, "Cannot open file: %s", path); // file_utils.cpp:116-118
        struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
        fseek(f, 0, SEEK_END);
        const long size = ftell(f);
        fseek(f, 0, SEEK_SET);
        const int64_t n = size / (4 * (long)sizeof(float)); // file_utils.cpp:127
        *n_out = n;
        if (!d_out_xyz) return ICPMI_OK;
        if (cap < n) return fail(ctx, ICPMI_ERR_CAPACITY, "output holds %lld rows, needs %lld", (long long)cap, (long long)n);
        if (n == 0) return ICPMI_OK;
        std::vector<float> rec(4 * (size_t)n, 0.f); // (a short read leaves zeros, like the host loader)
        if (fread(rec.data(), 4 * sizeof(float), (size_t)n, f) != (size_t)n) { /* zeros stay */ }
        return icpmi_upload_points_f32(ctx, rec.data(), n, 4, d_out_xyz);
    }
    // PLY: header rules and the ASCII payload are host work (icpmi_load_cloud); fp64 rows go up as they are
    int64_t n = 0;
    if ((rc = icpmi_load_cloud(path, nullptr, 0, &n))) return fail(ctx, rc, "%s", icpmi_last_error(nullptr));
    *n_out = n;
    if (!d_out_xyz) return ICPMI_OK;
    if (cap < n) return fail(ctx, ICPMI_ERR_CAPACITY, "output holds %lld rows, needs %lld", (long long)cap, (long long)n);
    if (n == 0) return ICPMI_OK;
    std::vector<double> host(3 * (size_t)n);
    if ((rc = icpmi_load_cloud(path, host.data(), n, &n))) return fail(ctx, rc, "%s", icpmi_last_error(nullptr));
    HIP_TRY(ctx, hipMemcpy(d_out_xyz, host.data(), sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice));
    return ICPMI_OK;
}

int icpmi_estimate_normals_rows(icpmi_ctx *ctx, const double *points_xyz, int64_t n, int32_t k, int64_t row0,
                                int64_t row1, double *normals_xyz)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!points_xyz || !normals_xyz) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n <= 0) return fail(ctx, ICPMI_ERR_EMPTY_TARGET, "empty cloud");
    if (n > (int64_t)700000000) return fail(ctx, ICPMI_ERR_ARG, "cloud larger than 7e8 points");
    if (k < 1 || k > 64) return fail(ctx, ICPMI_ERR_ARG, "k %d outside [1,64]", k);
    if (row0 < 0 || row1 < row0 || row1 > n) return fail(ctx, ICPMI_ERR_ARG, "rows [%lld,%lld) outside [0,%lld)", (long long)row0, (long long)row1, (long long)n);
    if (row1 == row0) return ICPMI_OK;
    const int m = (int)n, rows = (int)(row1 - row0);
    if ((rc = reserve(ctx, ctx->stage_c, sizeof(double) * 3 * (size_t)m))) return rc;
    if ((rc = reserve(ctx, ctx->nrm, sizeof(double) * 3 * (size_t)m))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_c.p, points_xyz, sizeof(double) * 3 * (size_t)m, hipMemcpyHostToDevice, s));
    // a slice is always taken in point order (what a rank of a caller-sharded job wants back)
    const bool keep_pruned = ctx->nn_pruned;
    if ((rc = prepare_nn(ctx, (const double *)ctx->stage_c.p, m, m))) return rc;
    ctx->nn_pruned = false;
    rc = launch_normals(ctx, (const double *)ctx->stage_c.p, m, k, (int)row0, (int)row1, (double *)ctx->nrm.p, false, true);
    ctx->nn_pruned = keep_pruned;
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(normals_xyz, (const double *)ctx->nrm.p + 3 * (size_t)row0, sizeof(double) * 3 * (size_t)rows,
                                hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    harvest_profile(ctx);
    return ICPMI_OK;
}

// ---- odometry stream: the registration part of SlamNode::process_frame (slam_node.cpp:122-152) ---------
int icpmi_stream_reset(icpmi_ctx *ctx)
{
    if (!ctx) return ICPMI_ERR_NULL;
    ctx->stream_prev_n = -1;
    ctx->prep_valid = false;
    if (ctx->prep_helper) ctx->prep_helper->prep_valid = false; // (a preparation still in flight is waited for by the next push)
    return ICPMI_OK;
}

namespace {
int stream_check_args(icpmi_ctx *ctx, const icpmi_config *cfg, icpmi_result *result, double *error_history, int32_t history_cap,
                      icpmi_stream_info *info)
{
    if (!cfg || !result || !error_history || !info) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (cfg->max_iterations < 0) return fail(ctx, ICPMI_ERR_ARG, "max_iterations < 0");
    if (history_cap < cfg->max_iterations + 1)
        return fail(ctx, ICPMI_ERR_CAPACITY, "error_history holds %d entries, needs %d", history_cap, cfg->max_iterations + 1);
    memset(result, 0, sizeof(*result));
    for (int i = 0; i < 16; ++i) result->transformation[i] = (i % 5 == 0) ? 1.0 : 0.0;
    memset(info, 0, sizeof(*info));
    return ICPMI_OK;
}
int stream_register(icpmi_ctx *ctx, int64_t n_cur, int64_t min_points, const icpmi_config *cfg, icpmi_result *result,
                    double *error_history, int32_t history_cap, icpmi_stream_info *info);
} // namespace

int icpmi_stream_push(icpmi_ctx *ctx, const double *d_raw_xyz, int64_t n_raw, double voxel_size, int64_t min_points,
                      const icpmi_config *cfg, icpmi_result *result, double *error_history, int32_t history_cap,
                      icpmi_stream_info *info)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!d_raw_xyz) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n_raw <= 0 || n_raw > 700000000) return fail(ctx, ICPMI_ERR_ARG, "n_raw out of range");
    if ((rc = stream_check_args(ctx, cfg, result, error_history, history_cap, info))) return rc;
    if (ctx->prefetch) { // what the worker filters the next file with
        std::lock_guard<std::mutex> lk(ctx->prefetch->mu);
        ctx->prefetch->voxel_hint = voxel_size;
    }
    // curr = voxel_downsample(raw)  (slam_node.cpp:122), into the buffer that is not the previous frame
    if ((rc = reserve(ctx, ctx->stream_cur, sizeof(double) * 3 * (size_t)n_raw))) return rc;
    int64_t n_cur = 0;
    if ((rc = voxel_downsample_device(ctx, d_raw_xyz, (int)n_raw, voxel_size, (double *)ctx->stream_cur.p, n_raw, &n_cur, false))) return rc;
    return stream_register(ctx, n_cur, min_points, cfg, result, error_history, history_cap, info);
}

namespace {
// everything of a push behind the filter: ctx->stream_cur holds the n_cur rows of the filtered scan
int stream_register(icpmi_ctx *ctx, int64_t n_cur, int64_t min_points, const icpmi_config *cfg, icpmi_result *result,
                    double *error_history, int32_t history_cap, icpmi_stream_info *info)
{
    int rc;
    info->n_filtered = n_cur;
    info->n_target = ctx->stream_prev_n < 0 ? 0 : ctx->stream_prev_n;
    ctx->stream_cur_n = n_cur;
    // The scan just filtered is the NEXT push's target (slam_node.cpp:128,152): its search structure and
    // normals (Morton sort, split frames, operand packing, 20-NN, PCA: ~135 us of small kernels) are built on a
    // helper context -- a stream and workspace of their own -- BESIDE this push's registration, queued once the
    // registration's first iterations are (the host's launch time for them then hides behind the device's
    // work), and the next push adopts them by swapping buffers behind an event: this context's stream carries the
    // registration only.  (Round 2 queued the preparation on this stream behind the result: registration and
    // preparation in a row were the 0.31 ms of a frame.)
    const bool early = ctx->opt.profile == 0 && !(ctx->comm != nullptr || ctx->cb_allreduce != nullptr);
    if (early && !ctx->prep_helper) {
        icpmi_options o = ctx->opt;
        icpmi_ctx *h = nullptr;
        if (icpmi_create(&o, &h) == ICPMI_OK) ctx->prep_helper = h;
        if (ctx->prep_helper && !ctx->prep_done) (void)hipEventCreateWithFlags(&ctx->prep_done, hipEventDisableTiming);
        if (ctx->prep_helper && !ctx->scan_ready) (void)hipEventCreateWithFlags(&ctx->scan_ready, hipEventDisableTiming);
    }
    icpmi_ctx *helper = early && ctx->prep_done && ctx->scan_ready ? ctx->prep_helper : nullptr;
    // everything that produced the scan on this context's stream is queued by now: the helper's stream waits for it
    if (helper && n_cur > 0) HIP_TRY(ctx, hipEventRecord(ctx->scan_ready, ctx->stream));
    auto prepare_next = [](icpmi_ctx *c) -> int {
        icpmi_ctx *h = c->prep_helper;
        if (!h || c->stream_cur_n <= 0) return ICPMI_OK;
        struct Clock { icpmi_ctx *c; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                       ~Clock() { c->t_prep_queue += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } clock{c};
        if (hipStreamWaitEvent(h->stream, c->scan_ready, 0) != hipSuccess) return ICPMI_ERR_HIP;
        const int r = prepare_target(h, (const double *)c->stream_cur.p, (int)c->stream_cur_n, (int)c->stream_cur_n);
        c->helper_busy = true; // (whatever was queued reads the scan: somebody has to wait for it before the scan's buffer is reused)
        if (r != ICPMI_OK || hipEventRecord(c->prep_done, h->stream) != hipSuccess) {
            h->prep_valid = false;
            (void)hipStreamSynchronize(h->stream);
            c->helper_busy = false;
            return ICPMI_ERR_HIP;
        }
        return ICPMI_OK;
    };
    // the target of THIS registration: prepared by the helper during the previous push?  Adopt it.
    if (helper && helper->prep_valid && ctx->stream_prev_n > 0 && helper->prep_tgt == (const double *)ctx->stream_prev.p &&
        helper->prep_m == (int)ctx->stream_prev_n) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->prep_done, 0));
        adopt_target(ctx, helper);
        ctx->helper_busy = false; // (this push's registration runs behind it, and the push waits for the registration)
    } else if (ctx->helper_busy) {
        // a preparation nobody is going to use (another scan was pushed in between, a stream reset): it still reads a
        // scan buffer of this stream -- wait for it before anything is queued that could write there
        HIP_TRY(ctx, hipEventSynchronize(ctx->prep_done));
        ctx->helper_busy = false;
        if (helper) helper->prep_valid = false;
    }
    bool queued_next = false;
    if (ctx->stream_prev_n < 0) {
        info->status = ICPMI_STREAM_FIRST_FRAME;          // slam_node.cpp:69-72: nothing to register against yet
    } else if (n_cur < min_points || n_cur <= 0 || ctx->stream_prev_n <= 0) { // (an empty cloud on either side is UB in the reference)
        info->status = ICPMI_STREAM_TOO_FEW_POINTS;       // slam_node.cpp:125-130: the caller repeats its last pose
    } else {
        // source = curr, target = prev (slam_node.cpp:132-133): both already in HBM; the target's
        // search structure and normals are the adopted ones (prepared during the previous push, as a rule)
        info->status = ICPMI_STREAM_REGISTERED;
        queued_next = helper != nullptr;
        static const bool span = getenv("ICPMI_STREAM_STATS") != nullptr;
        static hipEvent_t e0 = nullptr, e1 = nullptr;
        if (span && !e0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
        if (span) (void)hipEventRecord(e0, ctx->stream);
        const auto ta = std::chrono::steady_clock::now();
        rc = align_device(ctx, (const double *)ctx->stream_cur.p, n_cur, (const double *)ctx->stream_prev.p,
                          ctx->stream_prev_n, cfg, result, error_history, history_cap, nullptr,
                          helper ? +prepare_next : nullptr);
        ctx->t_align += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count();
        if (span) {
            (void)hipEventRecord(e1, ctx->stream);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ctx->t_gpu_span += ms;
        }
        if (rc) {
            //  after_first may have queued the preparation of stream_cur on the helper: the buffers are not
            // swapped on this path, so the NEXT push filters into stream_cur again -- not under the helper's reads
            if (ctx->helper_busy) {
                (void)hipEventSynchronize(ctx->prep_done);
                ctx->helper_busy = false;
                if (helper) helper->prep_valid = false;
            }
            return rc;
        }
    }
    if (!queued_next && helper) (void)prepare_next(ctx); // (a failure only means the next push prepares its target itself)
    std::swap(ctx->stream_prev, ctx->stream_cur);         // prev_points_ = curr (slam_node.cpp:128,152), no copy
    ctx->stream_prev_n = n_cur;
    return ICPMI_OK;
}
} // namespace

namespace {
bool is_bin(const char *path)
{
    const size_t len = strlen(path);
    return len >= 4 && strcmp(path + len - 4, ".bin") == 0;
}
void prefetch_worker(FilePrefetch *pf)
{
    (void)hipSetDevice(pf->device);
    std::unique_lock<std::mutex> lk(pf->mu);
    for (;;) {
        pf->cv.wait(lk, [pf] { return pf->quit || pf->busy; });
        if (pf->quit) return;
        const std::string path = pf->want;
        // the slot to fill: an empty one, else the older of two files that were prefetched but never pushed.  A push
        // may have read the slot: its kernels are behind the event it recorded (another stream than this thread's)
        const int slot = pf->taking >= 0 ? 1 - pf->taking
                                         : (pf->ready[0].empty() ? 0 : (pf->ready[1].empty() ? 1 : (pf->ready_seq[0] < pf->ready_seq[1] ? 0 : 1)));
        pf->ready[slot].clear();
        const bool wait_for_push = pf->slot_used[slot];
        const double voxel = pf->voxel_hint;
        lk.unlock();
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double ta = now();
        if (wait_for_push) (void)hipEventSynchronize(pf->slot_done[slot]);
        const double tb = now();
        double tc = tb, td = tb, te = tb;
        bool ok = false;
        int64_t n = 0;
        if (FILE *f = fopen(path.c_str(), "rb")) {
            fseek(f, 0, SEEK_END);
            const long size = ftell(f);
            fseek(f, 0, SEEK_SET);
            n = size / (4 * (long)sizeof(float)); // file_utils.cpp:127
            if (n > 0 && n <= 700000000) {
                const size_t bytes = 4 * sizeof(float) * (size_t)n;
                if (pf->h_cap[slot] < bytes) {
                    // (the copies out of this slot's staging buffer finished long ago: the slot's previous file has been pushed)
                    if (pf->h_buf[slot]) (void)hipHostFree(pf->h_buf[slot]);
                    pf->h_buf[slot] = nullptr;
                    pf->h_cap[slot] = 0;
                    if (hipHostMalloc(&pf->h_buf[slot], bytes + bytes / 4, hipHostMallocDefault) == hipSuccess) pf->h_cap[slot] = bytes + bytes / 4;
                }
                if (!pf->h_runs && hipHostMalloc((void **)&pf->h_runs, 4 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) pf->h_runs = nullptr;
                if (pf->d_cap[slot] < (size_t)n) {
                    if (pf->stream) (void)hipStreamSynchronize(pf->stream);
                    if (pf->d_f32[slot]) (void)hipFree(pf->d_f32[slot]);
                    if (pf->d_raw[slot]) (void)hipFree(pf->d_raw[slot]);
                    pf->d_f32[slot] = nullptr;
                    pf->d_raw[slot] = nullptr;
                    pf->d_cap[slot] = 0;
                    const size_t cap = (size_t)n + (size_t)n / 4;
                    if (hipMalloc(&pf->d_f32[slot], 4 * sizeof(float) * cap) == hipSuccess &&
                        hipMalloc((void **)&pf->d_raw[slot], 3 * sizeof(double) * cap) == hipSuccess)
                        pf->d_cap[slot] = cap;
                }
                if (!pf->stream) (void)hipStreamCreateWithFlags(&pf->stream, hipStreamNonBlocking);
                if (pf->stream && !pf->filled[slot]) (void)hipEventCreateWithFlags(&pf->filled[slot], hipEventDisableTiming);
                if (pf->h_cap[slot] >= bytes && pf->d_cap[slot] >= (size_t)n && pf->stream && pf->filled[slot] && pf->h_runs) {
                    // the file in two parts, the first one on its way to the device while the second is read; then the
                    // widening and the filter are queued behind the copies and an event behind them all: this thread
                    // waits for nothing of it
                    char *h = (char *)pf->h_buf[slot];
                    const size_t half = (bytes / 2) / 16 * 16;
                    const size_t got = fread(h, 1, half, f);
                    if (got < half) memset(h + got, 0, bytes - got); // a short read leaves zeros, like the synchronous path
                    bool up = hipMemcpyAsync(pf->d_f32[slot], h, half, hipMemcpyHostToDevice, pf->stream) == hipSuccess;
                    if (got == half) {
                        const size_t got2 = fread(h + half, 1, bytes - half, f);
                        if (got2 < bytes - half) memset(h + half + got2, 0, bytes - half - got2);
                    }
                    tc = now();
                    up = up && hipMemcpyAsync((char *)pf->d_f32[slot] + half, h + half, bytes - half, hipMemcpyHostToDevice, pf->stream) == hipSuccess;
                    if (up) {
                        hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pf->stream,
                                           (const float *)pf->d_f32[slot], (int)n, 4, pf->d_raw[slot]);
                        td = now();
                        pf->filtered_voxel[slot] = 0.0;
                        pf->pend_filtered[slot] = false;
                        if (voxel > 0.0 && reserve_raw(pf->filtered[slot], sizeof(double) * 3 * (size_t)n) == hipSuccess) {
                            const VoxelScratch vs{&pf->sc_bbox, &pf->sc_box, &pf->sc_keys, &pf->sc_vals, &pf->sc_tmp, 0};
                            const char *msg = "";
                            int64_t unused = 0;
                            pf->h_runs[2 * slot] = 0u;
                            pf->h_runs[2 * slot + 1] = 1u; // (reads as "could not be keyed" until the filter's own words arrive)
                            pf->pend_filtered[slot] = voxel_filter_core(pf->stream, vs, pf->d_raw[slot], (int)n, voxel, (double *)pf->filtered[slot].p, n,
                                                                        &unused, true, &msg, true, pf->h_runs + 2 * slot) == ICPMI_OK;
                            pf->pend_voxel[slot] = voxel;
                            // (a filter that could not be queued, or whose outcome says the grid cannot be keyed: the push
                            // filters the raw points itself and reports what is wrong with them)
                        }
                        ok = hipEventRecord(pf->filled[slot], pf->stream) == hipSuccess && hipGetLastError() == hipSuccess;
                    }
                }
            }
            fclose(f);
        }
        te = now();
        pf->t_wait += tb - ta, pf->t_read += tc - tb, pf->t_upload += td - tc, pf->t_filter += te - td, pf->files += 1;
        lk.lock();
        if (ok) {
            pf->ready[slot] = path;
            pf->ready_n[slot] = n;
            pf->ready_seq[slot] = ++pf->seq;
            pf->pending[slot] = true; // (the push waits for `filled[slot]`)
        }
        pf->want.clear();
        pf->busy = false;
        pf->cv.notify_all();
    }
}
} // namespace

int icpmi_stream_prefetch_file(icpmi_ctx *ctx, const char *path)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!path) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (!is_bin(path)) return ICPMI_OK; // a PLY goes through the host parser when it is pushed: nothing to do ahead
    if (!ctx->prefetch) {
        ctx->prefetch = new FilePrefetch();
        ctx->prefetch->device = ctx->opt.device;
        ctx->prefetch->worker = std::thread(prefetch_worker, ctx->prefetch);
    }
    FilePrefetch *pf = ctx->prefetch;
    {
        std::unique_lock<std::mutex> lk(pf->mu);
        if (pf->ready[0] == path || pf->ready[1] == path || (pf->busy && pf->want == path)) return ICPMI_OK; // already there / on its way
        pf->cv.wait(lk, [pf] { return !pf->busy; }); // one read at a time
        pf->want = path;
        pf->busy = true;
    }
    pf->cv.notify_all();
    return ICPMI_OK;
}

int icpmi_stream_push_file(icpmi_ctx *ctx, const char *path, double voxel_size, int64_t min_points, const icpmi_config *cfg,
                           icpmi_result *result, double *error_history, int32_t history_cap, icpmi_stream_info *info)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!path) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    struct PushClock { icpmi_ctx *c; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                       ~PushClock() { c->t_push += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); c->pushes += 1; } } push_clock{ctx};
    if (!is_bin(path)) {
        // PLY: host parser (header rules, ASCII numbers), then the host-pointer form
        int64_t n = 0;
        if ((rc = icpmi_load_cloud(path, nullptr, 0, &n))) return fail(ctx, rc, "%s", icpmi_last_error(nullptr));
        if (n <= 0) return fail(ctx, ICPMI_ERR_ARG, "no points in %s", path);
        std::vector<double> host(3 * (size_t)n);
        if ((rc = icpmi_load_cloud(path, host.data(), n, &n))) return fail(ctx, rc, "%s", icpmi_last_error(nullptr));
        return icpmi_stream_push_host(ctx, host.data(), n, voxel_size, min_points, cfg, result, error_history, history_cap, info);
    }
    // KITTI .bin: the file is read into pinned memory and everything behind it -- copy, widening,
    // voxel filter, registration -- is queued on the context's stream without a wait in between
    if (FilePrefetch *pf = ctx->prefetch) { // already on the device (or on its way) thanks to the worker?
        std::unique_lock<std::mutex> lk(pf->mu);
        if (pf->busy && pf->want == path) pf->cv.wait(lk, [pf] { return !pf->busy; });
        const int slot = pf->ready[0] == path ? 0 : (pf->ready[1] == path ? 1 : -1);
        if (slot >= 0) {
            const int64_t n = pf->ready_n[slot];
            pf->taking = slot; // the worker fills the other one meanwhile
            lk.unlock();
            if (pf->pending[slot]) {
                // the worker announced the file with its copies, the widening and the filter still queued: wait for them here,
                // then look at what the filter left in pinned memory
                pf->pending[slot] = false;
                const auto tw = std::chrono::steady_clock::now();
                const bool arrived = hipEventSynchronize(pf->filled[slot]) == hipSuccess;
                ctx->t_wait_file += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
                if (arrived && pf->pend_filtered[slot] && pf->h_runs[2 * slot + 1] == 0u && (int64_t)pf->h_runs[2 * slot] <= n) {
                    pf->filtered_n[slot] = (int64_t)pf->h_runs[2 * slot];
                    pf->filtered_voxel[slot] = pf->pend_voxel[slot];
                }
                if (!arrived) { // the slot's contents cannot be trusted: the synchronous path below reads the file itself
                    (void)hipGetLastError();
                    lk.lock();
                    pf->ready[slot].clear();
                    pf->slot_used[slot] = false;
                    pf->taking = -1;
                    lk.unlock();
                    goto read_it_here;
                }
            }
            int rc2;
            if (pf->filtered_voxel[slot] > 0.0 && pf->filtered_voxel[slot] == voxel_size) {
                // the worker filtered it too: its buffer becomes the current scan (ours goes to the worker in exchange;
                // it holds the scan before the previous one, which nothing reads any more).  The arguments are
                // checked before anything is taken, and a registration that fails hands the buffer back: the
                // frame stays in its slot and a second push of the path finds it.
                if ((rc2 = stream_check_args(ctx, cfg, result, error_history, history_cap, info)) == ICPMI_OK) {
                    std::swap(ctx->stream_cur, pf->filtered[slot]);
                    rc2 = stream_register(ctx, pf->filtered_n[slot], min_points, cfg, result, error_history, history_cap, info);
                    if (rc2 != ICPMI_OK) std::swap(ctx->stream_cur, pf->filtered[slot]);
                }
                std::lock_guard<std::mutex> lk2(pf->mu);
                pf->voxel_hint = voxel_size;
            } else {
                rc2 = icpmi_stream_push(ctx, pf->d_raw[slot], n, voxel_size, min_points, cfg, result, error_history, history_cap, info);
            }
            if (!pf->slot_done[slot]) (void)hipEventCreateWithFlags(&pf->slot_done[slot], hipEventDisableTiming);
            const bool recorded = pf->slot_done[slot] && hipEventRecord(pf->slot_done[slot], ctx->stream) == hipSuccess;
            if (!recorded) (void)hipStreamSynchronize(ctx->stream); // (then the slot is simply free)
            lk.lock();
            pf->slot_used[slot] = recorded;
            if (rc2 == ICPMI_OK) pf->ready[slot].clear(); // free for the worker, behind the event (a failed push leaves the frame there)
            pf->taking = -1;
            lk.unlock();
            return rc2;
        }
        // not there (never asked for, or the worker could not read it): the synchronous path reads it and reports
    }
read_it_here:
    FILE *f = fopen(path, "rb");
    if (!f) return fail(ctx, ICPMI_ERR_ARG, "Cannot open file: %s", path); // file_utils.cpp:116-118
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    const int64_t n = size / (4 * (long)sizeof(float)); // file_utils.cpp:127
    if (n <= 0 || n > 700000000) return fail(ctx, ICPMI_ERR_ARG, "%lld points in %s", (long long)n, path);
    const size_t bytes = 4 * sizeof(float) * (size_t)n;
    if (ctx->h_file_cap < bytes) {
        if (ctx->h_file) (void)hipHostFree(ctx->h_file);
        ctx->h_file = nullptr;
        ctx->h_file_cap = 0;
        HIP_TRY(ctx, hipHostMalloc(&ctx->h_file, bytes + bytes / 4, hipHostMallocDefault));
        ctx->h_file_cap = bytes + bytes / 4;
    }
    // (the previous frame's copy out of this buffer finished long ago: every push ends with a stream wait)
    const size_t got = fread(ctx->h_file, 1, bytes, f);
    if (got < bytes) memset((char *)ctx->h_file + got, 0, bytes - got); // a short read leaves zeros, like the host loader
    if ((rc = reserve(ctx, ctx->f32_stage, bytes))) return rc;
    if ((rc = reserve(ctx, ctx->stage_a, sizeof(double) * 3 * (size_t)n))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->f32_stage.p, ctx->h_file, bytes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)ctx->f32_stage.p, (int)n, 4,
                       (double *)ctx->stage_a.p);
    HIP_TRY(ctx, hipGetLastError());
    return icpmi_stream_push(ctx, (const double *)ctx->stage_a.p, n, voxel_size, min_points, cfg, result, error_history,
                             history_cap, info);
}

int icpmi_stream_push_host(icpmi_ctx *ctx, const double *raw_xyz, int64_t n_raw, double voxel_size, int64_t min_points,
                           const icpmi_config *cfg, icpmi_result *result, double *error_history, int32_t history_cap,
                           icpmi_stream_info *info)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!raw_xyz) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n_raw <= 0 || n_raw > 700000000) return fail(ctx, ICPMI_ERR_ARG, "n_raw out of range");
    if ((rc = reserve(ctx, ctx->stage_a, sizeof(double) * 3 * (size_t)n_raw))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_a.p, raw_xyz, sizeof(double) * 3 * (size_t)n_raw, hipMemcpyHostToDevice, ctx->stream));
    return icpmi_stream_push(ctx, (const double *)ctx->stage_a.p, n_raw, voxel_size, min_points, cfg, result, error_history,
                             history_cap, info);
}

// ---- the map side of process_frame (slam_node.cpp:147-153, :211-221) -------------------------------------
void icpmi_grid_config_default(icpmi_grid_config *grid)
{
    if (!grid) return;
    grid->resolution = 0.2;   // slam_node.hpp:36-39
    grid->height_min = 0.3;
    grid->height_max = 2.0;
    grid->max_range = 40.0;
}

int icpmi_occupancy_update_device(icpmi_ctx *ctx, const double *d_world_xyz, int64_t n, const double sensor_xyz[3],
                                  const icpmi_grid_config *grid, int64_t *n_cells)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if ((!d_world_xyz && n != 0) || !sensor_xyz || !grid) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n < 0 || n > 700000000) return fail(ctx, ICPMI_ERR_ARG, "n out of range");
    Range range("icpmi:occupancy_update");
    if ((rc = grid_update_queue(ctx, d_world_xyz, (int)n, sensor_xyz, grid))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    grid_finish(ctx);
    if (n_cells) *n_cells = ctx->grid_n;
    return ICPMI_OK;
}

int icpmi_occupancy_update(icpmi_ctx *ctx, const double *world_xyz, int64_t n, const double sensor_xyz[3],
                           const icpmi_grid_config *grid, int64_t *n_cells)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if ((!world_xyz && n != 0) || !sensor_xyz || !grid) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n < 0 || n > 700000000) return fail(ctx, ICPMI_ERR_ARG, "n out of range");
    if ((rc = reserve(ctx, ctx->world, sizeof(double) * 3 * (size_t)std::max<int64_t>(n, 1)))) return rc;
    if (n > 0)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->world.p, world_xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    return icpmi_occupancy_update_device(ctx, (const double *)ctx->world.p, n, sensor_xyz, grid, n_cells);
}

int icpmi_occupancy_cells(icpmi_ctx *ctx, int32_t *cells_xy, int64_t cap_cells, int64_t *n_cells)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!n_cells) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    *n_cells = ctx->grid_n;
    if (!cells_xy || ctx->grid_n == 0) return ICPMI_OK;
    if (cap_cells < ctx->grid_n)
        return fail(ctx, ICPMI_ERR_CAPACITY, "output holds %lld cells, needs %lld", (long long)cap_cells, (long long)ctx->grid_n);
    const int n = (int)ctx->grid_n;
    if ((rc = reserve(ctx, ctx->grid_out, sizeof(int) * 2 * (size_t)n))) return rc;
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(k_grid_decode, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned long long *)ctx->grid_set.p, n,
                       (int *)ctx->grid_out.p);
    HIP_TRY(ctx, hipMemcpyAsync(cells_xy, ctx->grid_out.p, sizeof(int) * 2 * (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    HIP_TRY(ctx, hipGetLastError());
    return ICPMI_OK;
}

int icpmi_occupancy_clear(icpmi_ctx *ctx)
{
    if (!ctx) return ICPMI_ERR_NULL;
    ctx->grid_n = 0;
    return ICPMI_OK;
}

int icpmi_stream_current_scan(icpmi_ctx *ctx, double *out_xyz, int64_t cap, int64_t *n_out)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!n_out) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (ctx->stream_prev_n < 0) return fail(ctx, ICPMI_ERR_ARG, "no resident frame: call icpmi_stream_push first");
    *n_out = ctx->stream_prev_n;
    if (!out_xyz || ctx->stream_prev_n == 0) return ICPMI_OK;
    if (cap < ctx->stream_prev_n)
        return fail(ctx, ICPMI_ERR_CAPACITY, "output holds %lld rows, needs %lld", (long long)cap, (long long)ctx->stream_prev_n);
    HIP_TRY(ctx, hipMemcpyAsync(out_xyz, ctx->stream_prev.p, sizeof(double) * 3 * (size_t)ctx->stream_prev_n, hipMemcpyDeviceToHost,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ICPMI_OK;
}

int icpmi_stream_map_update(icpmi_ctx *ctx, const double pose[16], const icpmi_grid_config *grid, double *world_out,
                            int64_t world_cap, int64_t *n_world, int64_t *n_cells)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!pose) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (ctx->stream_prev_n < 0) return fail(ctx, ICPMI_ERR_ARG, "no resident frame: call icpmi_stream_push first");
    const int n = (int)ctx->stream_prev_n;
    if (n_world) *n_world = n;
    if (world_out && world_cap < n)
        return fail(ctx, ICPMI_ERR_CAPACITY, "world_out holds %lld rows, needs %d", (long long)world_cap, n);
    Range range("icpmi:map_update");
    hipStream_t s = ctx->stream;
    if ((rc = reserve(ctx, ctx->world, sizeof(double) * 3 * (size_t)std::max(n, 1)))) return rc;
    if (n > 0) {
        // world = curr * R^T + t^T (slam_node.cpp:147), the resident filtered scan -> ctx->world
        memset(ctx->h_state, 0, sizeof(IcpState));
        memcpy(ctx->h_state->total, pose, sizeof(double) * 16);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(IcpState), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_transform, dim3(std::min(2048, (n + 255) / 256)), dim3(256), 0, s, (const double *)ctx->stream_prev.p,
                           (double *)ctx->world.p, n, ctx->d_state, 1, 0);
        if (world_out)
            HIP_TRY(ctx, hipMemcpyAsync(world_out, ctx->world.p, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, s));
    }
    if (grid) {
        const double sensor[3] = {pose[3], pose[7], pose[11]}; // new_pose.t() (slam_node.cpp:153)
        if ((rc = grid_update_queue(ctx, (const double *)ctx->world.p, n, sensor, grid))) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));   // the call's one wait: world points and the set's size
    HIP_TRY(ctx, hipGetLastError());
    if (grid) grid_finish(ctx);
    if (n_cells) *n_cells = ctx->grid_n;
    return ICPMI_OK;
}

int icpmi_scan_context(icpmi_ctx *ctx, const double *cloud_xyz, int64_t n, double *desc_out)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!cloud_xyz || !desc_out) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (n < 0 || n > 700000000) return fail(ctx, ICPMI_ERR_ARG, "n out of range");
    const size_t bytes = sizeof(double) * 3 * (size_t)std::max<int64_t>(n, 1);
    if ((rc = reserve(ctx, ctx->stage_a, bytes))) return rc;
    if ((rc = reserve(ctx, ctx->vox_out, sizeof(double) * kScCells))) return rc;
    hipStream_t s = ctx->stream;
    if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_a.p, cloud_xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_scan_context, dim3(1), dim3(1024), 0, s, (const double *)ctx->stage_a.p, (int)n,
                       (double *)ctx->vox_out.p);
    HIP_TRY(ctx, hipMemcpyAsync(desc_out, ctx->vox_out.p, sizeof(double) * kScCells, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    HIP_TRY(ctx, hipGetLastError());
    return ICPMI_OK;
}

int icpmi_scan_context_distances(icpmi_ctx *ctx, const double *query_desc, const double *hist_descs,
                                 int64_t count, double *dist_out)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!query_desc || (count > 0 && (!hist_descs || !dist_out))) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    if (count < 0 || count > 100000000) return fail(ctx, ICPMI_ERR_ARG, "count out of range");
    if (count == 0) return ICPMI_OK;
    const size_t hb = sizeof(double) * kScCells * (size_t)count;
    if ((rc = reserve(ctx, ctx->stage_a, hb))) return rc;
    if ((rc = reserve(ctx, ctx->stage_b, sizeof(double) * kScCells))) return rc;
    if ((rc = reserve(ctx, ctx->vox_out, sizeof(double) * (size_t)count))) return rc;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_a.p, hist_descs, hb, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_b.p, query_desc, sizeof(double) * kScCells, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sc_distances, dim3((unsigned)count), dim3(64), 0, s, (const double *)ctx->stage_b.p,
                       (const double *)ctx->stage_a.p, (int)count, (double *)ctx->vox_out.p);
    HIP_TRY(ctx, hipMemcpyAsync(dist_out, ctx->vox_out.p, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    HIP_TRY(ctx, hipGetLastError());
    return ICPMI_OK;
}

int icpmi_comm_unique_id(icpmi_ctx *ctx, void *id_out)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!id_out) return fail(ctx, ICPMI_ERR_NULL, "id_out is NULL");
    if ((rc = load_rccl(ctx))) return rc;
    static_assert(sizeof(ncclUniqueId) == ICPMI_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    RCCL_TRY(ctx, ctx->rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return ICPMI_OK;
}

int icpmi_comm_init(icpmi_ctx *ctx, int32_t n_ranks, int32_t rank, const void *id)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ctx, ICPMI_ERR_ARG, "bad rank %d of %d", rank, n_ranks);
    if (ctx->comm) return fail(ctx, ICPMI_ERR_ARG, "communicator already initialised");
    if (n_ranks == 1 && !id) { // no communicator wanted
        ctx->n_ranks = 1;
        ctx->rank = 0;
        return ICPMI_OK;
    }
    // (a 1-rank communicator is allowed: it runs the full sharded path, exchanges included)
    if (!id) return fail(ctx, ICPMI_ERR_NULL, "id is NULL");
    if ((rc = load_rccl(ctx))) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    RCCL_TRY(ctx, ctx->rccl.CommInitRank(&ctx->comm, n_ranks, uid, rank));
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return ICPMI_OK;
}

int icpmi_comm_init_callbacks(icpmi_ctx *ctx, int32_t n_ranks, int32_t rank,
                              icpmi_allreduce_fn allreduce, icpmi_allgather_fn allgather, void *user)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ctx, ICPMI_ERR_ARG, "bad rank %d of %d", rank, n_ranks);
    if (ctx->comm || ctx->cb_allreduce) return fail(ctx, ICPMI_ERR_ARG, "communicator already initialised");
    if (n_ranks > 1 && (!allreduce || !allgather)) return fail(ctx, ICPMI_ERR_NULL, "callback is NULL");
    ctx->cb_allreduce = allreduce;
    ctx->cb_allgather = allgather;
    ctx->cb_user = user;
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return ICPMI_OK;
}

// Who is in the communicator, asked of the communicator itself: its size and this rank from RCCL (ncclCommCount,
// ncclCommUserRank, ncclCommCuDevice), and every rank's device ordinal and PCI bus id gathered through the library's own
// all-gather -- the exchange the sharded normals use.  A collective: every rank must call it.
int icpmi_comm_info(icpmi_ctx *ctx, icpmi_comm_info_t *out)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (!out) return fail(ctx, ICPMI_ERR_NULL, "null argument");
    memset(out, 0, sizeof(*out));
    out->kind = ctx->comm ? 1 : (ctx->cb_allreduce ? 2 : 0);
    out->n_ranks = ctx->n_ranks;
    out->rank = ctx->rank;
    int device = ctx->opt.device;
    if (ctx->comm) {
        int v = 0;
        RCCL_TRY(ctx, ctx->rccl.CommCount(ctx->comm, &v));
        out->n_ranks = v;
        RCCL_TRY(ctx, ctx->rccl.CommUserRank(ctx->comm, &v));
        out->rank = v;
        RCCL_TRY(ctx, ctx->rccl.CommCuDevice(ctx->comm, &device));
    }
    const int nr = out->n_ranks;
    if (nr > ICPMI_MAX_RANKS_INFO) return fail(ctx, ICPMI_ERR_ARG, "more than %d ranks", ICPMI_MAX_RANKS_INFO);
    // one record of four doubles per rank: {device ordinal, 16 characters of PCI bus id, 0}
    constexpr size_t per = 4;
    double rec[per] = {0.0, 0.0, 0.0, 0.0};
    char bus[32] = {0};
    HIP_TRY(ctx, hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device));
    rec[0] = (double)device;
    memcpy(&rec[1], bus, 16);
    if ((rc = reserve(ctx, ctx->stage_c, sizeof(double) * per * (size_t)nr))) return rc;
    double *d = (double *)ctx->stage_c.p;
    HIP_TRY(ctx, hipMemsetAsync(d, 0, sizeof(double) * per * (size_t)nr, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d + per * (size_t)out->rank, rec, sizeof(rec), hipMemcpyHostToDevice, ctx->stream));
    if (out->kind != 0 && (rc = exchange_allgather(ctx, d, per))) return rc;
    std::vector<double> host(per * (size_t)nr);
    HIP_TRY(ctx, hipMemcpyAsync(host.data(), d, sizeof(double) * host.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < nr; ++r) {
        out->device[r] = (int32_t)host[per * r];
        memcpy(out->pci_bus_id[r], &host[per * r + 1], 16);
        out->pci_bus_id[r][15] = '\0';
    }
    return ICPMI_OK;
}

int icpmi_comm_finalize(icpmi_ctx *ctx)
{
    int rc;
    if ((rc = check_common(ctx))) return rc;
    if (ctx->comm) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        RCCL_TRY(ctx, ctx->rccl.CommDestroy(ctx->comm));
        ctx->comm = nullptr;
    }
    ctx->cb_allreduce = nullptr;
    ctx->cb_allgather = nullptr;
    ctx->cb_user = nullptr;
    ctx->n_ranks = 1;
    ctx->rank = 0;
    return ICPMI_OK;
}

int icpmi_reset_profile(icpmi_ctx *ctx)
{
    if (!ctx) return ICPMI_ERR_NULL;
    if (ctx->ev_used) { // events of finished calls not read yet: they belong to what is being discarded
        (void)hipSetDevice(ctx->opt.device);
        (void)hipStreamSynchronize(ctx->stream);
        harvest_profile(ctx);
    }
    memset(&ctx->prof, 0, sizeof(ctx->prof));
    return ICPMI_OK;
}

int icpmi_get_profile(icpmi_ctx *ctx, icpmi_profile *out)
{
    if (!ctx || !out) return ICPMI_ERR_NULL;
    if (ctx->ev_used) { // the events of the calls since the last look are read now (see align_device)
        (void)hipSetDevice(ctx->opt.device);
        (void)hipStreamSynchronize(ctx->stream);
        harvest_profile(ctx);
    }
    *out = ctx->prof;
    out->coarse_minima_bytes = (int64_t)ctx->coarse.cap;
    return ICPMI_OK;
}

#ifdef ICPMI_DEBUG_LOOP
// diagnostic build only (scripts/loop_rows.py): the rows of the last registration as the loop left them -- matches,
// moved coordinates, and the order the rows were taken in (0..n-1 when they were not sorted)
extern "C" int icpmi_debug_loop_lists(icpmi_ctx *ctx, double *ub_out, int32_t *cnt_out, uint32_t *ent_out, int64_t n)
{
    if (!ctx || !ctx->nn_lists.p || kNnListRowBytes * (size_t)n > ctx->nn_lists.cap) return ICPMI_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return ICPMI_ERR_HIP;
    const NnListRows lr = nn_list_rows(ctx, (int)n);
    if (hipMemcpy(ub_out, lr.ub, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return ICPMI_ERR_HIP;
    if (hipMemcpy(cnt_out, lr.cnt, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return ICPMI_ERR_HIP;
    if (hipMemcpy(ent_out, lr.ent, (size_t)n * 4 * kNnEntCap, hipMemcpyDeviceToHost) != hipSuccess) return ICPMI_ERR_HIP;
    return ICPMI_OK;
}
extern "C" int icpmi_debug_loop_rows(icpmi_ctx *ctx, int32_t *idx_out, double *cur_out, uint32_t *perm_out, int64_t n)
{
    if (!ctx || (size_t)n * sizeof(int) > ctx->idx.cap || (size_t)n * 24 > ctx->cur.cap) return ICPMI_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return ICPMI_ERR_HIP;
    if (hipMemcpy(idx_out, ctx->idx.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return ICPMI_ERR_HIP;
    if (hipMemcpy(cur_out, ctx->cur.p, (size_t)n * 24, hipMemcpyDeviceToHost) != hipSuccess) return ICPMI_ERR_HIP;
    if (ctx->src_sort.p && ctx->src_sort.cap >= (size_t)n * 16) {
        if (hipMemcpy(perm_out, (const unsigned *)ctx->src_sort.p + 3 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return ICPMI_ERR_HIP;
    } else {
        for (int64_t i = 0; i < n; ++i) perm_out[i] = (uint32_t)i;
    }
    return ICPMI_OK;
}
#endif

#if defined(ICPMI_COARSE_CLOCKS) || defined(ICPMI_SMALL_CLOCKS) || defined(ICPMI_GROUPS_CLOCKS)
// diagnostic build only: the stamps of the last all-pairs 1-NN pass (4 words per workgroup:
// s_memtime, s_memrealtime at its start and at its end)
int icpmi_debug_coarse_clocks(icpmi_ctx *ctx, unsigned long long *out, int64_t words)
{
    if (!ctx || !out || (size_t)words * 8 > ctx->slotmin.cap) return ICPMI_ERR_ARG;
    if (hipMemcpy(out, ctx->slotmin.p, (size_t)words * 8, hipMemcpyDeviceToHost) != hipSuccess) return ICPMI_ERR_HIP;
    return ICPMI_OK;
}
#endif

} // extern "C"
