/*
 * icp_mi355x.h -- C ABI of libicp_mi355x.so: point-to-plane ICP registration on one
 * MI355X (gfx950), optionally source-sharded over several with RCCL.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference has no
 * FFI layer; its boundary is the header-only C++ function
 *     slam::icp_point_to_plane(const PointCloud&, const PointCloud&, const ICPConfig&)
 *         -> ICPResult                      (slam_viz/core/icp.hpp:157-161)
 * called from slam_node.cpp:138 and loop_closure.hpp:109.  Each entry point below
 * names the reference interface it replaces (paths relative to
 * slam_viz/include/slam_viz/core/).  Plain pointers and sizes only.
 *
 * Conventions
 *   - points: row-major N x 3 fp64, contiguous ("xyzxyz...", types.hpp:17).
 *   - 4x4 transforms: ROW-major double[16] here.  Eigen::Matrix4d is column-major
 *     (types.hpp:76); an adapter must convert element-wise, never memcpy.
 *   - every function returns ICPMI_OK (0) or a negative ICPMI_ERR_* code; the text
 *     of the last failure is available from icpmi_last_error().
 *   - one icpmi_ctx serves one caller thread at a time; distinct contexts may run
 *     concurrently.  Calls block until results are on the host.
 *   - stream ordering of the *_device entry points: the library works on a private
 *     non-blocking HIP stream that is NOT ordered against any stream of the caller
 *     (torch's current stream included).  Device inputs must be complete and visible
 *     before the call (synchronise the producing stream, or wait on its event, first);
 *     device outputs are complete when the call returns.  One exception in what "returns" means:
 *     icpmi_stream_push* hand back their RESULTS complete, but may leave kernels of the library's own
 *     queued on that stream -- the search structure and normals of the scan just filtered, the next
 *     push's target -- so that the device builds them while the caller digests the result.  They touch
 *     only the context's workspace; every later call on the context is ordered behind them, and a
 *     fault in them is reported by that next call as a failure of the preparation step.
 *   - there is no CPU fallback: with no usable HIP device icpmi_create fails.
 */
#ifndef ICP_MI355X_H
#define ICP_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICPMI_OK 0
#define ICPMI_ERR_NULL (-1)          /* a required pointer is NULL */
#define ICPMI_ERR_EMPTY_SOURCE (-2)  /* n_src <= 0 (reference: division by zero, icp.hpp:206) */
#define ICPMI_ERR_EMPTY_TARGET (-3)  /* n_tgt <= 0 (reference: row(-1) UB, kdtree.hpp:33-36,211) */
#define ICPMI_ERR_CAPACITY (-4)      /* error_history buffer smaller than max_iterations + 1 */
#define ICPMI_ERR_HIP (-5)           /* HIP runtime failure */
#define ICPMI_ERR_RCCL (-6)          /* RCCL failure */
#define ICPMI_ERR_ARG (-7)           /* argument out of range */
#define ICPMI_ERR_NO_DEVICE (-8)     /* no gfx950 device / kernels not loadable */

/* nearest-neighbour search engines (all return the exact fp64 nearest neighbour) */
#define ICPMI_SEARCH_AUTO 0
#define ICPMI_SEARCH_EXACT_F64 1     /* fp64 brute force, SGPR-broadcast targets */
#define ICPMI_SEARCH_MFMA_BF16 2     /* bf16 MFMA coarse pass over ALL pairs + certified fp64 resolve */
#define ICPMI_SEARCH_MFMA_PRUNED 3   /* the same, skipping (32-row tile, target split) pairs whose bounding
                                        boxes are farther apart than the tile's known neighbour distance -- the
                                        rule of kdtree.hpp:139,177 applied to groups; same exact result, not an
                                        all-pairs pass (registrations and normal estimation).  AUTO takes it for
                                        targets of more than 12 splits (24,576 points), engine 2 below that; targets
                                        of more than 3,072 splits (6.29M points) or sources of 2^25 rows and more
                                        run on engine 2 whichever of the two was asked for */

typedef struct icpmi_ctx icpmi_ctx;

typedef struct {
    int32_t device;     /* HIP device ordinal */
    int32_t normal_k;   /* neighbours for PCA normals; the reference hard-codes 20 (icp.hpp:170) */
    int32_t search;     /* ICPMI_SEARCH_* */
    int32_t profile;    /* 0 off; 1: HIP events around the call, the loop and every 4th launch of the
                           dominant kernel; 2: around every stage (icpmi_get_profile) */
} icpmi_options;

/* mirrors slam::ICPConfig, types.hpp:143-148 */
typedef struct {
    int32_t max_iterations;        /* default 50 */
    int32_t reserved;
    double tolerance;              /* default 1e-6 */
    double min_error;              /* default 1e-9 */
    double initial_transform[16];  /* row-major, default identity */
} icpmi_config;

/* mirrors slam::ICPResult, types.hpp:155-164 */
typedef struct {
    double transformation[16];     /* row-major, maps source -> target (icp.hpp:154-155) */
    int32_t converged;
    int32_t num_iterations;        /* error_history.size() - 1 (icp.hpp:255) */
    double final_error;
    int32_t history_len;           /* entries written to error_history */
    int32_t loop_iterations;       /* loop bodies entered (not a reference field) */
} icpmi_result;

/* per-stage device time from HIP events on the library's stream, accumulated since
 * the last icpmi_reset_profile(); only filled when options.profile != 0 */
typedef struct {
    double nn_ms;        int64_t nn_launches;        /* correspondence search passes (coarse + resolve) */
    double coarse_ms;    int64_t coarse_launches;    /* k_nn_coarse alone: the dominant kernel (the launches bracketed) */
    double reduce_ms;    int64_t reduce_launches;    /* residual + 6x6 accumulation + solve */
    double transform_ms; int64_t transform_launches;
    double normals_ms;   int64_t normals_launches;   /* k-NN + PCA */
    double total_ms;     int64_t calls;              /* whole icpmi_align* calls, device time */
    double loop_ms;                                  /* iteration loop + post-loop pass only */
    double setup_ms;                                 /* Morton sort + operand packing of the target */
    double nn_pairs;                                 /* (source,target) pairs evaluated by nn passes */
    int64_t nn_recheck_queries;                      /* extra 128-target slots scanned in fp64 (MFMA engine) */
    int64_t nn_fallback_queries;                     /* whole 2048-target splits re-scanned in fp64 */
    int64_t knn_fallback_rows;                       /* normal-estimation rows resolved by the exact k-NN kernel */
    int64_t nn_coarse_blocks;                        /* (512-query block, 2048-target split) workgroups launched */
    int64_t nn_pruned_blocks;                        /* of those, skipped by ICPMI_SEARCH_MFMA_PRUNED's box test */
    int64_t small_launches;                          /* iterations run by the small-cloud kernel (search + residuals + pose update in one launch) */
    int64_t bounded_launches;                        /* passes of the ICP loop searched behind a bound per row (lists instead of coarse minima) */
    int64_t nn_group_pairs;                          /* culled engine: (32-row tile, 2048-target split) pairs of its passes ... */
    int64_t nn_group_pairs_run;                      /* ... and those within reach, the ones the coarse pass evaluated */
    double exchange_ms;  int64_t exchange_launches;  /* sharded runs: the per-pass all-reduce of 30 doubles alone (the launches bracketed) */
    int64_t coarse_minima_bytes;                     /* bytes this context holds for (row, split) coarse minima (6 B each): only the
                                                        stand-alone nearest-neighbour search and ICPMI_NN_BOUNDED=0 reserve any -- no
                                                        registration does (a state, not a counter: icpmi_reset_profile leaves it) */
} icpmi_profile;

void icpmi_options_default(icpmi_options *opt);     /* device 0, normal_k 20 (icp.hpp:170), search AUTO or the
                                                        value of the environment variable ICPMI_SEARCH (0..3) */
void icpmi_config_default(icpmi_config *cfg);        /* types.hpp:143-148 defaults */

int icpmi_create(const icpmi_options *opt, icpmi_ctx **out);
void icpmi_destroy(icpmi_ctx *ctx);
const char *icpmi_last_error(const icpmi_ctx *ctx);  /* ctx may be NULL: last create error */
const char *icpmi_version(void);

/* Replaces slam::icp_point_to_plane (icp.hpp:157-258).  Host pointers.
 * error_history must hold max_iterations + 1 doubles (history_cap states its size). */
int icpmi_align(icpmi_ctx *ctx, const double *source_xyz, int64_t n_src,
                const double *target_xyz, int64_t n_tgt, const icpmi_config *cfg,
                icpmi_result *result, double *error_history, int32_t history_cap);

/* Same, with source/target already resident in this device's HBM (device pointers). */
int icpmi_align_device(icpmi_ctx *ctx, const double *d_source_xyz, int64_t n_src,
                       const double *d_target_xyz, int64_t n_tgt, const icpmi_config *cfg,
                       icpmi_result *result, double *error_history, int32_t history_cap);

/* Several independent registrations at once: the up-to-three ICP verifications of one
 * LoopClosureDetector::detect() (loop_closure.hpp:94-123), each a slam::icp_point_to_plane call of its own in
 * the reference.  Problem k runs on a stream and workspace of its own (helper contexts the library keeps
 * inside `ctx`, created on first use with ctx's options), driven by a host thread of its own, so the
 * registrations share the GPU: a filtered scan fills the chip for a part of each iteration only.
 * Every result is bit-identical to the same icpmi_align call made alone.  Host pointers; cfgs, results,
 * status: `count` entries; error_history: `count` rows of history_stride doubles (>= cfgs[k].max_iterations + 1).
 * status[k] is problem k's return code; the call returns ICPMI_OK if all are, else the first that is not
 * (its text in icpmi_last_error(ctx)).  1 <= count <= ICPMI_MAX_BATCH; not for a context with a communicator. */
#define ICPMI_MAX_BATCH 8
int icpmi_align_batch(icpmi_ctx *ctx, int32_t count, const double *const *sources_xyz, const int64_t *n_src,
                      const double *const *targets_xyz, const int64_t *n_tgt, const icpmi_config *cfgs,
                      icpmi_result *results, double *error_history, int32_t history_stride, int32_t *status);

/* Replaces KDTree(points) + KDTree::nearest_batch (kdtree.hpp:20-26,43-59): for each
 * query the index of, and squared distance to, its nearest target.  Host pointers;
 * dist_sq may be NULL. */
int icpmi_nearest_batch(icpmi_ctx *ctx, const double *targets_xyz, int64_t n_tgt,
                        const double *queries_xyz, int64_t n_qry, int32_t *indices,
                        double *dist_sq);

/* Replaces KDTree::k_nearest (kdtree.hpp:65-78) for a batch of query points: indices (n_qry x k,
 * row-major) of the k nearest targets of every query, closest first as kdtree.hpp:72-76 returns
 * them; equal distances in ascending index order.  A target used as a query finds itself
 * first (distance 0), like the reference.  1 <= k <= 64.  Entries a list does not reach
 * (k > n_tgt; a query with a NaN coordinate) are -1 with distance +infinity.  dist_sq (n_qry x
 * k) may be NULL.  Host pointers. */
int icpmi_k_nearest(icpmi_ctx *ctx, const double *targets_xyz, int64_t n_tgt,
                    const double *queries_xyz, int64_t n_qry, int32_t k, int32_t *indices,
                    double *dist_sq);

/* Replaces estimate_normals(points, tree, k) (icp.hpp:23-67).  Host pointers. */
int icpmi_estimate_normals(icpmi_ctx *ctx, const double *points_xyz, int64_t n, int32_t k,
                           double *normals_xyz);

/* Replaces solve_point_to_plane(source, target, normals) (icp.hpp:89-144): inputs are
 * three n x 3 arrays matched row by row; out is the row-major 4x4 update. */
int icpmi_solve_point_to_plane(icpmi_ctx *ctx, const double *source_xyz,
                               const double *target_xyz, const double *normals_xyz, int64_t n,
                               double transform_out[16]);

/* Replaces Transformation::apply(cloud) (types.hpp:110-115): out = in * R^T + t^T. */
int icpmi_transform_points(icpmi_ctx *ctx, const double transform[16], const double *in_xyz,
                           int64_t n, double *out_xyz);

/* Replaces voxel_downsample(points, voxel_size) (slam_viz/src/core/file_utils.cpp:148-196), the
 * step slam_node.cpp:122 runs before every registration: centroid of the points of each
 * occupied voxel, key = floor(coord / voxel_size), points summed in input order.  Voxels come
 * out sorted by key (the reference's order is std::unordered_map iteration order, i.e.
 * implementation-defined).  voxel_size <= 0 copies the input (file_utils.cpp:152).  out_cap
 * is in rows; n rows always suffice.  The grid may span at most 2^21 cells per axis; a point with a NaN or
 * infinite coordinate (undefined behaviour in the reference: the cast of floor(NaN)) is ICPMI_ERR_ARG. */
int icpmi_voxel_downsample(icpmi_ctx *ctx, const double *points_xyz, int64_t n, double voxel_size,
                           double *out_xyz, int64_t out_cap, int64_t *n_out);
/* Same on device pointers (the result can feed icpmi_align_device without leaving HBM). */
int icpmi_voxel_downsample_device(icpmi_ctx *ctx, const double *d_points_xyz, int64_t n,
                                  double voxel_size, double *d_out_xyz, int64_t out_cap,
                                  int64_t *n_out);

/* Replaces load_ply / load_bin (slam_viz/src/core/file_utils.cpp:20-108, 115-141; the node
 * calls load_ply at slam_node.cpp:69,121): a path ending in ".bin" is read as KITTI
 * (x, y, z, intensity float32, intensity dropped), anything else as PLY with the reference's
 * header rules.  Host-side only (no context, no device).  Two-call pattern: with out_xyz ==
 * NULL only *n_out is set.  A file that cannot be opened returns ICPMI_ERR_ARG (the
 * reference throws std::runtime_error). */
int icpmi_load_cloud(const char *path, double *out_xyz, int64_t cap, int64_t *n_out);
/* Replaces discover_frames + extract_timestamp (slam_viz/src/core/file_utils.cpp:203-247): the
 * entries of data_dir whose extension is ".ply" or ".bin" and whose name holds a run of digits in
 * front of that extension, sorted by that number (equal numbers by path).  Two-call pattern: with
 * stamps == paths == NULL only *n_frames and *paths_bytes are set; then stamps[n_frames] and
 * paths (NUL-terminated strings, one after the other, paths_bytes in all) are filled.  A
 * directory that cannot be opened returns ICPMI_ERR_ARG (the reference throws).  Host-side. */
int icpmi_discover_frames(const char *data_dir, int64_t *stamps, int64_t frames_cap, char *paths,
                          int64_t paths_cap, int64_t *n_frames, int64_t *paths_bytes);

/* The device form of load_bin (file_utils.cpp:115-141): host float32 records (x, y, z leading,
 * stride_floats apart: 4 for KITTI) are copied to the device as they are and widened to the
 * N x 3 fp64 layout there (static_cast<double> is exact either side): 16 instead of 24 bytes per
 * point cross the host link.  d_out_xyz: device pointer, n rows. */
int icpmi_upload_points_f32(icpmi_ctx *ctx, const float *records, int64_t n, int32_t stride_floats,
                            double *d_out_xyz);
/* icpmi_load_cloud into device memory: ".bin" through icpmi_upload_points_f32, PLY parsed on the
 * host (header rules, ASCII numbers) and uploaded as fp64.  Two-call pattern like icpmi_load_cloud
 * (d_out_xyz == NULL: only *n_out). */
int icpmi_load_cloud_device(icpmi_ctx *ctx, const char *path, double *d_out_xyz, int64_t cap,
                            int64_t *n_out);

/* estimate_normals (icp.hpp:23-67) for rows [row0, row1) of the cloud only, against the whole
 * cloud: what one rank of a job that shards the normal estimation itself computes.  normals_xyz
 * receives row1 - row0 rows.  Host pointers. */
int icpmi_estimate_normals_rows(icpmi_ctx *ctx, const double *points_xyz, int64_t n, int32_t k,
                                int64_t row0, int64_t row1, double *normals_xyz);

/* One step of frame-to-frame odometry with the clouds resident in HBM: the registration part of
 * SlamNode::process_frame (slam_viz/src/ros/slam_node.cpp:122-152).
 *     curr = voxel_downsample(raw, voxel_size)                       :122
 *     first frame: keep it, nothing to register                      :69-72   -> ICPMI_STREAM_FIRST_FRAME
 *     curr.rows() < min_points: keep it, caller repeats its last pose :125-130 -> ICPMI_STREAM_TOO_FEW_POINTS
 *     result = icp_point_to_plane(source = curr, target = prev, cfg) :132-138 -> ICPMI_STREAM_REGISTERED
 *     prev = curr                                                    :128,152
 * The context keeps the previous filtered scan in device memory (the target of frame t+1 is the
 * source of frame t; buffers are swapped, nothing is copied or uploaded twice), builds the
 * target's search structure and normals from that resident copy, and the caller applies the
 * reference's gate (!converged || final_error > 1.0 -> identity, :139-140) and pose update.
 * d_raw_xyz: device pointer to the raw scan (e.g. from icpmi_load_cloud_device).  In the first
 * two cases *result is the identity with converged = 0 and no history. */
#define ICPMI_STREAM_REGISTERED 0
#define ICPMI_STREAM_FIRST_FRAME 1
#define ICPMI_STREAM_TOO_FEW_POINTS 2
typedef struct {
    int32_t status;      /* ICPMI_STREAM_* */
    int32_t reserved;
    int64_t n_filtered;  /* rows of the filtered current scan (now the resident "previous" one) */
    int64_t n_target;    /* rows of the scan it was registered against */
} icpmi_stream_info;
int icpmi_stream_push(icpmi_ctx *ctx, const double *d_raw_xyz, int64_t n_raw, double voxel_size,
                      int64_t min_points, const icpmi_config *cfg, icpmi_result *result,
                      double *error_history, int32_t history_cap, icpmi_stream_info *info);
/* Same with the raw scan in host memory (uploaded once; the filtered scan still never leaves the device). */
int icpmi_stream_push_host(icpmi_ctx *ctx, const double *raw_xyz, int64_t n_raw, double voxel_size,
                           int64_t min_points, const icpmi_config *cfg, icpmi_result *result,
                           double *error_history, int32_t history_cap, icpmi_stream_info *info);
/* Same with the raw scan in a file (load_ply / load_bin, file_utils.cpp:20-141, what slam_node.cpp:121
 * reads): a KITTI ".bin" goes from disk through pinned memory to the device as float32 and
 * everything behind the read is queued without a wait in between; a PLY takes the host parser. */
int icpmi_stream_push_file(icpmi_ctx *ctx, const char *path, double voxel_size, int64_t min_points,
                           const icpmi_config *cfg, icpmi_result *result, double *error_history,
                           int32_t history_cap, icpmi_stream_info *info);
/* Start bringing the NEXT frame file to the device on a worker thread of the context and return at once:
 * the file is read into pinned memory (~140 us for a 1.8 MB scan out of the page cache), copied over on a
 * stream of the worker's own, widened and -- with the voxel size of the stream's last push -- filtered there;
 * the icpmi_stream_push_file of that same path then starts at the registration (it waits for the worker if it
 * is still busy with that file; with another voxel size it filters the raw points, already on the device, itself).  Call it BEFORE
 * pushing the current frame: read and copy then run beside the current frame's work.  KITTI ".bin" only (a
 * PLY is parsed when pushed: the call is a no-op); one file at a time; a finished file is kept until the
 * push of its path takes it (at most two wait, the older gives way to a third); a file that cannot be read
 * is reported by the push.  Never needed for correctness. */
int icpmi_stream_prefetch_file(icpmi_ctx *ctx, const char *path);
int icpmi_stream_reset(icpmi_ctx *ctx);   /* forget the resident frame (a new sequence starts) */

/* The map side of SlamNode::process_frame, once the caller has formed new_pose = poses.back() * delta
 * (slam_viz/src/ros/slam_node.cpp:142-153):
 *     world = curr * new_pose.R^T + new_pose.t^T                    :147
 *     update_occupancy_grid(world, new_pose.t)                      :153, :211-221
 * update_occupancy_grid marks, for every world point with height_min <= z <= height_max and
 * 0.5 <= hypot(x - sensor.x, y - sensor.y) <= max_range, the 2-D cell (floor(x / resolution),
 * floor(y / resolution)) in a set of cells (std::unordered_set<GridCell>, slam_node.hpp:45-58); the
 * defaults are OccupancyGridConfig's (slam_node.hpp:35-40).  Here the set lives in device memory
 * as a sorted array of unique cells that every update merges into.  Beyond the reference: a point
 * whose quotient is not finite or does not fit an int (undefined static_cast there) marks nothing. */
typedef struct {
    double resolution;   /* 0.2 */
    double height_min;   /* 0.3 */
    double height_max;   /* 2.0 */
    double max_range;    /* 40.0 */
} icpmi_grid_config;
void icpmi_grid_config_default(icpmi_grid_config *grid);
/* world points in host memory (n x 3) and the sensor position; *n_cells (may be NULL: no wait for
 * the device then) receives the size of the set after the update. */
int icpmi_occupancy_update(icpmi_ctx *ctx, const double *world_xyz, int64_t n, const double sensor_xyz[3],
                           const icpmi_grid_config *grid, int64_t *n_cells);
/* Same with the points in device memory. */
int icpmi_occupancy_update_device(icpmi_ctx *ctx, const double *d_world_xyz, int64_t n, const double sensor_xyz[3],
                                  const icpmi_grid_config *grid, int64_t *n_cells);
/* The set: cells_xy[2 i], cells_xy[2 i + 1] = (x, y) of cell i, sorted by x then y.  cells_xy may be
 * NULL (only *n_cells is set); fewer than *n_cells entries of capacity is ICPMI_ERR_CAPACITY.  What
 * cells_to_occupancy_grid_msg (slam_node.cpp:279-297) rasterises. */
int icpmi_occupancy_cells(icpmi_ctx *ctx, int32_t *cells_xy, int64_t cap_cells, int64_t *n_cells);
int icpmi_occupancy_clear(icpmi_ctx *ctx);   /* occupied_cells_.clear() (slam_node.cpp:224) */
/* The filtered scan the last icpmi_stream_push* left resident (`curr`, slam_node.cpp:122), copied to the host:
 * what the node hands to loop_detector_.addFrame and keeps in downsampled_clouds_ (:160).  out_xyz may be
 * NULL (only *n_out is set).  The stream itself never needs this copy. */
int icpmi_stream_current_scan(icpmi_ctx *ctx, double *out_xyz, int64_t cap, int64_t *n_out);
/* Both steps on the scan the last icpmi_stream_push* left resident (it never came to the host):
 * pose = new_pose, row-major 4 x 4.  world_out (host, may be NULL) receives the n_filtered x 3 world
 * points (what publish_current_scan sends, :155), *n_world their number; grid may be NULL (no grid
 * update; *n_cells is then the current size).  One wait for everything the call returns. */
int icpmi_stream_map_update(icpmi_ctx *ctx, const double pose[16], const icpmi_grid_config *grid,
                            double *world_out, int64_t world_cap, int64_t *n_world, int64_t *n_cells);

/* Replaces ScanContext::compute (core/scan_context.hpp:44-82): 20 rings x 60 sectors max-height
 * descriptor, row-major desc_out[ring * 60 + sector], empty bins 0. */
#define ICPMI_SC_RINGS 20
#define ICPMI_SC_SECTORS 60
int icpmi_scan_context(icpmi_ctx *ctx, const double *cloud_xyz, int64_t n, double *desc_out /* 1200 */);
/* Replaces the loop of ScanContext::distance calls in LoopClosureDetector::detect
 * (core/loop_closure.hpp:78-89, core/scan_context.hpp:90-142): dist_out[i] = min over the 60
 * column shifts of 1 - cosine(query, hist_descs + 1200 * i). */
int icpmi_scan_context_distances(icpmi_ctx *ctx, const double *query_desc, const double *hist_descs,
                                 int64_t count, double *dist_out);

/* Multi-GPU (new; the reference has no distributed path).  One process per GPU.  Rank 0
 * obtains an id, the host distributes it (e.g. torch.distributed broadcast), every rank
 * calls icpmi_comm_init.  Afterwards icpmi_align* treats `source` as this rank's shard
 * of the source cloud; the target is replicated.  Each iteration all-reduces 30 doubles
 * (21 J^T J + 6 J^T b + sum b^2 + count + the number of ranks whose loop has ended) over RCCL.
 * Every rank must make the same calls with the same icpmi_config and target.  A shard may be
 * empty (n_src == 0, source pointer ignored) as long as some rank holds points: it adds
 * nothing to the sums and takes part in every exchange.  The ranks stop on the exchanged
 * count of finished loops, never on a local decision, so they always queue the same number
 * of collectives; if they did not all finish at the same iteration (configs differ, or an
 * exchange that is not bit-identical on every rank) every rank returns ICPMI_ERR_RCCL from
 * that call instead of waiting for the others forever. */
#define ICPMI_UNIQUE_ID_BYTES 128
int icpmi_comm_unique_id(icpmi_ctx *ctx, void *id_out /* ICPMI_UNIQUE_ID_BYTES */);
int icpmi_comm_init(icpmi_ctx *ctx, int32_t n_ranks, int32_t rank, const void *id);
int icpmi_comm_finalize(icpmi_ctx *ctx);

/* Same sharded path with the two exchanges done by host callbacks instead of RCCL (the
 * buffers are host memory; the callee must leave the result in place).  Meant for
 * rehearsing the N > 1 path where RCCL cannot run (several ranks on one GPU, gloo). */
typedef int (*icpmi_allreduce_fn)(void *user, double *buf, int32_t count);            /* in-place sum */
typedef int (*icpmi_allgather_fn)(void *user, double *buf, int32_t count_per_rank);   /* in-place, rank-major */
int icpmi_comm_init_callbacks(icpmi_ctx *ctx, int32_t n_ranks, int32_t rank,
                              icpmi_allreduce_fn allreduce, icpmi_allgather_fn allgather,
                              void *user);

/* Who is in this context's communicator, asked of the communicator itself (new, like the rest of the multi-GPU
 * section: no reference counterpart): n_ranks / rank / this rank's device from ncclCommCount / ncclCommUserRank /
 * ncclCommCuDevice, and every rank's device ordinal and PCI bus id gathered through the library's own all-gather --
 * so that a scaling line can prove how many distinct GPUs its ranks ran on.  A collective: every rank calls it.
 * kind: 0 no communicator (n_ranks 1), 1 RCCL, 2 host callbacks (n_ranks as given to icpmi_comm_init_callbacks). */
#define ICPMI_MAX_RANKS_INFO 64
typedef struct {
    int32_t kind, n_ranks, rank, reserved;
    int32_t device[ICPMI_MAX_RANKS_INFO];
    char pci_bus_id[ICPMI_MAX_RANKS_INFO][16];
} icpmi_comm_info_t;
int icpmi_comm_info(icpmi_ctx *ctx, icpmi_comm_info_t *out);

/* profiling */
int icpmi_reset_profile(icpmi_ctx *ctx);
int icpmi_get_profile(icpmi_ctx *ctx, icpmi_profile *out);

#ifdef __cplusplus
}
#endif
#endif
