/*
 * include/mvr_hip.h -- C-ABI of the MI355X-native ICP hot path (libmvr_hip.so).
 *
 * The reference (fanxiaochen/Multi-View-Registration, `mvr`) has no FFI or
 * plugin interface for this path: the boundary is the set of PCL member
 * functions that Registrator calls.  Each entry point
 * below names the reference call site(s) it stands behind; the C++ shim in
 * include/mvr/ (source-compatible PCL-style classes) is the only intended
 * caller, and INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *  - plain C types only; every function returns 0 (MVR_OK) or a negative
 *    mvr_status; no exceptions cross this boundary.
 *  - points are 16-byte records {x,y,z,w} (pcl::PointXYZ layout,
 *    mvr/include/types.h:14) or packed 12-byte xyz; `stride_bytes` says which.
 *    Uploads COPY: the caller keeps ownership and may mutate or alias its
 *    clouds between calls, as the reference does (registrator.cpp:576, :920).
 *  - poses are 4x4 column-major, column-vector convention: the memory layout
 *    of Eigen::Matrix4f (what icp.getFinalTransformation() returns,
 *    registrator.cpp:573).  mvr/include/types.h:20-50 (PclMatrixCaster) is the
 *    transposing bridge to OSG's row-vector matrices.
 *  - one mvr_ctx per host thread and GPU; a ctx is not thread-safe (the
 *    reference drives the path from one worker thread, registrator.cpp:606).
 *  - clouds live in numbered device "slots" of the ctx.
 *  - the library fails loudly: with no usable GPU mvr_ctx_create returns
 *    MVR_E_HIP; there is NO CPU fallback anywhere behind this header.
 */
#ifndef MVR_HIP_H
#define MVR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvr_ctx mvr_ctx;

typedef enum {
  MVR_OK        =  0,
  MVR_E_ARG     = -1,   /* bad argument / slot / size                       */
  MVR_E_HIP     = -2,   /* HIP runtime error or no GPU (see mvr_last_error) */
  MVR_E_NOCORR  = -3,   /* < 3 correspondences (PCL min_number_correspondences_ = 3;
                           "Not enough correspondences found", App. A.1)   */
  MVR_E_NOMEM   = -4,
  MVR_E_SINGULAR= -5,   /* singular system in a host solve (LUM)            */
  MVR_E_RCCL    = -6    /* RCCL failure, or librccl not loadable (multi-GPU entry points only) */
} mvr_status;

#define MVR_MAX_SLOTS 256

/* ---- parameters of one IterativeClosestPoint::align ----------------------
 * Setters at mvr/src/registrator.cpp:551-560, :768-771, :901-904.           */
typedef struct {
  int    use_reciprocal;          /* setUseReciprocalCorrespondences(bool)  */
  double max_corr_dist;           /* setMaxCorrespondenceDistance(double)   */
  int    max_iterations;          /* setMaximumIterations(int)              */
  double transformation_epsilon;  /* setTransformationEpsilon(double)       */
  double euclidean_fitness_eps;   /* setEuclideanFitnessEpsilon(double)     */
  int    fma_dist;                /* 0 (spec): d2 = (dx*dx+dy*dy)+dz*dz, each op
                                     rounded (FLANN L2_Simple); 1: fma chain */
  int    point_to_plane;          /* 0 (the reference): TransformationEstimationSVD;
                                     1 (EXTENSION, BASELINE config 2; no counterpart in
                                     the reference): PCL's point-to-plane LLS estimator on
                                     the same correspondences; needs target normals */
} mvr_icp_params;

/* pcl::registration::DefaultConvergenceCriteria::ConvergenceState */
enum { MVR_CONV_NOT = 0, MVR_CONV_ITERATIONS = 1, MVR_CONV_TRANSFORM = 2,
       MVR_CONV_ABS_MSE = 3, MVR_CONV_REL_MSE = 4, MVR_CONV_NO_CORRESPONDENCES = 5 };

typedef struct {
  int    iterations;   /* nr_iterations_                                    */
  int    converged;    /* hasConverged()                                    */
  int    state;        /* MVR_CONV_*                                        */
  int    n_corr;       /* accepted correspondences, last iteration          */
  double mse;          /* mean squared correspondence distance, last iter.  */
  double evals;        /* point-pair distance evaluations performed (fwd + reciprocal) */
  double fwd_queries;  /* source queries of the forward NN passes           */
  double ms;           /* wall milliseconds spent inside the call           */
} mvr_icp_stats;

/* moments of one scan pair under its accepted correspondences (K5+K6):
 * what TransformationEstimationSVD accumulates (App. A.3).                 */
typedef struct {
  double n;            /* M = number of accepted correspondences           */
  double mean_src[3];  /* (1/M) sum p_i                                     */
  double mean_tgt[3];  /* (1/M) sum q_i                                     */
  double mse;          /* (1/M) sum d2_i                                    */
  double sigma[9];     /* (1/M) sum (q_i-mean_tgt)(p_i-mean_src)^T, row-major */
} mvr_pair_moments_t;

/* raw second moments of one scan pair (K8): everything LUM::computeEdge needs
 * for ANY pair of vertex poses (see mvr_lum_edge_from_moments).  Points are
 * taken relative to `origin` to keep magnitudes small.                     */
typedef struct {
  double n;
  double origin[3];
  double sp[3], sq[3];        /* sum p, sum q                               */
  double spp[6], sqq[6];      /* sum p p^T, sum q q^T : xx xy xz yy yz zz   */
  double spq[9];              /* sum p q^T, row-major                       */
  double sum_d2;              /* sum of the accepted correspondences' squared distances (the f32 values the search
                                 found, added in f64): the pair's residual; 0 when the caller supplied the pairs */
} mvr_pair_moments2_t;       /* 32 doubles: exactly one row of the device edge table */

/* ---- context -------------------------------------------------------------- */
int  mvr_ctx_create(mvr_ctx **ctx, int device_id);
/* as above, but all work is enqueued on `hip_stream` (a hipStream_t owned by
 * the caller, e.g. torch.cuda.current_stream().cuda_stream) instead of a
 * private stream; NULL = private stream. */
int  mvr_ctx_create_on_stream(mvr_ctx **ctx, int device_id, void *hip_stream);
int  mvr_ctx_destroy(mvr_ctx *ctx);
int  mvr_ctx_sync(mvr_ctx *ctx);                 /* wait for the ctx stream   */
const char *mvr_strerror(int status);
const char *mvr_last_error(const mvr_ctx *ctx);  /* detail of the last failure */
int  mvr_device_info(mvr_ctx *ctx, char *name, size_t name_cap, int *n_cu, int *clock_mhz);

/* ---- clouds ----------------------------------------------------------------
 * setInputSource/setInputTarget (registrator.cpp:566-567, 497-498, 645-646,
 * 776-777, 913-914) and lum.addPointCloud (:636) become an upload into a slot. */
int  mvr_cloud_upload(mvr_ctx *ctx, int slot, const float *xyz, size_t n, size_t stride_bytes);
int  mvr_cloud_download(mvr_ctx *ctx, int slot, float *xyz, size_t cap_points, size_t stride_bytes, size_t *n);
int  mvr_cloud_size(mvr_ctx *ctx, int slot, size_t *n);
int  mvr_cloud_reserve(mvr_ctx *ctx, int slot, size_t capacity_points);
int  mvr_cloud_copy(mvr_ctx *ctx, int dst_slot, int src_slot);
/* `*target += transformed_source` (registrator.cpp:576, :833, :982). */
int  mvr_cloud_append(mvr_ctx *ctx, int dst_slot, int src_slot);
int  mvr_cloud_clear(mvr_ctx *ctx, int slot);
/* PointCloud::denoise(segment_threshold, triangle_length) (mvr/src/point_cloud.cpp:423-465): link the points
 * whose distance is <= triangle_length (the same connected components as the reference's Delaunay edges of that
 * length: the Euclidean MST is a subgraph of the Delaunay triangulation), drop the components with fewer than
 * segment_threshold points; the cloud in `slot` is replaced by the kept points, component after component (ordered
 * by their smallest point index), points in index order -- the reference's output order.  kept_index (optional,
 * host, capacity = old size) receives the original indices of the kept points.  One documented difference: exact
 * duplicate points are ordinary members of their component here, while CGAL's insert keeps only the first of them as a
 * vertex (the later copies become isolated and are dropped by any segment_threshold > 1). */
int  mvr_cloud_denoise(mvr_ctx *ctx, int slot, int segment_threshold, double triangle_length,
                       size_t *n_kept, size_t *n_components, uint32_t *kept_index);
/* PointCloud::getTransformedPoints (point_cloud.cpp:290-303): f32 points times
 * the f64 pose (osg::Matrixd::preMult incl. the w divide), rounded to f32.
 * dst_slot may equal src_slot. */
int  mvr_cloud_transform(mvr_ctx *ctx, int dst_slot, int src_slot, const double T[16]);
/* the same for `count` clouds in ONE launch (the loop over all scans that poses them before a
 * global iteration, registrator.cpp:630-637): T = count x 16 doubles.  The entries run concurrently: a destination
 * slot may not appear twice, nor be the source of ANOTHER entry (MVR_E_ARG); dst[k] == src[k] poses in place.
 * mvr_ring_step / mvr_ring_run hand their posed_slots / raw_slots to this call and inherit the rule. */
int  mvr_cloud_transform_batch(mvr_ctx *ctx, int count, const int *dst_slots, const int *src_slots, const double *T);
/* pcl transformPointCloud / ICP::transformCloud (inside align, App. A.1):
 * x' = ((T00 x + T01 y) + T02 z) + T03 in f32, no contraction. */
int  mvr_cloud_transform_f32(mvr_ctx *ctx, int dst_slot, int src_slot, const float T[16]);
/* EXTENSION (point-to-plane): unit normals of the cloud in `slot`, one per point
 * (n must equal the cloud's size; stride 16 {nx,ny,nz,*} or 12).  Normals follow
 * the cloud through copy / transform (rotated) / append (kept only if both
 * clouds carry normals); a new upload of points drops them. */
int  mvr_cloud_upload_normals(mvr_ctx *ctx, int slot, const float *nxyz, size_t n, size_t stride_bytes);
int  mvr_cloud_download_normals(mvr_ctx *ctx, int slot, float *nxyz, size_t cap_points, size_t stride_bytes, size_t *n);

/* ---- the hot path ------------------------------------------------------------ */
/* exact 1-NN of every point of q_slot in t_slot: tree_->nearestKSearch(p,1,..)
 * (inside icp.align registrator.cpp:569,920,1012,1024 and :502,:649).
 * idx/d2 are host arrays of n_q entries (either may be NULL).  Ties -> lowest
 * index.  idx = UINT32_MAX and d2 = +inf when the target is empty. */
int  mvr_nn(mvr_ctx *ctx, int q_slot, int t_slot, int fma_dist, uint32_t *idx, float *d2);

/* CorrespondenceEstimation::determineReciprocalCorrespondences(corrs, max_d)
 * (registrator.cpp:496-502, 644-649); reciprocal = 0 gives
 * determineCorrespondences.  Outputs are host arrays of capacity `cap`
 * (pcl::Correspondence fields index_query / index_match / distance(squared)),
 * in ascending query order.  *m = number found (may exceed cap: truncated). */
int  mvr_correspondences(mvr_ctx *ctx, int src_slot, int tgt_slot, double max_dist,
                         int reciprocal, int fma_dist,
                         int32_t *query, int32_t *match, float *dist2, size_t cap, size_t *m);

/* correspondences + TransformationEstimationSVD accumulation (K2,K3,K5,K6) of
 * one scan pair; the correspondences stay on the device.  This is the
 * shardable per-pair unit of the ring/global mode (registrator.cpp:482-502,
 * 640-651).  q_begin/q_count restrict the SOURCE queries to a sub-range (for
 * splitting one pair over ranks; sums are then partial: n, n*mean, n*mse and
 * raw second moments add across sub-ranges) -- pass 0, SIZE_MAX for all. */
int  mvr_pair_moments(mvr_ctx *ctx, int src_slot, int tgt_slot, double max_dist,
                      int reciprocal, int fma_dist, mvr_pair_moments_t *out);
int  mvr_pair_moments2(mvr_ctx *ctx, int src_slot, int tgt_slot, double max_dist,
                       int reciprocal, int fma_dist, size_t q_begin, size_t q_count,
                       const double origin[3], mvr_pair_moments2_t *out);
/* device-output variant: writes 32 doubles {n, origin[3], sp, sq, spp, sqq,
 * spq, 0} to `dev_out` (device pointer, e.g. a row of a torch tensor that is
 * all-reduced over RCCL afterwards) without any host synchronisation. */
int  mvr_pair_moments2_dev(mvr_ctx *ctx, int src_slot, int tgt_slot, double max_dist,
                           int reciprocal, int fma_dist, size_t q_begin, size_t q_count,
                           const double origin[3], double *dev_out);
/* The whole edge list of one global iteration in one call (the `for` loops over
 * scan pairs of registrator.cpp:482-502 and :640-651): pair k = (src[k], dst[k])
 * with query range q_begin[k], q_count[k] (null arrays: all queries).  The pairs
 * are independent: with the culled search every stage (forward searches, flagging
 * of matched targets, reverse searches, filter + raw moments, final sums) is ONE
 * launch for all pairs (blockIdx.y = pair), six launches in all, so the pairs fill
 * the chip together; with the brute-force search (or pair_fused = 0) each pair runs
 * on one of `pair_streams` worker HIP streams forked from and joined back into the
 * context's stream (mvr_ctx_tune).  Results are identical to
 * n_pairs calls of mvr_pair_moments2.  out (host, [n_pairs]) and/or dev_out
 * (device, [n_pairs][32] doubles, no host synchronisation) receive the sums. */
int  mvr_pair_moments2_batch(mvr_ctx *ctx, int n_pairs, const int *src_slots, const int *tgt_slots,
                             double max_dist, int reciprocal, int fma_dist, const size_t *q_begin,
                             const size_t *q_count, const double origin[3],
                             mvr_pair_moments2_t *out, double *dev_out);

/* The accepted correspondences of pair k of the LAST fused mvr_pair_moments2_batch / mvr_ring_step on this context (the
 * culled / grid search with pair_fused, the default), as determineReciprocalCorrespondences lists them for that pair
 * (registrator.cpp:496-502, :644-649): ascending query index, squared distances; *m = number found (may exceed cap).
 * Read back from the keys the batch left on the device: valid until the next batch or any change of the pair's clouds
 * (MVR_E_ARG then).  What computeError shows and what the tests compare with the oracle's lists pair by pair. */
int  mvr_pair_batch_correspondences(mvr_ctx *ctx, int k, int32_t *query, int32_t *match, float *dist2, size_t cap, size_t *m);

/* ---- target sharding over ranks (sequential mode, registrator.cpp:563-577, is loop-carried and does not
 * shard by pair: the growing TARGET is split by points, every rank holds the full source).  A target slot
 * can be a shard: mvr_cloud_set_global_base / mvr_cloud_append_range record which GLOBAL point numbers its
 * local points carry.  One ICP iteration over G ranks is then
 *   mvr_nn_forward_keys        -> dev_keys[Ns] signed 64-bit (d2 bits << 32 | GLOBAL target index; INT64_MAX = none)
 *   all-reduce MIN over ranks    (RCCL ncclInt64/ncclMin: d2 >= 0, so signed order = (d2, index) order and
 *                                 ties go to the lowest global index -- the single-GPU rule)
 *   mvr_pair_moments2_from_keys -> this rank's share of the sums: reciprocal check, acceptance and raw
 *                                 second moments of the matches whose target point it OWNS (32 doubles,
 *                                 [31] = sum of d2), device output
 *   all-reduce SUM of the 32 doubles; then the usual host solve (mvr_moments_from_moments2 +
 *   mvr_umeyama_from_moments).  multi-view-registration_amd/seq.py is that loop. */
int  mvr_cloud_set_global_base(mvr_ctx *ctx, int slot, size_t global_begin);
int  mvr_cloud_append_range(mvr_ctx *ctx, int dst_slot, int src_slot, size_t src_begin, size_t count,
                            size_t global_begin);
int  mvr_nn_forward_keys(mvr_ctx *ctx, int src_slot, int tgt_slot, double max_dist, int fma_dist,
                         long long *dev_keys);
int  mvr_pair_moments2_from_keys(mvr_ctx *ctx, int src_slot, int tgt_slot, const long long *dev_keys,
                                 double max_dist, int reciprocal, int fma_dist, const double origin[3],
                                 double *dev_out);

/* The native host of that loop (csrc/mvr_ctx.hip, collectives in csrc/mvr_world.cpp).  mvr_seq_align_sharded: ONE
 * IterativeClosestPoint::align (registrator.cpp:569) of the full source in src_slot against the sharded target in tgt_slot
 * (this rank's shard), collectives on the context's communicator (mvr_ctx_comm_init / mvr_world_create; none = a world of
 * one): per iteration ncclAllReduce(ncclInt64, ncclMin) of the Ns keys and ncclAllReduce(ncclDouble, ncclSum) of the sums, on
 * the context's stream, then the host solve and the convergence criteria.  out_slot receives final * input (may equal
 * src_slot; < 0: none).  Every rank calls it with the same arguments and gets the same T_out / stats.
 * mvr_seq_run_sharded: Registrator::registrationICP (registrator.cpp:526-588) on top of it -- views 1, V-1, 2, ... aligned
 * one after the other against everything merged so far, pose_v <- T_icp * pose_v (:574), and `*target += aligned source`
 * (:576) as "every rank appends its slice [n g / G, n (g + 1) / G) of the aligned scan to its shard".  poses: n_views x 16
 * column-major, in / out; the per-align outputs (capacity repeat * (n_views - 1), any may be NULL) log the view, the
 * transformation and the statistics of every align.  PCL's "not enough correspondences" does not stop the driver
 * (registrator.cpp:569-574 never checks hasConverged()).
 * Failures: a rank whose local work fails still joins the iteration's collectives (neutral keys, zero sums, a raised failure
 * count), so every rank leaves the iteration together -- the failing one with its own status, the others with MVR_E_RCCL;
 * a rank whose peers never arrive gives up after "wait_timeout_ms" (mvr_ctx_tune), aborts the communicator
 * (ncclCommAbort) and returns MVR_E_RCCL.  The same holds for mvr_ring_run_sharded.  Never a hang. */
int  mvr_seq_align_sharded(mvr_ctx *ctx, int src_slot, int tgt_slot, int out_slot, const mvr_icp_params *params,
                           const double origin[3], float T_out[16], mvr_icp_stats *stats);
int  mvr_seq_run_sharded(mvr_ctx *ctx, int n_views, const int *raw_slots, int target_slot, int source_slot, int out_slot,
                         const mvr_icp_params *params, const double origin[3], int repeat, double *poses, int *align_view,
                         float *align_T, mvr_icp_stats *align_stats, int *n_aligns);
/* The same driver on ONE GPU around mvr_icp_align (Registrator::registrationICP, registrator.cpp:526-588, as one native call):
 * `repeat` sweeps; each poses view 0 into target_slot (reserved once for all the scans) and aligns views 1, V-1, 2, ... against
 * everything merged so far, pose_v <- T_icp * pose_v, target += aligned source.  poses: n_views x 16 column-major, in / out; the
 * per-align outputs as above (capacity repeat * (n_views - 1), any may be NULL).  MVR_E_NOCORR of an align does not stop it. */
int  mvr_seq_run(mvr_ctx *ctx, int n_views, const int *raw_slots, int target_slot, int source_slot, int out_slot,
                 const mvr_icp_params *params, int repeat, double *poses, int *align_view, float *align_T,
                 mvr_icp_stats *align_stats, int *n_aligns);

/* raw second moments of caller-supplied correspondences (lum.setCorrespondences,
 * registrator.cpp:650): query[k] indexes src_slot, match[k] indexes tgt_slot. */
int  mvr_pair_moments2_from_corr(mvr_ctx *ctx, int src_slot, int tgt_slot, const int32_t *query,
                                 const int32_t *match, size_t m, const double origin[3],
                                 mvr_pair_moments2_t *out);

/* host-side solves on the moments (3x3 SVD stays on the host) */
/* TransformationEstimationSVD / Eigen::umeyama (App. A.3). */
int  mvr_umeyama_from_moments(const mvr_pair_moments_t *mom, float T[16], double sv[3]);
int  mvr_moments_from_moments2(const mvr_pair_moments2_t *m2, mvr_pair_moments_t *out);

/* pcl::IterativeClosestPoint<PointXYZ,PointXYZ>::align(out)
 * (registrator.cpp:569, :920, :1012, :1024).  out_slot receives
 * final * input (may equal src_slot: the aliased align(*source_) of :920);
 * out_slot < 0 skips it.  T_out = getFinalTransformation().  Returns
 * MVR_E_NOCORR (and stats->state = NO_CORRESPONDENCES, converged = 0) when an
 * iteration finds < 3 correspondences; T_out then holds the transformation
 * accumulated so far, as PCL leaves it. */
int  mvr_icp_align(mvr_ctx *ctx, int src_slot, int tgt_slot, int out_slot,
                   const mvr_icp_params *params, float T_out[16], mvr_icp_stats *stats);

/* Registration::getFitnessScore(max_range) (registrator.cpp:572,923,1015):
 * mean squared distance of T*input to its unbounded 1-NN in the target.
 * DBL_MAX when no point qualifies. */
int  mvr_fitness(mvr_ctx *ctx, int input_slot, int tgt_slot, const float T[16],
                 double max_range, int fma_dist, double *score);

/* ---- LUM (pcl::registration::LUM, registrator.cpp:627-663) ------------------- */
/* LUM::computeEdge for one edge from its raw moments and the two vertex poses
 * (x,y,z,roll,pitch,yaw): MM (6x6 row-major), MZ, ss, n_valid. */
int  mvr_lum_edge_from_moments(const mvr_pair_moments2_t *m2, const double pose_s[6],
                               const double pose_t[6], double MM[36], double MZ[6],
                               double *ss);
/* The same for FOUR edges in one pass (one per AVX2 lane; what mvr_lum_compute runs its edges through): m2[4],
 * pose_s / pose_t [4][6], MM [4][36], MZ [4][6], ss [4].  Bit-identical to four calls of the function above;
 * MVR_E_NOCORR (outputs untouched) when a lane needs one of its special paths (n < 3, an unsafe Cholesky pivot). */
int  mvr_lum_edge_from_moments_x4(const mvr_pair_moments2_t *m2, const double *pose_s, const double *pose_t,
                                  double *MM, double *MZ, double *ss);
/* LUM::compute on n vertices / ne edges given per-edge moments; poses n*6
 * in/out, vertex 0 fixed.  Returns iterations done in *iters. */
int  mvr_lum_compute(int n, int ne, const int *edge_src, const int *edge_tgt,
                     const mvr_pair_moments2_t *edge_m2, int max_iterations,
                     double convergence_threshold, double *poses, int *iters);
/* the host side of one global step of registrationLUM (registrator.cpp:650-662)
 * from the (all-reduced) ne x 32 edge table: per-pair Umeyama + residual,
 * LUM::compute, pose_v <- LUM_v * pose_v (poses: n_views x 16 column-major,
 * in/out; vertex 0 fixed).  pair_T (ne x 16) may be NULL. */
int  mvr_ring_host_step(int n_views, int ne, const int *edge_src, const int *edge_tgt, const double *rows,
                        const double origin[3], int lum_iterations, double *poses, double *lum_pose,
                        float *pair_T, double *pair_n, double *pair_mse, int *lum_iters);
/* One outer pass of Registrator::registrationLUM (registrator.cpp:625-664) in ONE call, for a single process:
 *   posed_slots[v] <- poses[v] applied to raw_slots[v]            (mvr_cloud_transform_batch)
 *   per edge e: reciprocal correspondences of posed[edge_src[e]] in posed[edge_tgt[e]] + raw moments
 *                                                                 (mvr_pair_moments2_batch)
 *   host: per-pair Umeyama + residual, LUM::compute, pose_v <- LUM_v * pose_v   (mvr_ring_host_step)
 * edge_src / edge_tgt are VIEW indices (0..n_views-1).  poses: n_views x 16 column-major, in/out.  Optional
 * outputs (may be NULL): rows (ne x 32, the edge table as mvr_pair_moments2_dev lays it out), pair_T (ne x 16),
 * timing_ms[3] = {enqueue, wait for the GPU + copy of the table, host solve}.  Synchronises the context's stream. */
int  mvr_ring_step(mvr_ctx *ctx, int n_views, const int *posed_slots, const int *raw_slots, int ne, const int *edge_src,
                   const int *edge_tgt, double max_dist, int reciprocal, int fma, const double origin[3], int lum_iterations,
                   double *poses, double *lum_pose, float *pair_T, double *pair_n, double *pair_mse, int *lum_iters,
                   double *rows, double *timing_ms);
/* n_steps outer passes (the loop of registrator.cpp:625-664), each exactly one mvr_ring_step; the outputs are those
 * of the LAST pass, timing_ms the SUM over the passes.  Stops at the first pass that fails and returns its status.
 * (pair_T is computed for the last pass only -- a 3 x 3 SVD per pair that no pass in between needs: it is left untouched when
 * the run stops early; pair_n, pair_mse, lum_pose and rows are those of the last pass that was solved.)
 * In steady state (no allocation, no index or grid to build: from the third or fourth pass of a registration on) the
 * passes are PIPELINED: the launch chain of pass k+1 is queued behind a gate while pass k runs and released by the host's
 * solve with one store (tune key "pipeline"); same results, bit for bit. */
int  mvr_ring_run(mvr_ctx *ctx, int n_steps, int n_views, const int *posed_slots, const int *raw_slots, int ne,
                  const int *edge_src, const int *edge_tgt, double max_dist, int reciprocal, int fma, const double origin[3],
                  int lum_iterations, double *poses, double *lum_pose, float *pair_T, double *pair_n, double *pair_mse,
                  int *lum_iters, double *rows, double *timing_ms);
/* LUM::incidenceCorrection(pose) (inside lum.compute(), registrator.cpp:654): the 6x6 (row-major) matrix H with
 * d(R p + t)/d pose = M(p') H for R = Rx Ry Rz and the M of LUM::computeEdge; exposed so that its sign pattern can be
 * pinned by a numeric Jacobian without a GPU. */
void mvr_lum_incidence(const double pose[6], double H[36]);
/* ---- multi-GPU: the outer passes of registrationLUM (registrator.cpp:625-664) sharded over the GPUs of one node ----
 * The shardable unit is the scan pair of the loop at registrator.cpp:640-651.  The V * Ns source queries of all pairs
 * are dealt to the ranks in contiguous equal ranges (mvr_ring_segments; a pair may be split between two ranks, its
 * sums are additive); every rank holds every scan, no point crosses the fabric.  Per pass each rank runs its share
 * of the fused searches + sums, then ONE ncclAllReduce(sum) of the [edges][32] f64 table over RCCL/xGMI on the
 * context's stream, then the (tiny) host solve -- redundantly on every rank from identical bits, so no broadcast.
 * RCCL is loaded at run time (dlopen: a copy the process already holds, e.g. PyTorch-ROCm's, else the system one,
 * else $MVR_RCCL_LIB); without it these entry points return MVR_E_RCCL and nothing else is affected.
 *
 * (a) one process per GPU (the usual launcher shape): rank 0 calls mvr_comm_unique_id, the 128 bytes travel to the
 *     other ranks by the launcher's own means, every rank calls mvr_ctx_comm_init on its context, then all ranks
 *     call mvr_ring_run_sharded with the same arguments.  A context without a communicator is a world of one
 *     (mvr_ring_run_sharded == mvr_ring_run). */
#define MVR_UNIQUE_ID_BYTES 128
int  mvr_comm_unique_id(char id[MVR_UNIQUE_ID_BYTES]);
int  mvr_ctx_comm_init(mvr_ctx *ctx, const char id[MVR_UNIQUE_ID_BYTES], int rank, int world);   /* ncclCommInitRank: collective */
int  mvr_ctx_comm_destroy(mvr_ctx *ctx);
/* rank / world of the context and the rank count RCCL itself reports for its communicator (0: none) */
int  mvr_ctx_comm_info(mvr_ctx *ctx, int *rank, int *world, int *rccl_ranks);
const char *mvr_rccl_library(void);      /* what was loaded, or why nothing was */
/* The library keeps the device and pinned-host blocks its contexts free in a process-wide cache (by device and size class) and
 * serves later allocations from it: a context's ~60 allocations cost 0.6-0.7 ms of a registration's first pass, and every free is a
 * device-wide synchronisation.  mvr_pool_trim gives all idle blocks back to the runtime and returns the bytes freed; stats (may be
 * NULL) receives {bytes still cached, requests served from the cache, requests that went to the runtime}.  Environment: MVR_POOL=0
 * switches the cache off, MVR_POOL_CAP_MB (default 16384) bounds what it holds. */
unsigned long long mvr_pool_trim(unsigned long long stats[3]);
/* the partition: edge_queries[e] = source points of edge e; rank's ranges (edge, first query, count), at most ne of them */
int  mvr_ring_segments(int ne, const size_t *edge_queries, int world, int rank, int *seg_edge, size_t *seg_begin, size_t *seg_count, int *n_seg);
/* arguments as mvr_ring_run; every rank passes the same poses in and gets the same poses out.  timing_ms = {enqueue,
 * wait for the GPU incl. the all-reduce, host solve}, summed over the passes.  A rank whose local work fails in a pass
 * reports it THROUGH that pass's all-reduce (a status row of the table), so all ranks return from the same pass -- the
 * failing rank with its own status, its peers with MVR_E_RCCL; a rank whose peers never arrive gives up after
 * "wait_timeout_ms" (mvr_ctx_tune, default 60 000), aborts the communicator (ncclCommAbort) and returns MVR_E_RCCL
 * (the context then refuses the multi-GPU entry points until mvr_ctx_comm_destroy + a new communicator).  The replicated
 * solve relies on RCCL's all-reduce handing every rank the SAME bits (one reduction order per element, whatever the
 * algorithm: ring and tree both reduce an element once and broadcast the result). */
int  mvr_ring_run_sharded(mvr_ctx *ctx, int n_steps, int n_views, const int *posed_slots, const int *raw_slots, int ne,
                          const int *edge_src, const int *edge_tgt, double max_dist, int reciprocal, int fma, const double origin[3],
                          int lum_iterations, double *poses, double *lum_pose, float *pair_T, double *pair_n, double *pair_mse,
                          int *lum_iters, double *rows, double *timing_ms);
/* one rank's half of ONE pass, for launchers that reduce the table themselves (MPI, a test): pose what `rank` of `world`
 * needs and return ITS rows of the edge table (zero rows for edges it does not touch; rows of a split edge are partial
 * sums) on the host.  The sum over the ranks' tables, fed to mvr_ring_host_step, is the pass. */
int  mvr_ring_rows_sharded(mvr_ctx *ctx, int rank, int world, int n_views, const int *posed_slots, const int *raw_slots, int ne,
                           const int *edge_src, const int *edge_tgt, double max_dist, int reciprocal, int fma, const double origin[3],
                           const double *poses, double *rows);
/* PROJECTION of one rank's share on ONE GPU (measurement aid, tools/rank_share_bench.py; no reference counterpart): after
 * mvr_ctx_project(ctx, world, rank, peer_rows, ne) mvr_ring_run_sharded plans and runs as `rank` of `world` -- its share of the
 * queries of registrator.cpp:640-651's edges, the collective of the context's communicator (a world of one) really issued --
 * and adds peer_rows ([ne][32], what the absent ranks would contribute: the sum of their mvr_ring_rows_sharded tables at the
 * poses the run starts from) to the table behind the all-reduce, so that the solve sees a whole table.  What it times is a
 * rank's critical path (its chain, the collective's launch, the replicated solve), NOT the fabric.  world <= 1: off. */
int  mvr_ctx_project(mvr_ctx *ctx, int world, int rank, const double *peer_rows, int ne);
/* (b) ONE process, all GPUs: n_dev contexts (device_ids NULL: 0 .. n_dev-1) + ncclCommInitAll;
 *     mvr_world_ring_run drives one host thread per device through mvr_ring_run_sharded and checks that every rank
 *     arrived at bit-identical poses.  mvr_world_upload puts a cloud into the same slot of every rank. */
typedef struct mvr_world mvr_world;
int  mvr_world_create(mvr_world **w, int n_dev, const int *device_ids);
int  mvr_world_destroy(mvr_world *w);
int  mvr_world_size(const mvr_world *w);
mvr_ctx *mvr_world_ctx(mvr_world *w, int rank);
const char *mvr_world_last_error(const mvr_world *w);
int  mvr_world_upload(mvr_world *w, int slot, const float *xyz, size_t n, size_t stride_bytes);
int  mvr_world_ring_run(mvr_world *w, int n_steps, int n_views, const int *posed_slots, const int *raw_slots, int ne, const int *edge_src,
                        const int *edge_tgt, double max_dist, int reciprocal, int fma, const double origin[3], int lum_iterations,
                        double *poses, double *lum_pose, float *pair_T, double *pair_n, double *pair_mse, int *lum_iters, double *rows,
                        double *timing_ms);
/* pcl::getTransformation(x,y,z,roll,pitch,yaw) -> column-major 4x4. */
void mvr_pose_to_mat4(const double pose[6], double T[16]);

/* ---- turntable prior (PointCloud::initRotation point_cloud.cpp:400-413,
 *      Registrator::getRotationMatrix registrator.cpp:331-342) --------------- */
double mvr_turntable_angle(int view, int n_views);
/* Registrator::refineAxis (registrator.cpp:402-455, with math_solvers::least_squares, math_solvers.cpp:12-38): the
 * turntable axis and pivot that best explain the poses of the n REGISTERED views (poses: n x 16, column-major,
 * column-vector convention).  Axis: least squares of (R_i - I) x = 0 with the row u + v + w = 1, normalised; pivot:
 * least squares of (R_i - I) p = -t_i with the row p_y = pivot_y (the current pivot's y).  Results are floats, as the
 * reference's osg::Vec3.  MVR_E_ARG for n <= 0, MVR_E_SINGULAR for a rank-deficient system (outputs untouched). */
int    mvr_refine_axis(int n, const double *poses, float pivot_y, float axis_out[3], float pivot_out[3]);
void   mvr_axis_rotation(const double pivot[3], const double axis[3], double angle, double T[16]);
void   mvr_mat4d_mul(const double A[16], const double B[16], double C[16]);
void   mvr_mat4f_mul(const float A[16], const float B[16], float C[16]);

/* ---- instrumentation --------------------------------------------------------- */
/* per-kernel-family device time measured with HIP events on the ctx stream.
 * family: 0 = nn (brute-force NN, fwd + reciprocal), 1 = reductions (K5/K6/K8),
 * 2 = transform/copy, 3 = glue (mark/compact/weights). */
enum { MVR_K_NN = 0, MVR_K_REDUCE = 1, MVR_K_XFORM = 2, MVR_K_GLUE = 3,
       MVR_K_NN_GRID = 4,   /* the grid search of the fused pass: one thread per bounded query (work = point-pair evaluations) */
       MVR_K_NN_WIDE = 5,   /* its stragglers: wide bounded queries (a wave each) and flagged query sets (culled kernel over a set list) */
       MVR_K_COUNT = 6 };
/* knobs of the exact NN search.  "nn_mode": 1 (default) = spatially culled
 * kernel, 0 = brute-force kernel (also MVR_NN_MODE in the environment);
 * brute-force launch shape: "nn_q" (queries per lane: 2,4,6,8), "nn_sub"
 * (min-tracking sub-tile: 16,32,64), "nn_blocks_per_cu" (1..5); culled kernel:
 * "cull_q" (64-query groups per set: 1,2; 0 = auto), "cull_w" (waves sharing one
 * query set: 1,2,4; 0 = by launch size; also MVR_CULL_W); "pair_fused" (1, default: in culled mode
 * mvr_pair_moments2_batch runs every stage of all pairs as ONE launch; 0: one pair per worker stream;
 * also MVR_PAIR_FUSED), "pair_groups" (1..8, default 2: the fused pass runs its pairs in that many groups on
 * concurrent streams -- when there are at least four pairs per group -- so one group's small kernels overlap
 * another's searches; also MVR_PAIR_GROUPS),
 * "pair_streams" (worker streams, 1..16; also MVR_PAIR_STREAMS); "posed_refresh" (1, default: in culled mode
 * mvr_cloud_transform_batch also refreshes the index of the posed copies, from the sources' sorted copies;
 * 0: at the first search, by a gather; also MVR_POSED_REFRESH); "cull_slices" (1, 2, 4, 8; 0 = auto: a fused
 * launch deals each pair's query sets to the XCDs in that many interleaved slices, so that a pair's target is read
 * through that many of the eight L2s instead of all of them; 8 = every XCD visits every pair; also MVR_CULL_SLICES);
 * "seed_forward" (1, default: when a fused pass searches the very same point sets as the previous one on this context,
 * every forward search starts from the distance of its previous match; also MVR_SEED_FORWARD);
 * "ring_search" (1, default: in a fused pass every query that HAS a bound -- a forward search seeded by its previous
 * match, every reverse search -- walks a pose-invariant uniform grid over the target, one thread per query; 0: the
 * culled kernel answers everything; also MVR_RING_SEARCH), with its knobs "grid_cell_points" (points per occupied
 * cell the cell edge aims at, default 4; applies to grids built afterwards), "grid_light_rows" (rows of cells a thread
 * walks itself, default 24; wider balls leave the thread-per-query walk; "grid_light_rows_lone", default 12: the same in a
 * launch of one pair), "grid_wide" (1, default: a wide BOUNDED query
 * gets a wave of its own; 0: flagged for the culled kernel), "grid_cluster" (a wave of the walk with at least this many
 * wide queries hands them all to the culled kernel, default 8; 65: never), "grid_sets" (1, default: the 64-query
 * sets that hold a flagged query are answered a block per set over the grid while a set's union of balls is a few hundred
 * rows of cells; 2: always; 0: by the culled kernel over the set list), "grid_tail" (1, default: that launch and the
 * wave-per-query launch of a forward pass are one launch), "cull_list" (1, default: the culled kernel
 * visits only the query sets the walk listed; 0: a block per set), "cull_list_w" (waves per listed set: 1, 2, 4),
 * "grid_lanes" (lanes sharing a query: 1 (default), 2, 4, 8), "grid_wide_waves" (waves per CU of the wave-per-query
 * launch), "fused_mark" (1, default: the forward searches record the start bounds of the reverse searches themselves when
 * they are the grid walk; 2: always; 0: never -- a separate launch re-reads the keys), "grid_debug" (1: every pass prints how its queries split; synchronises);
 * "pipeline" (1, default: mvr_ring_run / mvr_ring_run_sharded enqueue pass k+1's whole launch chain while pass k runs,
 * behind a hipStreamWaitValue32 gate the host opens after its solve, the poses reaching the kernels through a device table,
 * once a pass has run without allocating or waiting; 0: every pass is enqueued after the previous solve; also MVR_PIPELINE);
 * "grid_probe" (1, default: a query of the grid walk whose ball is wide first looks into the 2 x 2 x 2 cells nearest to it -- a
 * point found there is a tighter, valid bound; what a pass after a large motion needs; 0: off), "grid_probe_rows" (the probe is
 * made for balls of more than this many rows of cells, default 12, at most "grid_light_rows"; a probed query whose new ball lies inside the
 * probed cells is answered by the probe alone);
 * "lazy_super" (1, default: a posing launch that also writes a view's grid-ordered coordinates leaves its super boxes -- read by
 * the culled kernel alone -- to the first culled launch that follows, if any; 0: refreshed by every posing launch);
 * the aligns of the sequential mode: "seq_search" (mvr_icp_align of a posed scan against a model made of posed scans: 1, default:
 * the reverse searches walk the source scan's cell grid; 2: the forward search goes through the merged scans' grids as well; 3: the
 * forward search walks ONE grid over the model's own coordinates -- built from scratch per align, exact, measured no faster than the
 * culled kernel even on a free grid: DESIGN.md 4.5 --, with "seq_cell_points" points per cell (default 4) and the flagged query sets
 * through the culled kernel's listed-set launch ("seq_model_tail" 1, default) or the grid's set kernel (0); 0: the culled kernel both ways), "seq_seed" (1, default: an align's forward searches start from the distance, now, of the point
 * each query matched when the same scan was last aligned on this context -- the sweeps of registrationICP, the rounds of
 * AutoReg; 0: off, and what the aligns so far have left is forgotten), "align_spin" (1, default: the last sums launch of a
 * point-to-point iteration stores the iteration's row into mapped pinned memory itself and the host spins on a sequence word;
 * 0: a copy behind the launch and hipStreamSynchronize), "seq_rider" (1, default: mvr_seq_run has the launch that poses the next
 * source refresh the grown model's index tail on the way; 0: a launch of its own inside the align), "reduce_rows" (blocks, = partial rows, per pair of the launches that carry the 29 raw sums;
 * 0, default: a quarter of the compute units, at least 32 -- the sums are added in an order that depends on it, so two runs
 * compare bit for bit only at the same value);
 * multi-GPU: "wait_timeout_ms" (how long a rank waits for a pass that contains a collective before it aborts its
 * communicator), and the test hooks "inject_fail_pass" / "inject_stall_pass" (the k-th sharded pass or iteration from now:
 * this rank's local work fails / its stream stalls in front of the collective as if a peer never arrived; -1: off).
 * Round 4: "grid_stage" (the staged walk: a wave of the grid walk copies the cells its lanes want into LDS -- LDS-DMA loads -- and
 * walks them there -- the two waves of a block one region together; 2, default: every walk launch; 1: only the launches over a scan's
 * own query order, i.e. the forward searches; 0: off; also MVR_GRID_STAGE), "grid_stage_stat" (1: counters of the staged walk's waves by outcome, read with
 * mvr_ctx_stat "stage_staged" / "stage_rows" / "stage_width" / "stage_points"), "grid_index" (a grid's cell-start table, for grids
 * built from then on: 2, default = dense, built through the compact form and expanded; 1 = compact: a directory of 32-cell segments
 * + records of the occupied ones, a tenth of the bytes, walks 8-11 % slower; 0 = the round-3 dense build; also MVR_GRID_INDEX),
 * "order_batch" (1, default: the orderings several point sets need are built in one composite sort; 0: one after the other -- the
 * same permutations), "seed_delta_um" (forward start bounds from the previous distance + the clouds' motion while that is below
 * this many micrometres; 0, default: from the old match's coordinates), "rim_cert_um" (rim certificates with this margin; 0,
 * default: off), "pipeline_multi_rank" (1: passes with a collective over more than one rank may be queued ahead of their poses
 * too; 0, default, until a recorded two-GPU run).
 * Results never depend on them. */
int  mvr_ctx_tune(mvr_ctx *ctx, const char *key, int value);
/* wall milliseconds of every pass of the LAST mvr_ring_run / mvr_ring_run_sharded on this context (from the end of the
 * previous pass's solve to the end of this one's; the first from the start of the call): what the first passes of a
 * registration cost (orderings, grids) beside the steady state.  *n = passes logged (may exceed cap). */
int  mvr_ctx_pass_log(mvr_ctx *ctx, double *ms, int cap, int *n);
/* counters of the context, by name: "piped_passes" (passes of mvr_ring_run / mvr_ring_run_sharded whose launch chain was
 * enqueued ahead of their poses, see "pipeline"), "fused_passes" (fused pair batches run so far), "blocking_events"
 * (times the library waited for its stream or re-allocated a buffer: a pass without any is in steady state) */
int  mvr_ctx_stat(mvr_ctx *ctx, const char *key, double *value);
/* diagnostics: the ordering of the cloud in `slot` (sorted position -> original index); *n = its length (0: none yet) */
int  mvr_debug_order(mvr_ctx *ctx, int slot, uint32_t *perm, size_t cap, size_t *n);
/* diagnostics of the culled kernel: {pair evaluations of the last launch,
 * running total, max tiles processed by one wave, max tiles tested by one wave} */
int  mvr_debug_counters(mvr_ctx *ctx, uint64_t out[4], int reset);
/* per-launch HIP-event timing: 0 off, 1 all kernel families, 2 the NN search kernels only */
int  mvr_prof_enable(mvr_ctx *ctx, int on);
int  mvr_prof_reset(mvr_ctx *ctx);
/* launches, total ms, point-pair evals (nn families) / bytes (others) */
int  mvr_prof_get(mvr_ctx *ctx, int family, uint64_t *launches, double *ms, double *work);

/* ---- synthetic turntable scans (host only, no GPU needed) --------- */
typedef struct {
  int      n_views;        /* V: views at 2*pi/V                            */
  uint64_t seed;           /* base seed; view v uses seed + v               */
  double   noise_sigma;    /* along-normal noise, mm                        */
  double   pivot[3];       /* true turntable pivot (point_cloud.cpp:102)    */
  double   axis[3];        /* true turntable axis  (point_cloud.cpp:103)    */
} mvr_synth_params;
void mvr_synth_default(mvr_synth_params *p, int n_views, int config_id);
/* fills xyzw (n*4 floats, w=1) and, if non-NULL, normals (n*4 floats, w=0) of
 * the scan of `view`: object rotated by +view*2pi/V about (pivot, axis). */
int  mvr_synth_view(const mvr_synth_params *p, int view, size_t n, float *xyzw, float *normals);
/* the mis-calibrated prior handed to initRotation: pivot + (1.5,-1,2) mm and
 * the axis tilted by 0.5 deg about x. */
void mvr_synth_prior(const mvr_synth_params *p, double pivot[3], double axis[3]);

#ifdef __cplusplus
}
#endif
#endif /* MVR_HIP_H */
