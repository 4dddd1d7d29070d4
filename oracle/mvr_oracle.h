/*
 * oracle/mvr_oracle.h -- CPU ORACLE. TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT.
 *
 * A plain-C restatement of the reference's ICP hot path, used ONLY as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under multi-view-registration_amd/ or include/ may include, link or
 * call this file.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in a third-party
 * dependency that is absent from /root/reference and from this image: PCL
 * (`find_package(PCL REQUIRED common io registration kdtree search)`,
 * mvr/CMakeLists.txt:10, version NOT pinned; API usage implies PCL >= 1.7.0)
 * with FLANN and Eigen 3 underneath.  The reference holds no tests, fixtures
 * or golden vectors, so this restatement follows PCL's
 * published algorithm anchored on the reference's own call
 * sites, and is cross-checked only against independent implementations that
 * exist offline (scipy cKDTree for exact 1-NN, numpy SVD for Kabsch).
 *
 * Points are 16-byte PointXYZ records {x,y,z,w} exactly as pcl::PointXYZ
 * (mvr/include/types.h:14); all `const float *pts` arguments are arrays of
 * such records (stride 4 floats).  Poses are 4x4 column-major, column-vector
 * convention (Eigen::Matrix4f layout; mvr/include/types.h:20-50 is the
 * transposing bridge to OSG's row-vector matrices).
 */
#ifndef MVR_ORACLE_H
#define MVR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pcl::Correspondence {index_query, index_match, distance(=squared)} */
typedef struct { int32_t query; int32_t match; float dist2; } orc_corr;

/* Parameters of one pcl::IterativeClosestPoint::align, as set at
 * mvr/src/registrator.cpp:551-560 (and :768-771, :901-904). */
typedef struct {
  int    use_reciprocal;          /* setUseReciprocalCorrespondences     */
  double max_corr_dist;           /* setMaxCorrespondenceDistance        */
  int    max_iterations;          /* setMaximumIterations                */
  double transformation_epsilon;  /* setTransformationEpsilon            */
  double euclidean_fitness_eps;   /* setEuclideanFitnessEpsilon          */
  int    fma_dist;                /* 0: d2=(dx*dx+dy*dy)+dz*dz rounded per op (spec) ; 1: fma chain */
  int    use_kdtree;              /* 0: brute force NN ; 1: kd-tree NN (identical results) */
} orc_icp_params;

enum { ORC_CONV_NOT = 0, ORC_CONV_ITERATIONS = 1, ORC_CONV_TRANSFORM = 2,
       ORC_CONV_ABS_MSE = 3, ORC_CONV_REL_MSE = 4, ORC_CONV_NO_CORRESPONDENCES = 5 };

typedef struct {
  int    iterations;
  int    converged;
  int    state;        /* ORC_CONV_* */
  int    n_corr;       /* correspondences of the last iteration */
  double mse;          /* mean squared corr. distance of the last iteration */
  double evals;        /* distance evaluations a brute-force search would do */
} orc_icp_stats;

void   orc_icp_default_params(orc_icp_params *p);

/* float L2, accumulated x,y,z. */
float  orc_dist2(const float *a, const float *b, int fma);

/* K1: PCL transformPointCloud / ICP::transformCloud semantics (float):
 * x' = ((T00*x + T01*y) + T02*z) + T03, no contraction; w := 1. in==out ok. */
void   orc_transform_f32(const float T[16], const float *in, float *out, size_t n);
/* a1: PointCloud::getTransformedPoints (mvr/src/point_cloud.cpp:290-303):
 * osg::Matrixd::preMult(Vec3f) in double incl. the perspective divide, then
 * cast to float.  T is the column-vector 4x4 (transpose of the OSG matrix). */
void   orc_transform_f64(const double T[16], const float *in, float *out, size_t n);

/* exact 1-NN, ties -> lowest index. idx = UINT32_MAX, d2 = +inf when nt == 0 */
void   orc_nn_brute (const float *q, size_t nq, const float *t, size_t nt, int fma,
                     uint32_t *idx, float *d2);
void   orc_nn_kdtree(const float *q, size_t nq, const float *t, size_t nt, int fma,
                     uint32_t *idx, float *d2);

/* a5: CorrespondenceEstimation::determineReciprocalCorrespondences
 * (mvr/src/registrator.cpp:496-502, 644-649). out must hold ns entries.
 * reciprocal == 0 gives determineCorrespondences (one-way). */
size_t orc_correspondences(const float *src, size_t ns, const float *tgt, size_t nt,
                           double max_dist, int reciprocal, int fma, int use_kdtree,
                           orc_corr *out);
/* multi-threaded (OpenMP over queries) kd-tree variant: same output; a courtesy CPU baseline */
size_t orc_correspondences_mt(const float *src, size_t ns, const float *tgt, size_t nt, double max_dist,
                              int reciprocal, int fma, int threads, orc_corr *out);

/* a6: TransformationEstimationSVD (Umeyama, no scaling). Returns 0, or -1
 * when m < 3.  mom (may be NULL) receives the 20 moments:
 * [0]=M, [1..3]=mean src, [4..6]=mean tgt, [7]=mean d2, [8..16]=Sigma row-major
 * (dst*src^T / M), [17..19] = singular values. */
int    orc_umeyama(const float *src, const float *tgt, const orc_corr *c, size_t m,
                   float T[16], double *mom);
/* the same estimate in Eigen's own arithmetic (float sums, sequential means, optionally blocked product): a model
 * used to measure the distance between the f64-accumulating oracle and the f32 reference arithmetic. */
int    orc_umeyama_f32(const float *src, const float *tgt, const orc_corr *c, size_t m, int block, float T[16]);
/* host-side part only: from the 17 moments to T. */
void   orc_umeyama_from_moments(const double mean_src[3], const double mean_tgt[3],
                                const double sigma[9], float T[16], double sv[3]);
void   orc_svd3(const double A[9], double U[9], double S[3], double V[9]); /* row-major */

/* a4: IterativeClosestPoint::align. out may alias src. */
int    orc_icp_align(const float *src, size_t ns, const float *tgt, size_t nt,
                     const orc_icp_params *p, float *out, float T[16],
                     orc_icp_stats *st);

/* EXTENSION (BASELINE config 2; NO counterpart in the reference):
 * pcl::registration::TransformationEstimationPointToPlaneLLS on the same
 * correspondences: rows a = [p x n_q, n_q], d = n_q . (q - p), x = (A^T A)^-1 A^T d,
 * T = Rz(x2) Ry(x1) Rx(x0) + (x3,x4,x5).  tnrm: target normals (stride 4 floats).
 * Returns 0, -1 for < 3 pairs, -2 for a singular system. */
int    orc_p2plane(const float *src, const float *tgt, const float *tnrm, const orc_corr *c, size_t m,
                   float T[16], double *sums29);
/* orc_icp_align with the point-to-plane estimator instead of Umeyama. */
int    orc_icp_align_p2plane(const float *src, size_t ns, const float *tgt, const float *tnrm, size_t nt,
                             const orc_icp_params *p, float *out, float T[16], orc_icp_stats *st);

/* a8: Registration::getFitnessScore(max_range). input = cloud handed to
 * setInputSource; T = final transformation. */
double orc_fitness(const float *input, size_t ns, const float *tgt, size_t nt,
                   const float T[16], double max_range, int fma, int use_kdtree);

/* 4x4 helpers (column-major). */
void   orc_mat4f_mul(const float A[16], const float B[16], float C[16]);   /* C=A*B float */
void   orc_mat4d_mul(const double A[16], const double B[16], double C[16]);
/* a2: Registrator::getRotationMatrix (mvr/src/registrator.cpp:331-342) as a
 * column-vector matrix: x' = R(angle,axis) (x - pivot) + pivot. */
void   orc_axis_rotation(const double pivot[3], const double axis[3], double angle,
                         double T[16]);
/* a2: PointCloud::initRotation angle (mvr/src/point_cloud.cpp:409), generalised
 * from 12 views / 30 deg to n_views. */
double orc_turntable_angle(int view, int n_views);

/* a11: pcl::registration::LUM. */
void   orc_pose_to_mat4(const double pose[6], double T[16]);  /* pcl::getTransformation */
/* computeEdge: sums over correspondences. Returns number of valid pairs.
 * MM row-major 6x6, MZ 6, ss. cinv/cinvd are NOT divided here. */
size_t orc_lum_edge(const float *src, const float *tgt, const orc_corr *c, size_t m,
                    const double pose_s[6], const double pose_t[6],
                    double MM[36], double MZ[6], double *ss);
/* LUM::compute on a graph of n vertices and ne edges (es[e] -> et[e]).
 * clouds[v] points to vertex v's points; corr[e]/ncorr[e] the edge's
 * correspondences. poses: n*6, in/out (vertex 0 stays fixed). Returns the
 * number of iterations performed. */
int    orc_lum_compute(int n, const float *const *clouds, int ne, const int *es,
                       const int *et, const orc_corr *const *corr, const size_t *ncorr,
                       int max_iterations, double convergence_threshold, double *poses);

/* PointCloud::denoise (mvr/src/point_cloud.cpp:423-465, graph from :467-500).  The reference
 * links the points by the Delaunay edges no longer than triangle_length (sqrt of the squared distance, in double,
 * `> threshold -> skip`), takes connected components (boost: numbered by their smallest point index) and keeps the
 * components with at least segment_threshold points, component after component, points in index order.
 * Restated WITHOUT a triangulation: the Euclidean minimum spanning tree is a subgraph of the Delaunay triangulation,
 * so two points are joined by Delaunay edges <= r exactly when they are joined in the graph of ALL pairs <= r --
 * the components are identical (points in general position; exact duplicates, which CGAL collapses into one vertex,
 * are simply members of their component here).  out_index receives the original indices of the kept points in output
 * order; label (optional, n entries) the smallest index of each point's component.  Returns the number kept. */
size_t orc_denoise(const float *pts, size_t n, int segment_threshold, double triangle_length,
                   uint32_t *out_index, uint32_t *label, size_t *n_components);

/* LUM::incidenceCorrection (6x6 row-major): see the derivation at its definition. */
void   orc_lum_incidence(const double pose[6], double out[36]);

/* Registrator::refineAxis (mvr/src/registrator.cpp:402-455) with math_solvers::least_squares
 * (LAPACK dgels = Householder QR; mvr/src/math_solvers.cpp:12-38).  poses: n registered poses, 4x4 column-major,
 * column-vector convention.  Returns 0, or -1 (nothing written) for n == 0 / a rank-deficient system. */
int    orc_refine_axis(int n, const double *poses, float pivot_y, float axis_out[3], float pivot_out[3]);

/* dense solve helpers (exposed for tests) */
int    orc_solve_dense(int n, double *A /*row-major, destroyed*/, double *b /*in: rhs, out: x*/);
int    orc_invert6(const double A[36], double Ainv[36]);

#ifdef __cplusplus
}
#endif
#endif
