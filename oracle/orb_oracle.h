/*
 * orb_oracle.h -- C interface of the CPU ORACLE for the ORB extract + match hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a dependency-free CPU restatement of the
 * reference algorithm (elkuno213/orbslam3_cpp_fork: src/ORBextractor.cc,
 * src/ORBmatcher.cc, src/Frame.cc grid queries) plus the OpenCV >= 4.5
 * primitives the reference calls (cv::resize INTER_LINEAR 8U, cv::FAST 9/16 +
 * NMS, cv::GaussianBlur 7x7 sigma 2 fixed point, cv::fastAtan2, cvRound).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link,
 * load or call it -- as the checker / timed CPU baseline, never as the product.
 *
 * PARITY STATUS: "parity unpinned".  The reference cannot be compiled in the build
 * container (OpenCV and Eigen are absent), it ships no golden vectors or
 * known-answer tests for this path (its test/ directory never constructs an
 * extractor or matcher), and OpenCV itself is an un-vendored system dependency
 * (find_package(OpenCV 4.4), CMakeLists.txt:21).  The oracle is therefore pinned
 * only by analytic known-answer tests (tests/test_oracle_kats.py) and by
 * self-consistency; restatement-vs-OpenCV equality is unverified.
 */
#ifndef ORB_ORACLE_H_
#define ORB_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 28-byte layout as cv::KeyPoint {pt.x, pt.y, size, angle, response, octave, class_id}. */
typedef struct {
  float x, y, size, angle, response;
  int   octave, class_id;
} orao_keypoint_t;

typedef struct orao_extractor orao_extractor_t;

/* ORBextractor::ORBextractor (src/ORBextractor.cc:403-465) */
orao_extractor_t* orao_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast);
void              orao_destroy(orao_extractor_t*);

/* ORBextractor::operator() (src/ORBextractor.cc:1052-1138).  Returns monoIndex (>=0), -1 on an
 * empty image, -2 on bad arguments (nIni == 0 hazard, cap too small). */
int orao_extract(orao_extractor_t*, const uint8_t* img, int rows, int cols, size_t step, int lap0, int lap1,
                 orao_keypoint_t* kps, uint8_t* desc, int cap, int* n_out);

int  orao_get_levels(const orao_extractor_t*);
void orao_get_scale_tables(const orao_extractor_t*, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2);
void orao_get_features_per_level(const orao_extractor_t*, int* out);
void orao_get_umax(const orao_extractor_t*, int* out16);
/* mvImagePyramid[level] of the last extract (interior ROI, no border). */
int orao_pyramid_level(const orao_extractor_t*, int level, uint8_t* dst, size_t dst_step, int* rows, int* cols);
/* Blurred level of the last extract (what computeDescriptors reads). */
int orao_blurred_level(const orao_extractor_t*, int level, uint8_t* dst, size_t dst_step, int* rows, int* cols);
/* Candidates fed to DistributeOctTree for `level` in the last extract, in feed order
 * (coordinates relative to minBorder, as in src/ORBextractor.cc:842-848).  Returns the count. */
int orao_level_candidates(const orao_extractor_t*, int level, int* xs, int* ys, int* resp, int cap);
/* Keypoints of `level` after DistributeOctTree + orientation (level coordinates, natural order). */
int orao_level_keypoints(const orao_extractor_t*, int level, orao_keypoint_t* out, int cap);

/* ---- primitives, exposed for known-answer tests ---- */
void  orao_resize_linear_8u(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh, size_t dstep);
void  orao_gaussian_blur7_8u(const uint8_t* src, int w, int h, size_t sstep, uint8_t* dst, size_t dstep);
int   orao_fast9_16(const uint8_t* img, int w, int h, size_t step, int threshold, int* xs, int* ys, int* scores, int cap);
float orao_fast_atan2(float y, float x);
int   orao_cv_round_f(float v);
float orao_ic_angle(const uint8_t* img, size_t step, int x, int y);
void  orao_orb_descriptor(const uint8_t* blurred, size_t step, int x, int y, float angle_deg, uint8_t* desc32);
/* Tap-rotation contraction (src/ORBextractor.cc:105-106 as compiled): 1 = g++ -O3 -march=native on an FMA host (the reference's own
 * CMake flags; default), 0 = no contraction, 2 = the other fused form.  See orbDescriptor in orb_oracle.cpp. */
void  orao_orb_descriptor_c(const uint8_t* blurred, size_t step, int x, int y, float angle_deg, uint8_t* desc32, int contraction);
void  orao_set_tap_contraction(orao_extractor_t*, int mode);
/* DistributeOctTree (src/ORBextractor.cc:556-753): returns the number of kept points, their indices
 * into the input arrays in output (list) order.  -2 if nIni == 0. */
int orao_distribute_octtree(const float* xs, const float* ys, const float* resp, int n, int min_x, int max_x,
                            int min_y, int max_y, int n_target, int* out_idx, int cap);
/* This host's std::sort (libstdc++) on n 64-bit records ordered by (record >> 16) -- the low 16 bits ride along --, as DistributeOctTree's
 * sort of vPrevSizeAndPointerToNode (src/ORBextractor.cc:685) orders (size, UL.x) pairs.  std::sort is not stable: which of two equal
 * keys comes first is the library's business, and the quadtree's result depends on it; this pins the port's re-enactment directly. */
void orao_std_sort(unsigned long long* records, int n);
/* libm (cosf, sinf) of this host for x = deg * (float)(CV_PI/180.f), as computeOrbDescriptor uses them
 * (src/ORBextractor.cc:93,97-98).  Bulk form for the exhaustive device-port check. */
void orao_libm_sincos_deg(const float* deg, int n, float* cos_out, float* sin_out);
/* Number of float inputs `deg` in [bits_lo, bits_hi) (IEEE bit patterns of non-negative floats) for which
 * the given port (0 = plain double, 1 = FMA-contracted double) differs from this host's libm. Multi-threaded. */
long orao_sincos_port_mismatches(uint32_t bits_lo, uint32_t bits_hi, int fma_variant, int nthreads,
                                 uint32_t* first_bad /* up to 16 entries */, int* n_first_bad);

/* Frame::ComputeStereoMatches (src/Frame.cc:944-1113) on the pyramids of the last extract of `left` / `right`:
 * row-band candidates, best Hamming below TH_HIGH, 11x11 SAD over +-5 px on the keypoint's pyramid level, parabola
 * sub-pixel fit, median-distance cut.  Writes mvuRight / mvDepth (-1 = no match).  Returns the number of matches kept. */
int orao_stereo_matches(const orao_extractor_t* left, const orao_extractor_t* right, const orao_keypoint_t* kl, const uint8_t* dl,
                        int nl, const orao_keypoint_t* kr, const uint8_t* dr, int nr, float mb, float mbf, float* u_right,
                        float* depth);

/* ---- matcher ---- */
/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2075-2089) */
int orao_hamming(const uint8_t* a32, const uint8_t* b32);
/* ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:2038-2071) on bin counts. */
void orao_three_maxima(const int* counts, int L, int* ind1, int* ind2, int* ind3);
/* ORBmatcher::SearchByBoW, both overloads (src/ORBmatcher.cc:237-421 mono branch, 784-914), over flattened
 * FeatureVector groups.  Group g pairs query indices q_idx[q_off[g]..q_off[g+1]) with train indices
 * t_idx[t_off[g]..t_off[g+1]).  q_idx/t_idx may be NULL when ngroups == 1 (identity lists).
 * q_valid / t_valid: 1 = has a good MapPoint (NULL = all valid).  strict_lt: 0 -> best <= th (Frame
 * overload), 1 -> best < th (KeyFrame overload).  Outputs: match_t_of_q[nq] and match_q_of_t[nt]
 * (-1 = none; either may be NULL).  Returns nmatches. */
int orao_match_groups(const uint8_t* q, int nq, const uint8_t* t, int nt, int ngroups, const int* q_off,
                      const int* q_idx, const int* t_off, const int* t_idx, const uint8_t* q_valid,
                      const uint8_t* t_valid, const float* q_angle, const float* t_angle, int th, int strict_lt,
                      float ratio, int check_ori, int* match_t_of_q, int* match_q_of_t);
/* SearchByBoW(KeyFrame*, Frame&) for a two-camera frame (F.Nleft != -1, src/ORBmatcher.cc:305-333, 362-389): train features =
 * left keys [0, t_nleft) then right keys.  match_q_of_t[nt] = the query that claimed the train feature (vpMapPointMatches), -1 = none. */
int orao_match_groups_two_cameras(const uint8_t* q, int nq, const uint8_t* t, int nt, int t_nleft, int ngroups, const int* q_off,
                                  const int* q_idx, const int* t_off, const int* t_idx, const uint8_t* q_valid, const float* q_angle,
                                  const float* t_angle, int th, float ratio, int check_ori, int* match_q_of_t);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:666-782) with the Frame grid restated
 * (src/Frame.cc:524-553, 788-877; 64x48 grid over [min_x,max_x]x[min_y,max_y]).
 * kp1/kp2: keypoints (x, y, angle, octave); prev_matched: nq x 2 floats, updated in place
 * (:775-779).  matches12[n1] out.  Returns nmatches. */
int orao_match_window(const uint8_t* d1, const orao_keypoint_t* kp1, int n1, const uint8_t* d2,
                      const orao_keypoint_t* kp2, int n2, float* prev_matched, float min_x, float max_x, float min_y,
                      float max_y, int window, int th, float ratio, int check_ori, int* matches12);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBmatcher.cc:1684-1908,
 * CurrentFrame.Nleft == -1) at array level.  The host side of the reference loop -- MapPoint present, not an outlier,
 * Tcw * x3Dw, invzc >= 0, camera projection inside [mnMinX, mnMaxX] x [mnMinY, mnMaxY] (:1706-1729) -- is the caller's:
 * last_valid[i] says whether point i reaches the window search, last_uv / last_invz carry uv and invzc.
 * dir: 0 = neither, 1 = bForward, 2 = bBackward (:1701-1702).  cur_u_right NULL = monocular (mvuRight < 0 everywhere);
 * cur_occupied[i2] = (CurrentFrame.mvpMapPoints[i2] && Observations() > 0) before the call (NULL = none).  Every match
 * blocks its feature for later points, i.e. all tracked MapPoints are assumed to have observations (the zero-observation
 * temporal points of the stereo/RGB-D odometry are the caller's to handle).
 * Outputs: cur_of_last[n1] (feature that received last-frame point i, -1), last_of_cur[n2].  Returns nmatches. */
int orao_search_by_projection(const uint8_t* last_desc, const float* last_uv, const float* last_invz, const int* last_octave,
                              const float* last_angle, const uint8_t* last_valid, int n1, const uint8_t* cur_desc,
                              const orao_keypoint_t* cur_kp, const float* cur_u_right, const uint8_t* cur_occupied, int n2,
                              const float* scale_factors, float th, int dir, float mbf, float min_x, float max_x, float min_y,
                              float max_y, int th_dist, int check_ori, int* cur_of_last, int* last_of_cur);

/* ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)
 * (src/ORBmatcher.cc:38-227, F.Nleft == -1) at array level.  mp_valid[i] = mbTrackInView && !isBad() && !(far) (:50-62);
 * mp_uv = (mTrackProjX, mTrackProjY), mp_proj_xr = mTrackProjXR (needed only with cur_u_right), mp_level = mnTrackScaleLevel,
 * mp_radius = RadiusByViewingCos(mTrackViewCos) [* th if th != 1] (:67-71).  Best and second-best candidate, TH_HIGH, and the
 * ratio veto only when both sit on the same pyramid level (:137-142).  Matched features are closed to later points. */
int orao_search_by_projection_map(const uint8_t* mp_desc, const float* mp_uv, const float* mp_proj_xr, const int* mp_level,
                                  const float* mp_radius, const uint8_t* mp_valid, int n1, const uint8_t* cur_desc,
                                  const orao_keypoint_t* cur_kp, const float* cur_u_right, const uint8_t* cur_occupied, int n2,
                                  const float* scale_factors, float ratio, float min_x, float max_x, float min_y, float max_y, int th_dist,
                                  int* cur_of_mp, int* mp_of_cur);
/* The same search for a two-camera frame (F.Nleft != -1, src/ORBmatcher.cc:139-146, 152-226); see orb_oracle.cpp for the argument
 * conventions.  mp_of_cur[n]: F.mvpMapPoints after the call. */
int orao_search_by_projection_map_two_cameras(const uint8_t* mp_desc, int n1, const float* uv_l, const int* level_l, const float* radius_l,
                                              const uint8_t* valid_l, const float* uv_r, const int* level_r, const float* radius_r,
                                              const uint8_t* valid_r, const uint8_t* cur_desc, const orao_keypoint_t* cur_kp, int n, int nleft,
                                              const uint8_t* cur_occupied, const int* left_to_right, const int* right_to_left,
                                              const float* mvScaleFactors, float mfNNratio, float mnMinX, float mnMaxX, float mnMinY,
                                              float mnMaxY, int th_dist, int* mp_of_cur);


/* The windowed best-candidate search shared by SearchByProjection(KeyFrame*, Sim3f&, ...) (src/ORBmatcher.cc:423-538, :540-664),
 * Fuse (:1149-1328, :1330-1452) and the two directions of SearchBySim3 (:1499-1658): KeyFrame::GetFeaturesInArea
 * (src/KeyFrame.cc:780-838) with radius th * scale[q_level], keypoint levels [q_level - 1, q_level], first minimum, accepted
 * when <= th_dist.  inv_level_sigma2 != NULL adds Fuse's reprojection gate (5.99; 7.8 with t_u_right[idx] >= 0 and q_ur);
 * claim = 1 skips t_occupied features and closes a matched feature to later points (vpMatched), claim = 0 searches every
 * point on its own.  t_of_q[n1] out; returns the number of matched points. */
int orao_search_in_windows(const uint8_t* q_desc, const float* q_uv, const float* q_ur, const int* q_level, const uint8_t* q_valid, int n1,
                           const uint8_t* t_desc, const orao_keypoint_t* t_kp, const float* t_u_right, const uint8_t* t_occupied, int n2,
                           const float* scale_factors, const float* inv_level_sigma2, float th, int claim, float min_x, float max_x,
                           float min_y, float max_y, int th_dist, int* t_of_q);

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:916-1147), keyframes of one pinhole camera (mpCamera2 == NULL), over the
 * flattened common nodes of the two FeatureVectors (as orao_match_groups).  mp1 / mp2 = feature already has a MapPoint, stereo1 /
 * stereo2 = mvuRight >= 0 (NULL = none); F12[9] row major and ep[2] as the reference computes them (:931-934,
 * src/CameraModels/Pinhole.cpp:152-156); level_sigma2 = mvLevelSigma2.  vMatches12[n1] out (-1 = none); returns nmatches. */
int orao_search_for_triangulation(const uint8_t* d1, const orao_keypoint_t* kp1, const uint8_t* mp1, const uint8_t* stereo1, int n1,
                                  const uint8_t* d2, const orao_keypoint_t* kp2, const uint8_t* mp2, const uint8_t* stereo2, int n2, int ngroups,
                                  const int* q_off, const int* q_idx, const int* t_off, const int* t_idx, const float* F12, const float* ep,
                                  const float* scale_factors, const float* level_sigma2, int only_stereo, int coarse, int th_low,
                                  int check_ori, int* matches12);

/* Frame::UndistortKeyPoints (src/Frame.cc:886-913): cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK) on the keypoint
 * coordinates (OpenCV cvUndistortPointsInternal, 5 iterations, doubles); K = (fx, fy, cx, cy) floats, dist = k1 k2 p1 p2 [k3]
 * floats (ndist = 4 or 5).  dist[0] == 0 copies the keypoints (:887-890).  The other KeyPoint fields are copied. */
void orao_undistort_keypoints(const orao_keypoint_t* kps, int n, const float K[4], const float* dist, int ndist, orao_keypoint_t* out);
/* Frame::ComputeImageBounds (src/Frame.cc:915-943): out = {mnMinX, mnMaxX, mnMinY, mnMaxY}. */
void orao_image_bounds(int cols, int rows, const float K[4], const float* dist, int ndist, float out[4]);

/* ---- DBoW2 vocabulary tree: ORBVocabulary::transform (N2) ----
 * TemplatedVocabulary<FORB::TDescriptor, FORB> restated for the configuration ORB-SLAM3 ships (ORBvoc.txt header
 * "10 6 0 0": scoring L1_NORM, weighting TF_IDF).  Node i + 1 of the tree is entry i of the arrays (node 0 = root), in the
 * order of the lines of the text format (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1337-1420: "parent isLeaf 32 bytes
 * weight"); word ids number the leaves in that order (:1402-1409). */
typedef struct orao_vocab orao_vocab_t;
orao_vocab_t* orao_vocab_create(int k, int L, int n_nodes, const int* parent, const uint8_t* is_leaf, const uint8_t* desc32,
                                const double* weight);
void          orao_vocab_destroy(orao_vocab_t*);
/* transform(features, BowVector&, FeatureVector&, levelsup) (TemplatedVocabulary.h:1127-1194, single-feature descent
 * :1218-1259; BowVector::addWeight / normalize(L1), BowVector.cpp:34-78; FeatureVector::addFeature, FeatureVector.cpp:34-49).
 * Outputs: BowVector as (word id ascending, value); FeatureVector as node ids ascending with node_off[n_nodes_out + 1]
 * offsets into feat_idx.  word_of_feat / node_of_feat (may be NULL): per-feature descent result. Returns 0. */
int orao_vocab_transform(const orao_vocab_t*, const uint8_t* desc32, int n, int levelsup, uint32_t* word_ids, double* word_vals,
                         int* n_words, uint32_t* node_ids, int* node_off, uint32_t* feat_idx, int* n_nodes_out,
                         uint32_t* word_of_feat, uint32_t* node_of_feat);

#ifdef __cplusplus
}
#endif
#endif
