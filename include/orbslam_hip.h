/* ============================================================================
 * orbslam_hip.h -- C ABI of the MI355X (gfx950) ORB front-end and bundle-adjustment
 * back-end.  This is the drop-in boundary: the reference has no plugin/FFI layer,
 * its hot path is three C++ classes called directly, so each entry point below
 * names the reference member it replaces (file:line relative to the reference
 * tree).  C++ shims with the reference's own class names and signatures live in
 * ceres_mono_orb_slam2_amd/csrc/compat/ and call nothing but these functions.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ORBHIP_E* code on error;
 *     orbhip_last_error() gives a message for the calling thread's last failure;
 *   - the caller owns every buffer; handles are not thread-safe, distinct handles
 *     are; `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - "_device" entry points take device pointers and only ENQUEUE work on
 *     `stream` (no host synchronisation); the others take host pointers and return
 *     after the result is in host memory;
 *   - there is no CPU fallback: without a HIP device every compute entry point
 *     fails with ORBHIP_ENODEV.
 * ========================================================================== */
#ifndef ORBSLAM_HIP_H
#define ORBSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBHIP_OK 0
#define ORBHIP_EINVAL (-1)    /* bad argument */
#define ORBHIP_ENODEV (-2)    /* no HIP device / HIP runtime error */
#define ORBHIP_ENOMEM (-3)
#define ORBHIP_ECAP (-4)      /* caller's output capacity too small */
#define ORBHIP_EOVERFLOW (-5) /* internal candidate capacity exceeded (cannot happen for images up to 4095 x 4095: the candidate arrays
                                 are sized for the worst case; only the ORBHIP_KEYCAP test hook lowers them) */
#define ORBHIP_ENUMERIC (-6)  /* linear solve failed */
#define ORBHIP_ETIMEOUT (-7)  /* a workgroup of a persistent (flag-linked) kernel waited longer than its time limit for another
                                 one: the device is oversubscribed or hung.  Never reported as a numerical failure: the solve
                                 ends with ba_summary.termination 7 and the outputs hold the last accepted iterate */

const char* orbhip_last_error(void);
int orbhip_device_count(void);
/* library version / build tag */
const char* orbhip_version(void);
/* HIP device used by the entry points that take no handle (matcher host-pointer calls, all ba_* calls);
 * default 0.  hipSetDevice is per host thread, so every such call re-selects this device itself. */
int orbhip_set_default_device(int device);
/* The reference runs Tracking, LocalMapping and the loop closer's GlobalBundleAdjustemnt on three threads that here share ONE GPU
 * (src/LocalMapping.cc:89, src/LoopClosing.cc:590,656).  Every host thread has its own HIP stream; a thread that calls this with
 * high != 0 - the Tracking thread - gets the device's greatest stream priority for the calls it makes from then on (its short
 * per-frame kernels are dispatched ahead of another thread's queued bundle-adjustment work).  Results do not depend on it.   */
int orbhip_set_thread_priority(int high);
int orbhip_get_default_device(void);
/* Copy between PINNED host memory (hipHostMalloc / hipHostRegister: mapped into the device's address space) and device memory by a
 * KERNEL on `stream` - the way the library's own host-pointer entry points move their staging blocks.  For callers that feed
 * orbx_extract_batch_device from host frames (src/Frame.cc:116 hands operator() a host cv::Mat): upload, extract + match and download
 * of consecutive batches then overlap as ordinary concurrent kernels on three streams, without the runtime's copy-engine path
 * (which stalls host threads that copy concurrently, and whose overlap with compute depends on the process's hardware-queue count:
 * DESIGN.md section 6).  Either pointer may be the host one; no synchronisation.                                               */
int orbhip_copy_pinned_async(void* dst, const void* src, size_t bytes, void* stream);

/* ---------------------------------------------------------------- extractor --
 * Replaces ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-111,
 * src/ORBextractor.cc:410-470 ctor, :1043-1105 operator()).                     */
typedef struct orbx_ctx orbx_ctx;

/* cv::KeyPoint's seven fields, 28 bytes (pt.x, pt.y, size, angle, response, octave, class_id) */
typedef struct orbx_keypoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} orbx_keypoint;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (src/ORBextractor.cc:410-470); `device` = HIP device ordinal.  Capacity: per-level quotas up to about 3200 keypoints keep
 * the octree's node arrays in LDS; larger ones (nfeatures beyond ~15000) run the same kernel on a global scratch row per
 * (frame, level), up to 32752 keypoints in one level (16-bit node indices; ORBHIP_EINVAL from the first extract call beyond
 * that - an error, never silent).  The number of FAST corners per level is not limited (candidate arrays are sized for the
 * worst case of the image geometry).                                                                                      */
int orbx_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast, int device,
                orbx_ctx** out);
int orbx_destroy(orbx_ctx* ctx);

/* GetLevels / GetScaleFactor(s) / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:63-83) plus the per-level quotas
 * mnFeaturesPerLevel (:435-446).  Any pointer may be NULL.                        */
int orbx_get_levels(const orbx_ctx* ctx);
int orbx_get_tables(const orbx_ctx* ctx, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* features_per_level);
/* upper bound on keypoints per frame (nfeatures + 3 per level): size outputs with this */
int orbx_max_keypoints(const orbx_ctx* ctx);

/* operator()(image, mask, keypoints, descriptors) for ONE host image (src/ORBextractor.cc:1043-1105).
 * img: CV_8UC1 rows of `stride` bytes.  Empty image (w<=0 or h<=0 or img==NULL) -> returns 0 with *n = 0 and
 * kps / desc32 untouched (":1046": the reference returns without touching its outputs; the C++ shim keeps the caller's
 * vector as it is in that case).  kps[cap], desc32[cap*32].                                */
int orbx_extract(orbx_ctx* ctx, const uint8_t* img, int w, int h, int stride, orbx_keypoint* kps,
                 uint8_t* desc32, int cap, int* n);

/* Batched, device-resident form: nframes images of the same size, frame f at
 * d_imgs + f*frame_stride_bytes.  Outputs (device): d_kps[nframes*cap], d_desc[nframes*cap*32],
 * d_counts[nframes] (keypoints per frame; -1 = that frame hit ORBHIP_EOVERFLOW).
 * Only enqueues on `stream`.                                                      */
int orbx_extract_batch_device(orbx_ctx* ctx, const uint8_t* d_imgs, int w, int h, int stride,
                              size_t frame_stride_bytes, int nframes, orbx_keypoint* d_kps, uint8_t* d_desc,
                              int cap, int32_t* d_counts, void* stream);

/* mvImagePyramid (include/ORBextractor.h:85) of the last extract call: copies level `level` of frame
 * `frame` to host (dense, w*h bytes).  blurred=1 returns the 7x7 Gaussian-blurred level that BRIEF
 * sampled (src/ORBextractor.cc:1085-1086).  out may be NULL to query the size.   */
int orbx_get_level_image(orbx_ctx* ctx, int frame, int level, int blurred, uint8_t* out, int* w, int* h);
/* introspection of the last call, for stage-by-stage parity tests: per-level FAST candidates in
 * candidate order as int32 triples (x, y, score) in detection-window coordinates
 * (src/ORBextractor.cc:789-829) and per-level selected keypoints (after DistributeOctTree, :834) as
 * int32 triples in the same coordinates.  out may be NULL; *n returns the count. */
int orbx_get_level_candidates(orbx_ctx* ctx, int frame, int level, int32_t* out, int cap, int* n);
int orbx_get_level_selected(orbx_ctx* ctx, int frame, int level, int32_t* out, int cap, int* n);

/* Measurement hook (no reference counterpart): when enabled, every batch call records HIP events on
 * its stream around its five stages {pyramid, FAST cells, octree, blur, orientation+BRIEF};
 * orbx_get_stage_ms synchronises on the last event, returns the per-stage sums (ms) over the calls
 * made since enabling and the number of calls, then resets.                           */
/* Which OpenCV the pixel arithmetic restates.  The reference's front-end arithmetic lives in OpenCV (cv::resize, cv::GaussianBlur,
 * cv::FAST), whose fixed-point rounding differs between versions; the library's default is what OpenCV 2.4.x / 3.0 - 3.4.1 compute
 * (the reference's README names 2.4.11 and 3.2).  An integrator whose OpenCV is newer selects its GaussianBlur here - tools/pin
 * (pin_extractor) tells which one the installed OpenCV implements:
 *   ORBX_CV_BLUR_8BIT    7x7 sigma-2 taps cvRound(g * 256) = {18,34,49,55,49,34,18} (sum 257), two passes, (sum + 2^15) >> 16
 *   ORBX_CV_BLUR_FIXED16 the "bit-exact" ufixedpoint16 path (OpenCV >= 3.4.2 / 4.x): 8.8 fixed-point taps, error-diffused so that they
 *                        sum to 256: {18,34,48,56,48,34,18}; same two passes and final rounding.  Restated from the published
 *                        algorithm, NOT verified against a build of that OpenCV here: run tools/pin before relying on it.
 * Applies to the calls made after it; the oracle has the same switch (orc_set_blur_variant).                                 */
#define ORBX_CV_BLUR_8BIT 0
#define ORBX_CV_BLUR_FIXED16 1
int orbx_set_opencv_variant(orbx_ctx* ctx, int blur_variant);

#define ORBX_NSTAGES 5
int orbx_set_profiling(orbx_ctx* ctx, int enable);
int orbx_get_stage_ms(orbx_ctx* ctx, float* ms /*[ORBX_NSTAGES]*/, int* ncalls);

/* ------------------------------------------------------------------ matcher --
 * Replaces the distance core of ORB_SLAM2::ORBmatcher (src/ORBmatcher.cc).        */

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1422-1437); host pointers, 32 bytes each. */
int orbm_descriptor_distance(const uint8_t* a, const uint8_t* b);

/* best / second-best Hamming distance of every query against its candidate list, first minimum
 * wins (the inner loop of every Search* method, e.g. src/ORBmatcher.cc:192-208).
 * cand_offsets (nq+1) / cand_idx = CSR candidate lists; both NULL = brute force over all nt targets.
 * Outputs per query: best_idx (-1 = no candidate), best_d, second_d (256 = none). Device pointers. */
int orbm_hamming_best2_device(const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                              const uint32_t* d_cand_offsets, const uint32_t* d_cand_idx, int32_t* d_best_idx,
                              int32_t* d_best_d, int32_t* d_second_d, void* stream);
/* same with host pointers (synchronous) */
int orbm_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint32_t* cand_offsets,
                       const uint32_t* cand_idx, int32_t* best_idx, int32_t* best_d, int32_t* second_d);

/* Brute-force frame-to-frame matching, batched: pair p matches the keypoints of frame a[p] against
 * frame b[p] of one extract batch: best/second-best over ALL keypoints of b, accept iff
 * best <= th && best < ratio*second (src/ORBmatcher.cc:210-212), then the rotation-consistency
 * histogram filter (ComputeThreeMaxima, :1386-1418; idiom :217-252) when check_ori != 0.
 * d_kps/d_desc/d_counts are orbx_extract_batch_device's outputs (cap = per-frame capacity).
 * d_match12[npairs*cap]: index into frame b or -1.  d_nmatch[npairs].              */
int orbm_match_frames_batch_device(const orbx_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_counts,
                                   int cap, const int32_t* d_pair_a, const int32_t* d_pair_b, int npairs,
                                   float ratio, int th, int check_ori, int32_t* d_match12, int32_t* d_nmatch,
                                   void* stream);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:363-468) on flattened frames (host pointers).
 * kps = n x 4 floats (x, y, octave, angle) of the undistorted keypoints; bounds2 = {min_x,max_x,min_y,max_y}
 * of frame 2 (Frame::ComputeImageBounds); prev_matched (n1 x 2) is updated in place; matches12[n1].
 * The frame grid, the window lists and every candidate distance are computed on the GPU; the order-dependent pass
 * (":408", ":421-427") runs on the host in frame-1 order.  Returns the match count in *nmatches. */
int orbm_search_for_initialization(const float* kps1, const uint8_t* desc1, int n1, const float* kps2,
                                   const uint8_t* desc2, int n2, const float* bounds2, float* prev_matched,
                                   int window, float nnratio, int check_ori, int32_t* matches12, int* nmatches);

/* Projection-guided searches on flattened data (host pointers).  One engine covers
 *   SearchByProjection(Frame&, vector<MapPoint*>&, th)            src/ORBmatcher.cc:42-119   (mode_best2 = 1, th = TH_HIGH)
 *   SearchByProjection(Frame& cur, const Frame& last, th)         :1161-1271  (best only, TH_HIGH, rotation check)
 *   SearchByProjection(Frame&, KeyFrame*, set&, th, ORBdist)      :1273-1384  (best only, th = ORBdist, rotation check)
 *   SearchByProjection(KeyFrame*, Scw, points, matched, th)       :258-361    (q_pred_level post-filter, TH_LOW)
 *   the candidate selection of Fuse (:724-954; chi2_gate = 5.99) and of SearchBySim3 (:956-1159, called twice).
 * The caller keeps the reference's own geometry (projection, frustum / distance / viewing-angle gates,
 * PredictScale) and passes per query: projected position, window radius, the GetFeaturesInArea level
 * range (-1,-1 = none), an optional predicted level for the [pred-1, pred] post-filter, the descriptor,
 * q_valid (0 = query skipped).  Targets: kps4 = n x {x, y, octave, angle} of the undistorted keypoints,
 * bounds = {min_x, max_x, min_y, max_y}.  Candidates come from the 64x48 grid (Frame::GetFeaturesInArea,
 * src/Frame.cc:243-307), distances from the GPU, and the order-dependent greedy pass runs in query order:
 * `taken` (nullable, n bytes, in/out) marks targets that already hold a map point (skipped; set on
 * assignment, cleared again if the rotation histogram rejects the match).  Outputs: q_match[nq] = target
 * index or -1, q_best_dist[nq] (nullable), *nmatches.                                        */
/* q_valid[q] (nullable = all 1): 0 = skip the query; bit 0 = search; bit 1 (value 3) = search, but a match does NOT close the
 * target for later queries - the reference closes a feature only while its map point has Observations() > 0
 * (src/ORBmatcher.cc:83-84, :1220-1221).  q_match[q] = target index, -1 = none, or -2 - target for a match the rotation
 * histogram removed (the reference had assigned that slot and resets it to nullptr, :1260-1264; *nmatches does not count it). */
int orbm_search_by_projection(const float* kps4, const uint8_t* desc, int n, const float* bounds, const float* q_uv,
                              const float* q_radius, const int32_t* q_min_level, const int32_t* q_max_level,
                              const int32_t* q_pred_level, const uint8_t* q_desc, const uint8_t* q_valid,
                              const float* q_angle, int nq, const float* inv_level_sigma2, float chi2_gate,
                              uint8_t* taken, int mode_best2, float ratio, int th, int check_ori, int32_t* q_match,
                              int32_t* q_best_dist, int* nmatches);

/* ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:956-1159) on flattened data (host pointers), whole entry point from the two
 * window searches on.  The caller keeps the reference's geometry (transform with sR21 / sR12, depth / image / distance gates,
 * PredictScale) and passes, per feature of keyframe 1 that holds a usable, not yet matched map point (q12_valid), its
 * projection into keyframe 2, the radius th * scale_factors_[predicted level] and the predicted level - and the same for
 * keyframe 2 into keyframe 1 (q21_*).  desc1 / desc2 = the keyframes' own descriptors (the search targets); q12_desc[n1][32] /
 * q21_desc[n2][32] = pMP->GetDescriptor() of the feature's map point (":1036", ":1112"; NULL = the keyframe's own row).  Each
 * direction takes the best candidate of KeyFrame::GetFeaturesInArea with level in [pred - 1, pred] and distance <= TH_HIGH; a
 * pair is kept iff the two directions agree (:1145-1157).  match12[n1] = index in keyframe 2 or -1; *nfound = the return value.
 * bounds1[4] / bounds2[4] = {min_x, max_x, min_y, max_y} of keyframe 1 / 2: each window search runs on the TARGET keyframe's own
 * grid (pKF2->GetFeaturesInArea :1022, pKF1->GetFeaturesInArea :1102). */
int orbm_search_by_sim3(const float* kps1, const uint8_t* desc1, int n1, const float* kps2, const uint8_t* desc2, int n2,
                        const float* bounds1, const float* bounds2, const float* q12_uv, const float* q12_radius, const int32_t* q12_pred,
                        const uint8_t* q12_valid, const uint8_t* q12_desc, const float* q21_uv, const float* q21_radius,
                        const int32_t* q21_pred, const uint8_t* q21_valid, const uint8_t* q21_desc, int32_t* match12, int* nfound);

/* SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:151-256; strict = 0) and SearchByBoW(KeyFrame*,
 * KeyFrame*, ...) (:470-580; strict = 1: best < th) on flattened data (host pointers).  The DBoW2 feature
 * vectors are passed as ascending node ids with CSR keypoint-index lists; valid1/valid2 (nullable) mark
 * keypoints that hold a usable map point.  match12[n1] = index in set 2 or -1.               */
int orbm_search_by_bow(const uint8_t* desc1, int n1, const uint8_t* valid1, const float* angle1, const uint8_t* desc2,
                       int n2, const uint8_t* valid2, const float* angle2, const uint32_t* fv1_node,
                       const uint32_t* fv1_off, const uint32_t* fv1_idx, int fv1_n, const uint32_t* fv2_node,
                       const uint32_t* fv2_off, const uint32_t* fv2_idx, int fv2_n, float ratio, int th, int strict,
                       int check_ori, int32_t* match12, int* nmatches);

/* ---- LocalMapping::CreateNewMapPoints, per-match body (src/LocalMapping.cc:267-378, monocular): ray-parallax
 * test, linear triangulation (null vector of the 4x4 system; the reference uses Eigen::JacobiSVD<Matrix4d>), positive depth in
 * both keyframes, chi-square reprojection gates (5.991 * level_sigma2), scale consistency.  Tcw = [Rcw | tcw] row-major 3x4 of
 * the current (1) and the neighbour (2) keyframe; kp1 / kp2 = n x {x, y, octave} floats of the matched undistorted keypoints
 * (matched_indices_ already applied); ratio_factor = 1.5f * scale_factor_.  ok[i] = 1 and x3D[i] valid for accepted matches. */
int orbm_triangulate_matches(const double* Tcw1, const double* Tcw2, const float* K1, const float* K2, const float* kp1,
                             const float* kp2, int n, const float* level_sigma2, const float* scale_factors, int n_levels,
                             float ratio_factor, double* x3D, uint8_t* ok);

/* ---- Frame::ComputeBoW: DBoW2 TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
 * (lib/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1260), TF_IDF weighting + L1 norm as ORBVocabulary uses.  The tree is given
 * flattened, node 0 = root: node_desc[n_nodes][32], children of node i = children[child_off[i] .. child_off[i+1]) in their
 * stored order (a leaf has none), word_id / weight of the leaves (word id of an inner node is ignored), L = depth.        */
typedef struct orbv_ctx orbv_ctx;
int orbv_create(const uint8_t* node_desc, const uint32_t* child_off /*[n_nodes+1]*/, const uint32_t* children,
                const int32_t* word_id, const double* weight, int n_nodes, int L, int device, orbv_ctx** out);
int orbv_destroy(orbv_ctx* ctx);
/* ORBVocabulary::loadFromTextFile (lib/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1423) - ORBvoc.txt straight into the device-resident
 * flattened tree: header "k L scoring weighting", then one node per line "parent is_leaf d0 .. d31 weight" (node ids in file
 * order from 1, children in file order, word ids in file order).  orbv_parse_text is the host half alone (arrays allocated with
 * malloc, release each with orbv_free_parsed); blank lines are skipped (the reference's eof loop turns the trailing one into a
 * spurious child of the root with an uninitialised descriptor - not reproduced).                                              */
int orbv_load_text(const char* path, int device, orbv_ctx** out);
int orbv_parse_text(const char* path, int32_t* k, int32_t* L, int32_t* scoring, int32_t* weighting, int32_t* n_nodes, int32_t* n_children,
                    uint8_t** node_desc, uint32_t** child_off, uint32_t** children, int32_t** word_id, double** weight);
void orbv_free_parsed(void* p);
/* desc[n][32] -> BowVector as ascending (bow_word[k], bow_value[k]), k < *n_words <= n, and FeatureVector as CSR:
 * node ids fv_node[m] ascending, features fv_idx[fv_off[m] .. fv_off[m+1]) ascending, m < *n_fv_nodes <= n (fv_off has n+1
 * entries) - the layout orbm_search_by_bow / orbm_search_for_triangulation take.  src/Frame.cc:322-327 uses levelsup = 4.
 * The node of a feature whose leaf lies above level L - levelsup is 0 (the reference leaves it uninitialised there).     */
int orbv_transform(orbv_ctx* ctx, const uint8_t* desc, int n, int levelsup, uint32_t* bow_word, double* bow_value,
                   int* n_words, uint32_t* fv_node, uint32_t* fv_off, uint32_t* fv_idx, int* n_fv_nodes);
/* the tree descent only, device-resident and batchable over frames: per feature its word id, idf weight and node id.
 * Enqueue only.                                                                                                          */
int orbv_descend_device(orbv_ctx* ctx, const uint8_t* d_desc, int n, int levelsup, int32_t* d_word, double* d_weight,
                        uint32_t* d_node, void* stream);
/* L1Scoring::score (lib/DBoW2/DBoW2/ScoringObject.cpp:23-68) between two BowVectors in the layout above (host arithmetic). */
double orbv_score_l1(const uint32_t* w1, const double* v1, int n1, const uint32_t* w2, const double* v2, int n2);

/* ---- the steps either side of extract -> match: undistortion, the 64 x 48 frame grid, window candidates,
 * frustum test.  kps4 = n x {x, y, octave, angle} floats of the UNDISTORTED keypoints; bounds = {min_x, max_x, min_y, max_y}
 * (Frame::ComputeImageBounds, src/Frame.cc:357-385).  Host pointers; the arithmetic runs on the device.                  */

/* Frame::UndistortKeyPoints (src/Frame.cc:329-355): cv::undistortPoints(mat, mat, K, dist, Mat(), K) on n points,
 * dist5 = {k1, k2, p1, p2, k3}; k1 == 0 copies the input, as the reference does.                                        */
int orbm_undistort_keypoints(const float* xy, int n, const float* K4 /*fx,fy,cx,cy*/, const float* dist5, float* xy_out);

/* Frame::AssignFeaturesToGrid (src/Frame.cc:158-173, PosInGrid :309-320): cell c = x * 48 + y holds keypoint indices
 * cell_idx[cell_offsets[c] .. cell_offsets[c+1]) in push_back order; keypoints outside the grid are skipped.            */
int orbm_assign_features_to_grid(const float* kps4, int n, const float* bounds, uint32_t* cell_offsets /*[64*48+1]*/,
                                 uint32_t* cell_idx /*[n]*/, int* n_assigned);

/* Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (src/Frame.cc:243-307) for nq queries -> CSR lists in the
 * reference's order (cells ix-major then iy, entries in keypoint-index order).  q_min_level / q_max_level NULL = -1 / -1,
 * which is also KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:575-622).  cand_idx may be NULL to size: *total is always set;
 * ORBHIP_ECAP if cap < *total with cand_idx given.                                                                       */
int orbm_features_in_area(const float* kps4, int n, const float* bounds, const float* q_xy, const float* q_radius,
                          const int32_t* q_min_level, const int32_t* q_max_level, int nq, uint32_t* cand_offsets /*[nq+1]*/,
                          uint32_t* cand_idx, int cap, int* total);

/* Frame::isInFrustum (src/Frame.cc:191-241) + MapPoint::PredictScale (src/MapPoint.cc:406-420) for n map points:
 * Rcw (row-major 3x3), tcw of the frame; P = world position, Pn = mean viewing direction, min_dist / max_dist = the map
 * point's raw min_distance_ / max_distance_ (the 0.8 / 1.2 invariance factors are applied here).  Outputs per point:
 * in_view (is_track_in_view_), uv (track_proj_x_, track_proj_y_), level (track_scale_level_), view_cos (track_view_cos_);
 * uv / level / view_cos are written for every point, meaningful where in_view is set.                                    */
int orbm_is_in_frustum(const double* Rcw, const double* tcw, const float* K4, const float* bounds, const double* P,
                       const double* Pn, const float* min_dist, const float* max_dist, int n, float viewing_cos_limit,
                       float log_scale_factor, int n_levels, uint8_t* in_view, float* uv, int32_t* level, float* view_cos);
/* The same gates fed from MapPoint's PUBLIC accessors: min / max = GetMinDistanceInvariance() / GetMaxDistanceInvariance()
 * (min_distance_ / max_distance_ are protected, include/MapPoint.h:125-151), and dist = |P - Ow| as float comes back so that
 * the caller runs MapPoint::PredictScale(dist, frame) itself for the points in view (src/Frame.cc:231).                   */
int orbm_is_in_frustum_gates(const double* Rcw, const double* tcw, const float* K4, const float* bounds, const double* P,
                             const double* Pn, const float* min_dist_invariance, const float* max_dist_invariance, int n,
                             float viewing_cos_limit, uint8_t* in_view, float* uv, float* view_cos, float* dist);

/* SearchForTriangulation (src/ORBmatcher.cc:582-722, mono) on flattened data (host pointers): BoW-node brute force
 * between keypoints WITHOUT a map point (unmapped1/2 flags), dist <= TH_LOW with later ties replacing earlier ones
 * (":654"), rejection near the epipole (ex, ey; ":658-664") and the epipolar-line gate CheckDistEpipolarLine
 * (":128-149") with F12 row-major (3x3 doubles), scale_factors / level_sigma2 of keyframe 2.  Queries are
 * independent (vbMatched2 is never set in this fork).  match12[n1] = index in keyframe 2 or -1.          */
int orbm_search_for_triangulation(const float* kps1, const uint8_t* desc1, const uint8_t* unmapped1, int n1,
                                  const float* kps2, const uint8_t* desc2, const uint8_t* unmapped2, int n2,
                                  const uint32_t* fv1_node, const uint32_t* fv1_off, const uint32_t* fv1_idx, int fv1_n,
                                  const uint32_t* fv2_node, const uint32_t* fv2_off, const uint32_t* fv2_idx, int fv2_n,
                                  const double* F12, float ex, float ey, const float* scale_factors,
                                  const float* level_sigma2, int check_ori, int32_t* match12, int* nmatches);

/* ---- Tracking::Relocalization, first stage (src/Tracking.cc:979-1029; round 5): current_frame_.ComputeBoW() and
 * matcher.SearchByBoW(keyframe, current_frame_, map_point_matches_vector[i]) - ORBmatcher(0.75, true) - for EVERY candidate keyframe of
 * KeyFrameDatabase::DetectRelocalizationCandidates in ONE call (the candidates are independent; the vocabulary descent of the frame
 * runs once).  img != NULL: the frame is extracted first and stays on the device (as orbt_track_reference_keyframe); NULL: the
 * frame an earlier orbt_* call of this thread left there.  Per candidate i: slot_owner[i][f] = the keyframe feature whose map
 * point vpMapPointMatches[f] holds (-1: none) after the rotation check, nmatches[i] = the return value (the caller discards the
 * candidate below 15, :1019).  What follows - PnPsolver::iterate per candidate, then PoseOptimization /
 * SearchByProjection(F, KF, found, th, ORBdist) rounds (:1040-1120) - starts from a PnP pose and stays with the caller: PnP is out of
 * scope, the rounds are ba_pose_optimization and the reloc_kf form of orbm_search_by_projection.             */
typedef struct orbt_reloc_keyframe {
  const uint8_t* desc; const uint8_t* valid; const float* angle; int n;      /* descriptors, 1 = usable map point (not NULL, not isBad()), keypoint angles */
  const uint32_t* fv_node; const uint32_t* fv_off; const uint32_t* fv_idx; int fv_n;      /* the keyframe's FeatureVector (orbv_transform's layout) */
} orbt_reloc_keyframe;
int orbt_relocalization_search_by_bow(orbx_ctx* ctx, orbv_ctx* voc, const uint8_t* img, int w, int h, int stride, const float* K4, const float* bounds,
                                      const orbt_reloc_keyframe* candidates, int n_candidates, float nnratio, int check_ori, orbx_keypoint* kps_out,
                                      uint8_t* desc_out, int cap, uint32_t* bow_word, double* bow_value, int* n_words, uint32_t* fv_node, uint32_t* fv_off,
                                      uint32_t* fv_idx, int* n_fv_nodes, int32_t* slot_owner /*[n_candidates][cap]*/, int32_t* nmatches /*[n_candidates]*/,
                                      int* n_keypoints);

/* ---- LocalMapping::CreateNewMapPoints, device-resident across the neighbour keyframes (src/LocalMapping.cc:196-396; round 5) ----
 * For every neighbour keyframe IN ORDER: ORBmatcher::SearchForTriangulation(current, neighbour, F12, pairs, false) with the matcher of
 * that call site, ORBmatcher(0.6, false) (:203: no orientation check; src/ORBmatcher.cc:582-722), then the per-match body (:267-378:
 * ray parallax, linear triangulation, depth / chi-square / scale gates) - and what makes the neighbours depend on each other: a
 * triangulated match gives the current keyframe's keypoint a map point (AddMapPoint, :383), so the next neighbour's search skips it
 * (src/ORBmatcher.cc:621-623).  One upload, one kernel per neighbour on the calling thread's stream, one download; the map mutation
 * (:380-393) is the caller's loop over the accepted entries (csrc/compat/orbslam_dropin.h: CreateNewMapPoints).
 *   current keyframe: kps1[n1][4] = {x, y, octave, angle} undistorted keypoints, desc1[n1][32], unmapped1[n1] (1 = GetMapPoint(idx) ==
 *     NULL; NULL = all), its FeatureVector as ascending node ids + CSR lists (orbv_transform's layout), Tcw1 row-major 3x4, K1 = {fx, fy,
 *     cx, cy}.  Neighbours that fail the baseline test (:231-244) are left out by the caller.
 *   scale_factors / level_sigma2 [n_levels]: the extractor's tables (every keyframe copies them from the same ORBextractor);
 *     ratio_factor = 1.5f * current_keyframe_->scale_factor_ (:221).
 *   stop (nullable): a byte the caller may raise while the call runs = CheckNewKeyFrames() (:227): it is looked at ONCE before every
 *     neighbour after the first; the neighbours before it are complete, *n_processed says how many.
 *   out, per neighbour k and keypoint i of the current keyframe: match12[k][i] = SearchForTriangulation's partner or -1,
 *     ok[k][i] = 1 when the triangulation passed every gate, x3D[k][i] = the new point (zeros otherwise).                       */
typedef struct orbl_keyframe {
  const float* kps; const uint8_t* desc; const uint8_t* unmapped; int n;           /* as kps1 / desc1 / unmapped1 */
  const uint32_t* fv_node; const uint32_t* fv_off; const uint32_t* fv_idx; int fv_n;
  double Tcw[12]; float K4[4];
  double F12[9];              /* LocalMapping::ComputeF12(current, this) (:507-523), row-major */
  float ex, ey;               /* the epipole of the current keyframe's centre in this keyframe (src/ORBmatcher.cc:588-595) */
} orbl_keyframe;
int orbl_create_new_map_points(const float* kps1, const uint8_t* desc1, const uint8_t* unmapped1, int n1, const uint32_t* fv1_node,
                               const uint32_t* fv1_off, const uint32_t* fv1_idx, int fv1_n, const double* Tcw1, const float* K1,
                               const orbl_keyframe* neighbours, int n_neighbours, const float* scale_factors, const float* level_sigma2,
                               int n_levels, float ratio_factor, const volatile uint8_t* stop, int32_t* match12, uint8_t* ok, double* x3D,
                               int* n_processed);
/* ---- LocalMapping::SearchInNeighbors (:398-505): the candidate selection of ORBmatcher::Fuse (src/ORBmatcher.cc:724-842) for ALL
 * target keyframes of one loop in ONE call.  The caller keeps the reference's own projection (Rcw p + tcw, IsInImage, the distance
 * and viewing-angle gates, PredictScale: :739-778) and passes, per (keyframe t, map point m), q_uv[t][m] = (u, v), q_radius[t][m] =
 * th * scale_factors_[level], q_level[t][m] = nPredictedLevel or -1 when a gate failed; mp_desc[m] = pMP->GetDescriptor().  The call
 * does KeyFrame::GetFeaturesInArea on every keyframe's own 64 x 48 grid (src/KeyFrame.cc:575-622), the level window (:781), the
 * chi-square gate (:789) and the descriptor distances: best_idx[t][m] = the keypoint (-1: none), best_dist[t][m] (256: none); the
 * caller applies `bestDist <= TH_LOW` and the Replace / AddObservation mutation in the reference's order (:806-823).  A map point
 * whose descriptor changes through such a mutation (MapPoint::Replace recomputes it) is re-queried by the caller for the keyframes
 * that follow (csrc/compat/orbslam_dropin.h).                                                                                    */
typedef struct orbl_fuse_keyframe {
  const float* kps; const uint8_t* desc; int n;      /* undistorted keypoints {x, y, octave, angle}, descriptors */
  float bounds[4];                                   /* {min_x_, max_x_, min_y_, max_y_} */
} orbl_fuse_keyframe;
int orbl_fuse_batch(const orbl_fuse_keyframe* keyframes, int n_keyframes, const float* q_uv, const float* q_radius, const int32_t* q_level,
                    int n_map_points, const uint8_t* mp_desc, const float* inv_level_sigma2, int n_levels, int32_t* best_idx,
                    int32_t* best_dist);
/* ---- LoopClosing::SearchAndFuse (src/LoopClosing.cc:599-630): the candidate selection of ORBmatcher::Fuse(KeyFrame*, Scw, points, th,
 * replace) (src/ORBmatcher.cc:844-954) for ALL corrected keyframes of a loop closure in ONE call.  As orbl_fuse_batch, without the
 * chi-square gate (the Sim(3) form has none): the caller projects with the CORRECTED Sim(3) of every keyframe (Rcw = sRcw / s,
 * tcw = t / s, :854-859) and keeps the gates :873-906; best_idx / best_dist as above; `bestDist <= TH_LOW`, vpReplacePoint /
 * AddObservation and the Replace loop under the map mutex stay with the caller, keyframe after keyframe (csrc/compat/orbslam_dropin.h:
 * ORBmatcher::SearchAndFuse re-queries a point whose descriptor a Replace recomputed and re-reads GetMapPoints() per keyframe).   */
int orbl_fuse_batch_sim3(const orbl_fuse_keyframe* keyframes, int n_keyframes, const float* q_uv, const float* q_radius, const int32_t* q_level,
                         int n_points, const uint8_t* mp_desc, int n_levels, int32_t* best_idx, int32_t* best_dist);

/* ---- the per-frame Tracking step with the motion model, device-resident (src/Tracking.cc:616-646): Frame construction
 * (ORBextractor::operator(), AssignFeaturesToGrid; zero distortion: the undistorted keypoints are the raw ones, as for the
 * KITTI configurations), ORBmatcher::SearchByProjection(current_frame_, last_frame_, th) (src/ORBmatcher.cc:1161-1271,
 * monocular) and CeresOptimizer::PoseOptimization (src/CeresOptimizer.cc:275-342) in ONE call: the image and one packed block
 * go up, the kernels of all three stages run back to back on one stream (the greedy, order-dependent pass of the search as a
 * parallel fixpoint on the device, csrc/orb_track.hip), one block comes down.
 *   in   img: CV_8UC1 w x h; K4 = {fx, fy, cx, cy}; bounds = {min_x, max_x, min_y, max_y}; Tcw_pred = current_frame_.Tcw_
 *        after SetPose(velocity_ * last_frame_.Tcw_), row-major 3x4 (or the first 12 of a 4x4); per last-frame feature i
 *        (n_last <= 4096): last_Xw = map point position, last_desc = its descriptor (MapPoint::GetDescriptor), last_octave =
 *        LastFrame.keypoints_[i].octave, last_angle = LastFrame.undistort_keypoints_[i].angle, last_valid = 0 (no map point,
 *        or is_outliers_[i]), 1 (map point with Observations() > 0) or 3 (without: it is matched but does not close the
 *        feature for later points, :1220-1221); th = 15 (the caller repeats the call with 2 * th when nmatches < 20,
 *        src/Tracking.cc:635-641); check_ori = mbCheckOrientation.
 *   out  kps / desc [cap >= orbx_max_keypoints(ctx)]: the frame's keypoints and descriptors; match[n_last]: index of the
 *        current feature matched to last-frame feature i, -1 none, -2 - index removed by the rotation check; owner[n_kp]:
 *        which last-frame feature's map point CurrentFrame.map_points_[f] holds, or -1; outlier[n_kp] = is_outliers_ after
 *        PoseOptimization; res: counts, the optimised pose [tx, ty, tz, qx, qy, qz, qw] (the predicted one when fewer than
 *        3 correspondences, as the reference leaves the pose alone then).                                                  */
typedef struct orbt_result {
  int32_t n_keypoints, nmatches, n_correspondences, n_inliers, greedy_rounds, reserved;
  double pose7[7];
} orbt_result;
int orbt_track_with_motion_model(orbx_ctx* ctx, const uint8_t* img, int w, int h, int stride, const float* K4, const float* bounds,
                                 const double* Tcw_pred, const double* last_Xw, const uint8_t* last_desc, const int32_t* last_octave,
                                 const float* last_angle, const uint8_t* last_valid, int n_last, float th, int check_ori,
                                 orbx_keypoint* kps_out, uint8_t* desc_out, int cap, int32_t* match_out, int32_t* owner_out,
                                 uint8_t* outlier_out, orbt_result* res);

/* ---- the second stage of Tracking on the same frame, device-resident (src/Tracking.cc:673-750 TrackLocalMap): SearchLocalPoints
 * (:793-842) = Frame::isInFrustum(pMP, 0.5) (src/Frame.cc:191-241, MapPoint::PredictScale src/MapPoint.cc:406-420) over the local
 * map points, ORBmatcher::SearchByProjection(Frame&, vpMapPoints, th) (src/ORBmatcher.cc:42-119: window radius by viewing cosine,
 * levels [nPredictedLevel - 1, nPredictedLevel], best / second best + level rule, nnratio, the :83-84 claim rule), then
 * CeresOptimizer::PoseOptimization over EVERY slot that holds a point - in ONE call on the frame (keypoints, descriptors, grid)
 * that orbt_track_with_motion_model left on the device: the same host thread must have called it for this frame.
 *   in   Tcw = current_frame_.Tcw_ (row-major 3x4: the pose the first stage optimised); log_scale_factor = Frame::log_scale_factor_;
 *        per local map point (n_mp <= 16384, vector order of local_map_points_): position, GetNormal(), min_distance_ /
 *        max_distance_ (the 0.8 / 1.2 invariance factors are applied inside), GetDescriptor(), mp_state = 0 skip (isBad(), or
 *        last_seen_frame_id_ == current_frame_.id_: it already sits in the frame, :816-817), 1 candidate with Observations() > 0,
 *        3 candidate without (it is matched but does not close the feature, :83-84); per keypoint of the frame
 *        (n_kp = the resident frame's count): slot_state = 0 empty, 1 holds a point with Observations() > 0 (closed), 3 holds one
 *        without; slot_Xw = that point's position (read where slot_state != 0); th = 1 (5 right after a relocalisation,
 *        :834-838), nnratio = 0.8.
 *   out  mp_in_view[n_mp] = isInFrustum result (the caller's IncreaseVisible, :822-825); mp_match[n_mp] = feature the point was
 *        written to or -1; slot_owner[n_kp] = index of the local map point the slot holds NOW if this call wrote it (the last
 *        writer, :110), else -1 (unchanged); outlier[n_kp] = is_outliers_ after PoseOptimization; res: nmatches, correspondences,
 *        inliers, pose (the input pose with fewer than 3 correspondences), reserved = points in view.                          */
int orbt_track_local_map(orbx_ctx* ctx, const float* K4, const float* bounds, const double* Tcw, float log_scale_factor,
                         const double* mp_Xw, const double* mp_normal, const float* mp_min_dist, const float* mp_max_dist,
                         const uint8_t* mp_desc, const uint8_t* mp_state, int n_mp, const double* slot_Xw, const uint8_t* slot_state,
                         int n_kp, float th, float nnratio, uint8_t* mp_in_view, int32_t* mp_match, int32_t* slot_owner,
                         uint8_t* outlier, orbt_result* res);

/* ---- Tracking::TrackReferenceKeyFrame's data-parallel core in one call (src/Tracking.cc:566-615): Frame::ComputeBoW (src/Frame.cc:
 * 322-327, the vocabulary descent), ORBmatcher::SearchByBoW(reference_keyframe_, current_frame_, ...) (src/ORBmatcher.cc:151-256,
 * nnratio 0.7, TH_LOW 50, rotation histogram) and CeresOptimizer::PoseOptimization from last_frame_.Tcw_.
 *   in   img != NULL: the frame is extracted first (as in orbt_track_with_motion_model; kps / desc [cap] receive it) and stays on
 *        the device for a following orbt_track_local_map; img == NULL: the frame an earlier orbt_* call of this thread left there.
 *        Per keyframe feature i < n_kf: descriptor, kf_valid = it holds a map point that is not bad, undistort_keypoints_[i].angle,
 *        the point's position; the keyframe's feature_vector_ as CSR (ascending node ids, as orbv_transform returns it).
 *   out  the frame's BowVector / FeatureVector in orbv_transform's layout (pass NULL for bow_word to skip them);
 *        match_kf[n_kf] = frame feature matched to keyframe feature i or -1; slot_owner[n_kp] = keyframe feature whose map point
 *        current_frame_.map_points_[f] holds, or -1; outlier[n_kp]; res->nmatches (the caller returns false below 15, :582),
 *        pose.  PoseOptimization runs whatever nmatches is.                                                                   */
int orbt_track_reference_keyframe(orbx_ctx* ctx, orbv_ctx* voc, const uint8_t* img, int w, int h, int stride, const float* K4,
                                  const float* bounds, const double* Tcw_last, const uint8_t* kf_desc, const uint8_t* kf_valid,
                                  const float* kf_angle, const double* kf_Xw, int n_kf, const uint32_t* kf_fv_node,
                                  const uint32_t* kf_fv_off, const uint32_t* kf_fv_idx, int kf_fv_n, float nnratio, int check_ori,
                                  orbx_keypoint* kps, uint8_t* desc, int cap, uint32_t* bow_word, double* bow_value, int* n_words,
                                  uint32_t* fv_node, uint32_t* fv_off, uint32_t* fv_idx, int* n_fv_nodes, int32_t* match_kf,
                                  int32_t* slot_owner, uint8_t* outlier, orbt_result* res);

/* Host wall time (ms) of the calling thread's most recent orbt_* call, measured inside the library (entry to return): the latency
 * of a Tracking step as the reference's C++ caller would see it, free of what a scripting host adds around the call.       */
double orbt_last_call_ms(void);

/* ------------------------------------------------------------ bundle adjust --
 * Replaces CeresOptimizer::{PoseOptimization, BundleAdjustment/GlobalBundleAdjustemnt,
 * LocalBundleAdjustment, CheckOutlier(s)} (src/CeresOptimizer.cc:49-599) and the Ceres solve
 * underneath (trust-region LM, Huber loss, quaternion manifold, Jacobi scaling; exact Schur
 * solve).  All arithmetic is fp64.  Poses are 7-vectors [tx,ty,tz,qx,qy,qz,qw]
 * (src/MatEigenConverter.cc:66-85); K4 = {fx,fy,cx,cy}.                              */
typedef struct ba_options {
  int32_t max_iterations;      /* options.max_num_iterations */
  double huber_delta;          /* sqrt(5.991) in the reference; applied where obs_robust != 0 */
  int32_t fix_points;          /* 1 = points are constants (PoseOptimization) */
  const volatile uint8_t* stop_flag; /* StopFlagCallback (include/CeresOptimizer.h:332-349); may be NULL.  Any host byte: it is
                                        polled before every enqueue and forwarded to the device while the solve drains, so a flag
                                        raised mid-solve ends it at the next LM iteration boundary with termination 4 and the last
                                        accepted iterate written back */
} ba_options;

/* MatEigenConverter::Matrix4dToMatrix_7_1 / Matrix_7_1_ToMatrix4d (src/MatEigenConverter.cc:66-85): T = row-major 4x4
 * [R t; 0 1] <-> [tx,ty,tz,qx,qy,qz,qw].  Encoding is Eigen::Quaterniond(R).coeffs() (trace / largest-diagonal branches, not
 * normalised); decoding normalises the quaternion first.  Host arithmetic.                                             */
int ba_matrix4d_to_pose7(const double* T, double* pose7);
int ba_pose7_to_matrix4d(const double* pose7, double* T);

typedef struct ba_summary {
  double initial_cost, final_cost;
  int32_t iterations;          /* LM iterations attempted */
  int32_t successful_steps;
  int32_t termination;         /* 0 max-iters 1 gradient 2 parameter 3 function tol 4 user stop 5 failure 6 min-radius
                                  7 device wait timed out (the call returns ORBHIP_ETIMEOUT) */
  double final_radius;
} ba_summary;

/* CeresOptimizer::PoseOptimization(Frame*) (src/CeresOptimizer.cc:275-342) for ONE frame, host pointers.
 * inv_sigma2[i] = frame->inv_level_sigma2s_[octave_i] (used un-square-rooted as in the reference, F7).
 * On return pose7 holds t and the NORMALISED quaternion (":336"), outlier[i] = CheckOutliers flag;
 * *n_inliers = n - n_bad, or 0 with the pose untouched when n < 3 (":330").          */
int ba_pose_optimization(const double* K4, double* pose7, const double* Xw, const double* uv,
                         const float* inv_sigma2, int n, uint8_t* outlier, int* n_inliers, ba_summary* summary);

/* batched, device-resident PoseOptimization: problem p uses observations [offsets[p], offsets[p+1]).
 * d_pose7[np*7] in/out, d_outlier[total], d_n_inliers[np], d_summary[np] (may be NULL). Enqueue only. */
int ba_pose_optimization_batch_device(const double* d_K4 /*np*4*/, double* d_pose7, const double* d_Xw,
                                      const double* d_uv, const float* d_inv_sigma2, const int32_t* d_offsets,
                                      int nproblems, uint8_t* d_outlier, int32_t* d_n_inliers,
                                      ba_summary* d_summary, void* stream);

/* Batches of INDEPENDENT problems (sub-maps) solved in lockstep: every kernel launch covers all problems of
 * the batch (one grid row per problem), each with its own device-side LM state, so small problems fill the 256 CUs together
 * instead of one at a time.  Results are identical to calling the single-problem entry points one by one.               */
typedef struct ba_problem {                 /* arguments of ba_solve */
  const double* K4; double* poses7; const uint8_t* cam_fixed; int32_t ncam;
  double* pts3; int32_t npts;
  const int32_t* obs_cam; const int32_t* obs_pt; const double* obs_uv; const double* obs_weight; const uint8_t* obs_robust;
  int32_t nobs;
} ba_problem;
int ba_solve_batch(const ba_problem* problems, int nproblems, const ba_options* opts, ba_summary* summaries /*[nproblems], may be NULL*/);

typedef struct ba_local_problem {           /* arguments of ba_local_bundle_adjustment */
  const double* K4; double* poses7; const uint8_t* cam_fixed; const uint8_t* cam_local; int32_t ncam;
  double* pts3; int32_t npts;
  const int32_t* obs_cam; const int32_t* obs_pt; const double* obs_uv; const float* obs_inv_sigma2; int32_t nobs;
  uint8_t* obs_erase;                       /* out [nobs] */
} ba_local_problem;
int ba_local_bundle_adjustment_batch(const ba_local_problem* problems, int nproblems, const volatile uint8_t* stop_flag,
                                     int duplicate_blocks, int* aborted, ba_summary* pass1 /*[nproblems] or NULL*/,
                                     ba_summary* pass2 /*[nproblems] or NULL*/);

/* CeresOptimizer::OptimizeSim3(KeyFrame*, KeyFrame*, vector<MapPoint*>& matches12, Sophus::Sim3d& S12, float th2,
 * bool bFixScale) (src/CeresOptimizer.cc:601-735; cost functor include/CeresOptimizer.h:168-236, parameterisation
 * src/CeresOptimizer.cc:24-47) for ONE keyframe pair, host pointers.  Correspondence i (the reference's accepted
 * (map_point_1, map_point_2) pairs, in loop order) carries
 *   P3D2c[i] = R2cw * P3D2w + t2cw, obs1[i] = keyframe_1 keypoint, inv_sigma2_1[i] = kf1->inv_level_sigma2s_[octave]  (:649-664)
 *   P3D1c[i] = R1cw * P3D1w + t1cw, obs2[i] = keyframe_2 keypoint, inv_sigma2_2[i] likewise                         (:666-683)
 * s12[7] in/out = Sophus::Sim3d::data() = [qx,qy,qz,qw (|q|^2 = scale), tx,ty,tz].  th2 -> HuberLoss(sqrt(th2)) and the
 * outlier threshold.  fix_scale is accepted and ignored, as the reference ignores bFixScale.  outlier[n] (may be NULL) =
 * is_outlier_12 || is_outlier_21 (:694-726).  *n_inliers = n - n_bad, or 0 when that is < 10 (:731).                     */
int ba_optimize_sim3(const double* K1 /*fx,fy,cx,cy*/, const double* K2, double* s12, const double* P3D2c,
                     const double* obs1, const float* inv_sigma2_1, const double* P3D1c, const double* obs2,
                     const float* inv_sigma2_2, int n, double th2, int fix_scale, uint8_t* outlier, int* n_inliers,
                     ba_summary* summary);

/* batched, device-resident OptimizeSim3 (one loop candidate per problem): problem p uses correspondences
 * [offsets[p], offsets[p+1]); d_K1/d_K2 [np*4], d_s12 [np*7] in/out, d_th2 [np], d_outlier [total] (may be NULL),
 * d_n_inliers [np], d_summary [np] (may be NULL).  Enqueue only.                                                         */
int ba_optimize_sim3_batch_device(const double* d_K1, const double* d_K2, double* d_s12, const double* d_P3D2c,
                                  const double* d_obs1, const float* d_inv_sigma2_1, const double* d_P3D1c,
                                  const double* d_obs2, const float* d_inv_sigma2_2, const int32_t* d_offsets,
                                  const double* d_th2, int nproblems, uint8_t* d_outlier, int32_t* d_n_inliers,
                                  ba_summary* d_summary, void* stream);

/* CeresOptimizer::OptimizeEssentialGraph (src/CeresOptimizer.cc:737-957), the solve: n_kf Sim(3) vertices given as tangent
 * 7-vectors (Scw.log(), in/out), kf_fixed[v] != 0 for the constant loop keyframe, n_edges EssentialGraphErrorTerm blocks
 * (include/CeresOptimizer.h:266-330) as (vertex j, vertex i, Sji in the qt7 layout) in the reference's insertion order
 * (loop connections :797-821, then per keyframe its parent :839-852, loop edges :855-874 and covisibility edges :877-905;
 * the caller forms Sji = Sjw * Swi from the corrected / non-corrected Sim3 exactly as there).  Identity information, no
 * loss, Sim3Parameterization, <= max_iterations (100 in the reference) LM iterations; the normal equations are solved with
 * the dense FP64-MFMA Cholesky; the reduced system takes (7 n_free)^2 * 8 bytes of device memory (2340 keyframes: 2.1 GB,
 * 10000 keyframes: 39 GB) and ORBHIP_ECAP is returned only when that does not fit.                                         */
int ba_optimize_essential_graph(double* lie7, const uint8_t* kf_fixed, int n_kf, const int32_t* edge_j, const int32_t* edge_i,
                                const double* edge_Sji, int n_edges, int max_iterations, const volatile uint8_t* stop_flag,
                                ba_summary* summary);
/* its write-back arithmetic (:916-956): Tiw[v] = [R | t / s] (row-major 3x4) of exp(lie7_opt[v]); every map point
 * P <- corrected_Swr * (Srw_original * P) with r = pt_ref_kf[p].                                                            */
int ba_essential_graph_correct(const double* lie7_orig, const double* lie7_opt, int n_kf, double* Tiw /*[n_kf*12]*/,
                               const int32_t* pt_ref_kf, double* pts3, int npts);

/* Sophus::Sim3d::exp / log in the layout above (tangent = [upsilon, omega, sigma]); host arithmetic, for bindings that
 * cross the Sophus boundary (LoopClosing builds gScm from (s, R, t): src/LoopClosing.cc:322).                           */
int ba_sim3_exp(const double* tangent7, double* s12_out);
int ba_sim3_log(const double* s12, double* tangent7_out);
/* Sophus::Sim3d::operator* and ::inverse() in the same layout (src/CeresOptimizer.cc:806-813,828-849 form Sji = Sjw * Swi from
 * them; src/LoopClosing.cc:335,458 likewise); out may alias an input.  Host arithmetic.                                   */
int ba_sim3_mul(const double* a, const double* b, double* out);
int ba_sim3_inverse(const double* a, double* out);

/* Measurement hook (no reference counterpart): while enabled, every ba_solve / ba_solve_batch / ba_local_bundle_adjustment(_batch)
 * / call brackets its device work (first kernel .. last LM iteration, copies excluded) with HIP events on the calling thread's
 * stream; ba_get_profile returns the sums of THIS host thread since its last call (device ms, problems solved, LM iterations)
 * and resets them.                                                                                                       */
int ba_set_profiling(int enable);
int ba_get_profile(double* device_ms, int* nsolves, int* lm_iterations);
/* Measurement hook (no reference counterpart): which member of the factorisation family the LAST ba_solve(_batch) /
 * ba_local_bundle_adjustment(_batch) call of THIS host thread took (INTEGRATION.md section 7 has the decision table).  out12 =
 * { problems in the call, look-ahead form (0 none, 1 persistent launch k_chol_persist, 2 one workgroup per system k_chol_wg, 3 one
 * launch per 32-column step k_chol_la), two-level form (0 none, 1 persistent block launches k_chol_persist_blk, 2 hybrid step
 * kernels, 3 classic, 4 one launch), backward substitution (1 k_chol_bsolve_sky, 2 per super-block), largest padded system of the
 * look-ahead family, of the two-level family, widest skyline in 32-column tiles, workgroups of the persistent launch per problem,
 * persistent mode granted by the lease (0 none), a look-ahead system has > 1024 unknowns, every look-ahead system fits k_chol_wg, 0 }. */
int ba_get_last_plan(int32_t* out12);
/* Configuration (no reference counterpart): the time limit, in milliseconds, after which a workgroup of the persistent Cholesky
 * kernels stops waiting for another one and the solve ends with ORBHIP_ETIMEOUT / ba_summary.termination 7 (default 5000 ms;
 * ms <= 0 restores the default; resolution 10 ns).  A deployment with a frame deadline may want tens of milliseconds.  Process-wide,
 * for the default device, for launches made after the call; NOT thread-safe: it synchronises the device - call it at start-up,
 * not while other threads are solving.                                                                                    */
int ba_set_wait_limit_ms(double ms);

/* CeresOptimizer::BundleAdjustment (src/CeresOptimizer.cc:59-225) on flattened arrays, host pointers:
 * cameras with cam_fixed != 0 are constant (KF id 0, fixed KFs); obs_weight multiplies the pixel
 * residual (= invSigma2, F7); obs_robust selects the loss per observation: 0 = none, 1 = Huber, 2 = the observation
 * is present twice, under Huber AND without loss (LocalBA pass 2, F6), folded into one block.  poses7 / pts3 are
 * updated in place with the last accepted iterate (quaternions NOT re-normalised here).             */
int ba_solve(const double* K4_per_cam, double* poses7, const uint8_t* cam_fixed, int ncam, double* pts3,
             int npts, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv,
             const double* obs_weight, const uint8_t* obs_robust, int nobs, const ba_options* opts,
             ba_summary* summary);

/* CeresOptimizer::CheckOutlier (src/CeresOptimizer.cc:227-241); host, scalar. depth may be NULL. */
int ba_check_outlier(const double* K4, const double* pose7, const double* Xw, const double* uv,
                     double inv_sigma2, double thres, double* depth);

/* Optimisation core of CeresOptimizer::LocalBundleAdjustment (src/CeresOptimizer.cc:408-598): pass 1
 * Huber / <=5 iterations, outlier classification (chi2 > 5.991 or depth <= 0, local keyframes only),
 * pass 2 with the not-erased observations re-added loss-free ON TOP of pass 1's blocks when
 * duplicate_blocks != 0 (the reference's behaviour) / <=10 iterations, classification again.
 * obs_erase[i] = final to_erase membership.  *aborted = 1 if *stop_flag was set before a solve
 * (nothing written back, ":509-512").  Quaternions are normalised on write-back (":589").         */
int ba_local_bundle_adjustment(const double* K4_per_cam, double* poses7, const uint8_t* cam_fixed,
                               const uint8_t* cam_local, int ncam, double* pts3, int npts,
                               const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv,
                               const float* obs_inv_sigma2, int nobs, const volatile uint8_t* stop_flag,
                               int duplicate_blocks, uint8_t* obs_erase, int* aborted, ba_summary* pass1,
                               ba_summary* pass2);

/* ---- multi-GPU: the landmark merge (north_star: "a single RCCL all-gather over xGMI to merge landmark updates") ----
 * One process per GPU, one sub-map per rank; frames and BA sub-problems shard with NO data-path collective.  The one exchange is the
 * merge of the landmark updates after a batched GlobalBA, so that every rank holds the whole map (the reference is one process,
 * src/MonoORBSlam.cc:52-100, src/System.cc: a multi-GPU embedding calls this from the thread that owns the map - LoopClosing's
 * RunGlobalBundleAdjustment, src/LoopClosing.cc:656-740, is where a sub-map's GlobalBA ends).
 * RCCL is looked up at run time (dlopen "librccl.so" or $ORBHIP_RCCL_LIB): liborbslam_hip.so does not link against it.
 *   orbhip_comm_get_unique_id   rank 0 makes the 128-byte id and hands it to the other ranks by whatever the embedding uses (MPI, a file, a socket)
 *   orbhip_comm_create          ncclCommInitRank on `device`; collective: every rank calls it with the same id and world size (<= 64)
 *   orbhip_comm_adopt           wraps an ncclComm_t the caller already owns (not destroyed by orbhip_comm_destroy)
 *   orbhip_allgather_landmarks  ONE ncclAllGather: every rank contributes a slot { count, cap_per_rank x (X, Y, Z[, id]) } - the sub-maps
 *                               are ragged, cap_per_rank is the bound the ranks agree on (ORBHIP_ECAP when n_local exceeds it) - and gets
 *                               the landmarks of all ranks, dense, in rank order: d_pts3_all [cap_all x 3], d_ids_all [cap_all] (NULL on
 *                               both sides when the landmarks carry no ids), counts_out [world size] (host), *n_all = their sum.
 *                               Device pointers; runs on `stream` and synchronises it.  ORBHIP_ECAP when n_all > cap_all.            */
typedef struct orbhip_comm orbhip_comm;
int orbhip_comm_get_unique_id(uint8_t* id128);
int orbhip_comm_create(const uint8_t* id128, int world_size, int rank, int device, orbhip_comm** out);
int orbhip_comm_adopt(void* nccl_comm, int device, orbhip_comm** out);
int orbhip_comm_info(const orbhip_comm* comm, int* world_size, int* rank);
int orbhip_comm_destroy(orbhip_comm* comm);
int orbhip_allgather_landmarks(orbhip_comm* comm, const double* d_pts3_local, const int64_t* d_ids_local, int n_local, int cap_per_rank,
                               double* d_pts3_all, int64_t* d_ids_all, int cap_all, int32_t* counts_out, int* n_all, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ORBSLAM_HIP_H */
