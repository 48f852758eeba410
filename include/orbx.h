/*
 * orbx.h -- C ABI of the MI355X-native ORB front-end (orbx_*) and Hamming matcher (orbm_*).
 *
 * This is the drop-in boundary for ORB-SLAM3's hot path (reference: elkuno213/orbslam3_cpp_fork).
 * The reference has no FFI/plugin layer; the seam is the C++ class `ORB_SLAM3::ORBextractor`
 * (include/ORBextractor.h:41-119, sole call site src/Frame.cc:555-562) and the inner loops of
 * `ORB_SLAM3::ORBmatcher` (include/ORBmatcher.h:35-148).  Every entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only; no C++ or torch types.  The C++ adaptor
 * classes with the reference's own signatures live in include/orbslam3_amd/.
 *
 * Return codes: >= 0 success (orbx_extract: monoIndex as the reference returns it),
 *   -1 empty image (src/ORBextractor.cc:1060-1062), -2 bad arguments, -3 HIP runtime error
 *   (message via orbx_last_error()).  Nothing throws across this boundary.
 *
 * Threading (src/Frame.cc:202-205): one handle = one HIP stream and one set of device buffers; two
 * handles may be driven concurrently from two threads (stereo left/right), one handle may not.
 */
#ifndef ORBX_H_
#define ORBX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint layout (28 bytes): the adaptor memcpy's these into std::vector<cv::KeyPoint>. */
typedef struct {
  float x, y, size, angle, response;
  int   octave, class_id;
} orbx_keypoint_t;

typedef struct orbx_extractor orbx_t;

/* ---- ORBextractor ---------------------------------------------------------------------------- */

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (src/ORBextractor.cc:403-465; constructed at src/Tracking.cc:576-585). */
int  orbx_create(orbx_t** out, int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast, int device_id);
void orbx_destroy(orbx_t* h);
const char* orbx_last_error(void);

/* GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:63-85); any pointer may be NULL. */
int  orbx_get_levels(const orbx_t* h);
void orbx_get_scale_tables(const orbx_t* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2);
void orbx_get_features_per_level(const orbx_t* h, int* out);

/* Upper bound on keypoints one image of this size can yield (sum over levels of max(N_l + 3, 4 nIni_l));
 * output buffers of orbx_extract* must hold at least this many entries. -2 if the size is unsupported. */
/* Fuses the cv::resize(im, im, cv::Size(new_cols, new_rows)) that System::TrackMonocular / TrackStereo / TrackRGBD apply when
 * the settings ask for it (src/System.cc:472-478; Camera.newWidth / newHeight) into the extractor: images of any other size
 * are first resized on the device (same fixed-point INTER_LINEAR as the pyramid), and orbx_max_keypoints / the pyramid
 * accessors then refer to the resized frame.  0, 0 disables.  Shrink factors up to 1.32 (x) / 1.4 (y). */
int orbx_set_input_resize(orbx_t* h, int new_cols, int new_rows);

int orbx_max_keypoints(orbx_t* h, int rows, int cols);

/* How the two tap-rotation expressions of computeOrbDescriptor (src/ORBextractor.cc:105-106),
 *     cvRound(pattern[idx].x * b + pattern[idx].y * a)   and   cvRound(pattern[idx].x * a - pattern[idx].y * b),
 * are evaluated -- the one place where the reference's results depend on how it was compiled (CMakeLists.txt:10-17 builds with
 * -O3 -march=native and g++ contracts a * b + c into an FMA by default):
 *   1 (default)  fma(x, b, y * a) and fma(x, a, -(y * b)): what g++ 11 emits for these expressions with the reference's own flags on
 *                any host with FMA (x86-64 Haswell or later, aarch64), i.e. the reference as its CMake builds it;
 *   0            every product rounded, then the sum: the reference built for baseline x86-64 or with -ffp-contract=off;
 *   2            fma(y, a, x * b) and fma(-y, b, x * a): the other legal contraction.
 * The two forms differ in about one descriptor out of 10^5 (a tap landing on the other side of a rounding tie). */
int orbx_set_tap_contraction(orbx_t* h, int mode);

/* ORBextractor::operator()(image, mask [ignored], keypoints, descriptors, vLappingArea)
 * (src/ORBextractor.cc:1052-1138).  Host image in, host results out; synchronous.
 * kps / desc are caller-owned (cap entries, desc is cap x 32 bytes); *n_out = number of keypoints.
 * Returns monoIndex (:1137). */
int orbx_extract(orbx_t* h, const uint8_t* img, int rows, int cols, size_t step_bytes, int lap0, int lap1,
                 orbx_keypoint_t* kps, uint8_t* desc, int cap, int* n_out);

/* The same operator over a batch of frames that are already resident in device memory (a camera rig or
 * a stream chunk): frame f starts at d_imgs + f * frame_stride_bytes, rows of step_bytes.  Results stay
 * on the device: d_kps[f * cap + i], d_desc[(f * cap + i) * 32], d_counts[2 f] = n, d_counts[2 f + 1] =
 * monoIndex.  Only enqueues work on `stream` (a hipStream_t; NULL = the handle's own stream); the caller
 * synchronises.  Input buffers must stay valid until then. */
int orbx_extract_batch_device(orbx_t* h, const uint8_t* d_imgs, int nframes, int rows, int cols, size_t step_bytes,
                              size_t frame_stride_bytes, int lap0, int lap1, orbx_keypoint_t* d_kps, uint8_t* d_desc,
                              int* d_counts, int cap, void* stream);

/* Streaming form of the same operator for HOST-resident frames (a camera rig, a stream): the call site of the reference
 * (src/Frame.cc:202-205, 555-562) blocks per image; a host with N cameras or a recorded stream instead keeps several chunks in
 * flight so that the upload of chunk k + 1, the kernels of chunk k and the download of chunk k - 1 overlap (three HIP streams).
 *   orbx_extract_async  enqueues one chunk of nframes frames (frame f at imgs + f * frame_stride_bytes) and returns a ticket;
 *                       kps [nframes][cap], desc [nframes][cap][32] and counts [nframes][2] (n, monoIndex) are HOST buffers
 *                       that hold the results once orbx_wait(ticket) has returned.  Host memory obtained from orbx_host_alloc
 *                       (or registered with hipHostRegister) is read / written by DMA directly; pageable memory is staged
 *                       through pinned buffers of the handle (one extra host copy each way).  All buffers must stay valid until
 *                       the ticket has been waited for.  At most `depth` tickets (orbx_set_async_depth, default 3, <= 8) may be
 *                       outstanding; tickets are waited for in any order.
 *   orbx_async_device_buffers  the device-side records of a ticket in flight and the stream its kernels were enqueued on, so
 *                       that the caller can chain device work (orbm_match_pairs_device, orbv_transform_batch_device ...) behind
 *                       the extraction without another upload.  Valid until the slot is reused (depth submissions later).
 *   orbx_async_download enqueues one more device -> host copy for a ticket in flight, ordered behind everything enqueued so far
 *                       on the ticket's stream (the results of the chained work); host_dst must be pinned.  orbx_wait(ticket)
 *                       then covers it as well.
 * Returns 0, -1 (empty input), -2, -3 as above. */
void* orbx_host_alloc(size_t bytes); /* pinned host memory (hipHostMalloc); NULL on failure */
void  orbx_host_free(void* p);
int   orbx_set_async_depth(orbx_t* h, int depth);
int   orbx_extract_async(orbx_t* h, const uint8_t* imgs, int nframes, int rows, int cols, size_t step_bytes, size_t frame_stride_bytes,
                         int lap0, int lap1, orbx_keypoint_t* kps, uint8_t* desc, int* counts, int cap, int* ticket);
int   orbx_async_device_buffers(orbx_t* h, int ticket, const orbx_keypoint_t** d_kps, const uint8_t** d_desc, const int** d_counts,
                                void** stream);
int   orbx_async_download(orbx_t* h, int ticket, void* host_dst, const void* device_src, size_t bytes);
int   orbx_wait(orbx_t* h, int ticket);

/* orbx_extract as a replayed hipGraph (default on): the second orbx_extract call with the same (size, step, lapping area, source
 * buffer, device allocations) records its upload / 16 kernels / downloads by stream capture and later calls replay that graph with
 * one launch -- 0.191 -> 0.167 ms per 640x480 call on MI355X.  Any other shape, a reallocation of any device buffer, profiling or a
 * changed tap mode falls back to the eager sequence and records again.  enable = 0 keeps every call eager. */
int orbx_set_graph_replay(orbx_t* h, int enable);

/* A hipStream_t with a hardware queue of its own, for the `stream` argument of the *_device entry points when several chunks are kept in
 * flight: ordinary HIP streams share a few hardware queues and two streams on one queue run in submission order, so chunks meant to
 * overlap (one chunk's latency-bound quadtree / claim resolution beside another's FAST / blur / top-k) would not.  NULL on failure. */
void* orbx_stream_create(int device_id);
void  orbx_stream_destroy(void* stream);

/* The one exchange step of the path: when a rig's cameras are extracted on several GPUs -- one process per GPU, each
 * with its own orbx_t -- the fixed-capacity records of every rank's cameras are gathered onto the rank that tracks
 * (src/Tracking.cc:1329-1608 consumes all cameras of a step).  RCCL underneath, opened lazily (no RCCL dependency for hosts that never
 * create a communicator); direct sends to the root over each peer's own xGMI link in one group, no ring, no staging.
 *   orbx_comm_unique_id  one rank (normally 0) creates the 128-byte rendezvous token; the host hands it to every rank by its own
 *                        means (file, socket, MPI, environment).
 *   orbx_comm_create     collective: every rank calls it with the same token, its rank and the number of ranks.
 *   orbx_gather          collective: rank r contributes its nframes records (device arrays laid out as orbx_extract_batch_device
 *                        writes them: d_kps [nframes][cap], d_desc [nframes][cap][32], d_counts [nframes][2]); the root receives
 *                        d_all_kps [nranks][nframes][cap], d_all_desc [nranks][nframes][cap][32], d_all_counts [nranks][nframes][2]
 *                        (ignored on the other ranks, may be NULL there).  Enqueued on `stream` (a hipStream_t); returns at once.
 * Return 0, -2 (bad arguments), -3 (RCCL / HIP error; orbx_last_error has the text). */
#define ORBX_UNIQUE_ID_BYTES 128
typedef struct orbx_comm orbx_comm_t;
int  orbx_comm_unique_id(void* id_out);
int  orbx_comm_create(orbx_comm_t** out, int device_id, int rank, int nranks, const void* unique_id);
void orbx_comm_destroy(orbx_comm_t* c);
int  orbx_comm_rank(const orbx_comm_t* c);
int  orbx_comm_size(const orbx_comm_t* c);
int  orbx_gather(orbx_comm_t* c, const orbx_keypoint_t* d_kps, const uint8_t* d_desc, const int* d_counts, int nframes, int cap,
                 orbx_keypoint_t* d_all_kps, uint8_t* d_all_desc, int* d_all_counts, int root, void* stream);

/* mvImagePyramid[level] (public member, include/ORBextractor.h:87; read by Frame::ComputeStereoMatches,
 * src/Frame.cc:950,1039-1058) of frame `frame` of the last extract call: copies the level to host memory.
 * dst may be NULL to query the size. */
int orbx_pyramid_level(orbx_t* h, int frame, int level, uint8_t* dst, size_t dst_step, int* rows, int* cols);

/* enable != 0: every orbx_extract also brings levels >= 1 of its pyramid to pinned host memory, behind the records and before the
 * call's one synchronisation (one more copy in the same queue instead of a blocking round trip per level); orbx_pyramid_level(h, 0,
 * level >= 1, ...) is then a host copy.  Level 0 is the caller's own image (or, with orbx_set_input_resize, fetched on demand).
 * This is what the adaptor's mbDownloadPyramid (default true: an unchanged Frame::ComputeStereoMatches reads mvImagePyramid,
 * src/Frame.cc:950, 1039-1058) turns on.  Off by default at this level.  Returns 0, -2 for a null handle. */
int orbx_set_pyramid_download(orbx_t* h, int enable);

/* Frame::ComputeStereoMatches (reference src/Frame.cc:944-1113) on the device-resident pyramids of two extractors
 * that have just run orbx_extract on the left / right image (same size, same device): row-band candidate search,
 * best Hamming below TH_HIGH, 11 x 11 SAD over +-5 px on the keypoint's pyramid level, parabola sub-pixel fit,
 * median-distance cut.  Host pointers, synchronous.  Writes mvuRight / mvDepth (-1 = no match); mb, mbf = Frame::mb,
 * Frame::mbf.  Returns the number of stereo matches kept.  With this call the caller no longer needs
 * orbx_pyramid_level() copies of mvImagePyramid for the stereo case. */
int orbx_stereo_matches(orbx_t* left, orbx_t* right, const orbx_keypoint_t* kps_left, const uint8_t* desc_left, int n_left,
                        const orbx_keypoint_t* kps_right, const uint8_t* desc_right, int n_right, float mb, float mbf, float* u_right,
                        float* depth);

/* The same for every frame pair of two extraction batches, device-resident: the two handles must have run
 * orbx_extract_batch_device on same-sized batches of at least nframes frames (their pyramids and -- for level 0 -- the caller's
 * image buffers of those calls are read, so the images must still be there); d_kl / d_dl / d_counts_l and d_kr / d_dr /
 * d_counts_r are the output buffers of those calls (capacity cap each).  Frame f writes d_u_right / d_depth at [f * cap ..]
 * and d_kept[f].  The stream must be ordered after both extractions.  Enqueue only. */
int orbx_stereo_matches_batch_device(orbx_t* left, orbx_t* right, const orbx_keypoint_t* d_kl, const uint8_t* d_dl, const int* d_counts_l,
                                     const orbx_keypoint_t* d_kr, const uint8_t* d_dr, const int* d_counts_r, int nframes, int cap, float mb,
                                     float mbf, float* d_u_right, float* d_depth, int* d_kept, void* stream);

/* Stage timing with HIP events recorded on the launch stream between the kernels of every batch:
 * enable (resets the record), run batches, then read the mean milliseconds per stage over the batches
 * recorded since (at most the last 32); returns the number of batches averaged.  Stages: 0 pyramid
 * (all resize launches), 1 FAST, 2 quadtree, 3 layout, 4 blur (with ORBX_OVERLAP_BLUR=1: the wait for the blur's
 * side stream), 5 orientation+descriptor, 6 whole batch. */
#define ORBX_NUM_STAGES 7
void orbx_set_profiling(orbx_t* h, int enable);
int  orbx_get_stage_ms(orbx_t* h, float* ms /* ORBX_NUM_STAGES */);

/* Parity-test hooks (frame / level of the last call): blurred level, the candidate feed of
 * DistributeOctTree (packed x | y << 12 | response << 24, minBorder-relative), and the per-level
 * keypoints after the quadtree in list order (same packing).  Each returns the element count. */
int orbx_debug_blurred_level(orbx_t* h, int frame, int level, uint8_t* dst, size_t dst_step, int* rows, int* cols);
int orbx_debug_level_candidates(orbx_t* h, int frame, int level, uint32_t* out, int cap);
int orbx_debug_level_keypoints(orbx_t* h, int frame, int level, uint32_t* out, int cap);
/* Tuning hook (set ORBX_DBG_OCT=1 before orbx_create): per (frame, level) of the last batch
 * 8 values of the quadtree kernel: {sort cycles, total cycles, division steps, sorts, cycles in the passes over the
 * points, in the per-node scans, in the node rewrite, in the root set-up}. Returns the number of values. */
int orbx_debug_octree_times(orbx_t* h, long long* out, int cap);
/* PMC calibration: stream nbytes of scratch memory with `width` (1, 4 or 16) bytes per lane per load. */
int orbx_debug_calib_read(orbx_t* h, long long nbytes, int width);
/* Diagnosis: out[2 i], out[2 i + 1] = device address and size in bytes of the handle's i-th workspace buffer (names: tools/dbg_buffer_map.py);
 * returns the number of buffers.  For matching a GPU memory-fault address with the allocation it lies behind. */
int orbx_debug_buffers(orbx_t* h, unsigned long long* out, int cap);
/* Counts that drive loops over device memory (FAST cell counts, per-level keypoint counts, the counts of a stereo batch) are clamped to
 * the capacity of what they index; the first one that had to be is recorded: out = {code, frame, value, where} (codes: orbx_kernels.h).
 * Returns 1 and clears the record if there is one, 0 if none (a healthy run), < 0 on error.  Synchronises the device. */
int orbx_debug_fault(orbx_t* h, int out[4]);
/* VALU issue-rate calibration (csrc/orbx_calib.hip): independent chains of one instruction class with waves_per_simd (1 .. 8) waves
 * resident per SIMD on every CU.  out = {nanoseconds of the launch, median shader cycles per wave, wave-instructions per wave};
 * orbx_debug_valu_calib_name(op) names the instruction (NULL past the last one).  Tuning hook: tools/valu_calibrate.py. */
int         orbx_debug_valu_calib(int op, int waves_per_simd, int iters, double out[3]);
const char* orbx_debug_valu_calib_name(int op);
/* LDS read issue rate: ds_read_b32 (width 4) / ds_read_b64 (width 8) at lane * width + off; off != 0 = not naturally aligned.
 * out = {ns of the launch, median shader cycles per wave, LDS read instructions per wave, lanes that read wrong bytes}. */
int orbx_debug_lds_calib(int width, int off, int waves_per_simd, int iters, double out[4]);
/* Device run of DistributeOctTree alone on caller-supplied candidates (host arrays). */
int orbx_debug_octree_device(orbx_t* h, const uint32_t* cand, int n, int W, int H, int N, uint32_t* out, int* out_count);
/* The kernel's std::sort re-enactment alone (one wave) on n host records, in place: order = record >> 16, low 16 bits = payload;
 * depth_limit < 0 = libstdc++'s own introsort depth limit, small values reach its heap-sort path on any input. */
int orbx_debug_wave_sort(orbx_t* h, unsigned long long* records, int n, int depth_limit);
/* orbx_debug_octree_device, keeping the records of every std::sort re-enactment of the run: sorts[0] = their number, then per sort 1 + 2 * 256 words
 * (V, the V records before, the V records after); at most max_sorts are kept (sorts holds 1 + 513 * max_sorts words). */
int orbx_debug_octree_sorts(orbx_t* h, const uint32_t* cand, int n, int W, int H, int N, uint32_t* out, int* out_count, unsigned long long* sorts,
                            int max_sorts);

/* ---- ORBmatcher ------------------------------------------------------------------------------ */

typedef struct orbm_matcher orbm_t;

int  orbm_create(orbm_t** out, int device_id);
void orbm_destroy(orbm_t* h);
/* Message of the calling thread's last failed orbm_* call.  orbx_last_error() returns the most recent message of ANY of the three
 * families (orbx_ / orbm_ / orbv_): they share one slot per thread. */
const char* orbm_last_error(void);

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2075-2089); host-side, for the call sites that
 * compare single pairs (src/Frame.cc:1019, src/MapPoint.cc:426). */
int orbm_hamming(const uint8_t* a32, const uint8_t* b32);

/* ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:2038-2071) on bin counts. */
void orbm_three_maxima(const int* counts, int len, int* ind1, int* ind2, int* ind3);

/* ORBmatcher::SearchByBoW, both overloads (src/ORBmatcher.cc:237-421 [F.Nleft == -1] and 784-914), over
 * flattened DBoW2 FeatureVector groups: group g pairs the query (KeyFrame) features
 * q_idx[q_off[g] .. q_off[g+1]) with the train (Frame) features t_idx[t_off[g] .. t_off[g+1]).
 * ngroups == 1 with all four arrays NULL = plain brute force.  q_valid / t_valid: 1 = feature has a good
 * MapPoint (NULL = all).  strict_lt = 0: best <= th (:319), 1: best < th (:864).  Claims are greedy in query
 * order as in the reference.  Host pointers, synchronous.  match_t_of_q[nq] / match_q_of_t[nt]: -1 = none;
 * either may be NULL.  Returns nmatches. */
int orbm_match_groups(orbm_t* h, const uint8_t* q, int nq, const uint8_t* t, int nt, int ngroups, const int* q_off,
                      const int* q_idx, const int* t_off, const int* t_idx, const uint8_t* q_valid, const uint8_t* t_valid,
                      const float* q_angle, const float* t_angle, int th, int strict_lt, float ratio, int check_ori,
                      int* match_t_of_q, int* match_q_of_t);

/* The same overload for a two-camera frame (F.Nleft != -1: a second fisheye camera, src/ORBmatcher.cc:305-333 and 362-389).  The train
 * features are the frame's left keys [0, t_nleft) followed by its right keys (F.mDescriptors is stacked that way); a query keeps a
 * best / second-best pair per camera, the right-camera match needs only best <= th (the ratio test is disabled there in the reference)
 * and is only looked at when the left best is <= th, exactly as the reference nests it, so one query can claim two train features.
 * match_q_of_t[nt] (= vpMapPointMatches): the query that claimed the train feature, -1 = none.  nt <= 8192.  Returns nmatches. */
int orbm_match_groups_two_cameras(orbm_t* h, const uint8_t* q, int nq, const uint8_t* t, int nt, int t_nleft, int ngroups, const int* q_off,
                                  const int* q_idx, const int* t_off, const int* t_idx, const uint8_t* q_valid, const float* q_angle,
                                  const float* t_angle, int th, float ratio, int check_ori, int* match_q_of_t);

/* Device-resident brute-force form (ngroups == 1, all features valid) for streams whose descriptors never
 * leave the GPU: d_q / d_t are device descriptor arrays, d_q_kp / d_t_kp the device keypoint arrays written
 * by orbx_extract_batch_device (angles are read from them), d_nq / d_nt device ints with the live counts
 * (<= cap_q, cap_t).  Writes d_match_t_of_q[cap_q] and d_nmatches[1].  Enqueue only. */
int orbm_match_bruteforce_device(orbm_t* h, const uint8_t* d_q, const orbx_keypoint_t* d_q_kp, const int* d_nq, int cap_q,
                                 const uint8_t* d_t, const orbx_keypoint_t* d_t_kp, const int* d_nt, int cap_t, int th,
                                 int strict_lt, float ratio, int check_ori, int* d_match_t_of_q, int* d_nmatches, void* stream);

/* The same search for many frame pairs of one extraction batch in two launches (a camera stream: frame f-1
 * against frame f).  d_desc / d_kps / d_counts / cap are the buffers orbx_extract_batch_device filled;
 * d_pairs holds npairs x {query frame, train frame} (device ints).  Pair p writes d_match_t_of_q[p * cap ..]
 * and d_nmatches[p].  Enqueue only. */
int orbm_match_pairs_device(orbm_t* h, const uint8_t* d_desc, const orbx_keypoint_t* d_kps, const int* d_counts, int cap,
                            const int* d_pairs, int npairs, int th, int strict_lt, float ratio, int check_ori, int* d_match_t_of_q,
                            int* d_nmatches, void* stream);

/* SearchByBoW proper for the pairs of an extraction batch: as orbm_match_pairs_device, but every query feature only meets
 * the train features under the same vocabulary node, the groups and the query order (node ascending, feature ascending -- the
 * merge walk of src/ORBmatcher.cc:256-268) being taken from the FeatureVectors orbv_transform_batch_device wrote for the batch
 * (d_fv_nodes / d_fv_off / d_fv_feat / d_n_nodes, same cap).  Features of stopped words take no part, as in the reference.
 * Enqueue only. */
int orbm_match_pairs_bow_device(orbm_t* h, const uint8_t* d_desc, const orbx_keypoint_t* d_kps, const int* d_counts, int cap, const int* d_pairs,
                                int npairs, const uint32_t* d_fv_nodes, const int* d_fv_off, const uint32_t* d_fv_feat, const int* d_n_nodes,
                                int th, int strict_lt, float ratio, int check_ori, int* d_match_t_of_q, int* d_nmatches, void* stream);

/* Tuning hook for orbm_match_pairs_device: enable = 1 arms the counters {speculative rounds, exact rescans,
 * 64-query chunks}; enable = 0 synchronises the device and copies them to out[3]. */
int orbm_debug_counters(orbm_t* h, int enable, int* out);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:666-782) including Frame::GetFeaturesInArea over
 * the 64 x 48 grid (src/Frame.cc:524-553, 788-877).  kp1 / kp2 are undistorted keypoints; prev_matched is
 * n1 x 2 floats, updated in place (:775-779); bounds = Frame::mnMinX..mnMaxY.  Host pointers, synchronous. */
int orbm_match_window(orbm_t* h, const uint8_t* d1, const orbx_keypoint_t* kp1, int n1, const uint8_t* d2,
                      const orbx_keypoint_t* kp2, int n2, float* prev_matched, float min_x, float max_x, float min_y,
                      float max_y, int window, int th, float ratio, int check_ori, int* matches12);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBmatcher.cc:1684-1908,
 * CurrentFrame.Nleft == -1) with Frame::GetFeaturesInArea over the 64 x 48 grid (src/Frame.cc:788-865).
 * The caller does the geometry of the loop head (:1706-1729): last_valid[i] = MapPoint present, not an outlier, in front of
 * the camera and projecting inside the image bounds; last_uv[i] = projection, last_invz[i] = 1/z (needed only with
 * cur_u_right), last_octave / last_angle = the last frame's keypoint, last_desc = MapPoint::GetDescriptor().
 * cur_kp = undistorted keypoints; cur_u_right = mvuRight or NULL (monocular); cur_occupied[i2] = the feature already holds
 * a MapPoint with observations (NULL = none).  dir: 0 neither, 1 bForward, 2 bBackward (:1701-1702), 3 = levels
 * [octave - 1, octave] (the window of orbm_search_in_windows); th_dist = TH_HIGH.
 * A matched feature is closed to later points (all tracked points are taken to have observations; the zero-observation
 * temporal points of stereo/RGB-D odometry stay with the caller).  Outputs cur_of_last[n1], last_of_cur[n2] (may be NULL);
 * returns nmatches.  Host pointers, synchronous. */
int orbm_search_by_projection(orbm_t* h, const uint8_t* last_desc, const float* last_uv, const float* last_invz, const int* last_octave,
                              const float* last_angle, const uint8_t* last_valid, int n1, const uint8_t* cur_desc,
                              const orbx_keypoint_t* cur_kp, const float* cur_u_right, const uint8_t* cur_occupied, int n2,
                              const float* scale_factors, int nlevels, float th, int dir, float mbf, float min_x, float max_x,
                              float min_y, float max_y, int th_dist, int check_ori, int* cur_of_last, int* last_of_cur);

/* Frame::UndistortKeyPoints (src/Frame.cc:886-913): cv::undistortPoints(pts, pts, K, mDistCoef, cv::Mat(), mK)
 * on the keypoint coordinates -- OpenCV's five fixed-point iterations in double precision -- every other field copied.
 * K = {fx, fy, cx, cy}; dist = k1 k2 p1 p2 [k3 ...] (ndist <= 12); dist[0] == 0 copies the keypoints, as the reference does.
 * The device form works on the buffers of orbx_extract_batch_device (d_counts may be NULL: all cap records); enqueue only. */
int orbx_undistort_keypoints(orbx_t* h, const orbx_keypoint_t* kps, int n, const float K[4], const float* dist, int ndist, orbx_keypoint_t* out);
int orbx_undistort_keypoints_device(orbx_t* h, const orbx_keypoint_t* d_kps, const int* d_counts, int nframes, int cap, const float K[4],
                                    const float* dist, int ndist, orbx_keypoint_t* d_out, void* stream);
/* Frame::ComputeImageBounds (src/Frame.cc:915-943): out = {mnMinX, mnMaxX, mnMinY, mnMaxY}; host arithmetic, same core. */
int orbx_image_bounds(int cols, int rows, const float K[4], const float* dist, int ndist, float out[4]);

/* ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)
 * (src/ORBmatcher.cc:38-227, F.Nleft == -1) -- the local-map tracking search.  mp_valid[i] = mbTrackInView && !isBad() &&
 * !(bFarPoints && mTrackDepth > thFarPoints); mp_uv = (mTrackProjX, mTrackProjY); mp_proj_xr = mTrackProjXR (only read with
 * cur_u_right); mp_level = mnTrackScaleLevel; mp_radius = RadiusByViewingCos(mTrackViewCos) (* th when th != 1).  Window
 * mp_radius * scale[level] over levels [level - 1, level]; best and second-best unclaimed candidate; accepted when
 * best <= th_dist (TH_HIGH) unless both sit on one pyramid level and best > ratio * second (:137-142).  Same conventions for
 * cur_* and the outputs as orbm_search_by_projection.  Host pointers, synchronous. */
int orbm_search_by_projection_map(orbm_t* h, const uint8_t* mp_desc, const float* mp_uv, const float* mp_proj_xr, const int* mp_level,
                                  const float* mp_radius, const uint8_t* mp_valid, int n1, const uint8_t* cur_desc,
                                  const orbx_keypoint_t* cur_kp, const float* cur_u_right, const uint8_t* cur_occupied, int n2,
                                  const float* scale_factors, int nlevels, float ratio, float min_x, float max_x, float min_y, float max_y,
                                  int th_dist, int* cur_of_mp, int* mp_of_cur);

/* The same search for a two-camera frame (F.Nleft != -1; src/ORBmatcher.cc:139-146 and 152-226).  cur_desc / cur_kp hold the left keys
 * [0, nleft) followed by the right keys (mvKeys, mvKeysRight), cur_occupied likewise; the two grids of Frame::AssignFeaturesToGrid are
 * built from them.  Per map point: valid_l = mbTrackInView, valid_r = mbTrackInViewR && mnTrackScaleLevelR != -1 (both 0 for the points
 * the loop head skips, :50-62); uv_l / level_l / radius_l = (mTrackProjX, mTrackProjY), mnTrackScaleLevel, RadiusByViewingCos(mTrackViewCos)
 * (* th when th != 1); uv_r / level_r / radius_r = the ...R members, the radius without the factor (:158).  An accepted left match also
 * hands the point to left_to_right[best] + nleft (mvLeftToRightMatch, -1 = none), an accepted right match to right_to_left[best]; a
 * ratio veto in the left block skips the point's right search, as the reference's `continue` does.  mp_of_cur[n] = F.mvpMapPoints after
 * the call (map point index, -1 = untouched).  n <= 16384.  Returns nmatches (every assignment counts, as in the reference). */
int orbm_search_by_projection_map_two_cameras(orbm_t* h, const uint8_t* mp_desc, int n1, const float* uv_l, const int* level_l, const float* radius_l,
                                              const uint8_t* valid_l, const float* uv_r, const int* level_r, const float* radius_r,
                                              const uint8_t* valid_r, const uint8_t* cur_desc, const orbx_keypoint_t* cur_kp, int n, int nleft,
                                              const uint8_t* cur_occupied, const int* left_to_right, const int* right_to_left,
                                              const float* scale_factors, int nlevels, float ratio, float min_x, float max_x, float min_y, float max_y,
                                              int th_dist, int* mp_of_cur);

/* The windowed best-candidate search of the remaining projection matchers, on arrays:
 *   ORBmatcher::SearchByProjection(KeyFrame*, Sophus::Sim3f&, vpPoints, vpMatched, th, ratioHamming) (src/ORBmatcher.cc:423-538; :540-664
 *     is the same search) -- claim = 1, t_occupied[idx] = (vpMatched[idx] != NULL), th_dist = floor(TH_LOW * ratioHamming);
 *   ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th, bRight) (:1149-1328) -- claim = 0, inv_level_sigma2 = mvInvLevelSigma2 (chi-square gate 5.99,
 *     or 7.8 on the three-term error where t_u_right[idx] >= 0, q_ur = uv(0) - bf * invz), th_dist = TH_LOW;
 *   ORBmatcher::Fuse(KeyFrame*, Sim3f&, vpPoints, th, vpReplacePoint) (:1330-1452) -- claim = 0, no gate, th_dist = TH_LOW;
 *   each direction of ORBmatcher::SearchBySim3 (:1499-1577, :1580-1658) -- claim = 0, no gate, th_dist = TH_HIGH.
 * (SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist), :1910-2036, is orbm_search_by_projection with dir = 0,
 * last_octave = the predicted level, last_angle = pKF->mvKeysUn[i].angle and th_dist = ORBdist.)
 * The loop heads (bad / already found points, depth, IsInImage, distance invariance, viewing angle, PredictScale) are the caller's:
 * q_valid[i] (NULL = all) says whether point i reaches the search, q_uv its projection, q_level the predicted level, q_desc =
 * MapPoint::GetDescriptor().  Window = KeyFrame::GetFeaturesInArea(u, v, th * scale_factors[level]) (src/KeyFrame.cc:780-838) over the
 * 64 x 48 grid on [min_x, max_x] x [min_y, max_y]; keypoints on levels [level - 1, level]; first minimum in scan order; accepted when
 * <= th_dist.  claim = 1 skips occupied features and closes a matched feature to later points; claim = 0 lets every point search on
 * its own (the map updates that follow -- Replace / AddObservation -- stay with the caller, replayed in point order).
 * t_of_q[n1] = matched feature or -1; returns the number of matched points.  Host pointers, synchronous. */
int orbm_search_in_windows(orbm_t* h, const uint8_t* q_desc, const float* q_uv, const float* q_ur, const int* q_level, const uint8_t* q_valid, int n1,
                           const uint8_t* t_desc, const orbx_keypoint_t* t_kp, const float* t_u_right, const uint8_t* t_occupied, int n2,
                           const float* scale_factors, const float* inv_level_sigma2, int nlevels, float th, int claim, float min_x, float max_x,
                           float min_y, float max_y, int th_dist, int* t_of_q);

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (src/ORBmatcher.cc:916-1147)
 * for two keyframes of one pinhole camera (mpCamera2 == NULL, NLeft == -1), over the vocabulary nodes both FeatureVectors hold,
 * flattened as for orbm_match_groups: group g pairs KF1's features q_idx[q_off[g] .. q_off[g+1]) with KF2's t_idx[t_off[g] ..).
 * mp1 / mp2[i] = the feature already has a MapPoint (skipped, :980-984, :1007-1011); stereo1 / stereo2[i] = mvuRight[i] >= 0 (NULL =
 * monocular); F12 = K1^-T [t12]x R12 K2^-1, row major, exactly the matrix Pinhole::epipolarConstrain forms
 * (src/CameraModels/Pinhole.cpp:152-156); ep = pKF2->mpCamera->project(T2w * Cw) (:931-934); level_sigma2 = mvLevelSigma2;
 * th_low = TH_LOW; coarse = bCoarse (skips the epipolar test; F12 / level_sigma2 may then be NULL).  Per KF1 feature: the candidate
 * with the smallest distance <= th_low that is not within 100 * scale[octave] (squared pixels) of the epipole (monocular pairs) and
 * lies within 3.84 * sigma2 of the epipolar line, the last of equals in list order; then the rotation histogram (:1112-1129).
 * The reference never sets vbMatched2, so several KF1 features may receive the same KF2 feature -- reproduced.
 * matches12[n1] = KF2 feature or -1 (vMatchedPairs = the pairs (i, matches12[i]) in ascending i); returns nmatches.
 * KannalaBrandt8 / two-camera rigs use a triangulation-based test (src/CameraModels/KannalaBrandt8.cpp) and stay with the caller
 * unless bCoarse.  Host pointers, synchronous. */
int orbm_search_for_triangulation(orbm_t* h, const uint8_t* d1, const orbx_keypoint_t* kp1, const uint8_t* mp1, const uint8_t* stereo1, int n1,
                                  const uint8_t* d2, const orbx_keypoint_t* kp2, const uint8_t* mp2, const uint8_t* stereo2, int n2, int ngroups,
                                  const int* q_off, const int* q_idx, const int* t_off, const int* t_idx, const float F12[9], const float ep[2],
                                  const float* scale_factors, const float* level_sigma2, int nlevels, int only_stereo, int coarse, int th_low,
                                  int check_ori, int* matches12);

/* Frame::AssignFeaturesToGrid (src/Frame.cc:524-553) for every frame of an extraction batch, on the device: the (undistorted)
 * keypoints that fall inside the 64 x 48 grid over [min_x, max_x] x [min_y, max_y], in the scan order of GetFeaturesInArea
 * (cells (ix, iy) ascending, feature index ascending inside a cell).  Frame f writes d_g_idx / d_g_cx / d_g_cy at [f * cap ..]
 * (feature index, cell column, cell row) and d_ng[f].  cap <= 8192.  Enqueue only. */
int orbm_assign_features_to_grid_device(orbm_t* h, const orbx_keypoint_t* d_kps_un, const int* d_counts, int nframes, int cap, float min_x,
                                        float max_x, float min_y, float max_y, int* d_g_idx, uint8_t* d_g_cx, uint8_t* d_g_cy, int* d_ng,
                                        void* stream);
/* orbm_search_by_projection for npairs (last frame, current frame) pairs of one extraction batch, device-resident: d_desc /
 * d_kps_un / d_counts / cap are the batch buffers (keypoints undistorted); d_pairs = npairs x {last, current} frame indices;
 * the per-pair inputs d_last_uv [npairs][cap][2], d_last_invz, d_last_valid, d_occupied [npairs][cap] and the optional
 * d_last_desc [npairs][cap][32] (MapPoint descriptors; NULL = the last frame's own descriptors) follow the single-pair
 * meanings; d_u_right is [nframes][cap] or NULL.  The grid is built on the device.  Pair p writes d_cur_of_last /
 * d_last_of_cur at [p * cap ..] and d_nmatches[p].  scale_factors is a host array.  Enqueue only. */
int orbm_search_by_projection_batch_device(orbm_t* h, int npairs, const int* d_pairs, const uint8_t* d_desc, const orbx_keypoint_t* d_kps_un,
                                           const int* d_counts, int nframes, int cap, const uint8_t* d_last_desc, const float* d_last_uv,
                                           const float* d_last_invz, const uint8_t* d_last_valid, const float* d_u_right,
                                           const uint8_t* d_occupied, const float* scale_factors, int nlevels, float th, int dir, float mbf,
                                           float min_x, float max_x, float min_y, float max_y, int th_dist, int check_ori, int* d_cur_of_last,
                                           int* d_last_of_cur, int* d_nmatches, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * ORBVocabulary::transform -- DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> as
 * Frame::ComputeBoW uses it (src/Frame.cc:879-884), for the configuration of ORBvoc.txt (scoring L1_NORM, weighting
 * TF_IDF).  Node i + 1 of the tree is entry i of the arrays (node 0 = root) in the order of the text file's lines
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1337-1420); words number the leaves in that order.  k <= 32.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct orbv_vocab orbv_t;
int         orbv_create(orbv_t** out, int k, int L, int n_nodes, const int* parent, const uint8_t* is_leaf, const uint8_t* desc32,
                        const double* weight, int device_id);
/* TemplatedVocabulary::loadFromTextFile (:1337-1420); blank lines are skipped (see orbx_vocab.hip). */
int         orbv_load_text(orbv_t** out, const char* path, int device_id);
void        orbv_destroy(orbv_t* h);
const char* orbv_last_error(void);
int         orbv_info(const orbv_t* h, int* k, int* L, int* n_nodes, int* n_words);
/* transform(features, BowVector&, FeatureVector&, levelsup) (:1127-1194).  Host pointers, synchronous.  BowVector comes
 * back as n_words (word id ascending, value) pairs, FeatureVector as n_nodes node ids (ascending) with
 * node_off[n_nodes + 1] offsets into feat_idx (feature indices ascending inside a node).  Every output needs room for n
 * entries (node_off: n + 1). */
int orbv_transform(orbv_t* h, const uint8_t* desc32, int n, int levelsup, uint32_t* word_ids, double* word_vals, int* n_words,
                   uint32_t* node_ids, int* node_off, uint32_t* feat_idx, int* n_nodes);
/* The same for every frame of an extraction batch, device-resident: d_desc / d_counts / cap are the buffers
 * orbx_extract_batch_device filled.  Frame f writes d_bow_words / d_bow_vals / d_fv_nodes / d_fv_feat at [f * cap ..],
 * d_fv_off at [f * (cap + 1) ..], d_n_words[f], d_n_nodes[f].  Enqueue only. */
int orbv_transform_batch_device(orbv_t* h, const uint8_t* d_desc, const int* d_counts, int nframes, int cap, int levelsup,
                                uint32_t* d_bow_words, double* d_bow_vals, int* d_n_words, uint32_t* d_fv_nodes, int* d_fv_off,
                                uint32_t* d_fv_feat, int* d_n_nodes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
