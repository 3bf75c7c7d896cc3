/*
 * rgbdfe.h -- C ABI of the MI355X-native RGB-D SLAM visual front end.
 *
 * This is the drop-in boundary behind rgbdslam_v2's Node / GraphManager seam
 * (the reference has no FFI of its own; the seams are the C++ signatures cited
 * on each entry point, paths relative to the rgbdslam_v2 tree).  Plain pointers
 * and sizes only; no exceptions cross this boundary; every function returns an
 * rgbdfe_status (0 = ok, <0 = error) unless noted.  INTEGRATION.md shows the
 * reference-side binding.
 *
 * Threading: a context owns its HIP streams and is internally locked; calls on
 * one context from any number of threads serialise (this replaces
 * QtConcurrent::blockingMapped's barrier, graph_manager.cpp:548, one call = one
 * batch of pairs).  rgbdfe_last_error returns a per-thread copy of the message.
 */
#ifndef RGBDFE_H
#define RGBDFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGBDFE_MAX_MATCHES 320   /* capacity for max_matches (reference default 300) */
#define RGBDFE_MASK_WORDS 5      /* RGBDFE_MAX_MATCHES / 64 */
#define RGBDFE_MAX_KEYPOINTS 65535

typedef enum {
  RGBDFE_OK = 0,
  RGBDFE_ERR_INVALID_ARG = -1,
  RGBDFE_ERR_NO_DEVICE = -2,    /* no HIP device / kernels cannot run: never falls back to CPU */
  RGBDFE_ERR_HIP = -3,
  RGBDFE_ERR_UNKNOWN_NODE = -4,
  RGBDFE_ERR_CAPACITY = -5,
  RGBDFE_ERR_OUT_OF_MEMORY = -6,
  RGBDFE_ERR_INTERNAL = -7      /* a C++ exception was caught at the ABI (none ever crosses it) */
} rgbdfe_status;

/* Snapshot of the ParameterServer values the pair path reads at call time
 * (parameter_server.cpp:85,86,100,101,46 ). */
typedef struct {
  int32_t  max_matches;           /* "max_matches"           default 300 (<= RGBDFE_MAX_MATCHES) */
  int32_t  min_matches;           /* "min_matches"           default 20  */
  int32_t  ransac_iterations;     /* "ransac_iterations"     default 200 */
  float    max_dist_for_inliers;  /* "max_dist_for_inliers"  default 3.0 */
  double   depth_cov;             /* value depth_covariance() froze at its first call
                                     (misc2.h:30-35): (sigma_depth * z0^2)^2, default z0 = 1 m -> 1e-4 */
  uint32_t seed;                  /* replaces srand(clock()) (node.cpp:1102) */
  uint32_t g2o_iterations;        /* "g2o_transformation_refinement" default 0 = off (parameter_server.cpp:103): Gauss-
                                     Newton steps of the two-view refinement after RANSAC (node.cpp:1222-1268); needs the
                                     nodes' 2-D keypoints, rgbdfe_upload_node_keypoints */
} rgbdfe_params;

typedef struct {
  int32_t device_id;              /* HIP device ordinal */
  int32_t max_nodes;              /* resident node slots (each max_keypoints rows) */
  int32_t max_keypoints;          /* rows per node slot, <= RGBDFE_MAX_KEYPOINTS */
  int32_t max_pairs_per_batch;    /* pairs per kernel batch (results/keys staging) */
  rgbdfe_params params;
} rgbdfe_config;

/* Per-pair result: the MatchingResult POD (matching_result.h:24-46, edge.h:24-32).
 * This exact layout is what lives in HBM, what is all-gathered between ranks
 * and what rgbdfe_match_* copies to the host. */
typedef struct {
  int32_t  id1, id2;              /* edge.id1 = older node, edge.id2 = newer node; -1,-1 = no edge
                                     (node.cpp:1337-1338, 1419-1422) */
  int32_t  n_all;                 /* |all_matches| after keepStrongestMatches (node.cpp:674) */
  int32_t  n_inl;                 /* |inlier_matches| */
  float    rmse;                  /* MatchingResult::rmse */
  float    trafo[16];             /* ransac_trafo == final_trafo, Eigen::Matrix4f column-major,
                                     maps the newer node's frame into the older node's frame */
  uint32_t pad0;
  double   info_scale;            /* edge.informationMatrix = I6 * info_scale (node.cpp:1335) */
  int32_t  valid_iterations;      /* diagnostics (node.cpp:1216) */
  int32_t  real_iterations;
  uint16_t all_q[RGBDFE_MAX_MATCHES];  /* DMatch.queryIdx, sorted by (hd, queryIdx) */
  uint16_t all_t[RGBDFE_MAX_MATCHES];  /* DMatch.trainIdx */
  uint8_t  all_hd[RGBDFE_MAX_MATCHES]; /* Hamming distance; DMatch.distance = hd/256.0f */
  uint64_t inlier_mask[RGBDFE_MASK_WORDS]; /* bit m set <=> all_*[m] is in inlier_matches */
} rgbdfe_match_result;

/* The same record without its all_matches lists: what GraphManager consumes of a MatchingResult besides GUI drawing
 * (edge ids / transform / information scale, rmse, |inlier_matches|, graph_manager.cpp:554-606; all_matches only feeds
 * graph_mgr_io.cpp:739-1107).  This is the default payload of the multi-GPU gather: 144 B instead of 1744 B per pair.
 * Node features are replicated on every device, and a pair's result does not depend on the batch or the device it ran
 * on, so the full record of a pair whose lists are wanted (updateInlierFeatures, :409-419; drawing) is
 * rgbdfe_match_pair_list(ctx, &q, &t, 1, &full) on ANY device -- byte-identical to the owner's, no fetch needed. */
typedef struct {
  int32_t  id1, id2, n_all, n_inl;
  float    rmse;
  float    trafo[16];
  uint32_t pad0;
  double   info_scale;
  int32_t  valid_iterations, real_iterations;
  uint64_t inlier_mask[RGBDFE_MASK_WORDS];
} rgbdfe_compact_result;

/* The inlier form of a shard's results: what the consumer of the multi-GPU gather reads of a MatchingResult -- the edge
 * (ids, transform, information scale), rmse and counts as above, and the inlier matches' (queryIdx, trainIdx)
 * (GraphManager::updateInlierFeatures, graph_manager.cpp:409-419) -- without the all_matches lists and without a second
 * pair op on the receiving device.  A shard of n pairs is ONE byte stream:
 *   n_headers x rgbdfe_inlier_header   (the leading 104 bytes of rgbdfe_match_result; `first_inlier` = position of the pair's
 *                                       first inlier in the list block; headers n .. n_headers-1 are padding: ids -1, no inliers)
 *   total x uint32                     (the list block: query row | train row << 16 of every inlier, pair after pair, inliers
 *                                       in match order = ascending bit of inlier_mask)
 * total = sum of n_inl: 104 + 4 * n_inl bytes per pair instead of 1744 (configs[1]: ~260).  The default payload of
 * bench.py --gpus N. */
typedef struct {
  int32_t  id1, id2, n_all, n_inl;
  float    rmse;
  float    trafo[16];
  uint32_t first_inlier;
  double   info_scale;
  int32_t  valid_iterations, real_iterations;
} rgbdfe_inlier_header;

typedef struct rgbdfe_ctx rgbdfe_ctx;

/* ---- lifetime ---------------------------------------------------------- */
void rgbdfe_default_config(rgbdfe_config* cfg);
int  rgbdfe_create(const rgbdfe_config* cfg, rgbdfe_ctx** out);
void rgbdfe_destroy(rgbdfe_ctx* ctx);
/* ---- several GPUs behind ONE handle --------------------------------------------------
 * The reference's caller is one process (GraphManager::nodeComparisons, graph_manager.cpp:541-548), so the
 * drop-in form of "shard the candidate pairs over the 8 GPUs of a node" is a context that owns one device
 * context + one host thread per listed device (cfg->device_id is ignored).  With such a handle
 *   - rgbdfe_upload_node / _sift_node / _node_cloud / release / set_* act on every device (node features are
 *     replicated: 48 B per keypoint, trivial in 288 GB);
 *   - rgbdfe_match_node_pairs / _pair_list / _sift_pair_list shard the list pair k -> device k mod G; each device
 *     writes its results to the caller's out[k] directly and the call returns when all are done (the same
 *     barrier semantics as the single-device call; results do not depend on G);
 *   - rgbdfe_observation_likelihood shards its jobs the same way; frame-level calls (detect / describe /
 *     project_to_3d ...) run on the first device;
 *   - entry points that take device pointers are refused: use rgbdfe_device_context(ctx, i) for those.
 * rgbdfe_match_pair_list_allgather is the device-resident form: d_out[i] is a buffer on device i holding
 * G * per records, per = ceil(n_pairs / G) (returned in *records_per_device); afterwards EVERY buffer holds ALL
 * results -- pair k at record (k mod G) * per + k / G, unused tail records filled with 0xFF bytes (ids -1) --
 * exchanged with ONE ncclAllGather of the fixed-size PODs (RCCL over xGMI, loaded at run time), or with peer
 * copies when RCCL cannot be used (a device listed twice, RGBDFE_GATHER=p2p); rgbdfe_gather_transport tells which.
 * A device may be listed more than once (two shards on one GPU: how a 1-GPU box tests the sharding). */
int  rgbdfe_create_multi(const rgbdfe_config* cfg, const int32_t* device_ids, int32_t n_devices, rgbdfe_ctx** out);
int  rgbdfe_device_count(rgbdfe_ctx* ctx);                       /* 1 for rgbdfe_create handles */
rgbdfe_ctx* rgbdfe_device_context(rgbdfe_ctx* ctx, int32_t i);   /* the i-th device's own context (owned by ctx) */
int  rgbdfe_match_pair_list_allgather(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                      int32_t n_pairs, void* const* d_out, int32_t* records_per_device);
/* The same with only the ACCEPTED edges travelling (an all-pairs loop-closure sweep rejects most pairs: id1 == -1,
 * node.cpp:1419): every device compacts its shard in shard order, `*stride` = the largest count of a device; on return
 * d_out[j] holds device i's edges_per_device[i] records at [i * stride, ...), d_index[j] (optional, may be NULL) their
 * positions in the caller's pair list.  Buffers as above (n_devices * ceil(n_pairs / n_devices) records / int32). */
int  rgbdfe_match_pair_list_allgather_edges(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                            int32_t n_pairs, void* const* d_out, int32_t* const* d_index,
                                            int32_t* edges_per_device, int32_t* stride);
const char* rgbdfe_gather_transport(rgbdfe_ctx* ctx);            /* "rccl", "p2p" or "none" (last allgather) */
/* Exchanges the latest rgbdfe_match_pair_list_allgather_inliers call issued: 1 = the one collective sized before the
 * devices had counted their lists, 2 = a list had outgrown that size (and the group's first call: 1, after reading the
 * counts); 0 for other handles / before the first call. */
int  rgbdfe_gather_exchanges(rgbdfe_ctx* ctx);
/* Host time (microseconds) the calling thread spent enqueueing the latest sharded batch on all devices of a multi handle:
 * the shards are submitted by ONE thread, device after device.  With graph capture on (rgbdfe_set_graph_capture; OFF by
 * default) a batch's launch chain is a cached hipGraph per device -- one hipGraphLaunch (+ a read-back or pack enqueue) per
 * device; with the default it is the chain's ~10 kernel launches per device.  0 for single-device contexts. */
int  rgbdfe_group_submit_us(rgbdfe_ctx* ctx, double* us);
/* The compact form of rgbdfe_match_pair_list_allgather: d_out[i] holds G * per rgbdfe_compact_result, same placement
 * (pair k at (k mod G) * per + k / G, unused tail records 0xFF); 12x fewer bytes cross xGMI. */
int  rgbdfe_match_pair_list_allgather_compact(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                              int32_t n_pairs, void* const* d_out, int32_t* records_per_device);
/* The inlier form of the all-gather (rgbdfe_inlier_header below the result structs): every device packs its shard into an
 * inlier stream -- per = ceil(n_pairs / G) headers of 104 bytes, then (queryIdx | trainIdx << 16) of every inlier match --
 * and on return d_out[j] holds device i's stream at byte offset i * (*stride_bytes); list_entries[i] = entries of device
 * i's list block (*stride_bytes = per * 104 + 4 * C with C >= the largest of them: the exchange is sized from the lists of
 * the group's earlier calls, so that packing and the ONE collective are enqueued without a host read between them; C is the
 * largest list itself on a group's first call and whenever a list has outgrown the earlier ones by more than a quarter --
 * rgbdfe_gather_exchanges).  Pair k of the caller's list = header k / G of device k mod G.  What GraphManager reads of a MatchingResult (edge, rmse, counts, inlier_matches for
 * updateInlierFeatures, graph_manager.cpp:409-419) at ~260 bytes per pair instead of 1744.  Buffers: G * per *
 * (104 + 4 * RGBDFE_MAX_MATCHES) bytes each (the worst case). */
int  rgbdfe_match_pair_list_allgather_inliers(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                              int32_t n_pairs, void* const* d_out, int32_t* records_per_device,
                                              int32_t* list_entries, int64_t* stride_bytes);
/* d_records (n rgbdfe_match_result in HBM) -> d_compact (n rgbdfe_compact_result in HBM), enqueued on `stream`
 * (hipStream_t; NULL = the context's stream): what a one-process-per-GPU caller runs between rgbdfe_wait_ticket and its
 * own ncclAllGather (bench.py --gpus N).  Single-device contexts only. */
int  rgbdfe_pack_compact(rgbdfe_ctx* ctx, const void* d_records, int32_t n, void* d_compact, void* stream);
int  rgbdfe_sizeof_compact_result(void);
/* d_records (n rgbdfe_match_result in HBM) -> the inlier stream of the shard at d_stream (see rgbdfe_inlier_header; capacity
 * n_headers * 104 + 4 * sum of n_inl bytes, at most n_headers * 104 + n * 4 * RGBDFE_MAX_MATCHES), *d_total (device int32) =
 * entries of the list block; enqueued on `stream` (hipStream_t; NULL = the context's stream).  n_headers >= n: the
 * header count every rank of a gather pads its shard to.  Single-device contexts only. */
int  rgbdfe_pack_inliers(rgbdfe_ctx* ctx, const void* d_records, int32_t n, int32_t n_headers, void* d_stream,
                         int32_t* d_total, void* stream);
int  rgbdfe_sizeof_inlier_header(void);
int  rgbdfe_set_params(rgbdfe_ctx* ctx, const rgbdfe_params* p);
const char* rgbdfe_status_string(int status);
const char* rgbdfe_last_error(rgbdfe_ctx* ctx);

/* ---- node residency (replaces Node::feature_descriptors_ / feature_locations_3d_,
 *      node.h:167-178; lifetime follows GraphManager::addNode / clearFeatureInformation,
 *      graph_manager.h:156-161, node.cpp:1431-1443) ---------------------------- */
/* desc: n x 32 bytes, row-major, continuous (cv::Mat CV_8U, node.cpp:567-568);
 * xyz1: n x 4 float, (x,y,z,1) as Node::projectTo3D writes them (node.cpp:955). */
int rgbdfe_upload_node(rgbdfe_ctx* ctx, int32_t node_id, const uint8_t* desc,
                       const float* xyz1, int32_t n);
/* Many nodes in one call (offline runs: the nodes of a stretch of frames): the same residency as n_nodes calls of
 * rgbdfe_upload_node, with one pinned staging pass, back-to-back copies and expansion launches and ONE wait -- ~10 us per
 * node instead of ~65.  desc[i]: counts[i] x 32 bytes, xyz1[i]: counts[i] x 4 floats.  Nothing is uploaded when an argument
 * is bad, a node has more than max_keypoints rows, an id appears twice or the free slots do not suffice. */
int rgbdfe_upload_nodes(rgbdfe_ctx* ctx, int32_t n_nodes, const int32_t* node_ids, const uint8_t* const* desc,
                        const float* const* xyz1, const int32_t* counts);
/* same, sources already in device memory (device-to-device copy on `stream`, a hipStream_t).
 * Ordering: stream == NULL copies on the context's stream and returns when the node is resident.  With a caller
 * stream the copies are enqueued there and the call returns at once; every batch submitted to this context
 * afterwards (any entry point) waits for them on the device, and the sources must stay valid until `stream` has
 * passed the copies.  Overwriting a resident node first waits for the batches in flight. */
int rgbdfe_upload_node_device(rgbdfe_ctx* ctx, int32_t node_id, const void* d_desc,
                              const void* d_xyz1, int32_t n, void* stream);
/* Node::feature_locations_2d_ (KeyPoint.pt, n x 2 float) of a resident node: only the g2o refinement reads them
 * (edgeToFeature, transformation_estimation.cpp:95-125).  n must equal the node's row count. */
int rgbdfe_upload_node_keypoints(rgbdfe_ctx* ctx, int32_t node_id, const float* kp_xy, int32_t n);
int rgbdfe_release_node(rgbdfe_ctx* ctx, int32_t node_id);
int rgbdfe_node_count(rgbdfe_ctx* ctx, int32_t node_id); /* rows of a resident node or <0 */

/* ---- the pair op -------------------------------------------------------- */
/* Batched Node::matchNodePair (node.h:85, node.cpp:1305-1429): for one new node and
 * n candidates.  1:1 replacement of
 *   QtConcurrent::blockingMapped(nodes_to_comp, bind(&Node::matchNodePair,new_node,_1))
 * (graph_manager.cpp:548): synchronous, out[i] belongs to candidate_ids[i]. */
int rgbdfe_match_node_pairs(rgbdfe_ctx* ctx, int32_t new_node_id, const int32_t* candidate_ids,
                            int32_t n_pairs, rgbdfe_match_result* out);
/* general pair list (query = newer node, train = older node) */
int rgbdfe_match_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                           int32_t n_pairs, rgbdfe_match_result* out);
/* asynchronous, results stay in device memory (d_out: n_pairs rgbdfe_match_result in HBM);
 * enqueued on `stream` (hipStream_t; NULL = the context's stream).  n_pairs must be
 * <= max_pairs_per_batch. */
int rgbdfe_match_pair_list_device(rgbdfe_ctx* ctx, const int32_t* query_ids,
                                  const int32_t* train_ids, int32_t n_pairs, void* d_out,
                                  void* stream);
/* Fully asynchronous form for pipelined callers: the batch is enqueued on one of the
 * context's internal streams (consecutive batches alternate streams, so batch k+1 overlaps the
 * tail of batch k) and identified by *ticket.  rgbdfe_wait_ticket makes `stream` (a
 * hipStream_t) wait for that batch, or blocks the host when stream == NULL.  d_out must stay
 * valid and untouched until the ticket has been waited for. */
int rgbdfe_submit_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                            int32_t n_pairs, void* d_out, int64_t* ticket);
int rgbdfe_wait_ticket(rgbdfe_ctx* ctx, int64_t ticket, void* stream);
/* The asynchronous form with results in HOST memory -- what GraphManager consumes (graph_manager.cpp:409-419, 554-560) at
 * the rate the device produces it: batch k's results travel device -> host behind batch k on its internal stream while
 * batch k+1 (submitted before the wait) computes on the other one.  payload RGBDFE_HOST_RECORDS: n_pairs
 * rgbdfe_match_result; RGBDFE_HOST_INLIERS: the inlier stream of the batch (rgbdfe_inlier_header above: n_pairs headers,
 * then the list block -- ~260 instead of 1744 bytes per pair on configs[1]); out_bytes = capacity of `out` (the worst case
 * of the inlier stream is n_pairs * (104 + 4 * RGBDFE_MAX_MATCHES)).  `out` stays the library's until rgbdfe_wait_host
 * (ticket) returns; *bytes_written (may be NULL) = the payload's size.  When `out` is pinned (hipHostMalloc /
 * rgbdfe_host_register) the download goes straight into it, otherwise through a pinned stage and one memcpy inside
 * rgbdfe_wait_host.  At most two jobs in flight per context (one per internal stream): a third submit before a wait is
 * refused with RGBDFE_ERR_CAPACITY.  Same results as rgbdfe_match_pair_list.  Single-device contexts only.
 * One waiter per ticket (a second concurrent rgbdfe_wait_host of the same ticket is refused).  When the payload does not
 * fit `out` (the inlier stream's list block is sized by the results), rgbdfe_wait_host returns RGBDFE_ERR_CAPACITY with
 * *bytes_written = the size the payload needs and the job STAYS pending: rgbdfe_wait_host_into(ticket, out2, bytes2)
 * collects it into a larger buffer (pageable or pinned; through the library's pinned stage), out2 == NULL drops it. */
#define RGBDFE_HOST_RECORDS 0
#define RGBDFE_HOST_INLIERS 1
int rgbdfe_submit_pair_list_host(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs,
                                 void* out, size_t out_bytes, int payload, int64_t* ticket);
int rgbdfe_wait_host(rgbdfe_ctx* ctx, int64_t ticket, int64_t* bytes_written);
int rgbdfe_wait_host_into(rgbdfe_ctx* ctx, int64_t ticket, void* out, size_t out_bytes, int64_t* bytes_written);
int rgbdfe_synchronize(rgbdfe_ctx* ctx);

/* ---- SIFT (128-d float descriptor) nodes: matcher_type == "SIFTGPU" ------------------------
 * Replaces SiftGPUWrapper::match (sift_gpu_wrapper.h:61, sift_gpu_wrapper.cpp:169-227) inside
 * Node::featureMatching (node.cpp:553-557): u8-quantised dot products (exact on the bf16 MFMA),
 * acos distance / ratio tests, mutual best, DMatch.distance = float L2 of the raw descriptors.
 * desc128: n x 128 float, as Node::siftgpu_descriptors holds them (node.h:172). */
int rgbdfe_upload_sift_node(rgbdfe_ctx* ctx, int32_t node_id, const float* desc128,
                            const float* xyz1, int32_t n);
/* Batched matchNodePair for SIFT nodes.  out_dist (may be NULL): n_pairs x RGBDFE_MAX_MATCHES
 * floats, the DMatch.distance of out[i].all_q/all_t (all_hd is 0 on this path). */
int rgbdfe_match_sift_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                int32_t n_pairs, rgbdfe_match_result* out, float* out_dist);
/* same, asynchronous with results in HBM (see rgbdfe_submit_pair_list); d_out_dist may be NULL */
int rgbdfe_submit_sift_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                 int32_t n_pairs, void* d_out, void* d_out_dist, int64_t* ticket);
/* Twin of SiftGPUWrapper::match for two resident SIFT nodes: the mutual-best matches in ascending
 * query order (before keepStrongestMatches).  Arrays sized >= rows of the query node. */
int rgbdfe_sift_match_nodes(rgbdfe_ctx* ctx, int32_t query_id, int32_t train_id, int32_t* match_q,
                            int32_t* match_t, float* match_dist, int32_t* n_matches);

/* ---- float-descriptor nodes on Node::featureMatching's FLANN branch (node.cpp:610-667; a11) -------------------------
 * matcher_type == "FLANN" with a float extractor (SURF / SIFT / GFTT ...): knn-2 of every descriptor of the newer node
 * in the older node, ratio = dists[2i] / dists[2i+1] over FLANN's squared-L2 distances (:645), accepted when
 * nn_distance_ratio > ratio (:648), each train index once, first come first served in query order (:650-653),
 * DMatch.distance = the ratio (:657); then keepStrongestMatches and RANSAC as for every matcher.  The reference's
 * neighbours come from 4 randomised kd-trees searched with 16 checks (:493-505, :634) -- approximate and not
 * reproducible; here they are the EXACT two nearest (squared distance accumulated in flann::L2<float>'s order,
 * lowest row wins ties): the superset-quality replacement SURVEY.md 8(a) a11 names.
 * desc: n x dim float (Node::feature_descriptors_), dim a multiple of 4, <= 128.  out_dist (may be NULL) receives
 * the ratios of out[i].all_q/all_t (all_hd is 0 on this path). */
int rgbdfe_upload_float_node(rgbdfe_ctx* ctx, int32_t node_id, const float* desc, int32_t dim, const float* xyz1,
                             int32_t n);
int rgbdfe_match_flann_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs,
                                 double nn_distance_ratio, rgbdfe_match_result* out, float* out_dist);

/* ---- pieces of the pair op, exposed for A/B and parity ------------------- */
/* Batched bruteForceSearchORB (features.h:14, features.cpp:168-182) of every row of
 * query node against train node: out_hd[i] in [0,257], out_idx[i] (or -1), including
 * the reference's "last train row is never searched" behaviour. */
int rgbdfe_hamming_nn_nodes(rgbdfe_ctx* ctx, int32_t query_id, int32_t train_id,
                            int32_t* out_hd, int32_t* out_idx);
/* Drop-in twin of bruteForceSearchORB for host buffers (uploads, runs the kernel,
 * downloads; for A/B only -- the batched entry points are the product path). */
int rgbdfe_hamming_nn_host(rgbdfe_ctx* ctx, const uint8_t* qdesc, int32_t nq,
                           const uint8_t* tdesc, int32_t nt, int32_t* out_hd, int32_t* out_idx);

/* ---- per-frame depth filter + back-projection (removeDepthless node.cpp:67-97,
 *      projectTo3D node.cpp:900-965, backProject misc2.h:49-65) ---------------- */
/* kp_xy: n_kp x 2 float (KeyPoint.pt), depth: rows x cols float32 metres (NaN = invalid),
 * host buffers.  Writes kept_idx (indices into kp_xy, ascending) and xyz1 (n x 4) for the
 * first max_keypoints survivors; *n_out = survivors. */
int rgbdfe_project_to_3d(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* depth,
                         int32_t rows, int32_t cols, double fx, double fy, double cx, double cy,
                         double depth_scaling, int32_t max_keypoints, int32_t* kept_idx,
                         float* xyz1, int32_t* n_out);

/* a22 (i), the Node constructor that receives the sensor's organised point cloud (node.cpp:252-369):
 * rgbdfe_project_to_3d_cloud is Node::projectTo3D's point-cloud overload (node.cpp:855-898): lookup
 * point_cloud->at((int)x, (int)y) -- truncation --, drop when z > maximum_depth ("maximum_depth") or a coordinate is NaN,
 * the cloud's own (x, y, z, 1) is stored, cut at max_keypoints.  cloud: rows x cols x 4 float (x, y, z, rgb), host.
 * rgbdfe_detect_describe_cloud is that constructor's feature path: detect (:293) -> projectTo3D(cloud) (:308) ->
 * compute (:311), with the detector state / max_keypoints of rgbdfe_detector_configure; no removeDepthless, no
 * retainBest.  The 3-D points stay with their keypoints through compute()'s border filter and regrouping (the
 * reference leaves feature_locations_3d_ out of step there, its assert at :318). */
int rgbdfe_project_to_3d_cloud(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* cloud, int32_t rows,
                               int32_t cols, double maximum_depth, int32_t max_keypoints, int32_t* kept_idx, float* xyz1,
                               int32_t* n_out);

/* a20, SIFTGPU feature path: Node::projectTo3DSiftGPU (node.cpp:695-769) -- depth lookup with the
 * keypoint coordinates TRUNCATED to int (:733), no inside-the-image test (indices are clamped here
 * where the reference would read out of bounds), NaN depth drops the keypoint, stop at max_keypoints
 * (:748); the used descriptors are re-packed densely (:752-766) into
 *   siftgpu_descriptors  [n_out x 128] raw copy  = Node::siftgpu_descriptors, what rgbdfe_upload_sift_node takes
 *   feature_descriptors  [n_out x 128] (may be NULL) = Node::feature_descriptors_, RootSIFT-normalised by
 *                        squareroot_descriptor_space (node.cpp:1557-1571) when use_root_sift != 0
 *                        (parameter "use_root_sift", node.cpp:233-239).
 * kp_xy: n_kp x 2 f32, desc_in: n_kp x 128 f32, depth: rows x cols f32; outputs sized for
 * min(n_kp, max_keypoints) rows. */
int rgbdfe_sift_node_features(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* desc_in,
                              const float* depth, int32_t rows, int32_t cols, double fx, double fy,
                              double cx, double cy, double depth_scaling, int32_t max_keypoints,
                              int32_t use_root_sift, int32_t* kept_idx, float* xyz1,
                              float* siftgpu_descriptors, float* feature_descriptors, int32_t* n_out);
/* The same with "use_feature_min_depth" on (parameter_server.cpp:90, node.cpp:727-731): a keypoint's depth is
 * getMinDepthInNeighborhood(depth, pt, size) (misc.cpp:774-793); kp_size = cv::KeyPoint::size per keypoint (for SiftGPU
 * keypoints 12 * scale, what rgbdfe_sift_detect returns). */
int rgbdfe_sift_node_features_min_depth(rgbdfe_ctx* ctx, const float* kp_xy, const float* kp_size, int32_t n_kp,
                                        const float* desc_in, const float* depth, int32_t rows, int32_t cols, double fx,
                                        double fy, double cx, double cy, double depth_scaling, int32_t max_keypoints,
                                        int32_t use_root_sift, int32_t* kept_idx, float* xyz1, float* siftgpu_descriptors,
                                        float* feature_descriptors, int32_t* n_out);

/* Two schedules of a batch's RANSAC work give byte-identical results:
 *   record / replay       (default) recording waves each refine chunk_iterations iterations of a pair and write the
 *                         outcomes, then one wave per pair replays the records in iteration order with the reference's
 *                         bookkeeping (node.cpp:1171-1190).  Up to 256 pairs all iterations are recorded in one phase
 *                         (full speculation: one node against 20 candidates returns in 0.24-0.41 ms instead of 5.7 ms);
 *                         larger batches run up to four phases ([0,14), [14,70), [70,140), [140,200) for 200
 *                         iterations): each replay tells the next phase which pairs are finished and how many
 *                         iterations the others can still need, so recording stops where the reference stops
 *                         iterating.  After the first phase a pair whose loop has not jumped ahead (it += 10 / 20) and
 *                         whose hypotheses were mostly junk is recorded to the end in ONE launch (64-iteration shares,
 *                         junk hypotheses screened out lane-parallel before they take a slot), the others go on
 *                         phase by phase in short shares.  Batches beyond 2^24 / ransac_iterations (or 65535) pairs
 *                         run as pieces of that size;
 *   one wave per pair     a wave runs a pair's whole loop (windows of 7 iterations, replayed in order): exactly the
 *                         iterations the reference runs, but a pair takes ~4.6 ms however idle the chip is and a
 *                         launch lasts as long as its slowest pair.  Kept for A/B runs; never chosen by the library.
 * Batches of at most max_pairs pairs (ORB, SIFT, FLANN) take record / replay.  Defaults: max_pairs = INT32_MAX (every
 * batch), chunk_iterations = 0 (automatic: 4 up to 64 pairs, 7 up to 640, 14 up to 1280, 28 above; a phase is cut into
 * equal shares of at most that many iterations); max_pairs = 0 forces one wave per pair.
 * A negative chunk_iterations selects the phased plan for every batch size with |chunk_iterations| iterations
 * per recording wave (a testing aid: small batches are faster with the single phase). */
int rgbdfe_set_latency_mode(rgbdfe_ctx* ctx, int32_t max_pairs, int32_t chunk_iterations);

/* Which kernel computes the Hamming nearest neighbours of an ORB batch (identical keys, bit for bit):
 *   1            the descriptor bits as fp4 (+-1) operands of v_mfma_f32_32x32x64_f8f6f4: hd = (256 + dot) / 2, exact
 *                in the f32 accumulator, the row index folded into the accumulator's initial value (hamming_mfma.hip);
 *   3            the same contraction as a software pipeline inside every wave (the reduction of one accumulator and the
 *                LDS reads of the next train tile sit between the MFMAs of the other accumulator; train tiles arrive by
 *                global_load_lds through three LDS buffers);
 *   2            as 1, the row index added by the VALU instead of the matrix core's C operand;
 *   0            xor + popcount on the VALU (hamming_nn.hip) -- also what nodes with max_keypoints > 32768 get.
 * RGBDFE_HAMMING_MODE_DEFAULT is the mode of a new context; environment variable RGBDFE_HAMMING_MODE overrides it. */
#define RGBDFE_HAMMING_MODE_DEFAULT 3
int rgbdfe_set_hamming_mode(rgbdfe_ctx* ctx, int32_t mode);

/* ---- frame-level data either side of the pair path ----------------
 * rgbdfe_depth_to_mono8: depthToCV8UC1 (misc.cpp:414-430), the detection mask the listener derives from
 *   the depth image.  depth_is_u16 == 0: depth is rows x cols f32 (metres), mono8 = convertTo(CV_8UC1, 100)
 *   (:418), depth_m is ignored.  depth_is_u16 != 0: depth is u16 millimetres, mono8 = convertTo(CV_8UC1,
 *   0.05, -25) (:423) and depth_m (required) = the float image in metres (:424-425).
 * rgbdfe_upload_node_cloud: createXYZRGBPointCloud (misc.cpp:467-556) on the device.  Builds the node's
 *   structured cloud -- (rows/cloud_skip) x (cols/cloud_skip) points of 4 floats (x, y, z, rgb bits
 *   0x00RRGGBB) -- from its depth image (f32, metres x depth_scaling; Z < min_depth or NaN -> z = NaN with
 *   x, y at 1 m, :525-530) and keeps it resident under node_id (Node::pc_col) for the environment measurement
 *   model; rgb (rows x cols x rgb_channels u8, channels 1 or 3, may be NULL) only colours the points.
 *   cloud_skip = cloud_creation_skip_step must divide rows and cols (the reference crashes otherwise, :479).
 *   cloud_out (may be NULL) receives a copy.  rgbdfe_release_node also drops the node's cloud.
 * rgbdfe_observation_likelihood: observationLikelihood (misc.cpp:814-969) for a batch of directed edges:
 *   job i projects the cloud of new_ids[i], transformed by transforms[i] (16 floats, column-major, new ->
 *   old), into the raster of old_ids[i] and classifies every emm_skip_step-th point (parameter
 *   "emm__skip_step", 8) against the old depth.  pairwiseObservationLikelihood (node.cpp:1520-1554) is two
 *   jobs per edge, (newer, older, final_trafo) and (older, newer, final_trafo.inverse()), with the counts
 *   summed; matchNodePair then keeps the edge iff rgbdfe_observation_criterion_met(inliers, outliers,
 *   occluded + inliers + outliers, observability_threshold) (node.cpp:1341-1342, misc.cpp:1136-1148).
 *   depth_covariance() is params.depth_cov (the reference's frozen static, misc2.h:30-35). */
typedef struct rgbdfe_emm_counts {
  uint32_t inliers, outliers, occluded, all;
} rgbdfe_emm_counts;
int rgbdfe_depth_to_mono8(rgbdfe_ctx* ctx, const void* depth, int32_t depth_is_u16, int32_t rows, int32_t cols,
                          uint8_t* mono8, float* depth_m);
int rgbdfe_upload_node_cloud(rgbdfe_ctx* ctx, int32_t node_id, const float* depth, int32_t rows, int32_t cols,
                             const uint8_t* rgb, int32_t rgb_channels, int32_t encoding_bgr, double fx,
                             double fy, double cx, double cy, double depth_scaling, double min_depth,
                             int32_t cloud_skip, float* cloud_out);
int rgbdfe_release_node_cloud(rgbdfe_ctx* ctx, int32_t node_id);
int rgbdfe_observation_likelihood(rgbdfe_ctx* ctx, int32_t n, const int32_t* new_ids, const int32_t* old_ids,
                                  const float* transforms, int32_t emm_skip_step, rgbdfe_emm_counts* out);
int rgbdfe_observation_criterion_met(uint32_t inliers, uint32_t outliers, uint32_t all,
                                     double observability_threshold, double* quality);

/* ---- candidate selection for loop closure ----------------------------------
 * rgbdfe_potential_edge_targets is GraphManager::getPotentialEdgeTargetsWithDijkstra (graph_manager.cpp:204-324): the
 * ids of the earlier nodes a new node is to be compared with -- `sequential_targets` direct predecessors, then
 * `geodesic_targets` drawn (weighted by their distance in time, :266) from the graph neighbourhood of the predecessor
 * within `geodesic_depth` edges (g2o::HyperDijkstra with UniformCostFunction, :230-233), then `sampled_targets` drawn
 * uniformly from the remaining matchable keyframes (:297-317).  Order as in the reference's QList: sampled ids first
 * (latest draw first), sequential ones after, the predecessor last when include_predecessor != 0.  The list feeds
 * rgbdfe_match_node_pairs.  Host code, no device work.
 * The pose-graph object holds what the reference reads from graph_, camera_vertices, keyframe_ids_ and the optimizer's
 * edges: rgbdfe_pose_graph_add_node per Node added to the graph (id_, vertex_id_, matchable_, and whether it became a
 * keyframe, graph_manager.cpp:655), rgbdfe_pose_graph_add_edge per edge added to the optimizer (:811).
 * rand_fn(rand_state) replaces rand() (pass a wrapper of rand() for the reference's stream); NULL selects a
 * counter-based generator seeded with `seed`, which makes the selection reproducible.
 * Returns RGBDFE_ERR_CAPACITY (with *n_out = the needed size) when ids_out is too small. */
typedef struct rgbdfe_pose_graph rgbdfe_pose_graph;
typedef int (*rgbdfe_rand_fn)(void* state);
rgbdfe_pose_graph* rgbdfe_pose_graph_create(void);
void rgbdfe_pose_graph_destroy(rgbdfe_pose_graph* g);
int rgbdfe_pose_graph_add_node(rgbdfe_pose_graph* g, int32_t node_id, int32_t vertex_id, int32_t matchable,
                               int32_t keyframe);
int rgbdfe_pose_graph_add_edge(rgbdfe_pose_graph* g, int32_t node_id1, int32_t node_id2);
int rgbdfe_pose_graph_set_matchable(rgbdfe_pose_graph* g, int32_t node_id, int32_t matchable);
int rgbdfe_potential_edge_targets(const rgbdfe_pose_graph* g, int32_t sequential_targets, int32_t geodesic_targets,
                                  int32_t sampled_targets, int32_t geodesic_depth, int32_t predecessor_id,
                                  int32_t include_predecessor, rgbdfe_rand_fn rand_fn, void* rand_state, uint32_t seed,
                                  int32_t* ids_out, int32_t capacity, int32_t* n_out);

/* GPU prefilter in front of the pair path (second half): what loop_closing.cpp's
 * GraphManager::getNeighbours (:190-277, behind DO_LOOP_CLOSING, never wired into nodeComparisons) sketched -- every
 * descriptor of the new node votes `k_neighbours - rank` (:241) for the nodes holding its k nearest descriptors, a
 * node's votes are divided by its descriptor count (:263), the nodes are ranked by that score (:269) -- with EXACT
 * binary neighbours instead of an approximate kd-tree: a descriptor's k nearest nodes are the k candidates whose best
 * match (the Hamming stage's keys) has the smallest distance, ties to the candidate listed first; matches with
 * hd >= max_hd do not vote (128 = featureMatching's gate, node.cpp:572; 257 = every match votes).
 * out_ids / out_scores: the at most max_out best candidates in descending score order (ties: listed first); candidates
 * without a vote are not returned.  k_neighbours in [1, 8].  Feed out_ids to rgbdfe_match_node_pairs. */
int rgbdfe_place_recognition(rgbdfe_ctx* ctx, int32_t query_id, const int32_t* candidate_ids, int32_t n_candidates,
                             int32_t k_neighbours, int32_t max_hd, int32_t max_out, int32_t* out_ids, float* out_scores,
                             int32_t* n_out);
/* Many query nodes at once (an offline loop-closure sweep: ONE Hamming launch + ONE vote launch for all of them).
 * Query s has the candidates candidate_ids[candidate_offsets[s] .. candidate_offsets[s+1]-1] (offsets[0] = 0, at most
 * 65535 per query, offsets[n_queries] <= max_pairs_per_batch); its ranked list goes to out_ids / out_scores
 * [s * max_out ...], its length to out_counts[s]. */
int rgbdfe_place_recognition_batch(rgbdfe_ctx* ctx, const int32_t* query_ids, int32_t n_queries,
                                   const int32_t* candidate_offsets, const int32_t* candidate_ids, int32_t k_neighbours,
                                   int32_t max_hd, int32_t max_out, int32_t* out_ids, float* out_scores,
                                   int32_t* out_counts);

/* ---- per-frame feature path: detect + describe (Node::Node, node.cpp:139-210) -----------------
 * rgbdfe_detect_describe replaces, for one frame,
 *   detector->detect(gray, kps, mask)   the 3x3 grid of threshold-adaptive ORB detectors built by
 *                                       createDetector("ORB") (features.h:9, features.cpp:63-113,
 *                                       feature_adjuster.cpp:85-317); its per-cell FAST thresholds
 *                                       persist across frames inside the context, like the reference's
 *                                       detector_ object (openni_listener.h:195)
 *   removeDepthless + KeyPointsFilter::retainBest(max_keypoints)          (node.cpp:186-191)
 *   extractor->compute(gray, kps, desc) cv::ORB::create() defaults        (features.cpp:117-119)
 *   projectTo3D                                                           (node.cpp:900-965)
 * gray: rows x cols u8; mask: rows x cols u8 (depth_mono8, non-zero = usable) or NULL;
 * depth: rows x cols float32 metres (NaN = no measurement).  Outputs are sized by the caller for
 * max_keypoints entries: keypoints, descriptors (32 bytes each), xyz1 (4 floats each). */
typedef struct {
  float x, y;       /* cv::KeyPoint::pt (level-0 pixel coordinates) */
  float size;       /* 31 * scale of the octave */
  float angle;      /* degrees (cv::fastAtan2 of the intensity centroid) */
  float response;   /* Harris response */
  int32_t octave;   /* pyramid level */
} rgbdfe_keypoint;
/* parameter_server.cpp:83,87,89; resets the per-cell thresholds */
int rgbdfe_detector_configure(rgbdfe_ctx* ctx, int32_t max_keypoints, int32_t grid_resolution,
                              int32_t adjuster_max_iterations);
/* the current per-cell FAST thresholds (grid_resolution^2 doubles) */
int rgbdfe_detector_thresholds(rgbdfe_ctx* ctx, double* thresholds, int32_t* n_cells);
/* "use_feature_min_depth" (parameter_server.cpp:90, default off): a keypoint's depth is the nearest valid depth in its
 * neighbourhood (getMinDepthInNeighborhood, misc.cpp:774-793) instead of the pixel under it -- in removeDepthless
 * (node.cpp:82) and projectTo3D (:940).  rgbdfe_set_feature_min_depth switches rgbdfe_detect_describe(_batch) over;
 * rgbdfe_project_to_3d_min_depth is rgbdfe_project_to_3d in that mode (kp_size = cv::KeyPoint::size per keypoint).
 * The SIFTGPU call site (node.cpp:730) is rgbdfe_sift_node_features_min_depth. */
int rgbdfe_set_feature_min_depth(rgbdfe_ctx* ctx, int32_t on);
int rgbdfe_project_to_3d_min_depth(rgbdfe_ctx* ctx, const float* kp_xy, const float* kp_size, int32_t n_kp,
                                   const float* depth, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                   double cy, double depth_scaling, int32_t max_keypoints, int32_t* kept_idx,
                                   float* xyz1, int32_t* n_out);
int rgbdfe_detect_describe(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, const float* depth,
                           int32_t rows, int32_t cols, double fx, double fy, double cx, double cy,
                           double depth_scaling, rgbdfe_keypoint* keypoints, uint8_t* descriptors,
                           float* xyz1, int32_t* n_out);
/* Page-locks a range of the caller's host memory (hipHostRegister) / releases it.  rgbdfe_detect_describe copies image
 * buffers that are page-locked straight to the device; pageable ones (cv::Mat data as cv_bridge hands it over,
 * openni_listener.cpp) first go through the library's own pinned staging buffer, ~60 us per 640x480 frame.  An
 * integration that owns its image buffers registers them once.  ptr / bytes: as malloc'ed or page-aligned; registering
 * a range twice is an error. */
int rgbdfe_host_register(rgbdfe_ctx* ctx, void* ptr, size_t bytes);
int rgbdfe_host_unregister(rgbdfe_ctx* ctx, void* ptr);
/* A run of frames through the same detector state, in order: the same keypoints, descriptors and points as n_frames
 * calls of rgbdfe_detect_describe (the per-cell thresholds carry over from frame to frame, feature_adjuster.cpp:185-224),
 * with up to 7 frames sharing every kernel launch (64 / grid_resolution^2 frames per launch chain: the device pass runs at
 * floor thresholds and the adjuster is replayed over the scored corners on the host) and three launch chains
 * in flight.  For offline runs (bag files, OpenNIListener in "batch_processing" mode): 10-14 k frames/s at 640 x 480 for runs
 * of >= 56 frames.  Threads: the first call creates 11 worker threads inside the context (7 for the CPU halves of the
 * descriptions and the replay, 4 for staging copies; pure CPU work, they never enter the HIP runtime) that live until
 * rgbdfe_destroy, plus one helper thread per call.  mask may be NULL (no masks) or hold NULL entries; out_stride >= the configured
 * max_keypoints: frame f's outputs start at row f * out_stride of keypoints / descriptors (32 B rows) / xyz1 (4 floats),
 * n_out[f] of them. */
int rgbdfe_detect_describe_batch(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray,
                                 const uint8_t* const* mask, const float* const* depth, int32_t rows, int32_t cols,
                                 double fx, double fy, double cx, double cy, double depth_scaling, int32_t out_stride,
                                 rgbdfe_keypoint* keypoints, uint8_t* descriptors, float* xyz1, int32_t* n_out);
/* The same, and frame f's features also become the resident node node_ids[f] (>= 0; negative: no node for that frame) --
 * what Node::Node + GraphManager::addNode + rgbdfe_upload_node do, without the features' trip to the host and back: the
 * descriptors and points are copied into the node slabs from the description's device buffers (the host outputs are filled
 * as before).  A frame WITHOUT features becomes an EMPTY node (n = 0), exactly what rgbdfe_upload_node(id, ..., 0)
 * leaves: a fresh id takes a slot and is resident (a pair against it is matched and comes back without an edge), an id that
 * exists is rewritten to n = 0 -- its old features are gone.  An id that exists and gets features is rewritten in place.
 * Capacity is checked for the whole batch before the first frame is detected, counting every fresh id as one slot (empty
 * frames consume theirs, so the count is exact). */
int rgbdfe_detect_describe_batch_nodes(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray,
                                       const uint8_t* const* mask, const float* const* depth, int32_t rows, int32_t cols,
                                       double fx, double fy, double cx, double cy, double depth_scaling, int32_t out_stride,
                                       rgbdfe_keypoint* keypoints, uint8_t* descriptors, float* xyz1, int32_t* n_out,
                                       const int32_t* node_ids);
/* the point-cloud constructor's feature path (see rgbdfe_project_to_3d_cloud above) */
int rgbdfe_detect_describe_cloud(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, const float* cloud,
                                 int32_t rows, int32_t cols, double maximum_depth, rgbdfe_keypoint* keypoints,
                                 uint8_t* descriptors, float* xyz1, int32_t* n_out);
/* pieces, for A/B against cv::ORB: detect() of one image with a fixed FAST threshold (no grid,
 * feature_adjuster.cpp:94) and compute() for given keypoints (may drop border keypoints and
 * regroups them by octave, like cv::ORB::compute). */
int rgbdfe_orb_detect(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, int32_t rows, int32_t cols,
                      int32_t fast_threshold, rgbdfe_keypoint* keypoints, int32_t capacity, int32_t* n_out);
int rgbdfe_orb_compute(rgbdfe_ctx* ctx, const uint8_t* gray, int32_t rows, int32_t cols,
                       rgbdfe_keypoint* keypoints, int32_t n, uint8_t* descriptors, int32_t* n_out);

/* ---- SIFT extraction (feature_detector_type / feature_extractor_type == "SIFTGPU") ---------------------------------
 * rgbdfe_sift_detect replaces SiftGPUWrapper::detect (sift_gpu_wrapper.h:49, sift_gpu_wrapper.cpp:113-167; called from
 * Node::Node, node.cpp:149-152, 282-286) with an empty keypoint list: SiftGPU's scale-space extrema detection, orientation
 * assignment and 4x4x8 descriptors with the options the wrapper's constructor sets (:29-88) -- first octave -1 (the image
 * is up-sampled x2), 5 DoG levels per octave, edge threshold 10, sub-pixel localisation, two orientations per keypoint,
 * UNNORMALISED descriptors ("-unn"), at most ~max_keypoints features chosen from the coarse octaves down ("-tc2",
 * parameter "max_keypoints") -- on the pipeline of SiftGPU's CUDA back end (external/SiftGPU/src/SiftGPU/ProgramCU.cu,
 * PyramidCU.cpp).  gray: rows x cols u8; mask is ignored, as the reference ignores it.  keypoints[i]: pt = SiftGPU's
 * (x, y), size = 12 * scale, angle in degrees (the wrapper's conversion, :156-160), response = octave = 0; desc128: n x 128
 * floats = what the wrapper returns as `descriptors` and Node::projectTo3DSiftGPU / rgbdfe_sift_node_features take.
 * Returns RGBDFE_ERR_CAPACITY with *n_out = the number of features when `capacity` rows are too few.
 * The wrapper's second mode (a caller-provided keypoint list, :132-142) is rgbdfe_sift_describe below. */
int rgbdfe_sift_detect(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, int32_t rows, int32_t cols,
                       int32_t max_keypoints, rgbdfe_keypoint* keypoints, float* desc128, int32_t capacity, int32_t* n_out);
/* SiftGPUWrapper::detect with a non-empty keypoint list (sift_gpu_wrapper.cpp:132-142): feature_extractor_type == "SIFTGPU"
 * behind another detector (node.cpp:166-171) -- SiftGPU::SetKeypointList with its default "keys have orientation"
 * (SiftGPU.h:150), i.e. descriptors at the given positions, scales (size / 12) and orientations (angle, degrees), no
 * detection, no orientation assignment.  keypoints[n] are rewritten as the wrapper rebuilds them (size and angle through its
 * float conversions, response = octave = 0); desc128: n x 128 floats, unnormalised, row i for keypoint i. */
int rgbdfe_sift_describe(rgbdfe_ctx* ctx, const uint8_t* gray, int32_t rows, int32_t cols, rgbdfe_keypoint* keypoints, int32_t n,
                         float* desc128);
/* A run of frames of one size (offline processing of a recorded sequence): the results of n_frames single calls -- the
 * pipeline keeps no state between images -- with up to 8 frames sharing every launch (a frame alone is ~70 dependent
 * launches over planes of a few thousand pixels to 1.2 Mpixel and cannot fill the chip).  Frame f's keypoints / descriptors
 * go to row f * out_stride of keypoints / desc128 (128 floats per row), its count to n_out[f]; RGBDFE_ERR_CAPACITY when a
 * frame has more than out_stride features (n_out is complete, the rows of the other frames are valid). */
int rgbdfe_sift_detect_batch(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, int32_t rows, int32_t cols,
                             int32_t max_keypoints, int32_t out_stride, rgbdfe_keypoint* keypoints, float* desc128,
                             int32_t* n_out);
/* stage access for parity tests: the pyramid geometry of the latest frame, one Gaussian plane (octave index from 0, level
 * 0 .. levels-1; padded width x height floats), the keypoint candidates of one (octave, DoG level) as rows of
 * (x, y, extremum sign, dx, dy, ds) in list order, before the feature-count limit */
int rgbdfe_sift_geometry(rgbdfe_ctx* ctx, int32_t* octave_min, int32_t* octave_num, int32_t* levels, int32_t* dog_levels);
int rgbdfe_sift_debug_plane(rgbdfe_ctx* ctx, int32_t octave, int32_t level, float* out, int32_t capacity_floats, int32_t* w,
                            int32_t* h);
int rgbdfe_sift_debug_candidates(rgbdfe_ctx* ctx, int32_t octave, int32_t dog_level, float* out, int32_t capacity_rows,
                                 int32_t* n);

/* ---- measurement --------------------------------------------------------- */
/* When enabled, every launch of the dominant kernels is bracketed by HIP events on the
 * stream it runs on; totals are read back with rgbdfe_get_kernel_time. */
enum { RGBDFE_KERNEL_HAMMING = 0, RGBDFE_KERNEL_RANSAC = 1, RGBDFE_KERNEL_SIFT_DOT = 2,
       RGBDFE_KERNEL_SIFT_FINISH = 3, RGBDFE_KERNEL_EMM = 4, RGBDFE_KERNEL_COUNT = 5 };
int rgbdfe_set_profiling(rgbdfe_ctx* ctx, int enable);
int rgbdfe_get_kernel_time(rgbdfe_ctx* ctx, int which, double* total_ms, int64_t* launches,
                           int64_t* pairs);
int rgbdfe_reset_kernel_time(rgbdfe_ctx* ctx);
/* The hipGraph cache of the ORB pair path (one executable graph per distinct batch shape: pair count, Hamming launch
 * geometry, output buffer): out[0..n_out) = { captures, graph launches, misses (graphable batches whose shape was not
 * cached), batches issued as plain launches because the shapes kept missing, captures another thread invalidated,
 * cached graphs that failed to launch, graphs cached now, graphs enabled (RGBDFE_GRAPHS) }; summed over the devices
 * of a multi handle. */
#define RGBDFE_GRAPH_STATS 8
int rgbdfe_graph_stats(rgbdfe_ctx* ctx, int64_t* out, int32_t n_out);
/* Cached hipGraphs for the launch chain of ORB pair batches (one hipGraphLaunch instead of ~12 enqueues per batch: 2-3 %
 * on the 4000-pair headline, 20 % of the submission cost of a multi-device handle).  OFF by default: while a capture is
 * open, hipDeviceSynchronize() (torch.cuda.synchronize(), a synchronous hipMemcpy on the NULL stream ...) on any OTHER
 * thread of the process fails with hipErrorStreamCaptureUnsupported, and the library cannot know what the host
 * application's threads do.  Turn it on when every thread that calls HIP is yours (bench.py does).  RGBDFE_GRAPHS=1 / 0 in
 * the environment sets the initial value of new contexts. */
int rgbdfe_set_graph_capture(rgbdfe_ctx* ctx, int enable);

/* ABI self-description (lets bindings verify struct layout) */
int rgbdfe_sizeof_match_result(void);
int rgbdfe_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RGBDFE_H */
