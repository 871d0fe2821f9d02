/*
 * demon_hip.h -- C ABI of libdemon_hip.so, the MI355X-native (gfx950) DeMoN inference path.
 *
 * This is the drop-in boundary for the hot path of lmb-freiburg/demon.  The
 * reference has no FFI of its own for this path: it builds a TF1 graph in Python and crosses into
 * native code through tf.Session.run and the lmbspecialops custom-op library.  Each entry point
 * below names the reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types; no exceptions cross the ABI
 *   - every function returns 0 on success or a negative demon_status; text via demon_last_error()
 *   - all tensors are float32, NCHW ("channels_first"), contiguous; the Python shim converts NHWC
 *   - "host" pointers are caller-owned host memory; the context owns all device memory
 *   - a context is bound to one device and one HIP stream and is NOT thread safe
 *   - no device allocation happens inside any run call (hipGraph-capture safe)
 */
#ifndef DEMON_HIP_H
#define DEMON_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct demon_ctx demon_ctx;

enum demon_status {
    DEMON_OK = 0,
    DEMON_ERR_INVALID = -1,     /* bad argument / shape mismatch (TF: ValueError / InvalidArgumentError) */
    DEMON_ERR_HIP = -2,         /* a HIP runtime call failed */
    DEMON_ERR_NOT_READY = -3,   /* weights missing */
    DEMON_ERR_NOT_FOUND = -4    /* unknown variable / option */
};

/* Output bundle of one bootstrap / iterative evaluation; NULL members are skipped.
 * Replaces the fetches dict of BootstrapNet.eval / IterativeNet.eval
 * (python/depthmotionnet/networks_original.py:76-83, :182-189). Shapes for batch n at HxW input,
 * h2 = H/4, w2 = W/4, h5 = H/32, w5 = W/32:                                                       */
typedef struct demon_outputs {
    float *predict_flow5;       /* [n,2,h5,w5] */
    float *predict_conf5;       /* [n,2,h5,w5]  (attribute predict_conf5, networks_original.py:47) */
    float *predict_flow2;       /* [n,2,h2,w2] */
    float *predict_conf2;       /* [n,2,h2,w2] */
    float *predict_depth2;      /* [n,1,h2,w2] */
    float *predict_normal2;     /* [n,3,h2,w2] */
    float *predict_rotation;    /* [n,3] angle axis */
    float *predict_translation; /* [n,3] */
    float *predict_scale;       /* [n,1] */
} demon_outputs;

/* ---- lifecycle -------------------------------------------------------------------------------
 * Replaces tf.InteractiveSession + the three network constructors
 * (examples/example.py:70-77; networks_original.py:23, :93, :204).  height/width must be multiples
 * of 32; 192x256 is the reference's fixed size (networks_original.py:38-42).                      */
/* number of HIP devices visible to the library (0 when there is none or the runtime fails): replaces tf.test.is_gpu_available
 * (examples/example.py:45) for callers that must not load a second GPU runtime just to ask */
int demon_device_count(void);
int demon_create(demon_ctx **ctx, int device, int max_batch, int height, int width);
/* The retrained "v2" model: python/depthmotionnet/v2/networks.py:20-36, :80-122, :181-205 over v2/blocks.py (padding='same',
 * (24,32)/(48,64)/(96,128)/(192,256) separable pairs, 384 channels at level 5, dense5 bottleneck, motion_conv3..5b branch,
 * flow_to_depth2 + clip [0,50], predict_normal0).  Every other entry point works on such a context unchanged; the variable table
 * (demon_variable_info) is the v2 one.  demon_variant returns 1 or 2.                                                        */
int demon_create_v2(demon_ctx **ctx, int device, int max_batch, int height, int width);
/* A context without networks: one HIP stream and the split-K workspace, all the demon_op_* entry points below need
 * (replaces `import lmbspecialops` = tf.load_op_library($LMBSPECIALOPS_LIB), Dockerfile:26-27, for op-only callers).
 * Network entry points return DEMON_ERR_INVALID on it. */
int demon_create_ops(demon_ctx **ctx, int device);
int demon_variant(const demon_ctx *ctx);
int demon_destroy(demon_ctx *ctx);
const char *demon_last_error(const demon_ctx *ctx); /* ctx may be NULL: error of a failed create */
int demon_device(const demon_ctx *ctx);

/* ---- weights ---------------------------------------------------------------------------------
 * Replaces tf.train.Saver().restore (examples/example.py:82-83).  Variables carry their TF names
 * and TF layouts ("netFlow1/conv1y/kernel" [9,1,6,32] HWIO; deconv [4,4,Cout,Cin]; dense [in,out];
 * SURVEY.md appendix B); the library repacks them for the MFMA kernels.                           */
int demon_num_variables(const demon_ctx *ctx);
int demon_variable_info(const demon_ctx *ctx, int index, char *name, int name_cap, int64_t dims[4], int *ndim);
int demon_set_weight(demon_ctx *ctx, const char *tf_name, const float *host, const int64_t *dims, int ndim);
/* flat blob = all variables, TF layout, in demon_variable_info order */
int64_t demon_weights_blob_size(const demon_ctx *ctx);
int demon_set_weights_blob(demon_ctx *ctx, const float *host_blob, int64_t nfloats);
/* same blob already in device memory of this context's device (e.g. after an RCCL broadcast) */
int demon_set_weights_blob_device(demon_ctx *ctx, const void *device_blob, int64_t nfloats);

/* ---- options ---------------------------------------------------------------------------------
 * "hipgraph" 0/1 (default 1), "flow_to_depth_method" 0 = DLT/SVD, 1 = closed form (default 0),
 * "reuse_image_features" 0/1 (default 0): in demon_full / demon_run_full compute conv1 / conv2 of the iterative nets (which
 * depend on the images only) once per forward pass instead of once per iteration -- same results, 16 launches fewer;
 * "side_branches" 0/1 (default 1): the motion head and the level-5 flow head run on a second HIP stream next to the decoder
 * they do not depend on (fork / join by events, captured into the same hipGraph) -- same kernels, same results;
 * "fused_pairs" 0/1 (default 1): the first k x 1 / 1 x k pair (conv1, 6 input channels, the largest intermediate) as one launch,
 * the intermediate staying in LDS (conv_pair.hip) -- same arithmetic, other summation order;
 * "fused_inputs" 0/1 (default 1): the extra-input assembly of the iterative blocks / the refinement input as one launch each;
 * "tune_lanes" 1..8 (default 1): demon_autotune times every candidate as that many CONCURRENT replays on as many streams
 * ("throughput mode") -- the right cost when several passes are in flight on the GPU (demon_amd/lanes.py) */
int demon_set_option(demon_ctx *ctx, const char *key, int value);
int demon_get_option(const demon_ctx *ctx, const char *key, int *value);
/* Times every applicable kernel variant (im2col / patch-staged, tile shape, split-K) of every layer at batch n on
 * this GPU and keeps the fastest per layer (~1 s; results do not change, only launch plans). */
int demon_autotune(demon_ctx *ctx, int n);
/* read back / install launch plans, e.g. to ship the result of one autotune run as a file.  Entry = (kind, tile, ksplit):
 *   kind 0 im2col kernel (conv_mfma.hip), 1 patch-staged kernel (conv_patch.hip; ksplit + 1000 * (pixel-tile shape + 1)),
 *   8 minimal-filtering transposed conv (conv_wino.hip; tile = variant: 0..3 = 16 output channels x 32 / 64 / 48 / 16 tiles per
 *      workgroup, 4 / 5 = 32 output channels x 16 / 32 tiles),
 *   10 1-D minimal filtering for k x 1 / 1 x k convs (3 taps stride 1; 5 / 7 / 9 taps stride 2) and 3 x 3 stride-1 convs as three
 *      1 x 3 filters (conv_wino.hip; tile = workgroup shape 0..10); 9 is not used (a removed experiment),
 *   11 weight-streaming kernel for the dense layers (dense_stream.hip: dense5 of v2, motion_fc1; tile 0 / 1 = default /
 *      non-temporal weight loads, ksplit = K slices), 12 the blocks' first layer (9 x 1, stride 2, <= 6 input channels) with the weights
 *      in registers (conv_thin.hip; tile 0) -- on that layer it also means: the conv1 pair runs as two launches, not as conv_pair.hip's one,
 *   13 1 x 7 / 1 x 9 stride-2 conv with at most 32 channels on both sides, whole reduction out of LDS (conv_row.hip; tile 0),
 *        3 small-Cout VALU kernel, 4 register-streaming kernel (conv_stream.hip), 5 fragment-tiled kernel (conv_frag.hip),
 *        6 / 7 on the k x 1 layer of a stride-1 pair: the pair runs as ONE chained launch of conv_frag / conv_stream variant `tile`;
 *        14 marker on the k x 1 layer of a conv_pair.hip pair: the fused launch measured faster at this batch size;
 *        15 3 x 3 conv as three 1 x 3 minimal-filtering row filters with the transformed input rows stationary (conv_wino3.hip;
 *           stride 1: tile = workgroup shape 0..7 on F(2,3) tiles, 8..15 on F(4,3) tiles; stride 2 (rows of a multiple of 8 pixels):
 *           tile 16..19, polyphase F(4,2) + F(4,1)),
 *        16 k x 1 / 1 x k conv with four outputs per window (conv_wino4.hip: F(4,3) for 3 taps stride 1, F(4,3) + F(4,2) for 5 taps
 *           stride 2; tile = workgroup shape 0..13; ksplit field = launch form: 1 one workgroup per tile, 2 tile-walking workgroups, 3 flat
 *           line order -- the lines of all images of the batch as one sequence, for maps whose lines per image do not fill a workgroup;
 *           a form that does not exist for the shape, or would save nothing, runs as form 1);
 *   tile = tile / variant id of that kernel; ksplit = K slices across workgroups, combined by a conv_splitk_reduce launch.
 *   demon_plan_get returns DEMON_ERR_NOT_FOUND for an untuned layer. */
int demon_num_layers(const demon_ctx *ctx);
int demon_plan_get(const demon_ctx *ctx, int n, int layer_index, char *name, int name_cap, int *kind, int *tile, int *ksplit);
int demon_plan_set(demon_ctx *ctx, int n, const char *layer_name, int kind, int tile, int ksplit);
/* forgets every plan entry of batch size n (the layers fall back to the nearest tuned batch size / the heuristics): call it before
 * installing ANOTHER plan for the same n, so that a layer the new plan does not mention (e.g. a kind-14 marker of the old one)
 * does not survive the swap */
int demon_plan_clear(demon_ctx *ctx, int n);

/* ---- networks, host buffers in / host buffers out ------------------------------------------------
 * demon_bootstrap  replaces BootstrapNet.eval   (networks_original.py:60-88)
 * demon_iterative  replaces IterativeNet.eval   (networks_original.py:154-198)
 * demon_refine     replaces RefinementNet.eval  (networks_original.py:236-255)
 * demon_full       replaces the loop of examples/example.py:87-99 without host round trips     */
int demon_bootstrap(demon_ctx *ctx, int n, const float *image_pair, const float *image2_2, const demon_outputs *out);
int demon_iterative(demon_ctx *ctx, int n, const float *image_pair, const float *image2_2, const float *depth2,
                    const float *normal2, const float *rotation, const float *translation,
                    const demon_outputs *out);
int demon_refine(demon_ctx *ctx, int n, const float *image1, const float *depth2, float *predict_depth0);
int demon_full(demon_ctx *ctx, int n, const float *image_pair, const float *image2_2, int iterations,
               const demon_outputs *out, float *predict_depth0);

/* ---- device-resident path (throughput) ------------------------------------------------------------
 * inputs stay in HBM; image1 is image_pair[:,0:3].  demon_run_full enqueues bootstrap +
 * `iterations` x iterative + refine on the context stream (one hipGraph launch when enabled) and
 * returns without synchronising.                                                                */
int demon_upload_inputs(demon_ctx *ctx, int n, const float *image_pair, const float *image2_2);
int demon_run_full(demon_ctx *ctx, int n, int iterations);
int demon_run_bootstrap(demon_ctx *ctx, int n);
int demon_synchronize(demon_ctx *ctx);
/* Stream <-> hardware-queue mapping: the HIP runtime binds a stream to one of a few hardware queues at creation, by a rule that
 * depends on every stream alive in the process, and two busy streams on one queue serialise.  A group of contexts that measures a
 * poor mapping (demon_amd/lanes.py: LaneGroup.calibrate) releases all its streams, optionally creates placeholder streams
 * (demon_create_ops contexts), and acquires new ones context by context.  Nothing may be in flight; cached hipGraph execs are dropped
 * (a graph captured across a context's two streams does not survive them) and captured again at the next run call.
 * Between release and acquire the context must not be used: every entry point that would enqueue work returns
 * DEMON_ERR_NOT_READY (nothing falls onto the null stream); demon_synchronize, demon_set_weight*, demon_plan_* still work.   */
int demon_release_streams(demon_ctx *ctx);
int demon_acquire_streams(demon_ctx *ctx);
/* Lanes (several contexts = several passes in flight on ONE GPU; the reference's counterpart is a caller that loops over batches,
 * examples/evaluation.py:225-256).  How many lanes pay off, and behind how many idle "placeholder" streams their streams must be
 * created to land on hardware queues of their own, is MEASURED (it depends on every stream alive in the process: another
 * library's, RCCL's, torch's):
 *   demon_lanes_apply     : every lane releases its streams, `placeholder_streams` idle streams are created (owned by ctxs[0], freed
 *                           with it or by the next call), the lanes acquire new streams in array order.  Nothing may be in flight.
 *   demon_lanes_calibrate : for 0 .. max_placeholders placeholder streams and k = 1 .. nctx lanes: the rate of steps_per_lane * k
 *                           forward passes (demon_run_full(n, iterations), or demon_run_bootstrap when bootstrap_only) fed round
 *                           robin to the first k contexts over their RESIDENT inputs (demon_upload_inputs first), best of two
 *                           rounds by the host clock.  lanes_mask: bit k set = k lanes may win (0: any).  Fills `result` (winner +
 *                           the whole table) and leaves the contexts on the winning mapping, VERIFIED: the queue a new stream gets
 *                           also depends on how many streams the process created before, so the winner is applied and measured
 *                           again, with one throw-away stream more per attempt, until its rate is back (verified_pairs_per_s,
 *                           attempts).  Closing the contexts beyond result->lanes is the caller's business.  A winner measured once
 *                           can be tried in a later process of the same kind with demon_lanes_apply (measure it there too).
 * Contexts of a group: same device, side_branches off (demon_set_option) when nctx > 1, one host thread. */
#define DEMON_LANES_MAX 8
#define DEMON_LANES_MAX_PLACEHOLDERS 7
#define DEMON_LANES_TABLE_CAP 64
#define DEMON_LANES_MAX_ATTEMPTS 16
typedef struct demon_lanes_entry {
    int lanes, placeholder_streams;
    float pairs_per_s;
} demon_lanes_entry;
typedef struct demon_lanes_result {
    int lanes, placeholder_streams;   /* the winner */
    float pairs_per_s;                /* its rate in the sweep */
    float verified_pairs_per_s;       /* its rate measured AGAIN in the state the contexts are left in */
    int attempts;                     /* how often the winner had to be applied until that rate was within 2.5 % of the sweep's */
    int ntable;
    demon_lanes_entry table[DEMON_LANES_TABLE_CAP];
} demon_lanes_result;
int demon_lanes_apply(demon_ctx *const *ctxs, int nctx, int placeholder_streams);
int demon_lanes_calibrate(demon_ctx *const *ctxs, int nctx, int n, int iterations, int bootstrap_only, int steps_per_lane,
                          int max_placeholders, unsigned lanes_mask, demon_lanes_result *result);
/* Two ways of running a lane group WITHOUT depending on where the runtime puts a stream (round 6; measured against the calibrated
 * round-robin form):
 *   demon_set_cu_mask     : the context's streams are re-created on a compute-unit mask (hipExtStreamCreateWithCUMask; nwords 32-bit
 *                           words, bit i = CU slot i / 8 of XCD i % 8; nwords = 0: every CU again).  Nothing may be in flight; cached
 *                           graphs are dropped.  A mask must leave every XCD at least one CU (DEMON_ERR_INVALID otherwise).  Lanes on
 *                           disjoint masks do not compete for compute units, whatever hardware queue their streams share.
 *   demon_lanes_run_group : ONE hipGraph holding the forward passes of ctxs[0 .. nctx) (demon_run_full(n, iterations), or
 *                           demon_run_bootstrap when bootstrap_only) as parallel branches, launched on ctxs[0]'s stream: one call =
 *                           nctx steps, enqueued without synchronising; demon_synchronize(ctxs[0]) waits for all of them.
 *   demon_hw_queues_hint  : the GPU_MAX_HW_QUEUES value the lanes were measured best with (8 in a plain process, 16 when a
 *                           torch.distributed / RCCL launcher's streams exist first); a host exports it BEFORE its first HIP call
 *                           (the runtime reads it once, at initialisation) -- the library never changes the process environment. */
int demon_set_cu_mask(demon_ctx *ctx, const uint32_t *mask, int nwords);
int demon_lanes_run_group(demon_ctx *const *ctxs, int nctx, int n, int iterations, int bootstrap_only);
int demon_hw_queues_hint(int under_launcher);
int demon_download_outputs(demon_ctx *ctx, int n, const demon_outputs *out, float *predict_depth0);
/* Pipelining across contexts (copy / compute overlap): the _async variants only enqueue on the context's stream; the host buffers
 * must be page-locked (demon_host_register pins an existing allocation, e.g. a numpy array) and stay untouched until
 * demon_synchronize.  Two contexts fed alternately keep the copy engines busy under the kernels of the other one
 * (demon_amd/pipeline.py).                                                                                              */
int demon_upload_inputs_async(demon_ctx *ctx, int n, const float *image_pair, const float *image2_2);
int demon_download_outputs_async(demon_ctx *ctx, int n, const demon_outputs *out, float *predict_depth0);
int demon_host_register(void *ptr, int64_t bytes);
int demon_host_unregister(void *ptr);
/* v2 contexts only: predict_normal0 [n,3,H,W] of the last refinement run (v2/networks.py:223-226; v2/blocks.py:560-562). */
int demon_download_normal0(demon_ctx *ctx, int n, float *predict_normal0);
/* time `steps` back-to-back demon_run_full calls with hip events on the context stream */
int demon_time_full(demon_ctx *ctx, int n, int iterations, int steps, float *total_ms);

/* ---- per-launch profile (hip events around every kernel launch of one full pass, eager) ---------- */
typedef struct demon_launch_record {
    char name[64];     /* e.g. "netFlow1/conv1y" */
    char kernel[32];   /* the kernel that ran: "conv_mfma<128x32>+splitk" (= conv_mfma_kernel<128, 32, ..> followed by the split-K
                          reduce), "conv_patch<64x128,t5>", "deconv4<32x128>", "conv_pair", "conv_small", "warp2d", ... */
    double flops;      /* algorithmic 2*MAC of this launch (0 for non conv ops) */
    double bytes;      /* algorithmic bytes read + written */
    float ms;          /* whole step, including the split-K reduce launch where one follows */
    float reduce_ms;   /* of which: the conv_splitk_reduce launch (0 when the step has none) -- ms - reduce_ms is the kernel alone */
} demon_launch_record;
/* Under option "tune_lanes" > 1 every step is timed as that many CONCURRENT replays on as many streams, all writing the same output
 * buffers and split-K workspace: valid for timing only -- the context's resident activations and outputs are garbage afterwards (run a
 * pass again before downloading anything). */
int demon_profile_full(demon_ctx *ctx, int n, int iterations, int repeats, demon_launch_record *records, int cap,
                       int *count);

/* Diagnostic builds only (hipcc -DDEMON_TIMELINE, tools/timeline.py): per-workgroup wall-clock records (8 x uint64 each: entry,
 * prologue done, K loop done, stores drained [100 MHz ticks], HW_ID, XCC_ID, 0, 0) of one launch of the named layer at batch n;
 * the product build returns DEMON_ERR_INVALID. */
int demon_debug_timeline(demon_ctx *ctx, const char *layer_name, int n, uint64_t *records, int cap, int *count, float *ms,
                         char *kernel, int kernel_cap);

/* ---- lmbspecialops-level entry points (host buffers) -------------------------------------------------
 * Replace the lmbspecialops custom ops the reference calls:
 *   depth_to_flow            blocks_original.py:155-162 (+ gate :163-168 when gate != 0)
 *   flow_to_depth            blocks_original.py:344-351 (method 0) / v2/blocks.py:362-378 flow_to_depth2 (method 1)
 *   warp2d                   blocks_original.py:171-176, :336   (border_mode 0 clamp, 1 value)
 *   leaky_relu               helpers.py:60-63
 *   replace_nonfinite        v2/losses.py:49
 *   scale_invariant_gradient v2/losses.py:76-79   out [nc, 2, h, w] (channels fold into the batch; 0 = x, 1 = y), the deltas of one
 *                            call are summed with their weights (lmbspecialops contract; the reference concatenates one-delta calls)
 *   median3x3_downsample     examples/evaluation.py:173, v2/helpers.py:102   (NaN sorts last)
 *   depth_to_normals         v2/losses.py:336-337   out [n, 3, h, w], NaN at the border / invalid depth.  UNVERIFIED against
 *                            lmbspecialops (source absent): pixel centres +0.5, one-sided differences with the smaller depth change,
 *                            normal towards the camera are this library's decisions; pinned only by analytic plane tests   */
int demon_op_depth_to_flow(demon_ctx *ctx, float *out, const float *depth, const float *intrinsics,
                           const float *rotation, const float *translation, int n, int h, int w,
                           int inverse_depth, int normalize_flow, int gate);
int demon_op_flow_to_depth(demon_ctx *ctx, float *out, const float *flow, const float *intrinsics,
                           const float *rotation, const float *translation, int n, int h, int w,
                           int inverse_depth, int normalized_flow, int method);
int demon_op_warp2d(demon_ctx *ctx, float *out, const float *input, const float *displacements, int n, int c,
                    int h, int w, int normalized, int border_mode, float border_value);
int demon_op_leaky_relu(demon_ctx *ctx, float *out, const float *in, int64_t count, float leak);
int demon_op_replace_nonfinite(demon_ctx *ctx, float *out, const float *in, int64_t count, float value);
int demon_op_scale_invariant_gradient(demon_ctx *ctx, float *out, const float *in, int nc, int h, int w,
                                      const int *deltas, const float *weights, int ndeltas, float epsilon);
int demon_op_median3x3_downsample(demon_ctx *ctx, float *out, const float *in, int nc, int h, int w);
int demon_op_depth_to_normals(demon_ctx *ctx, float *out, const float *depth, const float *intrinsics, int n, int h, int w,
                              int inverse_depth);
/* pointwise_l2_loss of v2/losses.py:33-54 (NCHW): mean over pixels of sqrt(sum_c replace_nonfinite(inp - gt)^2 + epsilon) */
int demon_op_pointwise_l2_loss(demon_ctx *ctx, float *loss, const float *inp, const float *gt, int n, int c, int h, int w,
                               float epsilon);

/* ---- multi-GPU: one process per GPU, RCCL through the C ABI ------------------------------------------
 * image pairs shard across ranks with no data-path collective; the only collective is one
 * broadcast of the weights at start-up (the reference has no multi-GPU inference path; tf.train.Saver.restore,
 * examples/example.py:82-83, is what each process would otherwise repeat from disk).
 *   demon_comm_get_unique_id : ncclGetUniqueId -- call on ONE rank, ship the 128 bytes to the others by any means
 *                              (MPI, a file, torch.distributed's store ...)
 *   demon_comm_init_rank     : ncclCommInitRank on `device`; *nccl_comm is a plain ncclComm_t, usable with rccl.h directly
 *   demon_broadcast_weights  : ONE ncclBroadcast (float32) of the packed device-resident weight slab from `root` on the
 *                              context's stream, then marks every variable as set; `nccl_comm` may be any ncclComm_t whose
 *                              rank `rank` is this process.  Collective: every rank of the communicator must call it.
 *                              The collective is preceded by a 12-byte header broadcast + a 4-byte MIN all-reduce: the root's
 *                              slab layout (demon_weights_slab_layout) must equal every rank's, else ALL ranks return
 *                              DEMON_ERR_INVALID without entering the big broadcast.
 *   demon_comm_count         : ncclCommCount -- how many ranks the communicator really has (bench.py prints it)
 *   demon_copy_weights_from  : the receiver side without a second GPU / process: device-to-device copy of `src`'s packed
 *                              slab into `dst` (same layout), then exactly the epilogue of a non-root rank of
 *                              demon_broadcast_weights.  `dst` needs no demon_set_weight call.
 * librccl.so is dlopen'ed on first use; without it these return DEMON_ERR_HIP and nothing else is affected. */
#define DEMON_COMM_ID_BYTES 128
int demon_comm_get_unique_id(char *id /* [DEMON_COMM_ID_BYTES] */);
int demon_comm_init_rank(void **nccl_comm, int nranks, const char *id /* [DEMON_COMM_ID_BYTES] */, int rank, int device);
int demon_comm_destroy(void *nccl_comm);
int demon_broadcast_weights(demon_ctx *ctx, void *nccl_comm, int root, int rank);
int64_t demon_weights_slab_bytes(const demon_ctx *ctx);
uint64_t demon_weights_slab_layout(const demon_ctx *ctx);  /* hash of model variant, image size and the per-layer slab offsets; 0: no networks */
int demon_comm_count(void *nccl_comm, int *nranks);
int demon_copy_weights_from(demon_ctx *dst, const demon_ctx *src);

/* ---- layer-level entry points (host buffers; TF weight layouts) -------------------------------------
 * Replace the tf.layers calls of helpers.py:85-94 / :128-153 (conv2d on a zero padded input),
 * blocks_original.py:64-75 / :97-110 (conv2d_transpose k4 s2 + crop) and :390-410 (dense).
 * conv2d: ph, pw = kh/2, kw/2 (zeros on both sides, then VALID), or ph = pw = -1 for padding='same' of v2/helpers.py:24-35
 * (out = ceil(n/s), pad_total/2 zeros in front).                                                    */
int demon_op_conv2d(demon_ctx *ctx, float *out, const float *in, const float *w_hwio, const float *bias, int n,
                    int cin, int h, int w, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int lrelu);
int demon_op_deconv4x4s2(demon_ctx *ctx, float *out, const float *in, const float *w_hwoi, const float *bias,
                         int n, int cin, int h, int w, int cout, int lrelu);
int demon_op_dense(demon_ctx *ctx, float *out, const float *in, const float *w_io, const float *bias, int n,
                   int cin, int cout, int lrelu);

/* ---- tuning / diagnostics ----------------------------------------------------------------------------
 * Times one contraction layer (kind 0 conv, 1 transposed conv k4 s2, 2 dense) on device-resident random
 * data with hip events; tile < 0 / ksplit <= 0 select the automatic plan (tile 0..7 im2col tiles, 100 + t patch tiles,
 * 200 + v streaming-kernel variants, 300 + v fragment-tiled variants, 400 + v minimal-filtering variants (transposed conv / k x 1, 1 x k, 3 x 3 conv) resp. 400 = the weight-streaming kernel on a dense layer / the first-layer kernel, 500 = conv_row.hip).
 * Not on the reference's path. */
int demon_bench_layer(demon_ctx *ctx, int kind, int n, int cin, int h, int w, int cout, int kh, int kw, int sh,
                      int sw, int tile, int ksplit, int iters, float *avg_ms, double *flops);

/* Poison harness (tests/test_poison_gpu.py).  With DEMON_POISON_GUARD=1 in the environment when a context is created (or when a
 * demon_op_conv2d / deconv4x4s2 / dense call runs), every device allocation -- activations, every weight form, workspaces -- is placed
 * flush between two 4 MiB zones filled with a quiet-NaN canary and the activation buffers lose their slack planes: a kernel that
 * reads outside a tensor and USES the value produces NaN, a kernel that writes outside one is found here (the layer-level
 * entry points check by themselves and return DEMON_ERR_HIP).  *violations = guard zones that no longer hold the canary;
 * demon_last_error names the first.  DEMON_ERR_INVALID on a context created without the switch.  Not on the reference's path. */
int demon_debug_check_guards(demon_ctx *ctx, int *violations);

/* Diagnostic: the tag of the contraction kernel the calling thread launched last (the names demon_profile_full reports, e.g.
 * "wino_deconv<16x64>+splitk", "conv_frag<128x32,v6>"); tests use it to see that a forced variant really ran.  Returns the tag
 * length (0: nothing launched yet). */
int demon_last_kernel(char *tag, int tag_cap);

#ifdef __cplusplus
}
#endif
#endif /* DEMON_HIP_H */
