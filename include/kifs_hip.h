/*
 * kifs_hip.h -- C ABI of the MI355X-native raymarching library (libkifs_hip.so).
 *
 * Drop-in boundary for the one hot path of LesbianLemon/kifs-raymarching: the
 * per-pixel sphere-tracing fragment shader.  The reference has no FFI today;
 * the seam this ABI replaces is `GraphicState` (src/render/graphics.rs:25-37):
 *
 *   reference call (file:line)                         this ABI
 *   -------------------------------------------------  ---------------------------
 *   GraphicState::new            graphics.rs:183-232   kifs_create
 *   update_screen_data           graphics.rs:262-266   kifs_set_screen   (12 B image)
 *   zoom_camera / rotate_camera  graphics.rs:268-302   kifs_set_camera   (64 B image)
 *   update_options               graphics.rs:304-308   kifs_set_options  (80 B image)
 *   render (set_pipeline by fractal_group, bind group,
 *           draw(0..3, 0..2))    graphics.rs:310-325   kifs_render / kifs_render_async
 *   drop(RenderState)            application.rs:86-91  kifs_destroy
 *
 * The three uniform structs are byte-identical to the reference's Pod structs
 * (src/data.rs:17-49) and to the WGSL declarations
 * (src/shaders/dependencies/bindings.wgsl:1-35), so a Rust host passes
 * `bytemuck::bytes_of(&data.into_buffer_data())` unchanged.
 *
 * Plain pointers and sizes only; no exceptions cross the boundary; every entry
 * point returns a KifsStatus.  A context is not thread-safe: one caller thread
 * per context, mirroring the reference's single winit thread
 * (src/application.rs:37-48).
 */
#ifndef KIFS_HIP_H
#define KIFS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KIFS_ABI_VERSION 4

/* ---- uniform images (data.rs:17-49) --------------------------------------- */

typedef struct KifsScreenUniform { /* ScreenUniformData, data.rs:17-23 */
    float width;
    float height;
    float aspect_ratio;
} KifsScreenUniform; /* 12 bytes */

typedef struct KifsCameraUniform { /* CameraUniformData, data.rs:25-31 */
    float origin[3];
    uint32_t _padding;
    float matrix[3][4]; /* mat3x3 as 3 columns with 16-byte stride (packed.rs:78-92) */
} KifsCameraUniform; /* 64 bytes */

typedef struct KifsOptionsUniform { /* OptionsUniformData, data.rs:33-49 */
    int32_t max_iterations;
    float max_distance;
    float epsilon;
    uint32_t _padding1;
    float fractal_color[3]; /* linear RGB */
    uint32_t _padding2;
    float background_color[3];
    uint32_t is_heatmap;
    uint32_t fractal_group_id; /* KifsFractalGroup, scene.rs:4-11 */
    uint32_t primitive_id;     /* KifsPrimitiveShape, scene.rs:35-45 */
    float power;
    uint32_t _padding3;
    float constant[4]; /* quaternion (real, i, j, k) */
} KifsOptionsUniform; /* 80 bytes */

typedef enum KifsFractalGroup { /* data/scene.rs:4-11 */
    KIFS_GROUP_KIFS = 0,
    KIFS_GROUP_JULIA = 1,
    KIFS_GROUP_GEN_JULIA = 2
} KifsFractalGroup;

typedef enum KifsPrimitiveShape { /* data/scene.rs:35-45 */
    KIFS_PRIM_SPHERE = 0,
    KIFS_PRIM_CYLINDER = 1,
    KIFS_PRIM_BOX = 2,
    KIFS_PRIM_TORUS = 3,
    KIFS_PRIM_SIERPINSKI = 4,
    KIFS_PRIM_BUNNY = 5
} KifsPrimitiveShape;

/* Colour-target encoding: the reference picks the first sRGB surface format if
 * any, else formats[0] (render.rs:72-80); blend REPLACE (graphics.rs:87-91). */
typedef enum KifsEncode {
    KIFS_ENCODE_UNORM = 0, /* linear -> UNORM8 */
    KIFS_ENCODE_SRGB = 1   /* linear -> sRGB OETF -> UNORM8 (alpha stays linear) */
} KifsEncode;

/* Status codes; taxonomy follows src/error.rs:66-92. */
typedef enum KifsStatus {
    KIFS_OK = 0,
    KIFS_ERR_NO_DEVICE = 1,    /* ~ RenderStateError::RequestAdapter */
    KIFS_ERR_DEVICE_INIT = 2,  /* ~ RenderStateError::RequestDevice */
    KIFS_ERR_BAD_SIZE = 3,     /* ~ RenderError::SurfaceMissized; zero size (render.rs:211) */
    KIFS_ERR_UNCONFIGURED = 4, /* render before all three uniforms were set */
    KIFS_ERR_RUNTIME = 5,      /* HIP error: fatal for the context */
    KIFS_ERR_COMM = 6,         /* an inter-GPU transfer failed: RCCL (init, group, send/recv) or a peer copy */
    KIFS_ERR_BAD_ARG = 7       /* null pointer, unknown enum value, bad range */
} KifsStatus;

typedef struct kifs_ctx kifs_ctx;

/* ---- lifetime --------------------------------------------------------------
 * Replaces GraphicState::new (graphics.rs:183-232) + adapter/device request
 * (render.rs:42-69).  Binds the context to HIP device `device_ordinal`, creates
 * its stream and events and uploads the sRGB threshold table.  Returns NULL
 * and sets *status on failure. */
kifs_ctx* kifs_create(int device_ordinal, int* status);

/* Replaces the explicit drop (application.rs:86-91).  NULL is a no-op. */
void kifs_destroy(kifs_ctx* ctx);

/* ---- uniforms (each call COPIES its argument, like queue.write_buffer) ------ */
int kifs_set_screen(kifs_ctx* ctx, const KifsScreenUniform* screen);    /* graphics.rs:262 */
int kifs_set_camera(kifs_ctx* ctx, const KifsCameraUniform* camera);    /* graphics.rs:268,280 */
int kifs_set_options(kifs_ctx* ctx, const KifsOptionsUniform* options); /* graphics.rs:304 */

/* The reference hard-codes JULIA_ITERATIONS = 100, JULIA_NORMAL_ITERATIONS = 10
 * (julia.wgsl:2-3, gen_julia.wgsl:2-3) and 10 Sierpinski folds (kifs.wgsl:72).
 * Those are the defaults; BASELINE configs override them.  All must be >= 0. */
int kifs_set_iters(kifs_ctx* ctx, int sdf_iters, int normal_iters, int fold_iters);

/* ---- extension: soft shadows ---------------------------------------------------------
 * NOT part of the reference (its whole shading model is entry.wgsl:6-29); BASELINE config 5
 * names "soft-shadow secondary rays", so it exists as an explicit opt-in whose semantics this
 * library defines.  With soft_shadow != 0, every hit pixel marches a secondary ray from
 * p + 2*epsilon*n towards L = normalize((1,1,1)) (the direction of the reference's light
 * vector, entry.wgsl:17):
 *     res = 1; t = shadow_t0
 *     up to shadow_steps times: h = scene_SDF(start + t*L); if h < epsilon -> res = 0, stop;
 *                               res = min(res, shadow_k*h/t); t += h; stop if t > shadow_max_t
 * and the direct term is attenuated: diffuse = 0.1 + 0.9 * clamp(n.(1,1,1), 0, 1) * res.  A hit whose direct term
 * is not positive (it faces away from the light) marches no secondary ray (res = 1: there is nothing to attenuate).
 * All zero (the default) is the reference's behaviour exactly. */
typedef struct KifsExtensions {
    uint32_t soft_shadow;
    int32_t shadow_steps;
    float shadow_k;
    float shadow_t0;
    float shadow_max_t;
} KifsExtensions;
int kifs_set_extensions(kifs_ctx* ctx, const KifsExtensions* ext);

/* ---- render ---------------------------------------------------------------
 * Replaces GraphicState::render (graphics.rs:310-325): pipeline chosen by
 * options.fractal_group_id, one kernel launch instead of draw(0..3, 0..2).
 *
 * Renders rows [y0, y1) of the W x H frame (W, H from the screen uniform; pixel
 * coordinates stay global, so a band is bit-identical to the same rows of the
 * full frame).  Row y goes to out + (y - y0) * pitch_bytes, 4 bytes per pixel,
 * R G B A.  pitch_bytes >= 4*W and a multiple of 4.
 *
 * The library launches on non-blocking streams (its own, or the caller's `hip_stream`): work the caller has
 * pending on OTHER streams for the destination (a fill, a previous consumer) is not ordered before the render
 * unless the caller orders it, as with any asynchronous HIP work.  kifs_order_after does that in one call: the
 * context's NEXT work on `hip_stream` (NULL = the context's own stream, the one kifs_render uses) runs after
 * everything `producer_stream` (a hipStream_t of the context's device; NULL = the legacy default stream) holds
 * when the call is made -- an event recorded there and waited for on the launch stream; nothing blocks the host;
 * the same stream twice is a no-op.  (The reference has one queue and copies on every update,
 * util/uniform.rs:23-35, so the question does not arise there.)
 *
 * kifs_render: `out` may be a device pointer of the context's device or a host
 * pointer; returns once `out` holds the pixels.
 * kifs_render_async: `out` must be device memory; the launch is enqueued on
 * `hip_stream` (a hipStream_t; NULL = the context's own stream) and the call
 * returns without synchronising. */
int kifs_render(kifs_ctx* ctx, uint8_t* out_rgba8, size_t pitch_bytes, int y0, int y1,
                int encode);
int kifs_render_async(kifs_ctx* ctx, void* hip_stream, uint8_t* dev_out_rgba8,
                      size_t pitch_bytes, int y0, int y1, int encode);
int kifs_order_after(kifs_ctx* ctx, void* hip_stream, void* producer_stream);

/* A batch of frames in one launch: frame i is rendered with cameras[i] (the context's screen,
 * options, iteration counts and extensions; the context's own camera is neither used nor
 * changed) into dev_outs[i], same pitch / band / encoding for all.  1 <= count <= KIFS_MAX_BATCH.
 * This is the throughput path for sequences of independent frames (the frames of an orbit,
 * cf. rotate_camera graphics.rs:280-302): one frame of these scenes keeps most of the device
 * idle -- its run time is the critical path of a few long rays -- and the workgroups of a batch
 * are interleaved frame by frame, so the long rays of all its frames march side by side.
 * Every frame is bit-identical to the same frame rendered alone. */
#define KIFS_MAX_BATCH 512
int kifs_render_batch_async(kifs_ctx* ctx, void* hip_stream, int count,
                            const KifsCameraUniform* cameras, uint8_t* const* dev_outs_rgba8,
                            size_t pitch_bytes, int y0, int y1, int encode);

/* Contiguous row-band partition used for multi-GPU frames: rank r
 * of `world` owns rows [y0, y1); bands differ by at most one row. */
int kifs_band_range(int height, int rank, int world, int* y0, int* y1);

/* ---- row shards: load-balanced multi-GPU partition ----------------------------
 * Pixels are independent (entry.wgsl:49-59), so any set of rows can be rendered anywhere.  The
 * expensive rows of these scenes sit in the middle of the frame (the projected bounding sphere), so
 * contiguous bands leave the outer ranks idle.  A row SHARD is instead a list of 8-row stripes
 * (KIFS_STRIPE_ROWS = the height of the kernels' 32 x 8 pixel tiles) dealt to the ranks in turn:
 * stripe s covers frame rows [8 s, min(H, 8 s + 8)).
 *
 * kifs_shard_stripes: the stripes of `rank` when the frame's stripes are dealt to `world` ranks by
 * smooth weighted round robin -- weights NULL: equal shares, rank r gets stripes r, r + world, ..;
 * weights[r] >= 0: rank r gets weights[r] stripes in every sum(weights), spread evenly (a root that
 * receives everybody else's rows over point-to-point xGMI links can be given a larger share, so
 * that the peers' transfers take as long as the root's rendering).  Writes the ascending stripe
 * indices to stripes[0 .. *n_stripes) (stripes may be NULL to only count) and the shard's total row
 * count to *rows.  Every rank computes every rank's list from the same arguments.
 *
 * kifs_render_shard_async: kifs_render_batch_async for a shard.  Frame i (cameras[i]; cameras NULL
 * with count 1 = the context's camera) is rendered into dev_outs[i]:
 *   in_place == 0: a PACKED shard -- stripe k of the list occupies rows [8 k, 8 k + 8) of the
 *                  buffer (n_rows x pitch_bytes: what a peer sends to the root in one message);
 *   in_place != 0: dev_outs[i] is a whole frame and every row lands at its frame position (the
 *                  root's own shard needs no copy).
 * Pixel coordinates are global either way: shards are bit-identical to the frame's rows.
 *
 * kifs_unpack_shard_async: the root's side of the gather.  Copies `count` packed shards
 * (dev_shards + i * shard_stride, rows shard_pitch apart) into the frames dev_frames + i *
 * frame_stride (rows frame_pitch apart), stripe k to frame rows [8 stripes[k], ..).  Frame size from
 * the context's screen; all strides in bytes, multiples of 4. */
#define KIFS_STRIPE_ROWS 8
int kifs_shard_stripes(int height, int world, const int* weights, int rank, int* stripes,
                       int max_stripes, int* n_stripes, int* rows);
int kifs_render_shard_async(kifs_ctx* ctx, void* hip_stream, int count,
                            const KifsCameraUniform* cameras, uint8_t* const* dev_outs_rgba8,
                            size_t pitch_bytes, const int* stripes, int n_stripes, int in_place,
                            int encode);
int kifs_unpack_shard_async(kifs_ctx* ctx, void* hip_stream, int count, uint8_t* dev_frames,
                            size_t frame_pitch, size_t frame_stride, const uint8_t* dev_shards,
                            size_t shard_pitch, size_t shard_stride, const int* stripes,
                            int n_stripes);

/* ---- sparse shards: a peer's rows without their background tiles -----------------------------
 * The root of a gather takes each peer's rows over ONE xGMI link (~19 Gpixel/s inbound per link), a
 * GPU renders several times faster, and nine tenths of a 1080p frame of these scenes are the
 * background colour (the clear colour of the reference's render pass, render.rs:300-330).  So a peer
 * sends only the 32 x 8 tiles of its packed shards that hold a pixel other than the background, and
 * the root writes the background itself.  Lossless for any frame: a frame with no background at
 * all costs 1.6 % more than its dense form.
 *
 * A RECORD is KIFS_SPARSE_RECORD_BYTES = 1040 bytes: uint32 tile id, three zero words, then the tile's
 * 8 rows of 32 RGBA8 pixels (pixels outside the frame hold the background).
 * Tile id = (shard i * n_stripes + stripe slot k) * tiles_x + tile column, tiles_x = ceil(W / 32).
 *
 * kifs_pack_sparse_async: appends one record per non-background tile of the `count` packed shards
 *   (layout as kifs_unpack_shard_async's dev_shards; `stripes` the shard's list) to dev_records
 *   (16-byte aligned, room for capacity_records >= count * n_stripes * tiles_x records), in no
 *   particular order, and leaves their number in *dev_n_records (zeroed by the call, on the
 *   stream); host_n_records, when not NULL (pinned host memory), receives an asynchronous copy of
 *   it -- valid once the stream has reached that point.  `encode` selects the background pixel
 *   (the context's options, encoded as the render would).
 * kifs_unpack_sparse_async: the root's side.  Writes records [0, n_records) to their rows of the
 *   frames dev_frames + i * frame_stride; records whose id is out of range are skipped.
 * kifs_fill_shard_async: the background over the rows of the listed stripes of `count` frames
 *   (what the records leave out; any order with kifs_unpack_sparse_async's stream, before it).
 * kifs_erase_sparse_async: the background over the tiles of records [0, n_records) only -- frame
 *   buffers that are reused need it back just where the previous frames' records went (2 % of the
 *   headline's tiles) instead of a fill of every row. */
#define KIFS_SPARSE_RECORD_BYTES 1040
int kifs_pack_sparse_async(kifs_ctx* ctx, void* hip_stream, int count, const uint8_t* dev_shards,
                           size_t shard_pitch, size_t shard_stride, const int* stripes, int n_stripes,
                           int encode, uint8_t* dev_records, size_t capacity_records,
                           uint32_t* dev_n_records, uint32_t* host_n_records);
int kifs_unpack_sparse_async(kifs_ctx* ctx, void* hip_stream, int count, uint8_t* dev_frames,
                             size_t frame_pitch, size_t frame_stride, const uint8_t* dev_records,
                             size_t n_records, const int* stripes, int n_stripes);
int kifs_fill_shard_async(kifs_ctx* ctx, void* hip_stream, int count, uint8_t* dev_frames,
                          size_t frame_pitch, size_t frame_stride, const int* stripes, int n_stripes,
                          int encode);
int kifs_erase_sparse_async(kifs_ctx* ctx, void* hip_stream, int count, uint8_t* dev_frames,
                            size_t frame_pitch, size_t frame_stride, const uint8_t* dev_records,
                            size_t n_records, const int* stripes, int n_stripes, int encode);

/* ---- single-process multi-GPU ------------------------------------------------------
 * For a host that drives all GPUs of a node from one process (the reference's host is one
 * process, application.rs:37-48).  A kifs_multi owns one context per listed device; a render deals
 * the frame's 8-row stripes to the devices (kifs_shard_stripes; equal shares unless
 * kifs_multi_set_weights says otherwise), launches every shard on its own device concurrently --
 * the ROOT device (the first one listed) straight into the frame -- and the root pulls the other
 * shards over xGMI (hipMemcpyPeerAsync) and moves their stripes to their frame rows.  The frame is
 * bit-identical to a single-GPU frame.  The multi-process form of the same sharding (one process
 * per GPU, RCCL point-to-point gather) lives above the ABI in kifs_raymarching_amd/bands.py.  A
 * device may be listed more than once (testing on one GPU). */
typedef struct kifs_multi kifs_multi;
kifs_multi* kifs_multi_create(const int* device_ordinals, int n_devices, int* status);
void kifs_multi_destroy(kifs_multi* m);
int kifs_multi_set_screen(kifs_multi* m, const KifsScreenUniform* screen);
int kifs_multi_set_camera(kifs_multi* m, const KifsCameraUniform* camera);
int kifs_multi_set_options(kifs_multi* m, const KifsOptionsUniform* options);
int kifs_multi_set_iters(kifs_multi* m, int sdf_iters, int normal_iters, int fold_iters);
int kifs_multi_set_extensions(kifs_multi* m, const KifsExtensions* ext);
/* Shares of the devices, one integer per device in the order they were listed (NULL: equal). */
int kifs_multi_set_weights(kifs_multi* m, const int* weights);
/* `out_rgba8`: host memory or device memory of the root device; full frame, `pitch_bytes`
 * per row.  Returns when the frame is complete.  KIFS_ERR_COMM: a peer copy failed. */
int kifs_multi_render(kifs_multi* m, uint8_t* out_rgba8, size_t pitch_bytes, int encode);
/* Shard i of the configured screen: device ordinal, number of stripes, total rows. */
int kifs_multi_shard(kifs_multi* m, int i, int* device_ordinal, int* n_stripes, int* rows);
/* Kernel time of shard i in the last kifs_multi_render, ms (load balance across devices). */
double kifs_multi_shard_ms(kifs_multi* m, int i);

/* ---- batches of frames over several devices, gathered on the root (SURVEY 8b `kifs_render_multi`, 8e) ---------
 * The throughput form of kifs_multi_render for the one caller the reference has (one process, one thread:
 * application.rs:37-48, graphics.rs:310-325): `count` frames (cameras[i]; screen, options, iteration counts and
 * extensions as set) are rendered as ONE launch per device -- each device its row shard of every frame -- and
 * collected into `dev_frames` on the root device: frame i at dev_frames + i * frame_stride, rows frame_pitch
 * apart (caller-owned memory of the ROOT device, free to be overwritten when the call is made).
 *
 *   gather     KIFS_GATHER_SPARSE (default): the other devices send only the 32 x 8 tiles that hold a pixel
 *              other than the background (kifs_pack_sparse_async's records), the root writes the background
 *              under their rows itself and scatters the records over it.  KIFS_GATHER_DENSE: every row.
 *   transport  KIFS_TRANSPORT_RCCL: RCCL inside the library -- ncclCommInitAll over the listed devices once,
 *              then per step ONE group of point-to-point transfers, `ncclGroupStart(); every other device:
 *              ncclSend(its records, root); root: ncclRecv per device; ncclGroupEnd()` (RCCL has no gather;
 *              each sender -> root pair rides its own xGMI link).  librccl.so.1 is opened on first use; a
 *              device listed twice, a missing library or a failing call gives KIFS_ERR_COMM.
 *              KIFS_TRANSPORT_COPY: hipMemcpyPeerAsync per device (the copy engines; the only form that works
 *              with a device listed more than once, i.e. for testing on one GPU).
 *              KIFS_TRANSPORT_AUTO (default): RCCL when every listed device is distinct and there are at
 *              least two, else COPY.
 *
 * Asynchronous, two steps deep: kifs_multi_render_batch_async returns once the step's launches are enqueued
 * and writes the step's number to *step (0, 1, 2, ..).  A step's transfers are posted when the NEXT step has
 * been enqueued (their sizes are known on the host only once the senders have packed, and by then every device
 * is already rendering the next step), or by a wait.  kifs_multi_wait blocks until the frames of `step` are
 * complete in their dev_frames; kifs_multi_stream_wait makes a stream of the root device wait for them instead.
 * At most two steps are in flight: submitting step k first completes step k - 2.  A consumer therefore reads
 * the frames of step k after kifs_multi_wait(m, k) and before it hands the same buffer back in a later call.
 *
 * flags: KIFS_MULTI_FRAMES_UNTOUCHED -- dev_frames is the buffer of the submission two steps ago (same
 * pointer, pitch, stride, count, encoding) and nothing but this library has written to it since: with the sparse
 * gather the root then restores the background only under that step's records instead of under every row of
 * the other devices.  Without the flag (or when anything differs, the options' background included) every such
 * row is filled.  Each frame is bit-identical to the frame a single device renders.
 * kifs_multi_render_batch = submit + wait. */
typedef enum KifsGather { KIFS_GATHER_SPARSE = 0, KIFS_GATHER_DENSE = 1 } KifsGather;
typedef enum KifsTransport { KIFS_TRANSPORT_AUTO = 0, KIFS_TRANSPORT_RCCL = 1, KIFS_TRANSPORT_COPY = 2 } KifsTransport;
#define KIFS_MULTI_FRAMES_UNTOUCHED 1
int kifs_multi_set_gather(kifs_multi* m, int gather, int transport);
int kifs_multi_render_batch_async(kifs_multi* m, int count, const KifsCameraUniform* cameras, uint8_t* dev_frames,
                                  size_t frame_pitch, size_t frame_stride, int encode, int flags, uint64_t* step);
int kifs_multi_wait(kifs_multi* m, uint64_t step);
int kifs_multi_wait_all(kifs_multi* m);
int kifs_multi_stream_wait(kifs_multi* m, uint64_t step, void* hip_stream);
/* The counterpart of kifs_order_after for the frames a kifs_multi writes on its ROOT device: whatever the object
 * enqueues there from now on (the root's renders, the fill / erase and the scatter of the gather) runs after
 * everything `producer_stream` (a stream of the root device; NULL = the legacy default stream) holds now. */
int kifs_multi_order_after(kifs_multi* m, void* producer_stream);
int kifs_multi_render_batch(kifs_multi* m, int count, const KifsCameraUniform* cameras, uint8_t* dev_frames,
                            size_t frame_pitch, size_t frame_stride, int encode);
/* What the gather moved since creation (or the last call with reset != 0): steps completed, records received
 * and the tiles they stand for (sparse), payload bytes into the root, and the transport in use
 * (KIFS_TRANSPORT_RCCL / _COPY; AUTO until the first step decides). */
typedef struct KifsMultiStats {
    uint64_t steps;
    uint64_t records_received;
    uint64_t tiles_covered;
    uint64_t bytes_received;
    int32_t transport;
    int32_t gather;
    int32_t rccl_version; /* ncclGetVersion, 0 if RCCL was never opened */
    int32_t comm_ranks;   /* ranks of the communicator, 0 if none */
} KifsMultiStats;
int kifs_multi_stats(kifs_multi* m, KifsMultiStats* out, int reset);
/* Transport check: every other device sends `bytes` of a known pattern to the root through the configured
 * transport exactly as a step's gather would (with one device: the root sends to itself and receives from
 * itself in one group) and the root verifies what arrived.  KIFS_ERR_COMM on any mismatch or failing call. */
int kifs_multi_comm_selftest(kifs_multi* m, size_t bytes);

/* Device time of the most recent kifs_render on this context in ms (HIP events
 * on the launch stream), or a negative value if none completed. */
double kifs_last_kernel_ms(kifs_ctx* ctx);

/* Per-launch timing for benchmarks.  With enable != 0 every subsequent render launch is
 * bracketed by a pair of HIP events recorded on the launch stream immediately around the
 * render kernel (after any stream waits), kept in a ring of the latest 4096 launches;
 * enable = n > 1 times only every n-th launch (an event pair costs a few microseconds).
 * kifs_profile_read synchronises, reports how many launches were timed and their mean,
 * minimum and maximum kernel duration in ms, and clears the ring. */
int kifs_set_profiling(kifs_ctx* ctx, int enable);
int kifs_profile_read(kifs_ctx* ctx, int* launches, double* mean_ms, double* min_ms, double* max_ms);

/* Scheduling hint.  n = how many frames the caller keeps in flight on this device at once
 * (several contexts, one stream each; the reference queues up to
 * desired_maximum_frame_latency = 2 frames, render.rs:108).  With n = 1 (default) a launch is
 * tuned for the latency of a lone frame: frames whose run time is the critical path of a few
 * long rays cap their own residency so that those rays have a SIMD to themselves.  With n > 1
 * the device is shared on purpose and the cap is dropped (throughput over latency).  Pixels do
 * not depend on it.  Returns KIFS_ERR_BAD_ARG for n < 1. */
int kifs_set_frames_in_flight(kifs_ctx* ctx, int n);

/* Blocks until everything the context enqueued on its own stream is done. */
int kifs_synchronize(kifs_ctx* ctx);

const char* kifs_strerror(int status);
int kifs_abi_version(void);

/* ---- point evaluation (parity tests and tooling) -----------------------------
 * Evaluates scene_SDF and get_normal of the current options at `n` points
 * (xyz triples, host pointers) on the device.  Either output may be NULL. */
int kifs_eval_points(kifs_ctx* ctx, const float* points_xyz, int n, float* sdf_out,
                     float* normal_out_xyz);

/* Evaluates one of the library's f32 elementary functions on the device over
 * `n` host values.  fn: 0 log, 1 log2, 2 exp2, 3 sin, 4 cos, 5 acos,
 * 6 pow(x, y) with y = `param`, 7 sRGB-encode (result as float code),
 * 8 UNORM-encode, 9 / 10 the mid-range reciprocal and square root of the generalised-Julia
 * step (correctly rounded for 2^-60 <= x < 2^60; tests check them exhaustively), 11 the branch-free
 * form of sin the bunny network uses (same values as 3), 12 / 13 / 14 / 15 / 16 the straight-line cores of the
 * generalised-Julia step -- exp2, log2, sin, cos, acos for ORDINARY arguments only (|x| < 128; 2^-60 <= x < 2^60;
 * |x| <= 2^20; |x| <= 1), where they must equal 2 / 1 / 3 / 4 / 5 bit for bit. */
int kifs_eval_math(kifs_ctx* ctx, int fn, const float* in, float param, float* out, int n);

/* Diagnostics: with enable != 0, subsequent Julia renders write one record per wave into a
 * device buffer sized for the current screen; each call zeroes the buffer (out[8] receives
 * the 8 reserved header words).  enable == 0 frees it.  Not for timed runs. */
int kifs_debug_counters(kifs_ctx* ctx, int enable, unsigned long long out[8]);
/* Round length (march steps) of the ray re-queuing used by the context's latest launch; 0 = that
 * launch marched one wave per 8x8 block.  For tests. */
int kifs_debug_last_round_steps(kifs_ctx* ctx);
/* Shape of the throughput path in the context's latest launch: 0 = one single-wave workgroup per
 * tile (render_wave_kernel), 1 or 2 = tiles per 256-thread workgroup (render_group_kernel), -1 = the
 * launch did not re-queue rays at all.  For tests and bench.py's kernel name. */
int kifs_debug_last_group_tiles(kifs_ctx* ctx);
/* Which render kernel the context's latest launch used (-1 before the first).  For tests and bench.py. */
enum KifsKernel {
    KIFS_KERNEL_BLOCK = 0,       /* render_kernel: one wave per 8x8 block, whole rays (the latency path) */
    KIFS_KERNEL_GROUP = 1,       /* render_group_kernel: rays re-queued by a 256-thread workgroup */
    KIFS_KERNEL_WAVE = 2,        /* render_wave_kernel: rays re-queued, one wave per tile */
    KIFS_KERNEL_BUNNY_QUAD = 3,  /* render_bunny_quad_kernel: whole rays, four lanes per pixel */
    KIFS_KERNEL_BUNNY_COOP = 4   /* render_bunny_coop_kernel: rays re-queued, four waves per 64 rays */
};
int kifs_debug_last_kernel(kifs_ctx* ctx);
/* The bunny's throughput form in the context's latest launch: 0 = four lanes per ray with every weight in VGPRs, 1 = four
 * waves per 64 rays, 2 = four lanes per ray with layer 2 of the network in LDS; -1 = not a re-queued bunny launch. */
int kifs_debug_last_bunny_form(kifs_ctx* ctx);
/* Tuning hooks: read / replace the order in which workgroups take the tiles of the full
 * frame (a permutation of (tile_x | tile_y << 16)); the order only affects speed. */
int kifs_debug_get_tile_order(kifs_ctx* ctx, uint32_t* order, size_t max_count, size_t* count);
int kifs_debug_set_tile_order(kifs_ctx* ctx, const uint32_t* order, size_t count);
/* Per-wave records of the last counted render: 4 words per wave (wave = 4 * workgroup + wave
 * in workgroup, workgroups in dispatch order): total s_memtime ticks, ticks in the long-ray
 * loop, long-ray steps | (entries << 32), general steps.  Copies up to max_waves records. */
int kifs_debug_wave_records(kifs_ctx* ctx, unsigned long long* out, size_t max_waves,
                            size_t* n_waves);

/* ---- host model: the reference's scene -> uniform packing --------------------
 * C++ restatement of the caller side of the boundary so a harness without the
 * Rust host produces the same 156 bytes.
 *   kifs_host_screen   ScreenData::into_buffer_data        data.rs:66-81
 *   kifs_host_camera   CameraData::into_buffer_data        data.rs:91-129
 *   kifs_host_options  OptionsData::from(GuiData) + pack   data.rs:176-220, packed.rs:116-139
 *   kifs_host_rotate   GraphicState::rotate_camera         graphics.rs:280-302
 *   kifs_host_zoom     GraphicState::zoom_camera           graphics.rs:268-278
 */
typedef struct KifsGuiData { /* GuiData, data.rs:131-143 */
    uint32_t max_iterations;
    float max_distance;
    float epsilon;
    uint8_t fractal_color[3];    /* sRGB bytes */
    uint8_t background_color[3]; /* sRGB bytes */
    uint8_t is_heatmap;
    uint8_t _reserved;
    uint32_t fractal_group;
    uint32_t primitive_shape;
    float power;
    float constant[4];
} KifsGuiData;

typedef struct KifsCameraData { /* CameraData, data.rs:83-88 */
    float origin_distance;
    float min_distance;
    float phi;   /* angles.0 */
    float theta; /* angles.1 */
} KifsCameraData;

void kifs_host_gui_default(KifsGuiData* out);       /* GuiData::default, data.rs:145-160 */
void kifs_host_camera_default(KifsCameraData* out); /* CameraData::default, data.rs:105-113 */
int kifs_host_screen(uint32_t width, uint32_t height, KifsScreenUniform* out);
int kifs_host_camera(const KifsCameraData* camera, KifsCameraUniform* out);
int kifs_host_options(const KifsGuiData* gui, KifsOptionsUniform* out);
int kifs_host_rotate(KifsCameraData* camera, float delta_phi, float delta_theta);
int kifs_host_zoom(KifsCameraData* camera, float distance);
/* Mouse semantics of render.rs:255-270 / :239-250: degrees = -dx/10, +dy/10. */
int kifs_host_mouse_motion(KifsCameraData* camera, double dx, double dy);
float kifs_host_radians_from_degrees(float degrees); /* math.rs:429-431 */
void kifs_host_camera_matrix(const KifsCameraData* camera, float m_colmajor[9]);
void kifs_host_rotation_matrix(int axis, float radians, float m_colmajor[9]); /* math.rs:386-416 */
void kifs_host_mat3_mul(const float a[9], const float b[9], float out[9]);    /* math.rs:326-353 */
void kifs_host_mat3_vec(const float a[9], const float v[3], float out[3]);    /* math.rs:355-367 */

#ifdef __cplusplus
}
#endif
#endif /* KIFS_HIP_H */
