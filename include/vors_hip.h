/* vors_hip.h — C ABI of the MI355X-native direct RGB-D alignment hot path (libvors_hip.so).
 *
 * Drop-in boundary for the ONE hot path of mpizenberg/visual-odometry-rs ("vors"): pyramidal
 * inverse-compositional direct image alignment. Each entry point cites the reference interface it replaces
 * (paths relative to the reference repository root). The reference has no FFI today (100 % safe Rust); these
 * are the symbols a `-sys` style Rust binding would declare (see INTEGRATION.md for the Rust shim).
 *
 * Conventions
 *  - Plain C types only. Caller owns every buffer. No callbacks, no exceptions/aborts across the ABI.
 *  - Every function returns a vors_status (0 = ok, <0 = error); vors_last_error() gives the message of the last
 *    error on the calling thread.
 *  - Handles are NOT thread-safe (one thread per handle, mirroring `&mut self`); distinct handles are independent.
 *  - Images: gray u8 and depth u16 (TUM scale, 0 = unknown), rows x cols, `layout` = VORS_ROW_MAJOR (decoder order)
 *    or VORS_COL_MAJOR (nalgebra DMatrix::as_slice(), element (row,col) at col*rows+row).
 *  - Poses / models: 7 floats  tx ty tz qx qy qz qw  (nalgebra Isometry3<f32>: translation + unit quaternion
 *    coords [i,j,k,w]; same order as the TUM trajectory line, src/dataset/tum_rgbd.rs:76-86).
 *  - There is NO CPU fallback: every compute entry point fails with VORS_ERR_NO_DEVICE when no HIP device exists.
 */
#ifndef VORS_HIP_H
#define VORS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vors_status {
    VORS_OK = 0,
    VORS_ERR_INVALID_ARGUMENT = -1,
    VORS_ERR_NO_DEVICE = -2,        /* no HIP device / HIP runtime error */
    VORS_ERR_HIP = -3,
    VORS_ERR_PYRAMID_TOO_SHORT = -4, /* image too small for nb_levels: the reference panics (inverse_compositional.rs:124,183) */
    VORS_ERR_UNSUPPORTED = -5
} vors_status;

enum { VORS_ROW_MAJOR = 0, VORS_COL_MAJOR = 1 };

/* Candidate mask source. 0 reproduces the reference (candidates::coarse_to_fine, inverse_compositional.rs:120-125).
 * 1 = dense: all-true level-0 mask (extension for BASELINE config "dense candidates"; not in the reference).
 * 2 = DSO-style selection (src/core/candidates/dso.rs with the parameters of examples/candidates_dso.rs:40-59) as the
 *     level-0 mask source (BASELINE config 3; the reference's Tracker never uses it). Its random sub-sampling branch uses a
 *     counter-based hash instead of the reference's unseeded thread_rng. */
enum { VORS_CANDIDATES_COARSE_TO_FINE = 0, VORS_CANDIDATES_DENSE = 1, VORS_CANDIDATES_DSO = 2 };

/* Arithmetic of the LM evaluation (f32 in every mode; the integer stages, the inside test and the LM control flow do not depend on it).
 * The reference has no such choice (Config, inverse_compositional.rs:37-49): ZERO — what a zero-initialised vors_config, the Rust shim of
 * INTEGRATION.md, the vors_track CLI, host/tracker.hpp and vors_amd.Config select — is the mode that reproduces it.
 * 0 = REFERENCE: every per-point expression in the reference's evaluation order without FMA contraction AND the reference's summation:
 *     candidates in extract_z's column-major order (inverse_compositional.rs:260-279), `energy_sum += r * r`, `gradient += jac * r`,
 *     `hessian += hes` (lm_optimizer.rs:72-84,94-100) as sequential f32 multiply-then-add chains in that order, the optical-flow sum of
 *     the keyframe test likewise, sinf / cosf of se3::exp as glibc computes them. The device takes the oracle's LM path decision for
 *     decision: iteration counts equal at every level, poses BIT-IDENTICAL to the oracle's (tests/test_gpu_reference.py, bench.py
 *     `parity_reference`: 4096 / 4096 pairs per candidate mode, 64 / 64 sequences). Every candidate mode, Huber, the trackers and the
 *     operator level support it. Cost (640x480, 6 levels, 4096 pairs, MI355X): see `reference` in bench.py's line.
 * 1 = EXACT: the same per-point arithmetic (inverse depths, Jacobians, residuals bit-identical to the reference's), sums in the
 *     device's tree order. The LM loop's accept / stop comparisons (lm_optimizer.rs:144,179) are decided at ties, so a different order
 *     of the additions forks the loop in some pairs: measured tail beyond 1e-4 rad / 1e-4 m of the oracle — per 4096 pairs:
 *     coarse-to-fine 3, DSO 14, dense 0; per 64 sequences x 39 frames: 1-2 coarse-to-fine, 6-8 DSO (the oracle against its own
 *     f64-accumulation build shows the same counts: it is the order, not the precision).
 * 2 = FUSED: algebraically equivalent shorter forms (warp through the homography K R K^-1 plus _z K t with one hardware reciprocal,
 *     lerp-form bilinear interpolation, factored Jacobian; FMA) on the levels of MANY points; a level of at most 2500 points
 *     (VORS_FUSED_EXACT_POINTS) and every near-identity model is evaluated in the EXACT arithmetic. Per-point values agree
 *     to a few ulp; the tail beyond 1e-4 equals EXACT's (per 4096 pairs: coarse-to-fine 3, DSO 10, dense 0). The fastest mode: what
 *     bench.py's headline times. Pyramids of < 5 levels on large images exceed 1e-4 from summation order alone in modes 1 and 2. */
enum { VORS_ARITH_REFERENCE = 0, VORS_ARITH_EXACT = 1, VORS_ARITH_FUSED = 2 };

/* Per-pair tracking status. Mirrors `optimization_went_well` (inverse_compositional.rs:180,195-199,206-208). */
enum { VORS_TRACK_OK = 0, VORS_TRACK_OPTIMIZER_FAILED_POSE_KEPT = 1 };

/* Replaces `pub struct Config` (src/core/track/inverse_compositional.rs:37-49) with `Intrinsics`
 * (src/core/camera.rs:84-91) flattened. The last two fields are extensions; zero reproduces the reference. */
typedef struct vors_config {
    int32_t nb_levels;                 /* Config::nb_levels */
    int32_t candidates_diff_threshold; /* Config::candidates_diff_threshold (u16) */
    float depth_scale;                 /* Config::depth_scale (5000 for TUM, tum_rgbd.rs:15) */
    float cu, cv;                      /* Intrinsics::principal_point */
    float fu, fv;                      /* Intrinsics::focal */
    float skew;                        /* Intrinsics::skew */
    float idepth_variance;             /* Config::idepth_variance */
    int32_t candidates_mode;           /* extension: VORS_CANDIDATES_* */
    float huber_delta;                 /* extension: Huber threshold on |r| in grey levels; <= 0 = plain L2 (reference) */
    int32_t arithmetic;                /* extension: VORS_ARITH_* (how the per-point f32 expressions are evaluated) */
} vors_config;

#define VORS_MAX_LEVELS 8

/* Per-pair diagnostics of one track() (none of this exists in the reference API; it is what its commented-out
 * eprintln!s would show, lm_optimizer.rs:162,171,184, plus the counters the byte model of DESIGN.md needs). */
typedef struct vors_pair_stats {
    float lm_model[7];                 /* final lm_model (keyframe camera -> current camera), inverse_compositional.rs:177,193 */
    float optical_flow;                /* inverse_compositional.rs:213-221 */
    int32_t change_keyframe;           /* optical_flow >= 1.0 (inverse_compositional.rs:224) */
    int32_t nb_iter[VORS_MAX_LEVELS];  /* iterations returned by iterative_solve per level (optimizer.rs:57-70); 0 = level not run */
    int32_t n_points[VORS_MAX_LEVELS]; /* usable candidates per level (extract_z, inverse_compositional.rs:260-279) */
    float energy[VORS_MAX_LEVELS];     /* energy of the state kept at each level */
    int32_t nb_grad_evals[VORS_MAX_LEVELS]; /* of the nb_iter + 1 evaluations of a level, those for which the reference also forms g and H
                                             * (compute_eval_data, lm_optimizer.rs:90-107,147): the initial one and every accepted candidate */
} vors_pair_stats;

const char* vors_last_error(void);
/* Number of visible HIP devices (0 when none / no runtime). Never fails. */
int vors_device_count(void);
/* Peak shader clock (hipDeviceAttributeClockRate, kHz), compute units and device memory of a HIP device (all nullable). */
vors_status vors_device_info(int device, int* clock_khz, int* compute_units, uint64_t* memory_bytes);
/* Self-check of the DSO selector's gradient-magnitude root (dso_kernels.hip isqrt_floor_u16: the hardware square root + 0.001, truncated):
 * the number of arguments 0 .. 65535 for which it differs from floor(sqrt(n)) on this device — 0 on gfx950 (tests/test_gpu_parity.py). */
vors_status vors_selfcheck_isqrt(int* mismatches);
/* ABI version of this header: bump on any signature change. */
int vors_abi_version(void);  /* 2: vors_config.arithmetic, vors_pair_stats.nb_grad_evals, vors_batch_eval_level
                              * 3: vors_trackers_*, vors_synth_render_frames, vors_multi_rccl_version, vors_pipeline_*, vors_device_info,
                              *    vors_tracker_track_checked
                              * 4: VORS_ARITH_REFERENCE, vors_obs.arithmetic, vors_ref_sincos
                              * 5: VORS_ARITH_* renumbered: 0 = REFERENCE (a zero-initialised vors_config reproduces the reference), 1 = EXACT, 2 = FUSED
                              *    (+ vors_selfcheck_isqrt, added without a signature change) */

/* ------------------------------------------------------------------------------------------------------------
 * 1. Tracker: one sequence, host buffers.  Replaces
 *      Config::init(self, f64, &DMatrix<u16>, f64, DMatrix<u8>) -> Tracker      inverse_compositional.rs:74-100
 *      Tracker::track(&mut self, f64, &DMatrix<u16>, f64, DMatrix<u8>)           inverse_compositional.rs:170-240
 *      Tracker::current_frame(&self) -> (f64, Iso3)                              inverse_compositional.rs:243-248
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct vors_tracker vors_tracker;

vors_status vors_tracker_create(const vors_config* cfg, double depth_time, const uint16_t* depth, double img_time,
                                const uint8_t* gray, int rows, int cols, int layout, vors_tracker** out);
/* Returns VORS_OK and writes the VORS_TRACK_* status of this frame to *track_status (nullable). */
vors_status vors_tracker_track(vors_tracker* t, double depth_time, const uint16_t* depth, double img_time,
                               const uint8_t* gray, int* track_status);
/* Same with the frame's shape stated: fails with VORS_ERR_INVALID_ARGUMENT when rows x cols differ from the shape the tracker was created
 * with (vors_tracker_track trusts the caller's buffers to hold rows * cols elements, like the reference trusts its DMatrix arguments). */
vors_status vors_tracker_track_checked(vors_tracker* t, double depth_time, const uint16_t* depth, double img_time, const uint8_t* gray,
                                       int rows, int cols, int* track_status);
vors_status vors_tracker_current_frame(const vors_tracker* t, double* timestamp, float pose7[7]);
/* Diagnostics of the last track() and keyframe pose (not in the reference API). */
vors_status vors_tracker_last_stats(const vors_tracker* t, vors_pair_stats* stats);
vors_status vors_tracker_keyframe(const vors_tracker* t, double* timestamp, float pose7[7]);
void vors_tracker_destroy(vors_tracker* t);

/* ------------------------------------------------------------------------------------------------------------
 * 1b. N sequences advancing in lock-step, device resident — what a host that tracks many cameras / many replays calls. For every
 *     sequence s the calls below are exactly
 *        tracker[s] = cfg.init(depth0[s], gray0[s])        vors_trackers_init     (vors_track.rs:46)
 *        tracker[s].track(depth_k[s], gray_k[s])            vors_trackers_track    (vors_track.rs:54-59), k = 1, 2, ...
 *        tracker[s].current_frame()                         vors_trackers_current_frames / _state  (vors_track.rs:62)
 *     with the WHOLE state machine of Tracker::track on the device: the initial guess from the poses of the previous frame
 *     (inverse_compositional.rs:177), the LM loop, the pose composition (:203-208), the optical-flow keyframe test (:211-224) and — for
 *     exactly the sequences whose flow reached the threshold — the promotion of the current frame to keyframe (:227-239:
 *     precompute_multires_data on the current pyramid and THIS call's depth map, keyframe_pose <- current_frame_pose). No host round
 *     trip, no synchronisation: init and track only enqueue work on hip_stream. (A trackers-owned batch keeps no pointer into the
 *     caller's frames in the sparse modes: keyframe inspection through vors_batch_* is not available for it.) Results per sequence are bit-identical to a vors_tracker fed
 *     the same frames for handles of fewer than 512 sequences (tests/test_gpu_trackers.py); larger handles are scheduled differently
 *     (threads per sequence, evaluation rounds), which changes the ORDER of the f32 sums and with it the last bits, nothing else.
 *     Frames: DEVICE buffers, row-major, sequence s at offset s * rows * cols; they are read by the work this call enqueues and by
 *     nothing later (dense mode copies a promoted frame into the handle), so the caller may reuse them once the stream has passed
 *     the call. Timestamps stay with the caller: vors_trackers_state / _current_frames report, per sequence, the index of the frame
 *     that is its keyframe (0 = the init frame, k = the k-th track call).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct vors_trackers vors_trackers;
vors_status vors_trackers_create(const vors_config* cfg, int n_sequences, int rows, int cols, vors_trackers** out);
/* Same on an explicit HIP device (several GPUs = one handle per device, each with its own sequences: "replicas only", like batches). */
vors_status vors_trackers_create_on(int device, const vors_config* cfg, int n_sequences, int rows, int cols, vors_trackers** out);
int vors_trackers_count(const vors_trackers* t);
vors_status vors_trackers_init(vors_trackers* t, const uint8_t* d_gray, const uint16_t* d_depth, void* hip_stream);
vors_status vors_trackers_track(vors_trackers* t, const uint8_t* d_gray, const uint16_t* d_depth, void* hip_stream);
/* DEVICE pointers (valid for the life of the handle, contents valid once the stream has passed the last track call; all nullable):
 * current_frame_pose [n,7], keyframe_pose [n,7], VORS_TRACK_* status of the last track [n], keyframe frame index [n], diagnostics of the
 * last track [n]. */
vors_status vors_trackers_state(const vors_trackers* t, const float** d_current_poses7, const float** d_keyframe_poses7,
                                const int32_t** d_status, const int32_t** d_keyframe_index, const vors_pair_stats** d_stats);
/* Same to HOST buffers (nullable each); synchronises hip_stream. */
vors_status vors_trackers_current_frames(vors_trackers* t, float* poses7, int32_t* status, int32_t* keyframe_index, void* hip_stream);
/* Diagnostics of the last track of every sequence to a HOST buffer [n]; synchronises hip_stream. */
vors_status vors_trackers_last_stats(vors_trackers* t, vors_pair_stats* stats, void* hip_stream);
/* Stage timing as for a batch handle (stages 1 = keyframe promotion, 2 = current pyramid, 3 = LM). */
vors_status vors_trackers_enable_kernel_timing(vors_trackers* t, int ring);
vors_status vors_trackers_kernel_times(vors_trackers* t, int stage, float* ms_out, int capacity, int* n_out);
void vors_trackers_destroy(vors_trackers* t);

/* ------------------------------------------------------------------------------------------------------------
 * 2. Batch of independent frame pairs — the data-parallel hot path. For each pair p:
 *      tracker = cfg.init(kf_depth[p], kf_gray[p]); tracker.track(cur_gray[p]); pose[p] = tracker.current_frame()
 *    i.e. vors_track.rs:46-62 for a 2-frame sequence, for n_pairs sequences at once.
 *    Images of pair p start at offset p*rows*cols. prev_poses7 (nullable) = current_frame_pose before track()
 *    (identity in the reference's init; the initial guess is its inverse, inverse_compositional.rs:177).
 * ---------------------------------------------------------------------------------------------------------- */
/* Host buffers (copies in and out; PCIe-inclusive). */
vors_status vors_track_pairs(const vors_config* cfg, int n_pairs, const uint8_t* kf_gray, const uint16_t* kf_depth,
                             const uint8_t* cur_gray, int rows, int cols, int layout, const float* prev_poses7,
                             float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats /* nullable, n_pairs */);

/* Device-resident engine: buffers are DEVICE pointers (row-major), work is enqueued on `hip_stream`
 * (a hipStream_t passed as void*; NULL = default stream) and NOT synchronised: outputs are valid once the stream
 * reaches this point. Workspaces are allocated once at create() for up to max_pairs pairs. */
typedef struct vors_batch vors_batch;
/* Scheduling knobs (environment, read at create() — except VORS_DSO_SCAN, VORS_DSO_PLANES and VORS_PYRAMID_FUSED, which are read ONCE PER
 * PROCESS, at the first keyframe stage / pyramid; results stay within the stated tolerance whatever their value — they only
 * change how the same arithmetic is spread over the chip; tests/test_gpu_parity.py covers the variants):
 *   VORS_LM_BLOCK=64..1024       threads per frame pair in the per-pair LM kernel (default by batch size and mode)
 *   VORS_LM_SPLIT=0              dense mode: one per-pair kernel for all levels instead of evaluation rounds
 *   VORS_LM_SPLIT_LEVELS=n       dense mode: the n finest levels are solved by evaluation rounds (default: levels of >= 64 Ki pixels)
 *   VORS_LM_SPLIT_ROUNDS=n       rounds launched before per-pair workgroups finish the stragglers (default by batch size and by the number
 *                                of levels solved by rounds; the defaults of all these knobs come from tools/speed_sweep.py)
 *   VORS_LM_CHUNKS=n             partial-sum chunks per pair of a level-0 evaluation (default by batch size)
 *   VORS_KF_R=1|2|4|8            tree roots per wavefront in the coarse-to-fine keyframe kernel (default 4)
 *   VORS_NO_FASTDIV=1            plain IEEE division by the focal lengths (the verified 3-instruction form is bit-identical)
 *   VORS_DSO_PLANES=1            DSO mode: keyframe records through per-level inverse-depth planes instead of the sorted pick list
 *                                (same candidates and values; the lists then come out in raster instead of Morton order)
 *   VORS_DSO_SCAN=1              DSO mode: the usable picks from a pass over the stamp plane instead of the selection rounds' own list
 *                                (identical lists)
 *   VORS_DSO_ROUNDS_THREADS=n    DSO mode: threads per pair in the selection-rounds kernel, a multiple of 64 (default 768 from 512 pairs on,
 *                                else 1024: profiles/r04_dso_rounds_threads.log; read per launch)
 *   VORS_DSO_SORT=bitonic        DSO mode: the pick list ordered by round 3's bitonic network instead of the bucket sort (identical lists;
 *                                read per launch)
 *   VORS_REF_SORT_REGCAP=n       REFERENCE arithmetic: lists longer than n records take the multi-pass form of the column-major sort
 *                                (default 4096 = what the register-resident form holds; identical lists; read per launch)
 *   VORS_DSO_RECORDS_THREADS=512|1024  DSO mode: threads per pair in the sparse records kernel (default 512 from 512 pairs on, else 1024)
 *   VORS_PYRAMID_FUSED=0         mean pyramid one level per launch instead of up to five halvings in one (bit-identical; read once per process)
 *   VORS_IDEPTH_LEVEL12=1        dense mode: inverse-depth levels 1-2 in one pass + a halving launch instead of levels 1-3 in one (bit-identical)
 *   VORS_FUSED_EXACT_POINTS=n    FUSED arithmetic: levels of at most n points take (u, v) from the reference's own warp chain (candidate
 *                                lists) or run the whole EXACT evaluation (dense pixel levels); default 2500, re-derived in round 4 on
 *                                six draws of 64 sequences + 4096 pairs (profiles/r04_parity_sequences.md); 0 = the round-2 behaviour
 *   VORS_FUSED_SMALL=exact       FUSED arithmetic, candidate lists: the whole EXACT evaluation on those levels (round 3's rule) */
vors_status vors_batch_create(const vors_config* cfg, int max_pairs, int rows, int cols, vors_batch** out);
/* Same on an explicit HIP device (vors_batch_create = the calling thread's current device). The handle remembers its device: every
 * entry point switches to it for the call and restores the caller's current device; a hip_stream of another device is rejected with
 * VORS_ERR_INVALID_ARGUMENT. Device buffers passed to the handle must live on that device. */
vors_status vors_batch_create_on(int device, const vors_config* cfg, int max_pairs, int rows, int cols, vors_batch** out);
vors_status vors_batch_device(const vors_batch* b, int* device);
vors_status vors_batch_track_pairs(vors_batch* b, int n_pairs, const uint8_t* d_kf_gray, const uint16_t* d_kf_depth,
                                   const uint8_t* d_cur_gray, const float* d_prev_poses7 /* nullable */,
                                   float* d_out_poses7, int32_t* d_out_status,
                                   vors_pair_stats* d_out_stats /* nullable */, void* hip_stream);
/* The three stages of the above, separately (keyframe data persists in the handle between calls):
 *   prepare_keyframes = mean_pyramid + precompute_multires_data        inverse_compositional.rs:83-85,105-161
 *   track_current     = mean_pyramid + coarse->fine LM + keyframe test  inverse_compositional.rs:177-224
 * LIFETIME CONTRACT (zero copy, like the reference, which MOVES the image into the pyramid as level 0, multires.rs:14-15):
 * the handle keeps the caller's POINTERS to level 0 and to the depth map, not copies.
 *   - VORS_CANDIDATES_DENSE: d_kf_gray and d_kf_depth are re-read by every track_current (points are recomputed from them on the
 *     fly); they must stay allocated and unchanged until the next prepare_keyframes on this handle or its destruction.
 *   - coarse-to-fine / DSO: they are read during prepare_keyframes only (everything later needs is in the handle's records);
 *     vors_batch_get_keyframe_image(level 0) still reads d_kf_gray.
 * track_current accepts n_pairs <= the n_pairs of the last prepare_keyframes (more would read keyframe slots never prepared) and
 * fails with VORS_ERR_INVALID_ARGUMENT otherwise. */
vors_status vors_batch_prepare_keyframes(vors_batch* b, int n_pairs, const uint8_t* d_kf_gray, const uint16_t* d_kf_depth,
                                         void* hip_stream);
vors_status vors_batch_track_current(vors_batch* b, int n_pairs, const uint8_t* d_cur_gray, const float* d_prev_poses7,
                                     float* d_out_poses7, int32_t* d_out_status, vors_pair_stats* d_out_stats,
                                     void* hip_stream);
/* Bytes of device workspace held by the handle. */
vors_status vors_batch_workspace_bytes(const vors_batch* b, uint64_t* bytes);
/* Per-stage kernel timing with HIP events recorded on hip_stream (non-blocking during a step).
 * ring = number of most recent steps kept per stage (0 disables). Stages: 0 keyframe pyramid, 1 keyframe
 * precompute, 2 current pyramid, 3 LM kernel (the dominant one). kernel_times() synchronises on the events it reads
 * and returns the durations (ms) of the last min(steps, ring) steps, oldest first. */
vors_status vors_batch_enable_kernel_timing(vors_batch* b, int ring);
vors_status vors_batch_kernel_times(vors_batch* b, int stage, float* ms_out, int capacity, int* n_out);
/* Most recent step only; pyramid_ms = keyframe + current pyramids; a value < 0 = not measured. */
vors_status vors_batch_last_kernel_ms(vors_batch* b, float* lm_ms, float* keyframe_ms, float* pyramid_ms);
void vors_batch_destroy(vors_batch* b);

/* Throughput mode for a continuous feed of independent batches: a ring of `depth` batch handles, each on its own internal stream.
 * The tail of a step leaves the GPU partly idle (dependent straggler rounds of the dense LM stage, the last workgroups of the per-pair
 * kernel), its body VALU- or bandwidth-bound: with consecutive steps on different streams the GPU fills one with the other. Measured on one
 * MI355X (round 6, profiles/r06_stage_times_512_vs_4096.log; bench.py `pipelined` reports the current figures), depth 3 — the optimum; 2 is
 * within 10 %, 4 and 6 are no better: 4096-pair steps +3 % (dense FUSED), +10 % (coarse-to-fine, DSO), +25-33 % (dense in the default REFERENCE
 * arithmetic: the straggler tail of its one-wavefront-per-pair kernel); 512-pair steps — BASELINE config 4's share per GPU, which alone leave
 * most of the chip idle — +20-45 %: the 4096 / 512 step-time ratio goes from 4.3-5.5 to 6.3-7.0 (FUSED) and 5.0-5.3 (REFERENCE). Every step is a
 * plain vors_batch_track_pairs — same results bit for bit; the price is `depth` workspaces. NOTE: the HIP runtime maps a process's streams onto
 * GPU_MAX_HW_QUEUES hardware queues (ROCm default 4) in creation order; a process that holds other streams besides the ring's (torch, a
 * dense handle's side lane, a second ring) should start with GPU_MAX_HW_QUEUES=8 in its environment, or two slots can share one queue and
 * run one after the other (measured: 0.61 instead of 0.46 ms per 512-pair step; bench.py sets it).
 *   submit: the slot's stream first waits for everything enqueued on hip_stream so far (the inputs, and earlier readers of the output
 *           buffers), then runs the step; nothing is synchronised. Buffers as for vors_batch_track_pairs, and they must stay valid
 *           until the step has completed. *ticket (nullable) identifies the step.
 *   wait:   host_sync = 0: hip_stream waits for the step (its outputs are then ordered on hip_stream); host_sync != 0: the calling
 *           thread blocks until the step has completed.
 *   drain:  the same for every step submitted so far.
 * device < 0 = the calling thread's current device. */
typedef struct vors_pipeline vors_pipeline;
vors_status vors_pipeline_create(int device, const vors_config* cfg, int depth, int max_pairs, int rows, int cols, vors_pipeline** out);
vors_status vors_pipeline_submit(vors_pipeline* p, int n_pairs, const uint8_t* d_kf_gray, const uint16_t* d_kf_depth,
                                 const uint8_t* d_cur_gray, const float* d_prev_poses7 /* nullable */, float* d_out_poses7,
                                 int32_t* d_out_status, vors_pair_stats* d_out_stats /* nullable */, void* hip_stream, int64_t* ticket);
vors_status vors_pipeline_wait(vors_pipeline* p, int64_t ticket, void* hip_stream, int host_sync);
vors_status vors_pipeline_drain(vors_pipeline* p, void* hip_stream, int host_sync);
void vors_pipeline_destroy(vors_pipeline* p);

/* Inspection of the keyframe data held by a batch handle (device -> host copies; tests and debugging).
 * level image (mean_pyramid, multires.rs:21-31), row-major rows_l x cols_l: */
vors_status vors_batch_get_keyframe_image(vors_batch* b, int pair, int level, uint8_t* out, int* rows, int* cols);
vors_status vors_batch_get_current_image(vors_batch* b, int pair, int level, uint8_t* out, int* rows, int* cols);
/* usable candidates of a level (extract_z + warp_jacobians, inverse_compositional.rs:260-341): up to `capacity`
 * points, xy int32[2n], idepth f32[n], jac f32[6n], tmpl u8[n] (all nullable). Order is the device slot order,
 * NOT the reference's column-major order: sort by (x, y) to compare. *n = number of usable candidates. */
vors_status vors_batch_get_points(vors_batch* b, int pair, int level, int capacity, int32_t* xy, float* idepth, float* jac,
                                  uint8_t* tmpl, int* n);

/* ------------------------------------------------------------------------------------------------------------
 * 2b. Several GPUs from ONE process (no torch, no MPI): what a Rust host calls for BASELINE config 4 (4096 pairs over 8 MI355X).
 *     Frame pairs are independent (a Tracker is self-contained: inverse_compositional.rs:31-34), so pairs shard by contiguous blocks —
 *     pair i lives on device floor(i / ceil(n / G)) — each device runs its own vors_batch on its own stream with no data-path exchange,
 *     and the ONLY collective is one all-gather of 8 f32 per pair (pose 7 + status) over RCCL / xGMI (ncclAllGather; 16 KiB per GPU for
 *     4096 pairs on 8 GPUs), after which every device holds all results. RCCL is loaded at run time (librccl.so) and only when the
 *     handle spans more than one device.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct vors_multi vors_multi;
/* n_devices <= 0: all visible devices; device_ids NULL: 0 .. n_devices-1. */
vors_status vors_multi_create(const vors_config* cfg, int n_devices, const int* device_ids, int max_pairs_per_device, int rows, int cols,
                              vors_multi** out);
int vors_multi_device_count(const vors_multi* m);
/* Version code of the RCCL the handle bound at run time (ncclGetVersion), 0 when the handle spans one device (RCCL not loaded). */
int vors_multi_rccl_version(const vors_multi* m);
/* Block of pairs owned by device slot k for a batch of n_pairs_total: [*first, *first + *count). */
vors_status vors_multi_shard(const vors_multi* m, int n_pairs_total, int k, int* first, int* count);
/* Device-resident: d_*[k] = device slot k's block (row-major images of ITS pairs, allocated on that device). Runs all devices
 * concurrently, gathers, and returns poses (n_pairs_total x 7) and statuses on the host. Synchronous. */
vors_status vors_multi_track_pairs(vors_multi* m, int n_pairs_total, const uint8_t* const* d_kf_gray, const uint16_t* const* d_kf_depth,
                                   const uint8_t* const* d_cur_gray, float* out_poses7, int32_t* out_status);
/* Host buffers (row-major, all pairs contiguous): uploads each block to its device first (PCIe-inclusive). */
vors_status vors_multi_track_pairs_host(vors_multi* m, int n_pairs_total, const uint8_t* kf_gray, const uint16_t* kf_depth,
                                        const uint8_t* cur_gray, float* out_poses7, int32_t* out_status);
void vors_multi_destroy(vors_multi* m);

/* One evaluation — eval_energy + compute_eval_data (lm_optimizer.rs:68-107) — of level `level` of pair `pair` as the handle holds it
 * after prepare_keyframes + track_current, at an explicit model (HOST pointer, 7 floats), in the given VORS_ARITH_* mode whatever the
 * handle's own: sums29 (HOST) = sum r^2 (Huber loss with huber_delta), n_inside, g[6], H upper triangle row-wise [21]. Synchronises.
 * This is the operator-level window on the tracker's own point sources; tests compare EXACT and FUSED through it. */
vors_status vors_batch_eval_level(vors_batch* b, int pair, int level, const float model7[7], int arithmetic, float sums29[29]);

/* ------------------------------------------------------------------------------------------------------------
 * 3. Operator level — the optimizer trait's pieces for one pyramid level.  Replaces, for
 *    `impl optimizer::State<Obs, EvalState, Iso3, String> for LMOptimizerState` (lm_optimizer.rs:111-193):
 *      vors_lm_eval  = eval_energy + compute_eval_data at a model          lm_optimizer.rs:68-107 (the body of init/eval)
 *      vors_lm_step  = step()                                              lm_optimizer.rs:123-136
 *      vors_lm_solve = State::iterative_solve(&obs, model)                 src/math/optimizer.rs:57-70
 *    vors_obs replaces `pub struct Obs<'a>` (lm_optimizer.rs:43-58); hessians are not passed (J J^T is recomputed).
 *    Host buffers, row-major images.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct vors_obs {
    float cu, cv, fu, fv, skew; /* Obs::intrinsics (of this level) */
    int32_t rows, cols;         /* shape of template and image */
    const uint8_t* template_;   /* Obs::template (keyframe image of this level) */
    const uint8_t* image;       /* Obs::image (current image of this level) */
    int32_t n;                  /* number of candidates */
    const int32_t* coordinates; /* Obs::coordinates: (x, y) pairs, int32[2n] */
    const float* _z_candidates; /* Obs::_z_candidates: inverse depths, f32[n] */
    const float* jacobians;     /* Obs::jacobians: f32[6n] */
    float huber_delta;          /* extension, <= 0 = reference */
    int32_t arithmetic;         /* VORS_ARITH_EXACT (sums in the device's tree order) or VORS_ARITH_REFERENCE: sequential f32 sums in the
                                 * order of `coordinates` — the reference's eval on this Obs, bit for bit (lm_optimizer.rs:68-107) */
} vors_obs;

vors_status vors_lm_eval(const vors_obs* obs, const float model7[7], float* energy, int32_t* n_inside, float g[6],
                         float H[36] /* row-major 6x6 */, float* residuals /* nullable f32[n], NaN = outside */);
/* Host-only arithmetic (6x6 Cholesky + se3::exp + compose + renormalise). *chol_ok = 0 mirrors
 * Err("Error at Cholesky decomposition of hessian"). */
vors_status vors_lm_step(const float H[36], const float g[6], const float model7[7], float lm_coef, float out_model7[7],
                         int* chol_ok);
/* *solve_status: VORS_TRACK_OK or VORS_TRACK_OPTIMIZER_FAILED_POSE_KEPT (step error). */
vors_status vors_lm_solve(const vors_obs* obs, const float model7[7], float out_model7[7], int32_t* nb_iter, float* energy,
                          float* lm_coef, int* solve_status);

/* ------------------------------------------------------------------------------------------------------------
 * 4. Lie algebra helpers (host arithmetic; src/math/se3.rs:65-129, src/math/so3.rs:62-99). API parity only:
 *    the tracker itself only uses se3::exp.
 * ---------------------------------------------------------------------------------------------------------- */
void vors_se3_exp(const float xi[6], float out_iso7[7]);
void vors_se3_log(const float iso7[7], float out_xi[6]);
/* sinf / cosf as se3::exp evaluates them here, host and device alike (csrc/lie.h ref_sinf / ref_cosf: glibc's algorithm restated; equal to
 * the platform's sinf / cosf for every f32 in [0, 4), which tests/test_oracle_kat.py checks exhaustively). Outputs nullable. */
void vors_ref_sincos(const float* x, int n, float* sin_out, float* cos_out);
void vors_so3_exp(const float w[3], float out_q4[4]);
void vors_so3_log(const float q4[4], float out_w[3]);
void vors_iso_mul(const float a7[7], const float b7[7], float out7[7]);
void vors_iso_inverse(const float a7[7], float out7[7]);

/* ------------------------------------------------------------------------------------------------------------
 * 5. Synthetic scene renderer on the device (bench/test tooling). Renders, for pair i in
 *    [0, n_pairs), the keyframe (identity) and the current frame (exp(xi(seed0+i))) of the textured-plane scene
 *    into DEVICE buffers, and the ground-truth models (keyframe->current) into d_gt_models7 (nullable).
 * ---------------------------------------------------------------------------------------------------------- */
vors_status vors_synth_render_pairs(uint64_t seed0, int n_pairs, int rows, int cols, const double cam5[5],
                                    double motion_scale, int invalid_percent, uint8_t* d_kf_gray, uint16_t* d_kf_depth,
                                    uint8_t* d_cur_gray, uint16_t* d_cur_depth /* nullable */, float* d_gt_models7,
                                    void* hip_stream);

/* Frames of the same scene at explicit twists (sequence tooling): frame f = scene seeds[f], depth-dropout salt salts[f], camera at
 * exp(xi6[6 f .. 6 f + 5]) (twist (v, w), keyframe -> camera); host tables, DEVICE images [n_frames, rows, cols]. Synchronises. */
vors_status vors_synth_render_frames(int n_frames, const uint64_t* seeds, const uint64_t* salts, const double* xi6, int rows, int cols,
                                     const double cam5[5], int invalid_percent, uint8_t* d_gray, uint16_t* d_depth, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* VORS_HIP_H */
