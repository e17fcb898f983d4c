// Internal interface between the HIP kernels (kernels_*.hip) and the host engine / C ABI (capi.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vors_hip.h"
#include "lie.h"

namespace vors {

// Geometry of one pyramid level.
struct LevelGeom {
    int rows, cols;     // image shape at this level (floor halving, multires.rs:67-88)
    int img_off;        // byte offset of this level inside the per-pair "upper levels" buffer (levels >= 1); -1 for level 0
    int n_slots;        // candidate slots at this level (sparse: roots * 2^(L-1-l); dense: rows*cols)
    int slot_off;       // offset of this level's slots inside the per-pair planes (dense: IZ/V planes hold levels >= 1 only; -1 for level 0)
    Intr k;             // intrinsics of this level (camera.rs:106-123)
    FastDiv fu, fv;     // verified fast exact division by the focal lengths of this level (lie.h)
    // level constants of the FUSED arithmetic, formed once on the host in f64 (the kernels would otherwise redo these divisions in
    // every evaluation): 1 / fu, 1 / fv as f64 (for H = K R K^-1), and as f32 1 / fu, 1 / fv, s / (fu fv) (Jacobian)
    double inv_fu_d, inv_fv_d;
    float inv_fu, inv_fv, s_fuv;
};

// Everything a kernel needs to know about the batch layout. Passed by value.
struct Geom {
    int L;              // nb_levels
    int mode;           // VORS_CANDIDATES_*
    int arith;          // VORS_ARITH_*
    int thresh;         // candidates_diff_threshold (u16)
    float depth_scale, idepth_variance, huber_delta;
    int S0;             // rows*cols of level 0 (pair stride of level-0 images and depth maps)
    int upper_stride;   // bytes per pair of levels 1..L-1
    int slots_total;    // record slots per pair (all levels)
    int root_rows, root_cols;  // shape of the coarsest level (= roots of the selection quad-trees)
    int fast_idepth;    // scale / depth through idepth_of<true> (kernels.hip): proven bit-identical to the division for all 65535 depths
    // FUSED arithmetic: a level of at most this many points (dense: pixels of the level; sparse modes: candidates of the pair at the level)
    // is evaluated in the EXACT arithmetic (lm_kernels.hip lm_track_kernel); fused_exact_step != 0 also takes its step() with lm_step
    int fused_exact_points, fused_exact_step;
    int fused_small_warp;  // candidate-list modes: such a level takes only (u, v) from the reference's warp chain (lm_kernels.hip fused_stage_b<XW>)
    int ref_rank;       // REFERENCE arithmetic, coarse-to-fine: rank the keyframe kernel's staged regions directly (lm_reference.hip); resolved ONCE per
                        // handle from VORS_REF_RANK (development knob) so that the keyframe stage and the sort take the same decision
    int ref_inflight_x2;  // REFERENCE arithmetic, candidate lists: twice the number of steps whose LM stages share the chip (2 = a step on its own;
                          // 3 = a slot of a vors_pipeline ring: the LM stage is ~2/3 of a step, so about 1.5 of them overlap). The workgroup size of
                          // the workgroup-per-pair kernel is chosen for n_pairs x this / 2 RESIDENT pairs (lm_reference.hip refc_waves_per_pair): a
                          // ring wants the thinner workgroups whose LDS lets the LM kernels of consecutive steps share a CU. The sums do not depend on it
    int wide_loads_ok;  // set per launch: the caller's buffers are 16-byte aligned, so the dense quad source may use wide loads
    // Masked launches of the keyframe stage (vors_trackers: per-sequence keyframe promotion on the device). When sel_list is set, index k
    // of a kernel's pair dimension addresses pair sel_list[k] for k < *sel_count and nothing beyond (device_common.h select_pair).
    const int* sel_list;
    const int* sel_count;
    LevelGeom lv[VORS_MAX_LEVELS];
};

// Candidate record of the coarse-to-fine and generic-mask (DSO) modes: 12 bytes per point, everything else (the back-projected point,
// the warp Jacobian) is recomputed by the LM kernel from it — like the reference's Obs (lm_optimizer.rs:43-58) minus the Jacobians and
// Hessians it precomputes. Per pair and level the records are COMPACT: slots [0, n_used) hold points, nothing else is ever read.
struct SlimRec {
    uint32_t xy;  // x | y << 16
    float iz;     // inverse depth
    uint32_t tg;  // template grey level | (gx & 0x3ff) << 8 | (gy & 0x3ff) << 18   (integer gradients of the level, |g| <= 255)
};
__host__ __device__ inline uint32_t slim_pack_tg(int tmpl, int gx, int gy) {
    return (uint32_t)tmpl | (((uint32_t)gx & 0x3ffu) << 8) | (((uint32_t)gy & 0x3ffu) << 18);
}
__host__ __device__ inline int slim_gx(uint32_t tg) { return ((int)(tg << 14)) >> 22; }
__host__ __device__ inline int slim_gy(uint32_t tg) { return ((int)(tg << 4)) >> 22; }

// Operator-level record planes (explicit observations, vors_lm_eval / vors_lm_solve; also the inspection output of
// vors_batch_get_points). Structure of arrays, one entry per slot:
//   A = (X, Y, Z, tmpl)  back-projected keyframe point (camera.rs:135-140) + template grey level; tmpl < 0 = empty slot
//   B = (J0, J1, J2, J3) C = (J4, J5)   warp Jacobian (inverse_compositional.rs:313-341)
//   XY = x | y << 16     pixel coordinates (keyframe test, inspection)
//   IZ = inverse depth   (inspection only)
// REFERENCE arithmetic, dense mode: everything the LM kernel reads, in COLUMN-MAJOR order. The reference enumerates the pixels of a level
// column by column (DMatrix order, inverse_compositional.rs:260-279), so consecutive points of its order are consecutive ROWS: on the
// row-major planes a wavefront's 64 points touch 64 cache lines per load, here they touch one or two — and point i of the enumeration is
// element i of the level.
//   recs      per pixel of every level, 8 bytes: (inverse depth f32 — scale / depth at level 0 (inverse_depth.rs:24-29), the fused value
//             above, NaN = Unknown — and template | gx | gy packed like SlimRec.tg): the level's Obs entry minus what the LM kernel
//             recomputes (lm_optimizer.rs:43-58), written once per keyframe by lm_reference.hip ref_dense_records_kernel.
//             Pair stride S0 + upper_stride entries; level 0 at 0, level l >= 1 at S0 + img_off.
//   cur0/curu the current frame's pyramid, column-major level by level (laid out like the row-major pyramid).
struct RefDensePlanes {
    uint2* recs;
    int* n_valid;  // [pair][VORS_MAX_LEVELS] pixels with a known inverse depth per level, counted while the records are written (diagnostics)
    uint8_t* cur0;
    uint8_t* curu;
};

// REFERENCE arithmetic, large batches: hand-over of the pairs still iterating when most of the batch has finished (lm_reference.hip).
// One wavefront per pair fills the chip while every pair is alive; the pairs with the most iterations then run alone, one wavefront on a
// SIMD each, at a third of the chip's rate. Once `after` pairs are done, a wavefront that is about to start another evaluation saves its LM
// state instead and queues its pair; a second launch gives each queued pair a whole workgroup (the same chains, bit for bit).
struct RefResume {  // the state of optimizer::State::iterative_solve between two evaluations (lm_reference.hip RefLm) + the level
    float cur_model[7], cand[7];
    float kept[28];
    float cur_energy, lm_coef;
    int nb_iter, n_full, lvl;
};
struct RefHandoff {
    RefResume* state;  // [pairs]
    int* list;         // [pairs] queued pairs
    int* counters;     // [0] pairs queued, [1] pairs finished by the first launch
};

struct Records {
    float4* A;
    float4* B;
    float2* C;
    uint32_t* XY;
    float* IZ;
    float* V;  // dense mode only: fused weight ("variance") plane, < 0 = Unknown
    const float2* LUT;  // dense mode only: depth u16 -> (scale / depth, 1 / (scale / depth)), exact
    int* n_used;  // [pair][VORS_MAX_LEVELS]; sparse modes: slots in use per level (compact, no holes); dense mode: usable points
                  // per level counted by the keyframe stage (level 0 only when L >= 2)
    SlimRec* S;         // coarse-to-fine / generic-mask modes: compact candidate lists, pair stride slots_total, level offset slot_off
    SlimRec* stage;     // coarse-to-fine mode: the keyframe kernel's slot grid (compacted per wavefront region), input of the per-pair compaction
    int* region_cnt;    // coarse-to-fine mode: [pair][level][region] points per wavefront region
    int n_regions;      //   regions per level (= wavefronts of the keyframe kernel per pair)
    int kf_r;           //   roots per wavefront region
    RefDensePlanes dense_t;  // REFERENCE arithmetic, dense mode (all null otherwise)
    RefHandoff handoff;      // REFERENCE arithmetic (null otherwise)
    SlimRec* sort_tmp;  // REFERENCE arithmetic, sparse modes: scratch of the column-major sort (lm_reference.hip), laid out like S; the
                        // coarse-to-fine mode lends its staging grid, which is free once the regions have been compacted
};

// Workspace of the DSO-style selector (dso_kernels.hip), all per pair.
struct DsoState {
    int base_size, iterations_left, done, random_keep, count, final_round;
    int epoch;  // 1 .. 15: the selection this pair's pick stamps belong to (dso_kernels.hip: the stamp plane is cleared when it wraps, not per keyframe)
};
struct DsoWs {
    uint8_t* gmag;      // [S0] gradient magnitude (<= 180)
    uint16_t* median;   // [n_regions]
    uint16_t* thresh;   // [n_regions]
    uint8_t* max_g;     // [max_stride] block maxima of the 3 levels, concatenated
    uint32_t* max_pos;  // [max_stride] their pixel positions (row * cols + col)
    uint8_t* mask1;     // [mask_stride] block masks of levels 1 and 2 (+ the discarded mask after the last level)
    uint8_t* picked;    // [S0] 0 or (round << 2 | level + 1) of the pick
    DsoState* state;    // [1]
    uint32_t* pick_list;  // [list_cap] pixel positions picked in the round in progress (the last round executed = the final one)
    int n_regions, max_stride, mask_stride, list_cap;
};
// Evaluation rounds on the finest levels (dense mode, lm_kernels.hip "split" path): the coarse levels run in the per-pair
// kernel; then every ROUND is one launch that evaluates the energy of each still-active pair at ITS current level and
// candidate over (active pairs x chunks of the image), followed by a tiny per-pair launch that gives the verdict, takes the
// next step or moves the pair to the next level. All CUs stay busy whatever the pairs' iteration counts; the few pairs still
// iterating after `rounds` rounds finish inside the final per-pair kernel.
struct LmSplitState {  // per pair
    float entry[7];    // model on entry to the level: restored when step() fails (the level's progress is discarded)
    float model[7];    // kept model (the level's running estimate)
    float cand[7];     // candidate under evaluation
    float sums[32];    // sums of the kept state (energy sum, n, g[6], H upper triangle[21])
    float cur_energy, lm_coef;
    int nb_iter;
    int n_full;        // initial evaluation + accepted candidates of this level so far (statistics: vors_pair_stats.nb_grad_evals)
    int lvl;           // level being solved
    int phase;         // 0 init evaluation pending (full), 1 candidate's energy pending, 4 accepted candidate's g and H pending (full),
                       // 2 all levels finished
    int went_well;     // 0: a level failed (the pair skips the remaining levels)
    // FUSED arithmetic: the evaluation context (H = K R K^-1, K t: lm_kernels.hip FusedCtx, 21 floats) of the model the NEXT round
    // evaluates at level `lvl` — formed once per pair and round by whoever sets that model (the step kernel, the coarse-level kernel's
    // hand-over) instead of by every thread of every evaluation workgroup — and whether that model is the near-identity case that
    // runs in the exact arithmetic.
    float fctx[21];
    int fctx_exact;
};
#define VORS_SPLIT_MAX_ROUNDS 62
struct LmSplitWs {
    LmSplitState* state;  // [pairs]
    float* partials;      // [pairs][chunks][32]
    // Active pairs of a round, ping-pong per round. One array holds both kinds: pairs due a FULL evaluation (energy, g, H) fill it
    // from the front, pairs due an ENERGY-only evaluation of a candidate from the back — each kind has its own launch.
    int* list[2];         // [cap]
    int* count;           // [2][VORS_SPLIT_MAX_ROUNDS + 2]: full-kind / energy-kind pairs per round
    int cap;              // capacity of a list = pairs of the handle
    int chunks;           // partial-sum slots per pair = most chunks a pair is cut into; 0 = split path disabled
    int chunks0;          // chunks per pair at level 0 in this launch (level l: chunks0 >> 2l), <= chunks
    int n_split;          // levels 0 .. n_split-1 are solved this way
    int rounds;
    // SIDE LANE (large dense batches, two levels solved by rounds): the few pairs still iterating at level 1 once everybody else has moved on
    // to level 0 would get ONE cheap evaluation per round while each round lasts milliseconds (level-0 evaluations of 4096 pairs), and then
    // keep ~20 more rounds alive on their own. After round `side_round` the step kernel hands them to a per-pair kernel on a second stream
    // (one 1024-thread workgroup each, all their remaining level-1 iterations back to back, concurrent with the level-0 rounds of the
    // others); they join the rounds again at level 0 (`join_list`, merged into a later round's list).
    int side_round;       // -1: off
    int* side_list;       // [cap] pairs handed to the side lane;  count[SPLIT_SIDE_COUNT] of them
    int* join_list;       // [cap] pairs that finished level 1 there; count[SPLIT_JOIN_COUNT] of them
    hipStream_t side_stream;
    hipEvent_t ev_fork, ev_join;
};
#define SPLIT_SIDE_COUNT (2 * (VORS_SPLIT_MAX_ROUNDS + 2))
#define SPLIT_JOIN_COUNT (2 * (VORS_SPLIT_MAX_ROUNDS + 2) + 1)
#define SPLIT_COUNT_INTS (2 * (VORS_SPLIT_MAX_ROUNDS + 2) + 2)

// Per-pixel inverse-depth planes of the generic-mask keyframe path (all levels).
struct PixelPlanes {
    float* iz;  // NaN = Unknown
    float* v;   // < 0 = Unknown
    int off[VORS_MAX_LEVELS];
    int stride;
    // compaction workspace: usable pixels per chunk of VORS_CHUNK_PX pixels, [pair][chunks_total]; level l owns chunks
    // chunk_off[l] .. chunk_off[l + 1]
    int* counts;
    int chunk_off[VORS_MAX_LEVELS + 1];
    int chunks_total;
};
constexpr int VORS_CHUNK_PX = 4096;  // 256 threads x 16 consecutive pixels

// Image pyramid of a batch: level 0 is the caller's buffer (zero copy), levels >= 1 live in `upper`.
struct Pyramid {
    const uint8_t* level0;  // pair stride S0
    uint8_t* upper;         // pair stride upper_stride
};

void launch_transpose_u8(const uint8_t* src_colmajor, uint8_t* dst_rowmajor, int rows, int cols, int n, hipStream_t s);
void launch_transpose_u16(const uint16_t* src_colmajor, uint16_t* dst_rowmajor, int rows, int cols, int n, hipStream_t s);
void launch_pyramid(const Geom& g, Pyramid pyr, int n_pairs, hipStream_t s);
void launch_zero_ints(const Geom& g, int* base, int stride, int n_pairs, hipStream_t s);  // per-pair counters -> 0 (honours Geom::sel_list)
void launch_keyframe(const Geom& g, Pyramid kf, const uint16_t* depth, Records rec, int n_pairs, hipStream_t s);
void keyframe_region_geometry(const Geom& g, int* kf_r, int* n_regions);  // coarse-to-fine mode: roots per wavefront region, regions per pair
int count_isqrt_u16_mismatches(hipStream_t s);  // dso_kernels.hip: self-check of the gradient-magnitude root (0 = exact for every argument)
void launch_keyframe_dso(const Geom& g, Pyramid kf, const uint16_t* depth, DsoWs ws, uint8_t* mask, PixelPlanes pp, Records rec, int n_pairs,
                         hipStream_t s);
// Inspection: expand the slim records of one level of one pair into record planes (exact arithmetic).
void launch_slim_materialize(const Geom& g, int l, int pair, Records rec, int n, Records out, hipStream_t s);
void launch_dense_materialize(const Geom& g, int l, int pair, Pyramid kf, const uint16_t* depth, Records rec, Records out,
                              hipStream_t s);
// `kf` and `kf_depth` are read only in dense mode (points are recomputed from the keyframe image + depth on the fly).
void launch_lm_track(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, const float* prev_poses7, const float* kf_poses7,
                     float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats, int n_pairs, int block, LmSplitWs split, hipStream_t s);
// the two arithmetic modes of the above (lm_kernels.hip compiled with VORS_FUSED = 0 / 1)
void launch_lm_track_exact(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, const float* prev_poses7, const float* kf_poses7,
                           float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats, int n_pairs, int block, LmSplitWs split, hipStream_t s);
void launch_lm_track_fused(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, const float* prev_poses7, const float* kf_poses7,
                           float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats, int n_pairs, int block, LmSplitWs split, hipStream_t s);
// REFERENCE arithmetic (lm_reference.hip): the candidate lists of n_pairs pairs into extract_z's column-major order (no-op in dense mode;
// honours Geom::sel_list), and the tracker with the reference's sequential sums.
void launch_sort_colmajor(const Geom& g, Records rec, int n_pairs, hipStream_t s);
bool ref_rank_from_regions(const Geom& g, const Records& rec);  // coarse-to-fine: launch_sort_colmajor ranks the staged regions itself (no compaction first)
// dense mode: the column-major planes of the keyframe side (pyramid, depth map, inverse depths of levels >= 1; honours Geom::sel_list) and
// of the current frame's pyramid -> rec.dense_t
void launch_ref_dense_planes_keyframe(const Geom& g, Pyramid kf, const uint16_t* depth, Records rec, int n_pairs, hipStream_t s);
void launch_ref_dense_planes_current(const Geom& g, Pyramid cur, Records rec, int n_pairs, hipStream_t s);
void launch_lm_track_reference(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, const float* prev_poses7,
                               const float* kf_poses7, float* out_poses7, int32_t* out_status, vors_pair_stats* out_stats, int n_pairs, hipStream_t s);
void launch_lm_eval_level_reference(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, int pair, int lvl,
                                    const float* model7, float* out29, hipStream_t s);
void launch_lm_eval_obs_reference(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber_delta, const float* model7,
                                  float* out_energy_n_g_h, float* residuals, hipStream_t s);
void launch_lm_solve_obs_reference(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber_delta, const float* model7,
                                   float* out, hipStream_t s);
// One evaluation of one level of one pair of a prepared batch at an explicit model, per arithmetic mode -> 29 sums.
void launch_lm_eval_level_exact(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, int pair, int lvl,
                                const float* model7, float* out29, hipStream_t s);
void launch_lm_eval_level_fused(const Geom& g, Pyramid cur, Pyramid kf, const uint16_t* kf_depth, Records rec, int pair, int lvl,
                                const float* model7, float* out29, hipStream_t s);
// Operator level on explicit observations of one level (device buffers): eval at `model` -> out29 partial sums layout:
// [0]=sum r^2 (or Huber loss), [1]=n_inside (as float), [2..7]=g, [8..28]=H upper triangle row-wise.
void launch_lm_eval_obs(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber_delta,
                        const float* model7, float* out_energy_n_g_h /* 44 floats: e, n, g6, H36 */, float* residuals,
                        hipStream_t s);
void launch_lm_solve_obs(Intr k, int rows, int cols, const uint8_t* image, int n, Records rec, float huber_delta,
                         const float* model7, float* out /* model7, nb_iter, energy, lm_coef, status */, hipStream_t s);
// Records from explicit (x, y, idepth, jac, template) arrays: operator-level entry.
void launch_records_from_obs(Intr k, int rows, int cols, const uint8_t* tmpl, int n, const int32_t* xy, const float* iz,
                             const float* jac, Records rec, hipStream_t s);
// Exhaustive device check of div_uniform for divisor d (all 2^23 significands); returns true when it is exact.
bool verify_fastdiv(float d, float r, hipStream_t s);
// Exhaustive device check of idepth_of<true>(scale, d) == scale / d for d = 1 .. 65535.
bool verify_fast_idepth(float scale, hipStream_t s);
// depth -> (inverse depth, its reciprocal) table, 65536 float2 entries (dense mode, level 0).
void launch_build_depth_lut(float depth_scale, float2* lut, hipStream_t s);
void launch_synth_pairs(uint64_t seed0, int n_pairs, int rows, int cols, const double cam5[5], double motion_scale,
                        int invalid_percent, uint8_t* kf_gray, uint16_t* kf_depth, uint8_t* cur_gray, uint16_t* cur_depth,
                        float* gt_models7, hipStream_t s);

// Sequence tooling + the device side of vors_trackers_* (kernels.hip).
void launch_synth_frames(const void* d_frames /* {u64 seed, u64 salt, f64 xi[6]} x n */, int n_frames, int rows, int cols, const double cam5[5],
                         int invalid_percent, uint8_t* gray, uint16_t* depth, hipStream_t s);
void launch_tracker_pack_out(const float* pose7, const int32_t* status, const int32_t* kf_frame, const vors_pair_stats* stats, void* out,
                             hipStream_t s);
void launch_trackers_advance(int n_seq, int* frame_counter, const float* out_poses7, const vors_pair_stats* stats, float* cur_poses7,
                             float* kf_poses7, int32_t* kf_frame, int* promo_list, int* promo_count, hipStream_t s);
void launch_identity_poses(float* a, float* b, int n, hipStream_t s);  // two pose tables -> identity (inverse_compositional.rs:86-99)
void launch_promote_copy(const Geom& g, const void* src, size_t src_stride, void* dst, size_t dst_stride, size_t bytes, int n_pairs,
                         hipStream_t s);

}  // namespace vors
