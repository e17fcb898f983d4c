// ORACLE — TEST INFRASTRUCTURE ONLY (see vors_oracle.hpp header). C API over the CPU restatement so that
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes.
// All image buffers at this API are ROW-MAJOR (numpy order); they are converted to the reference's
// column-major DMatrix with from_row_slice exactly like src/bin/vors_track.rs:142 does.
#include <cstring>
#include <thread>

#include "../visual-odometry-rs_amd/csrc/synth_scene.h"
#include "vors_oracle.hpp"

using namespace vors_oracle;

extern "C" {

// Same field layout as `vors_config` in include/vors_hip.h so one ctypes.Structure serves both.
struct vo_config {
    int32_t nb_levels;
    int32_t candidates_diff_threshold;
    float depth_scale;
    float cu, cv, fu, fv, skew;
    float idepth_variance;
    int32_t candidates_mode;  // 0 coarse-to-fine (reference), 1 dense (extension)
    float huber_delta;        // <= 0: off (reference)
};

static track::Config to_config(const vo_config* c) {
    track::Config cfg;
    cfg.nb_levels = (size_t)c->nb_levels;
    cfg.candidates_diff_threshold = (uint16_t)c->candidates_diff_threshold;
    cfg.depth_scale = c->depth_scale;
    cfg.intrinsics = Intrinsics{c->cu, c->cv, c->fu, c->fv, c->skew};
    cfg.idepth_variance = c->idepth_variance;
    cfg.candidates_mode = c->candidates_mode;
    cfg.huber_delta = c->huber_delta;
    return cfg;
}
static void pose_to7(const Iso3& p, float out[7]) {
    out[0] = p.t.x; out[1] = p.t.y; out[2] = p.t.z;
    out[3] = p.q.i; out[4] = p.q.j; out[5] = p.q.k; out[6] = p.q.w;
}
static Iso3 pose_from7(const float in[7]) { return Iso3{{in[0], in[1], in[2]}, {in[3], in[4], in[5], in[6]}}; }

// ------------------------------------------------------------------ tracker handle
void* vo_tracker_create(const vo_config* cfg, double depth_t, const uint16_t* depth, double img_t, const uint8_t* gray,
                        int rows, int cols, int keep_debug) {
    auto* t = new track::Tracker();
    auto d = DMatrix<uint16_t>::from_row_slice(rows, cols, depth);
    auto g = DMatrix<uint8_t>::from_row_slice(rows, cols, gray);
    if (!track::Tracker::init(to_config(cfg), depth_t, d, img_t, std::move(g), keep_debug != 0, *t)) {
        delete t;
        return nullptr;
    }
    return t;
}
void vo_tracker_destroy(void* h) { delete static_cast<track::Tracker*>(h); }

int vo_tracker_track(void* h, double depth_t, const uint16_t* depth, double img_t, const uint8_t* gray) {
    auto* t = static_cast<track::Tracker*>(h);
    const int rows = t->keyframe_multires_data.img_multires[0].nrows, cols = t->keyframe_multires_data.img_multires[0].ncols;
    auto d = DMatrix<uint16_t>::from_row_slice(rows, cols, depth);
    auto g = DMatrix<uint8_t>::from_row_slice(rows, cols, gray);
    return t->track(depth_t, d, img_t, std::move(g));
}
void vo_tracker_current_frame(void* h, double* timestamp, float pose7[7]) {
    auto* t = static_cast<track::Tracker*>(h);
    *timestamp = t->current_frame_depth_timestamp;  // inverse_compositional.rs:243-247
    pose_to7(t->current_frame_pose, pose7);
}
void vo_tracker_keyframe_pose(void* h, double* timestamp, float pose7[7]) {
    auto* t = static_cast<track::Tracker*>(h);
    *timestamp = t->keyframe_depth_timestamp;
    pose_to7(t->keyframe_pose, pose7);
}
int vo_tracker_num_levels(void* h) { return (int)static_cast<track::Tracker*>(h)->keyframe_multires_data.img_multires.size(); }
void vo_tracker_level(void* h, int lvl, int* rows, int* cols, int* n_points, float intr5[5]) {
    auto& k = static_cast<track::Tracker*>(h)->keyframe_multires_data;
    *rows = k.img_multires[lvl].nrows;
    *cols = k.img_multires[lvl].ncols;
    *n_points = (int)k.usable_candidates_multires[lvl].second.size();
    const Intrinsics& i = k.intrinsics_multires[lvl];
    intr5[0] = i.cu; intr5[1] = i.cv; intr5[2] = i.fu; intr5[3] = i.fv; intr5[4] = i.skew;
}
void vo_tracker_get_image(void* h, int lvl, uint8_t* out) {
    static_cast<track::Tracker*>(h)->keyframe_multires_data.img_multires[lvl].to_row_slice(out);
}
int vo_tracker_get_gradients(void* h, int lvl, int16_t* gx, int16_t* gy, uint16_t* g2) {
    auto& k = static_cast<track::Tracker*>(h)->keyframe_multires_data;
    if (!k.keep_debug) return -1;
    k.gradients_multires[lvl].first.to_row_slice(gx);
    k.gradients_multires[lvl].second.to_row_slice(gy);
    k.gradients_squared_norm_multires[lvl].to_row_slice(g2);
    return 0;
}
int vo_tracker_get_mask(void* h, uint8_t* out) {
    auto& k = static_cast<track::Tracker*>(h)->keyframe_multires_data;
    if (!k.keep_debug) return -1;
    k.candidates_points.to_row_slice(out);
    return 0;
}
// xy: int32[2n] (x, y); idepth: float[n]; jac: float[6n]; reference enumeration order (column-major).
void vo_tracker_get_points(void* h, int lvl, int32_t* xy, float* idepth, float* jac) {
    auto& k = static_cast<track::Tracker*>(h)->keyframe_multires_data;
    const auto& c = k.usable_candidates_multires[lvl];
    for (size_t i = 0; i < c.second.size(); ++i) {
        xy[2 * i] = (int32_t)c.first[i].first;
        xy[2 * i + 1] = (int32_t)c.first[i].second;
        idepth[i] = c.second[i];
        if (jac) std::memcpy(jac + 6 * i, k.jacobians_multires[lvl][i].v, 6 * sizeof(float));
    }
}
// nb_iter/energy: arrays of nb_levels entries (index = level).
void vo_tracker_last(void* h, float lm_model7[7], float* flow, int* changed_keyframe, int* went_well, int32_t* nb_iter,
                     float* energy) {
    auto* t = static_cast<track::Tracker*>(h);
    pose_to7(t->last_lm_model, lm_model7);
    *flow = t->last_optical_flow;
    *changed_keyframe = t->last_changed_keyframe ? 1 : 0;
    *went_well = t->last_optimization_went_well ? 1 : 0;
    for (size_t l = 0; l < t->last_level_stats.size(); ++l) {
        if (nb_iter) nb_iter[l] = t->last_level_stats[l].nb_iter;
        if (energy) energy[l] = t->last_level_stats[l].energy;
    }
}

// ------------------------------------------------------------------ operator level (the trait's eval / solve)
struct ObsOwner {
    Intrinsics intr;
    DMatrix<uint8_t> tmpl, img;
    std::vector<std::pair<size_t, size_t>> coords;
    std::vector<Float> z;
    std::vector<Vec6> jac;
    std::vector<Mat6> hes;
    lm_optimizer::Obs obs;
};
static void make_obs(ObsOwner& o, const float intr5[5], const uint8_t* tmpl, const uint8_t* img, int rows, int cols, int n,
                     const int32_t* xy, const float* idepth, const float* jac, float huber_delta) {
    o.intr = Intrinsics{intr5[0], intr5[1], intr5[2], intr5[3], intr5[4]};
    o.tmpl = DMatrix<uint8_t>::from_row_slice(rows, cols, tmpl);
    o.img = DMatrix<uint8_t>::from_row_slice(rows, cols, img);
    o.coords.resize(n);
    o.z.assign(idepth, idepth + n);
    o.jac.resize(n);
    for (int i = 0; i < n; ++i) {
        o.coords[i] = {(size_t)xy[2 * i], (size_t)xy[2 * i + 1]};
        std::memcpy(o.jac[i].v, jac + 6 * i, 6 * sizeof(float));
    }
    o.hes = track::hessians_vec(o.jac);
    o.obs.intrinsics = &o.intr;
    o.obs.template_ = &o.tmpl;
    o.obs.image = &o.img;
    o.obs.coordinates = &o.coords;
    o.obs._z_candidates = &o.z;
    o.obs.jacobians = &o.jac;
    o.obs.hessians = &o.hes;
    o.obs.huber_delta = huber_delta;
}
// eval_energy + compute_eval_data at `model7` (lm_optimizer.rs:68-107). H36 row-major 6x6.
// residuals (nullable): float[n], NaN where the point is outside.
void vo_lm_eval(const float intr5[5], const uint8_t* tmpl, const uint8_t* img, int rows, int cols, int n, const int32_t* xy,
                const float* idepth, const float* jac, float huber_delta, const float model7[7], float* energy,
                int32_t* n_inside, float g6[6], float H36[36], float* residuals) {
    ObsOwner o;
    make_obs(o, intr5, tmpl, img, rows, cols, n, xy, idepth, jac, huber_delta);
    const Iso3 model = pose_from7(model7);
    auto pre = lm_optimizer::LMOptimizerState::eval_energy(o.obs, model);
    auto e = lm_optimizer::LMOptimizerState::compute_eval_data(o.obs, model, pre);
    *energy = e.energy;
    *n_inside = (int32_t)pre.inside_indices.size();
    std::memcpy(g6, e.gradient.v, sizeof(float) * 6);
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) H36[a * 6 + b] = e.hessian.m[a][b];
    if (residuals) {
        for (int i = 0; i < n; ++i) residuals[i] = NAN;
        for (size_t i = 0; i < pre.inside_indices.size(); ++i) residuals[pre.inside_indices[i]] = pre.residuals[i];
    }
}
// iterative_solve at one level (optimizer.rs:57-70). Returns 0 ok / 1 step error.
int vo_lm_solve(const float intr5[5], const uint8_t* tmpl, const uint8_t* img, int rows, int cols, int n, const int32_t* xy,
                const float* idepth, const float* jac, float huber_delta, const float model7_in[7], float model7_out[7],
                int32_t* nb_iter, float* energy, float* lm_coef) {
    ObsOwner o;
    make_obs(o, intr5, tmpl, img, rows, cols, n, xy, idepth, jac, huber_delta);
    lm_optimizer::LMOptimizerState st;
    size_t it = 0;
    std::string err;
    if (!lm_optimizer::LMOptimizerState::iterative_solve(o.obs, pose_from7(model7_in), st, it, err)) return 1;
    pose_to7(st.eval_data.model, model7_out);
    *nb_iter = (int32_t)it;
    *energy = st.eval_data.energy;
    *lm_coef = st.lm_coef;
    return 0;
}
// One LM step from given (H, g, model, lm_coef): lm_optimizer.rs:123-136. Returns 0 ok / 1 Cholesky failure.
int vo_lm_step(const float H36[36], const float g6[6], const float model7[7], float lm_coef, float out7[7], float delta6[6]) {
    lm_optimizer::LMOptimizerState st;
    st.lm_coef = lm_coef;
    for (int a = 0; a < 6; ++a) {
        st.eval_data.gradient.v[a] = g6[a];
        for (int b = 0; b < 6; ++b) st.eval_data.hessian.m[a][b] = H36[a * 6 + b];
    }
    st.eval_data.model = pose_from7(model7);
    if (delta6) {
        Mat6 h = st.eval_data.hessian;
        for (int a = 0; a < 6; ++a) h.m[a][a] *= 1.0f + lm_coef;
        if (!cholesky6(h)) return 1;
        const Vec6 d = cholesky6_solve(h, st.eval_data.gradient);
        std::memcpy(delta6, d.v, sizeof(float) * 6);
    }
    Iso3 out;
    std::string err;
    if (!st.step(out, err)) return 1;
    pose_to7(out, out7);
    return 0;
}

// ------------------------------------------------------------------ batch of independent pairs (the CPU baseline)
// For each pair: Config::init(keyframe) then Tracker::track(current). out_poses7 = current_frame() pose,
// out_models7 (nullable) = final lm_model, nb_iter (nullable) int32[n * nb_levels], n_points (nullable) same shape.
// n_threads > 1 splits pairs into contiguous blocks, one std::thread each (SURVEY.md §8d "all cores" variant).
int vo_track_pairs(const vo_config* cfg, int n_pairs, const uint8_t* kf_gray, const uint16_t* kf_depth,
                   const uint8_t* cur_gray, const uint16_t* cur_depth, int rows, int cols, const float* init_poses7,
                   float* out_poses7, int32_t* out_status, float* out_models7, int32_t* nb_iter, int32_t* n_points,
                   float* out_flow, int n_threads) {
    const track::Config config = to_config(cfg);
    const size_t S = (size_t)rows * cols;
    std::vector<uint16_t> zero_depth;
    if (!cur_depth) zero_depth.assign(S, 0);
    auto work = [&](int lo, int hi) {
        for (int p = lo; p < hi; ++p) {
            track::Tracker t;
            auto d = DMatrix<uint16_t>::from_row_slice(rows, cols, kf_depth + p * S);
            auto g = DMatrix<uint8_t>::from_row_slice(rows, cols, kf_gray + p * S);
            if (!track::Tracker::init(config, 0.0, d, 0.0, std::move(g), false, t)) {
                out_status[p] = -1;
                continue;
            }
            if (init_poses7) t.current_frame_pose = pose_from7(init_poses7 + 7 * p);  // initial guess = inverse of this
            auto d2 = DMatrix<uint16_t>::from_row_slice(rows, cols, cur_depth ? cur_depth + p * S : zero_depth.data());
            auto g2 = DMatrix<uint8_t>::from_row_slice(rows, cols, cur_gray + p * S);
            // Batch semantics: no keyframe switch is wanted, but running the keyframe test (and the switch when it
            // triggers) is part of Tracker::track and is timed as such.
            out_status[p] = t.track(1.0, d2, 1.0, std::move(g2));
            pose_to7(t.current_frame_pose, out_poses7 + 7 * p);
            if (out_models7) pose_to7(t.last_lm_model, out_models7 + 7 * p);
            if (out_flow) out_flow[p] = t.last_optical_flow;
            for (size_t l = 0; l < config.nb_levels; ++l) {
                if (nb_iter) nb_iter[p * config.nb_levels + l] = t.last_level_stats[l].nb_iter;
                if (n_points) n_points[p * config.nb_levels + l] = t.last_level_stats[l].n_points;
            }
        }
    };
    if (n_threads <= 1) {
        work(0, n_pairs);
    } else {
        std::vector<std::thread> th;
        const int per = (n_pairs + n_threads - 1) / n_threads;
        for (int k = 0; k < n_threads; ++k) {
            const int lo = k * per, hi = std::min(n_pairs, lo + per);
            if (lo < hi) th.emplace_back(work, lo, hi);
        }
        for (auto& x : th) x.join();
    }
    return 0;
}

// n_seq independent SEQUENCES of n_frames frames each, one Tracker per sequence (vors_track.rs:46-62 for each): frame k of sequence s
// at ((size_t)k * n_seq + s) * rows * cols (frame-major, the layout a lock-step host holds). Frame 0 initialises, frames 1 .. n_frames-1
// are tracked with depth timestamp = k. out_poses7 [n_seq][n_frames-1][7], out_status / out_switch [n_seq][n_frames-1].
int vo_track_sequences(const vo_config* cfg, int n_seq, int n_frames, const uint8_t* gray, const uint16_t* depth, int rows, int cols,
                       float* out_poses7, int32_t* out_status, int32_t* out_switch, int n_threads) {
    const track::Config config = to_config(cfg);
    const size_t S = (size_t)rows * cols;
    auto work = [&](int lo, int hi) {
        for (int s = lo; s < hi; ++s) {
            track::Tracker t;
            auto d = DMatrix<uint16_t>::from_row_slice(rows, cols, depth + (size_t)s * S);
            auto g = DMatrix<uint8_t>::from_row_slice(rows, cols, gray + (size_t)s * S);
            const bool ok = track::Tracker::init(config, 0.0, d, 0.0, std::move(g), false, t);
            for (int k = 1; k < n_frames; ++k) {
                const size_t o = (size_t)s * (n_frames - 1) + (k - 1);
                if (!ok) {
                    out_status[o] = -1;
                    continue;
                }
                const size_t f = ((size_t)k * n_seq + s) * S;
                auto dk = DMatrix<uint16_t>::from_row_slice(rows, cols, depth + f);
                auto gk = DMatrix<uint8_t>::from_row_slice(rows, cols, gray + f);
                out_status[o] = t.track((double)k, dk, (double)k, std::move(gk));
                pose_to7(t.current_frame_pose, out_poses7 + 7 * o);
                out_switch[o] = t.last_changed_keyframe ? 1 : 0;
            }
        }
    };
    if (n_threads <= 1) {
        work(0, n_seq);
    } else {
        std::vector<std::thread> th;
        const int per = (n_seq + n_threads - 1) / n_threads;
        for (int k = 0; k < n_threads; ++k) {
            const int lo = k * per, hi = std::min(n_seq, lo + per);
            if (lo < hi) th.emplace_back(work, lo, hi);
        }
        for (auto& x : th) x.join();
    }
    return 0;
}

// ------------------------------------------------------------------ stand-alone stages and KAT helpers
// mean_pyramid (multires.rs:21-31): writes levels concatenated row-major; returns the number of levels.
int vo_mean_pyramid(const uint8_t* img, int rows, int cols, int max_levels, uint8_t* out, int32_t* out_rows, int32_t* out_cols) {
    auto pyr = mean_pyramid((size_t)max_levels, DMatrix<uint8_t>::from_row_slice(rows, cols, img));
    size_t off = 0;
    for (size_t l = 0; l < pyr.size(); ++l) {
        pyr[l].to_row_slice(out + off);
        off += (size_t)pyr[l].nrows * pyr[l].ncols;
        out_rows[l] = pyr[l].nrows;
        out_cols[l] = pyr[l].ncols;
    }
    return (int)pyr.size();
}
// DSO-style level-0 mask (candidates/dso.rs with the parameters of examples/candidates_dso.rs). base_sizes (nullable, >= 3
// entries): the block sizes of the successive rounds; returns the number of rounds.
int vo_dso_mask(const uint8_t* img, int rows, int cols, int nb_target, uint64_t seed, uint8_t* mask_out, int32_t* base_sizes) {
    std::vector<size_t> trace;
    const auto m = dso::select_like_example(DMatrix<uint8_t>::from_row_slice(rows, cols, img), (size_t)nb_target, seed, &trace);
    m.to_row_slice(mask_out);
    if (base_sizes)
        for (size_t k = 0; k < trace.size() && k < 3; ++k) base_sizes[k] = (int32_t)trace[k];
    return (int)trace.size();
}
void vo_prune_with_thresh(int thresh, int a, int b, int c, int d, uint8_t out[4]) {
    bool r[4];
    candidates::prune_with_thresh((uint16_t)thresh, (uint16_t)a, (uint16_t)b, (uint16_t)c, (uint16_t)d, r);
    for (int k = 0; k < 4; ++k) out[k] = r[k];
}
// std::sin / std::cos on f32 exactly as se3::exp / so3::exp call them (= the platform libm's sinf / cosf, which is what Rust's f32::sin /
// f32::cos call): the yardstick for the product's restatement of that algorithm (csrc/lie.h ref_sinf / ref_cosf).
void vo_libm_sincos(const float* x, int n, float* s, float* c) {
    for (int i = 0; i < n; ++i) {
        s[i] = std::sin(x[i]);
        c[i] = std::cos(x[i]);
    }
}
void vo_se3_exp(const float xi[6], float out7[7]) { Vec6 v; std::memcpy(v.v, xi, 24); pose_to7(se3::exp(v), out7); }
void vo_se3_log(const float in7[7], float xi[6]) { const Vec6 v = se3::log(pose_from7(in7)); std::memcpy(xi, v.v, 24); }
void vo_se3_hat(const float xi[6], float out16[16]) { Vec6 v; std::memcpy(v.v, xi, 24); const Mat4 m = se3::hat(v); std::memcpy(out16, m.m, 64); }
void vo_se3_vee(const float in16[16], float xi[6]) { Mat4 m; std::memcpy(m.m, in16, 64); const Vec6 v = se3::vee(m); std::memcpy(xi, v.v, 24); }
void vo_so3_exp(const float w[3], float q4[4]) { const Quat q = so3::exp(Vec3{w[0], w[1], w[2]}); q4[0] = q.i; q4[1] = q.j; q4[2] = q.k; q4[3] = q.w; }
void vo_so3_log(const float q4[4], float w[3]) { const Vec3 v = so3::log(Quat{q4[0], q4[1], q4[2], q4[3]}); w[0] = v.x; w[1] = v.y; w[2] = v.z; }
void vo_so3_hat(const float w[3], float out9[9]) { const Mat3 m = so3::hat(Vec3{w[0], w[1], w[2]}); std::memcpy(out9, m.m, 36); }
void vo_so3_hat_2(const float w[3], float out9[9]) { const Mat3 m = so3::hat_2(Vec3{w[0], w[1], w[2]}); std::memcpy(out9, m.m, 36); }
void vo_so3_vee(const float in9[9], float w[3]) { Mat3 m; std::memcpy(m.m, in9, 36); const Vec3 v = so3::vee(m); w[0] = v.x; w[1] = v.y; w[2] = v.z; }
void vo_iso_mul(const float a7[7], const float b7[7], float out7[7]) { pose_to7(iso_mul(pose_from7(a7), pose_from7(b7)), out7); }
void vo_iso_inverse(const float a7[7], float out7[7]) { pose_to7(iso_inverse(pose_from7(a7)), out7); }
void vo_intrinsics_multires(const float intr5[5], int n, float* out5n) {
    auto v = Intrinsics{intr5[0], intr5[1], intr5[2], intr5[3], intr5[4]}.multi_res((size_t)n);
    for (size_t l = 0; l < v.size(); ++l) {
        out5n[5 * l] = v[l].cu; out5n[5 * l + 1] = v[l].cv; out5n[5 * l + 2] = v[l].fu; out5n[5 * l + 3] = v[l].fv; out5n[5 * l + 4] = v[l].skew;
    }
}

// ------------------------------------------------------------------ synthetic scene (test/bench tooling)
// Render one frame of scene `seed` seen from X_cam = exp(xi) X_key. cam5 = cu cv fu fv skew (double).
void vo_synth_frame(uint64_t seed, uint64_t frame_salt, const double cam5[5], const double xi[6], int rows, int cols,
                    int invalid_percent, uint8_t* gray, uint16_t* depth, int n_threads) {
    const vors_synth::CameraD cam{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]};
    const vors_synth::RigidD m = vors_synth::se3_exp_d(xi);
    auto work = [&](int y0, int y1) {
        for (int y = y0; y < y1; ++y)
            for (int x = 0; x < cols; ++x)
                vors_synth::render_pixel(seed, frame_salt, cam, m, x, y, invalid_percent, gray + (size_t)y * cols + x,
                                         depth + (size_t)y * cols + x);
    };
    if (n_threads <= 1) {
        work(0, rows);
    } else {
        std::vector<std::thread> th;
        const int per = (rows + n_threads - 1) / n_threads;
        for (int k = 0; k < n_threads; ++k) {
            const int lo = k * per, hi = std::min(rows, lo + per);
            if (lo < hi) th.emplace_back(work, lo, hi);
        }
        for (auto& x : th) x.join();
    }
}
void vo_synth_pair_twist(uint64_t seed, double motion_scale, double xi[6]) { vors_synth::pair_twist(seed, motion_scale, xi); }
// Ground-truth model (keyframe -> current camera) as tx ty tz qx qy qz qw.
void vo_synth_gt_model7(const double xi[6], float out7[7]) {
    const vors_synth::RigidD m = vors_synth::se3_exp_d(xi);
    vors_synth::rigid_to_pose7(m, xi, out7);
}

}  // extern "C"
