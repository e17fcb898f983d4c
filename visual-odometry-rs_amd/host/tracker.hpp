// C++ host-side mirror of the reference's tracking interface over the C ABI (include/vors_hip.h):
//   vors::Intrinsics                       reference src/core/camera.rs:84-91
//   vors::track::Config / Config::init     src/core/track/inverse_compositional.rs:37-49, 74-100
//   vors::track::Tracker::track / current_frame                      inverse_compositional.rs:170-248
//   vors::track::LMOptimizerState          src/core/track/lm_optimizer.rs:16-193, host-driven through the operator-level
//                                          entry points (eval on the device, step/stop on the host) — the shape a user
//                                          implementing the optimizer trait themselves would follow.
// Same names, argument meaning and error behaviour as the reference: track() returns void (failures keep the pose and
// are logged to stderr like the reference's eprintln!s); programmer errors (image too small for nb_levels) throw, where
// the reference panics.
#pragma once
#include <array>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../../include/vors_hip.h"
#include "optimizer.hpp"

namespace vors {

// Rust `{}` (Display) for floats: the shortest digits that round-trip, positional notation, `1.0` prints as `1`
// (what eprintln!("Optical_flow: {}", ..) and the trajectory writer tum_rgbd.rs:78-85 produce).
template <class T>
inline std::string rust_display(T v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[512];
    auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

using Float = float;              // src/misc/type_aliases.rs:10
using Iso3 = std::array<float, 7>;  // tx ty tz qx qy qz qw
using Vec6 = std::array<float, 6>;
using Mat6 = std::array<float, 36>;  // row-major

struct Intrinsics {  // camera.rs:84-91
    std::pair<Float, Float> principal_point;
    std::pair<Float, Float> focal;
    Float skew;
};

namespace tum_rgbd {  // src/dataset/tum_rgbd.rs:15-52
constexpr Float DEPTH_SCALE = 5000.0f;
inline Intrinsics INTRINSICS_ICL_NUIM() { return {{319.5f, 239.5f}, {481.20f, -480.00f}, 0.0f}; }
inline Intrinsics INTRINSICS_FR1() { return {{318.643040f, 255.313989f}, {517.306408f, 516.469215f}, 0.0f}; }
inline Intrinsics INTRINSICS_FR2() { return {{325.141442f, 249.701764f}, {520.908620f, 521.007327f}, 0.0f}; }
inline Intrinsics INTRINSICS_FR3() { return {{320.106653f, 247.632132f}, {535.433105f, 539.212524f}, 0.0f}; }
}  // namespace tum_rgbd

inline void check(vors_status st) {
    if (st != VORS_OK) throw std::runtime_error(std::string("vors_hip: ") + vors_last_error());
}

// A row-major or column-major image view (the reference passes nalgebra DMatrix = column-major).
template <class T>
struct ImageView {
    const T* data;
    int rows, cols;
    int layout;  // VORS_ROW_MAJOR / VORS_COL_MAJOR
};

namespace track {

class Tracker;

struct Config {  // inverse_compositional.rs:37-49
    std::size_t nb_levels;
    std::uint16_t candidates_diff_threshold;
    Float depth_scale;
    Intrinsics intrinsics;
    Float idepth_variance;
    // extensions (zero = reference behaviour)
    int candidates_mode = VORS_CANDIDATES_COARSE_TO_FINE;
    Float huber_delta = 0.0f;
    int arithmetic = VORS_ARITH_REFERENCE;  // the reference's arithmetic and summation order (bit-identical poses); VORS_ARITH_FUSED: the fastest, poses within 1e-4 but for a measured tail (include/vors_hip.h)

    vors_config to_c() const {
        return vors_config{(int32_t)nb_levels, (int32_t)candidates_diff_threshold, depth_scale, intrinsics.principal_point.first,
                           intrinsics.principal_point.second, intrinsics.focal.first, intrinsics.focal.second, intrinsics.skew,
                           idepth_variance, candidates_mode, huber_delta, arithmetic};
    }
    // Config::init (inverse_compositional.rs:74-100)
    Tracker init(double keyframe_depth_timestamp, ImageView<std::uint16_t> depth_map, double keyframe_img_timestamp,
                 ImageView<std::uint8_t> img) const;
};

class Tracker {  // inverse_compositional.rs:31-34
   public:
    Tracker(const Tracker&) = delete;
    Tracker& operator=(const Tracker&) = delete;
    Tracker(Tracker&& o) noexcept : h_(o.h_), rows_(o.rows_), cols_(o.cols_), layout_(o.layout_), log_(o.log_) { o.h_ = nullptr; }
    ~Tracker() { vors_tracker_destroy(h_); }

    // Tracker::track (inverse_compositional.rs:170-240): returns () like the reference.
    void track(double depth_time, ImageView<std::uint16_t> depth_map, double img_time, ImageView<std::uint8_t> img) {
        // the C ABI carries no dimensions after create(): a smaller frame would be read past its end
        if (depth_map.rows != rows_ || depth_map.cols != cols_ || img.rows != rows_ || img.cols != cols_)
            throw std::invalid_argument("Tracker::track: frame shape differs from the keyframe's");
        if (depth_map.layout != layout_ || img.layout != layout_) throw std::invalid_argument("Tracker::track: layout differs from init's");
        int status = 0;
        double keyframe_depth_timestamp = 0;
        Iso3 keyframe_pose_before;
        check(vors_tracker_keyframe(h_, &keyframe_depth_timestamp, keyframe_pose_before.data()));
        check(vors_tracker_track_checked(h_, depth_time, depth_map.data, img_time, img.data, img.rows, img.cols, &status));
        vors_pair_stats s;
        check(vors_tracker_last_stats(h_, &s));
        last_ = s;
        last_status_ = status;
        if (log_) {
            if (status != VORS_TRACK_OK) std::fprintf(stderr, "Error at Cholesky decomposition of hessian\n");  // :196
            // the reference's own lines, with Rust's float Display (inverse_compositional.rs:222,228-229)
            std::fprintf(stderr, "Optical_flow: %s\n", rust_display(s.optical_flow).c_str());
            if (s.change_keyframe)
                std::fprintf(stderr, "Changing keyframe after: %s seconds\n", rust_display(depth_time - keyframe_depth_timestamp).c_str());
        }
    }
    // Tracker::current_frame (inverse_compositional.rs:243-248): (depth timestamp, pose)
    std::pair<double, Iso3> current_frame() const {
        double t = 0;
        Iso3 p;
        check(vors_tracker_current_frame(h_, &t, p.data()));
        return {t, p};
    }
    const vors_pair_stats& last_stats() const { return last_; }
    int last_status() const { return last_status_; }
    void set_logging(bool on) { log_ = on; }

   private:
    friend struct Config;
    Tracker(vors_tracker* h, int rows, int cols, int layout) : h_(h), rows_(rows), cols_(cols), layout_(layout) {}
    vors_tracker* h_ = nullptr;
    int rows_ = 0, cols_ = 0, layout_ = 0;
    bool log_ = true;
    vors_pair_stats last_{};
    int last_status_ = 0;
};

inline Tracker Config::init(double keyframe_depth_timestamp, ImageView<std::uint16_t> depth_map, double keyframe_img_timestamp,
                            ImageView<std::uint8_t> img) const {
    if (depth_map.rows != img.rows || depth_map.cols != img.cols || depth_map.layout != img.layout)
        throw std::invalid_argument("Config::init: depth map and image differ in shape or layout");
    vors_config c = to_c();
    vors_tracker* h = nullptr;
    check(vors_tracker_create(&c, keyframe_depth_timestamp, depth_map.data, keyframe_img_timestamp, img.data, img.rows, img.cols,
                              img.layout, &h));
    return Tracker(h, img.rows, img.cols, img.layout);
}

// ---------------------------------------------------------------------------------------------------------------
// lm_optimizer.rs as an implementation of the trait, driven from the host.
// ---------------------------------------------------------------------------------------------------------------
struct EvalData {  // lm_optimizer.rs:31-40
    Mat6 hessian;
    Vec6 gradient;
    Float energy;
    Iso3 model;
};
using EvalState = std::variant<EvalData, Float>;  // Result<EvalData, Float> (lm_optimizer.rs:28)
using Obs = vors_obs;                             // lm_optimizer.rs:43-58

struct LMOptimizerState : optimizer::State<LMOptimizerState, Obs, EvalState, Iso3, std::string> {
    Float lm_coef;
    EvalData eval_data;

    static EvalData full_eval(const Obs& obs, const Iso3& model) {
        EvalData e;
        int32_t n_inside = 0;
        check(vors_lm_eval(&obs, model.data(), &e.energy, &n_inside, e.gradient.data(), e.hessian.data(), nullptr));
        e.model = model;
        return e;
    }
    static LMOptimizerState init(const Obs& obs, Iso3 model) {  // lm_optimizer.rs:113-118
        LMOptimizerState s;
        s.lm_coef = 0.1f;
        s.eval_data = full_eval(obs, model);
        return s;
    }
    bool step(Iso3* out, std::string* err) const {  // lm_optimizer.rs:123-136
        int ok = 0;
        check(vors_lm_step(eval_data.hessian.data(), eval_data.gradient.data(), eval_data.model.data(), lm_coef, out->data(), &ok));
        if (!ok) *err = "Error at Cholesky decomposition of hessian";
        return ok != 0;
    }
    EvalState eval(const Obs& obs, Iso3 model) const {  // lm_optimizer.rs:140-149 (device pass is fused: energy + g + H)
        EvalData e = full_eval(obs, model);
        if (e.energy > eval_data.energy) return EvalState(e.energy);
        return EvalState(e);
    }
    static std::pair<LMOptimizerState, optimizer::Continue> stop_criterion(LMOptimizerState self, std::size_t nb_iter,
                                                                           EvalState eval_state) {  // lm_optimizer.rs:156-192
        using optimizer::Continue;
        const bool too_many_iterations = nb_iter > 20;
        const bool is_err = std::holds_alternative<Float>(eval_state);
        if (is_err && too_many_iterations) return {self, Continue::Stop};
        if (too_many_iterations) {
            self.eval_data = std::get<EvalData>(eval_state);
            return {self, Continue::Stop};
        }
        if (is_err) {
            self.lm_coef *= 10.0f;
            return {self, Continue::Forward};
        }
        const EvalData& e = std::get<EvalData>(eval_state);
        const Float d_energy = self.eval_data.energy - e.energy;
        const Continue c = d_energy > 1.0f ? Continue::Forward : Continue::Stop;
        self.lm_coef = 0.1f * self.lm_coef;
        self.eval_data = e;
        return {self, c};
    }
};

}  // namespace track
}  // namespace vors
