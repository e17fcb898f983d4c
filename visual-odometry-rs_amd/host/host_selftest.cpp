// Compile-and-link check of the C++ host mirror against libvors_hip.so, plus a GPU self-test when a device exists:
// tracks a synthetic 2-frame sequence through vors::track::Config::init / Tracker::track / current_frame and solves one
// level through the optimizer trait (LMOptimizerState::iterative_solve). Exit code 0 = ok, 77 = no GPU (link check only).
#include <cmath>
#include <cstdio>
#include <vector>

#include "../csrc/synth_scene.h"
#include "tracker.hpp"

int main() {
    using namespace vors;
    // the trait skeleton works for any solver: 1-D toy problem (compile-time check of the CRTP contract)
    struct Toy : optimizer::State<Toy, double, std::pair<double, double>, double, int> {
        double x = 0, e = 0;
        static Toy init(const double& target, double m) { Toy t; t.x = m; t.e = (m - target) * (m - target); return t; }
        bool step(double* out, int*) const { *out = 0.5 * (x + 3.0); return true; }
        std::pair<double, double> eval(const double& target, double m) const { return {m, (m - target) * (m - target)}; }
        static std::pair<Toy, optimizer::Continue> stop_criterion(Toy self, std::size_t n, std::pair<double, double> ev) {
            const bool go = self.e - ev.second > 1e-9 && n < 60;
            self.x = ev.first; self.e = ev.second;
            return {self, go ? optimizer::Continue::Forward : optimizer::Continue::Stop};
        }
    };
    auto toy = Toy::iterative_solve(3.0, 11.0);
    if (!toy.ok() || std::fabs(toy.state->x - 3.0) > 1e-3) { std::fprintf(stderr, "trait skeleton failed\n"); return 1; }

    if (vors_device_count() < 1) { std::printf("host_selftest: link ok, no GPU (skipping device part)\n"); return 77; }
    const int rows = 120, cols = 160;
    const double s = cols / 640.0;
    const vors_synth::CameraD cam{s * (318.643040 + 0.5) - 0.5, s * (255.313989 + 0.5) - 0.5, s * 517.306408, s * 516.469215, 0.0};
    std::vector<uint8_t> g0(rows * cols), g1(rows * cols);
    std::vector<uint16_t> d0(rows * cols), d1(rows * cols);
    double xi[6], zero[6] = {0, 0, 0, 0, 0, 0};
    vors_synth::pair_twist(42, 1.0, xi);
    const auto id = vors_synth::se3_exp_d(zero), m = vors_synth::se3_exp_d(xi);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            vors_synth::render_pixel(42, 0, cam, id, x, y, 2, &g0[y * cols + x], &d0[y * cols + x]);
            vors_synth::render_pixel(42, 1, cam, m, x, y, 2, &g1[y * cols + x], &d1[y * cols + x]);
        }
    track::Config config{4, 7, tum_rgbd::DEPTH_SCALE, Intrinsics{{(float)cam.cu, (float)cam.cv}, {(float)cam.fu, (float)cam.fv}, 0.0f}, 0.0001f};
    track::Tracker tracker = config.init(0.0, {d0.data(), rows, cols, VORS_ROW_MAJOR}, 0.0, {g0.data(), rows, cols, VORS_ROW_MAJOR});
    tracker.set_logging(false);
    tracker.track(1.0, {d1.data(), rows, cols, VORS_ROW_MAJOR}, 1.0, {g1.data(), rows, cols, VORS_ROW_MAJOR});
    auto [t, pose] = tracker.current_frame();
    float gt[7];
    vors_synth::rigid_to_pose7(m, xi, gt);
    float err = 0;
    for (int k = 0; k < 7; ++k) err = std::fmax(err, std::fabs(tracker.last_stats().lm_model[k] - gt[k]));
    std::printf("host_selftest: t=%g pose=[%g %g %g | %g %g %g %g] max|model-gt|=%g status=%d\n", t, pose[0], pose[1], pose[2], pose[3],
                pose[4], pose[5], pose[6], err, tracker.last_status());
    return (t == 1.0 && err < 1e-2 && tracker.last_status() == 0) ? 0 : 1;
}
