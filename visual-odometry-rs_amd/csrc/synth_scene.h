// Deterministic synthetic RGB-D scene: a textured plane seen by a pinhole camera.
// Shared by the CPU tooling (oracle/, tests) and by the HIP renderer used by bench.py. Pure function of
// (seed, pixel), double arithmetic, integer-hash noise; no state. Not part of the reference; it only
// produces inputs in the format the reference consumes (gray u8 + depth u16 at TUM scale 5000,
// /root/reference/src/dataset/tum_rgbd.rs:15).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VORS_HD __host__ __device__ inline
#else
#define VORS_HD inline
#endif

namespace vors_synth {

struct CameraD {  // intrinsics in double: cu, cv, fu, fv, skew
    double cu, cv, fu, fv, skew;
};
struct RigidD {  // X_cam = R * X_key + t
    double R[3][3];
    double t[3];
};

VORS_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// uniform in [0,1)
VORS_HD double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

VORS_HD double lattice(uint64_t seed, int64_t ix, int64_t iy) {
    const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)ix * 0x100000001B3ull + 0x51ED27ull) ^
                                  splitmix64((uint64_t)iy * 0xC2B2AE3D27D4EB4Full + 0x7F4A7C15ull));
    return 2.0 * u01(h) - 1.0;
}
VORS_HD double value_noise(uint64_t seed, double p, double q) {
    const double fp = floor(p), fq = floor(q);
    const int64_t ix = (int64_t)fp, iy = (int64_t)fq;
    const double a = p - fp, b = q - fq;
    const double v00 = lattice(seed, ix, iy), v10 = lattice(seed, ix + 1, iy);
    const double v01 = lattice(seed, ix, iy + 1), v11 = lattice(seed, ix + 1, iy + 1);
    return (1 - b) * ((1 - a) * v00 + a * v10) + b * ((1 - a) * v01 + a * v11);
}
// Plane texture in metres on the plane (p, q) -> grey level (not yet clamped).
// Seeds with the top bit set select a PIECEWISE-CONSTANT texture (12 cm cells of random grey, plus a faint shading): mostly
// flat regions separated by strong edges, the kind of image the DSO-style selector (median-based thresholds) is made for.
VORS_HD double texture(uint64_t seed, double p, double q) {
    const double two_pi = 6.283185307179586476925;
    if (seed >> 63) {
        const double cell = 0.12;
        const int64_t ix = (int64_t)floor(p / cell), iy = (int64_t)floor(q / cell);
        return 128.0 + 100.0 * lattice(seed, ix, iy) + 3.0 * sin(two_pi * 0.4 * (p - q));
    }
    return 128.0 + 50.0 * sin(two_pi * 3.1 * p) + 40.0 * sin(two_pi * 7.3 * q + 1.3) +
           25.0 * sin(two_pi * 0.6 * (p + q) + 0.4) + 30.0 * value_noise(seed, p / 0.08, q / 0.08);
}
constexpr uint64_t BLOCKY = 1ull << 63;

// exp of a twist xi = (v, w) in double: X_cam = R X + t.
VORS_HD RigidD se3_exp_d(const double xi[6]) {
    const double wx = xi[3], wy = xi[4], wz = xi[5];
    const double th2 = wx * wx + wy * wy + wz * wz;
    double A, B, C;  // R = I + A W + B W^2 ; V = I + B W + C W^2
    if (th2 < 1e-16) {
        A = 1.0;
        B = 0.5;
        C = 1.0 / 6.0;
    } else {
        const double th = sqrt(th2);
        A = sin(th) / th;
        B = (1.0 - cos(th)) / th2;
        C = (th - sin(th)) / (th * th2);
    }
    const double W[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
    double W2[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W2[i][j] = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
    RigidD m;
    for (int i = 0; i < 3; ++i) {
        m.t[i] = 0;
        for (int j = 0; j < 3; ++j) {
            const double I = (i == j) ? 1.0 : 0.0;
            m.R[i][j] = I + A * W[i][j] + B * W2[i][j];
            m.t[i] += (I + B * W[i][j] + C * W2[i][j]) * xi[j];
        }
    }
    return m;
}

// The twist of pair `seed`: v uniform in +-0.02 m, w uniform in +-0.01 rad, scaled by `motion_scale`.
VORS_HD void pair_twist(uint64_t seed, double motion_scale, double xi[6]) {
    uint64_t s = splitmix64(seed ^ 0xA5A5A5A5DEADBEEFull);
    for (int k = 0; k < 6; ++k) {
        s = splitmix64(s);
        const double amp = (k < 3) ? 0.02 : 0.01;
        xi[k] = motion_scale * amp * (2.0 * u01(s) - 1.0);
    }
}

// Unit quaternion + translation (tx ty tz qx qy qz qw) of a RigidD, for ground-truth comparison.
VORS_HD void rigid_to_pose7(const RigidD& m, const double xi[6], float out[7]) {
    const double th = sqrt(xi[3] * xi[3] + xi[4] * xi[4] + xi[5] * xi[5]);
    double s = 0.5, c = 1.0;
    if (th > 1e-12) {
        s = sin(0.5 * th) / th;
        c = cos(0.5 * th);
    }
    out[0] = (float)m.t[0];
    out[1] = (float)m.t[1];
    out[2] = (float)m.t[2];
    out[3] = (float)(s * xi[3]);
    out[4] = (float)(s * xi[4]);
    out[5] = (float)(s * xi[5]);
    out[6] = (float)c;
}

// Render pixel (x, y) of the camera placed at X_cam = R X_key + t. Plane n.X = d in keyframe coordinates,
// n = normalize(0.1, -0.05, 1), d = 2 m. `invalid_percent` of depth pixels are 0 (unknown), by hash.
VORS_HD void render_pixel(uint64_t scene_seed, uint64_t frame_salt, const CameraD& cam, const RigidD& m, int x, int y,
                          int invalid_percent, uint8_t* gray, uint16_t* depth) {
    const double nn = sqrt(0.1 * 0.1 + 0.05 * 0.05 + 1.0);
    const double n[3] = {0.1 / nn, -0.05 / nn, 1.0 / nn};
    const double d = 2.0;
    // in-plane basis: e1 = normalize(ex - n (n.ex)), e2 = n x e1
    double e1[3] = {1.0 - n[0] * n[0], -n[1] * n[0], -n[2] * n[0]};
    const double e1n = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    e1[0] /= e1n;
    e1[1] /= e1n;
    e1[2] /= e1n;
    const double e2[3] = {n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0]};
    // ray through the pixel centre, camera frame, Z = 1
    const double Yc = ((double)y - cam.cv) / cam.fv;
    const double Xc = (((double)x - cam.cu) - cam.skew * Yc) / cam.fu;
    const double dc[3] = {Xc, Yc, 1.0};
    // camera centre and direction in keyframe coordinates: o = -R^T t, dk = R^T dc
    double o[3], dk[3];
    for (int i = 0; i < 3; ++i) {
        o[i] = -(m.R[0][i] * m.t[0] + m.R[1][i] * m.t[1] + m.R[2][i] * m.t[2]);
        dk[i] = m.R[0][i] * dc[0] + m.R[1][i] * dc[1] + m.R[2][i] * dc[2];
    }
    const double denom = n[0] * dk[0] + n[1] * dk[1] + n[2] * dk[2];
    const double s = (d - (n[0] * o[0] + n[1] * o[1] + n[2] * o[2])) / denom;  // = depth along the camera Z axis
    const double P[3] = {o[0] + s * dk[0], o[1] + s * dk[1], o[2] + s * dk[2]};
    const double p = e1[0] * P[0] + e1[1] * P[1] + e1[2] * P[2];
    const double q = e2[0] * P[0] + e2[1] * P[1] + e2[2] * P[2];
    double g = floor(texture(scene_seed, p, q) + 0.5);
    g = g < 0.0 ? 0.0 : (g > 255.0 ? 255.0 : g);
    *gray = (uint8_t)g;
    double dz = floor(s * 5000.0 + 0.5);
    dz = dz < 1.0 ? 1.0 : (dz > 65535.0 ? 65535.0 : dz);
    const uint64_t h = splitmix64(scene_seed ^ splitmix64(frame_salt) ^ splitmix64(((uint64_t)(uint32_t)y << 32) | (uint32_t)x));
    *depth = ((int)(h % 100ull) < invalid_percent) ? (uint16_t)0 : (uint16_t)dz;
}

}  // namespace vors_synth
