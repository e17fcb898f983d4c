// TUM RGB-D dataset plumbing, C++ mirror of the reference's src/dataset/tum_rgbd.rs (host side, outside the hot path):
//   Association / Frame                       tum_rgbd.rs:55-73
//   Frame::to_string  "t tx ty tz qx qy qz qw" tum_rgbd.rs:76-86  (Rust `Display` of f64 / f32: shortest digits that round-trip,
//                                              positional notation, `1.0` prints as `1`)
//   parse::associations                        tum_rgbd.rs:97-100,122-139 (nom grammar: a line is a `#` comment or
//                                              `double space path space double space path`; anything else — including an empty
//                                              line — is "Parsing error" for the whole file)
#pragma once
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <optional>
#include <string>
#include <vector>

#include "tracker.hpp"

namespace vors {
namespace tum_rgbd {

struct Association {  // tum_rgbd.rs:63-73
    double depth_timestamp;
    std::string depth_file_path;
    double color_timestamp;
    std::string color_file_path;
};

struct Frame {  // tum_rgbd.rs:55-61
    double timestamp;
    Iso3 pose;
};

using vors::rust_display;  // Rust `{}` for floats (tracker.hpp)

// tum_rgbd.rs:78-85
inline std::string to_string(const Frame& f) {
    std::string s = rust_display(f.timestamp);
    for (int k = 0; k < 7; ++k) s += " " + rust_display(f.pose[k]);
    return s;
}

namespace parse {
// nom's `double`: optional sign, digits with optional fraction (or fraction alone), optional exponent. Returns chars consumed.
inline size_t recognize_float(const std::string& s, size_t p) {
    const size_t start = p;
    if (p < s.size() && (s[p] == '+' || s[p] == '-')) ++p;
    size_t digits = 0;
    while (p < s.size() && std::isdigit((unsigned char)s[p])) ++p, ++digits;
    if (p < s.size() && s[p] == '.') {
        ++p;
        while (p < s.size() && std::isdigit((unsigned char)s[p])) ++p, ++digits;
    }
    if (digits == 0) return 0;
    if (p < s.size() && (s[p] == 'e' || s[p] == 'E')) {
        size_t q = p + 1;
        if (q < s.size() && (s[q] == '+' || s[q] == '-')) ++q;
        size_t ed = 0;
        while (q < s.size() && std::isdigit((unsigned char)s[q])) ++q, ++ed;
        if (ed > 0) p = q;
    }
    return p - start;
}
inline bool parse_double(const std::string& s, size_t& p, double& out) {
    const size_t n = recognize_float(s, p);
    if (n == 0) return false;
    out = std::strtod(s.substr(p, n).c_str(), nullptr);
    p += n;
    return true;
}
inline bool parse_space(const std::string& s, size_t& p) {  // nom `space`: one or more of ' ' '\t'
    const size_t start = p;
    while (p < s.size() && (s[p] == ' ' || s[p] == '\t')) ++p;
    return p > start;
}
inline bool parse_path(const std::string& s, size_t& p, std::string& out) {  // is_not!(" \t\r\n")
    const size_t start = p;
    while (p < s.size() && s[p] != ' ' && s[p] != '\t' && s[p] != '\r' && s[p] != '\n') ++p;
    if (p == start) return false;
    out = s.substr(start, p - start);
    return true;
}
// One line: nullopt-with-ok for a comment; false = parse error.
inline bool association_line(const std::string& line, std::optional<Association>& out) {
    out.reset();
    if (!line.empty() && line[0] == '#') return true;  // comment: tag!("#") >> many0!(anychar)
    Association a;
    size_t p = 0;
    if (!parse_double(line, p, a.depth_timestamp) || !parse_space(line, p) || !parse_path(line, p, a.depth_file_path) ||
        !parse_space(line, p) || !parse_double(line, p, a.color_timestamp) || !parse_space(line, p) ||
        !parse_path(line, p, a.color_file_path))
        return false;
    out = a;  // trailing characters are left unparsed, like nom's remaining input
    return true;
}
// tum_rgbd.rs:97-121: Err("Parsing error") as soon as one line fails. `str::lines()` splits on \n and strips a trailing \r.
inline bool associations(const std::string& content, std::vector<Association>& out, std::string& err) {
    out.clear();
    size_t pos = 0;
    while (pos < content.size()) {
        size_t nl = content.find('\n', pos);
        std::string line = content.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = (nl == std::string::npos) ? content.size() : nl + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::optional<Association> a;
        if (!association_line(line, a)) {
            err = "Parsing error";
            return false;
        }
        if (a) out.push_back(*a);
    }
    return true;
}
// Trajectory line: comment or `timestamp tx ty tz qx qy qz qw` (tum_rgbd.rs:141-194). The rotation goes through
// UnitQuaternion::from_quaternion, i.e. it is normalised (tum_rgbd.rs:191).
inline bool parse_float(const std::string& s, size_t& p, float& out) {
    double d;
    if (!parse_double(s, p, d)) return false;
    out = (float)d;
    return true;
}
inline bool trajectory_line(const std::string& line, std::optional<Frame>& out) {
    out.reset();
    if (!line.empty() && line[0] == '#') return true;
    Frame f;
    size_t p = 0;
    float v[7];
    if (!parse_double(line, p, f.timestamp)) return false;
    for (int k = 0; k < 7; ++k)
        if (!parse_space(line, p) || !parse_float(line, p, v[k])) return false;
    // nalgebra: q / sqrt(norm_squared), 4-vector dot special case (a + c) + (b + d)
    const float a = v[3] * v[3] + v[5] * v[5], b = v[4] * v[4] + v[6] * v[6];
    const float n = std::sqrt(a + b);
    f.pose = Iso3{v[0], v[1], v[2], v[3] / n, v[4] / n, v[5] / n, v[6] / n};
    out = f;
    return true;
}
// tum_rgbd.rs:102-105
inline bool trajectory(const std::string& content, std::vector<Frame>& out, std::string& err) {
    out.clear();
    size_t pos = 0;
    while (pos < content.size()) {
        size_t nl = content.find('\n', pos);
        std::string line = content.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = (nl == std::string::npos) ? content.size() : nl + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::optional<Frame> f;
        if (!trajectory_line(line, f)) {
            err = "Parsing error";
            return false;
        }
        if (f) out.push_back(*f);
    }
    return true;
}
}  // namespace parse

}  // namespace tum_rgbd
}  // namespace vors
