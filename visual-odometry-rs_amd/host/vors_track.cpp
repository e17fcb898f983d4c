// vors_track — C++ mirror of the reference's only binary, src/bin/vors_track.rs, on top of libvors_hip.so:
//   Usage: ./vors_track [fr1|fr2|fr3|icl] associations_file          (vors_track.rs:24)
// Reads a TUM RGB-D associations file, initialises the tracker with the first RGB-D frame, tracks every following frame
// and prints one trajectory line `timestamp tx ty tz qx qy qz qw` per tracked frame on stdout (vors_track.rs:46-64).
// Optional trailing flags (not in the reference): `--quiet` silences the per-frame stderr logs; `--arith exact|fused|reference` selects the
// per-point arithmetic (include/vors_hip.h VORS_ARITH_*, default reference); `--candidates c2f|dense|dso` the level-0 mask source
// (default c2f = the reference's coarse-to-fine selection).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "png_io.hpp"
#include "tum_rgbd.hpp"

using namespace vors;

static const char* USAGE = "Usage: ./vors_track [fr1|fr2|fr3|icl] associations_file";

static bool create_camera(const std::string& id, Intrinsics& out) {  // vors_track.rs:99-110
    if (id == "fr1") out = tum_rgbd::INTRINSICS_FR1();
    else if (id == "fr2") out = tum_rgbd::INTRINSICS_FR2();
    else if (id == "fr3") out = tum_rgbd::INTRINSICS_FR3();
    else if (id == "icl") out = tum_rgbd::INTRINSICS_ICL_NUIM();
    else return false;
    return true;
}
static std::string parent_of(const std::string& path) {
    const size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string("") : path.substr(0, p);
}
static std::string join(const std::string& parent, const std::string& rel) {  // Path::join: an absolute `rel` replaces `parent`
    if (!rel.empty() && rel[0] == '/') return rel;
    return parent.empty() ? rel : parent + "/" + rel;
}

int main(int argc, char** argv) {
    bool quiet = false;
    int arithmetic = VORS_ARITH_REFERENCE, candidates = VORS_CANDIDATES_COARSE_TO_FINE;
    bool bad_flag = false;
    for (int a = 3; a < argc && !bad_flag; ++a) {  // extension flags follow the reference's two positional arguments
        const std::string flag = argv[a], val = a + 1 < argc ? argv[a + 1] : "";
        if (flag == "--quiet") {
            quiet = true;
        } else if (flag == "--arith" && (val == "exact" || val == "fused" || val == "reference")) {
            arithmetic = val == "fused" ? VORS_ARITH_FUSED : (val == "exact" ? VORS_ARITH_EXACT : VORS_ARITH_REFERENCE);
            ++a;
        } else if (flag == "--candidates" && (val == "c2f" || val == "dense" || val == "dso")) {
            candidates = val == "dense" ? VORS_CANDIDATES_DENSE : (val == "dso" ? VORS_CANDIDATES_DSO : VORS_CANDIDATES_COARSE_TO_FINE);
            ++a;
        } else {
            bad_flag = true;
        }
    }
    if (argc > 3) argc = bad_flag ? 0 : 3;
    if (argc != 3) {  // vors_track.rs:75-96
        std::fprintf(stderr, "%s\n\"Wrong number of arguments\"\n", USAGE);
        return 0;  // the reference's main() only prints the error (vors_track.rs:17-22)
    }
    Intrinsics intrinsics;
    if (!create_camera(argv[1], intrinsics)) {
        std::fprintf(stderr, "%s\n\"Unknown camera id: %s\"\n", USAGE, argv[1]);
        return 0;
    }
    std::ifstream f(argv[2]);
    if (!f.good()) {
        std::fprintf(stderr, "%s\n\"The association file does not exist or is not reachable: %s\"\n", USAGE, argv[2]);
        return 0;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    std::vector<tum_rgbd::Association> associations;
    std::string err;
    if (!tum_rgbd::parse::associations(ss.str(), associations, err)) {
        std::fprintf(stderr, "\"%s\"\n", err.c_str());
        return 0;
    }
    if (associations.empty()) {
        std::fprintf(stderr, "thread 'main' panicked at 'index out of bounds'\n");  // associations[0] (vors_track.rs:43)
        return 101;
    }
    const std::string parent = parent_of(argv[2]);  // vors_track.rs:125-138
    try {
        auto read_images = [&](const tum_rgbd::Association& a, std::vector<uint16_t>& depth, std::vector<uint8_t>& gray, uint32_t& w,
                               uint32_t& h) {  // vors_track.rs:140-145
            uint32_t w2, h2;
            png_io::read_png_16bits(join(parent, a.depth_file_path), w, h, depth);
            png_io::read_luma8(join(parent, a.color_file_path), w2, h2, gray);
            if (w2 != w || h2 != h) throw std::runtime_error("depth and colour images differ in size");
        };
        // vors_track.rs:34-40
        track::Config config{6, 7, tum_rgbd::DEPTH_SCALE, intrinsics, 0.0001f};
        config.arithmetic = arithmetic;
        config.candidates_mode = candidates;
        std::vector<uint16_t> depth;
        std::vector<uint8_t> gray;
        uint32_t w = 0, h = 0;
        read_images(associations[0], depth, gray, w, h);
        track::Tracker tracker = config.init(associations[0].depth_timestamp, {depth.data(), (int)h, (int)w, VORS_ROW_MAJOR},
                                             associations[0].color_timestamp, {gray.data(), (int)h, (int)w, VORS_ROW_MAJOR});
        tracker.set_logging(!quiet);
        for (size_t k = 1; k < associations.size(); ++k) {  // vors_track.rs:49-64
            uint32_t w2, h2;
            read_images(associations[k], depth, gray, w2, h2);
            if (w2 != w || h2 != h) throw std::runtime_error("image size changed inside the sequence");
            tracker.track(associations[k].depth_timestamp, {depth.data(), (int)h, (int)w, VORS_ROW_MAJOR}, associations[k].color_timestamp,
                          {gray.data(), (int)h, (int)w, VORS_ROW_MAJOR});
            const auto cf = tracker.current_frame();
            std::printf("%s\n", tum_rgbd::to_string(tum_rgbd::Frame{cf.first, cf.second}).c_str());
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "\"%s\"\n", e.what());
    }
    return 0;
}
