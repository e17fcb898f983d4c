// CPU-only self-test of the TUM plumbing (no GPU, no libvors_hip call): parser grammar, Rust float Display, PNG round trips.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "png_io.hpp"
#include "tum_rgbd.hpp"

using namespace vors;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
    // ---- Rust Display of floats
    CHECK(tum_rgbd::rust_display(1.0f) == "1");
    CHECK(tum_rgbd::rust_display(-0.5f) == "-0.5");
    CHECK(tum_rgbd::rust_display(0.1f) == "0.1");
    CHECK(tum_rgbd::rust_display(1305031102.160407) == "1305031102.160407");
    CHECK(tum_rgbd::rust_display(1e-7f) == "0.0000001");
    CHECK(tum_rgbd::rust_display(16777216.0f) == "16777216");
    CHECK(tum_rgbd::rust_display(std::nanf("")) == "NaN");
    CHECK(tum_rgbd::to_string(tum_rgbd::Frame{1.5, {0.f, 1.f, -2.25f, 0.f, 0.f, 0.f, 1.f}}) == "1.5 0 1 -2.25 0 0 0 1");
    // ---- associations grammar (examples/README.md:27-31 of the reference)
    const std::string content =
        "# depth_timestamp depth_file_path rgb_timestamp rgb_file_path\n"
        "1305031102.160407 depth/1305031102.160407.png 1305031102.175304 rgb/1305031102.175304.png\n"
        "1305031102.226738\tdepth/1305031102.226738.png  1305031102.211214 rgb/1305031102.211214.png trailing junk\r\n";
    std::vector<tum_rgbd::Association> a;
    std::string err;
    CHECK(tum_rgbd::parse::associations(content, a, err));
    CHECK(a.size() == 2);
    CHECK(a[0].depth_timestamp == 1305031102.160407 && a[0].depth_file_path == "depth/1305031102.160407.png");
    CHECK(a[0].color_timestamp == 1305031102.175304 && a[0].color_file_path == "rgb/1305031102.175304.png");
    CHECK(a[1].color_file_path == "rgb/1305031102.211214.png");
    CHECK(!tum_rgbd::parse::associations("1.0 a 2.0 b\n\n3.0 c 4.0 d\n", a, err) && err == "Parsing error");  // blank line
    CHECK(!tum_rgbd::parse::associations(" 1.0 a 2.0 b\n", a, err));                                             // leading space
    CHECK(!tum_rgbd::parse::associations("1.0 a 2.0\n", a, err));                                                // missing field
    CHECK(tum_rgbd::parse::associations("", a, err) && a.empty());
    CHECK(tum_rgbd::parse::associations("1e3 a -2.5E-1 b", a, err) && a.size() == 1 && a[0].depth_timestamp == 1000.0 && a[0].color_timestamp == -0.25);
    // ---- trajectory grammar (examples/README.md:41-45 of the reference)
    {
        std::vector<tum_rgbd::Frame> fr;
        CHECK(tum_rgbd::parse::trajectory("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\n"
                                          "1305031098.6659 1.3563 0.6305 1.6380 0.6132 0.5962 -0.3311 -0.3986\n", fr, err));
        CHECK(fr.size() == 1 && fr[0].timestamp == 1305031098.6659 && fr[0].pose[0] == 1.3563f);
        const float n2 = fr[0].pose[3] * fr[0].pose[3] + fr[0].pose[4] * fr[0].pose[4] + fr[0].pose[5] * fr[0].pose[5] + fr[0].pose[6] * fr[0].pose[6];
        CHECK(std::fabs(n2 - 1.0f) < 1e-6f);  // rotation normalised like UnitQuaternion::from_quaternion
        CHECK(!tum_rgbd::parse::trajectory("1.0 0 0 0 0 0 0\n", fr, err));  // 7 numbers instead of 8
        // write -> parse round trip of a trajectory line
        const tum_rgbd::Frame f0{2.5, {0.1f, -0.2f, 0.3f, 0.0f, 0.0f, 0.6f, 0.8f}};
        CHECK(tum_rgbd::parse::trajectory(tum_rgbd::to_string(f0) + "\n", fr, err) && fr.size() == 1);
        for (int k = 0; k < 7; ++k) CHECK(std::fabs(fr[0].pose[k] - f0.pose[k]) < 1e-6f);
    }
    // ---- PNG round trips
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const uint32_t w = 37, h = 23;
    std::vector<uint16_t> d(w * h);
    std::vector<uint8_t> g(w * h), rgb(w * h * 3);
    for (uint32_t i = 0; i < w * h; ++i) {
        d[i] = (uint16_t)(i * 2654435761u >> 16);
        g[i] = (uint8_t)(i * 40503u >> 8);
        rgb[3 * i] = (uint8_t)i; rgb[3 * i + 1] = (uint8_t)(i >> 1); rgb[3 * i + 2] = (uint8_t)(255 - i);
    }
    png_io::write_gray16(dir + "/vors_t_d.png", w, h, d.data());
    png_io::write_gray8(dir + "/vors_t_g.png", w, h, g.data());
    png_io::write_rgb8(dir + "/vors_t_c.png", w, h, rgb.data());
    uint32_t w2, h2;
    std::vector<uint16_t> d2;
    std::vector<uint8_t> g2, l2;
    png_io::read_png_16bits(dir + "/vors_t_d.png", w2, h2, d2);
    CHECK(w2 == w && h2 == h && d2 == d);
    png_io::read_luma8(dir + "/vors_t_g.png", w2, h2, g2);
    CHECK(g2 == g);
    png_io::read_luma8(dir + "/vors_t_c.png", w2, h2, l2);
    for (uint32_t i = 0; i < w * h; ++i) {
        const float l = 0.2126f * rgb[3 * i] + 0.7152f * rgb[3 * i + 1] + 0.0722f * rgb[3 * i + 2];
        CHECK(l2[i] == (uint8_t)l);
    }
    bool threw = false;
    try { png_io::read_png_16bits(dir + "/vors_t_g.png", w2, h2, d2); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
    // ---- decoding of files written by an independent encoder (Pillow; adaptive filters 0-4, see tests/golden/make_png_fixtures.py)
    if (argc > 2) {
        const std::string fx = argv[2];
        for (const char* name : {"grey8", "rgb8", "rgba8", "greyalpha8", "palette8", "palette4", "grey1"}) {
            std::vector<uint8_t> got;
            png_io::read_luma8(fx + "/" + name + ".png", w2, h2, got);
            const std::vector<uint8_t> want = png_io::read_file(fx + "/" + name + ".u8");
            if (got != want) {
                std::fprintf(stderr, "FAILED: %s.png decodes differently from %s.u8\n", name, name);
                return 1;
            }
        }
        std::vector<uint16_t> dd;
        png_io::read_png_16bits(fx + "/depth16.png", w2, h2, dd);
        const std::vector<uint8_t> want = png_io::read_file(fx + "/depth16.u16le");
        CHECK(want.size() == dd.size() * 2);
        for (size_t i = 0; i < dd.size(); ++i) CHECK(dd[i] == (uint16_t)(want[2 * i] | want[2 * i + 1] << 8));
        threw = false;
        try { std::vector<uint8_t> l; png_io::read_luma8(fx + "/depth16.png", w2, h2, l); } catch (const std::exception&) { threw = true; }
        CHECK(threw);  // like image 0.19: no 16-bit DynamicImage
        // hostile headers: short IHDR, absurd dimensions, truncated stream
        std::vector<uint8_t> f = png_io::read_file(fx + "/grey8.png");
        auto expect_throw = [&](std::vector<uint8_t> bad) {
            try { png_io::decode(bad); } catch (const std::exception&) { return true; }
            return false;
        };
        {
            std::vector<uint8_t> bad = f;
            bad[8 + 3] = 12;  // IHDR length 12
            CHECK(expect_throw(bad));
            bad = f;
            bad[16] = 0x7f;  // width = 2^31-ish
            CHECK(expect_throw(bad));
            bad = f;
            bad.resize(bad.size() / 2);
            CHECK(expect_throw(bad));
            bad = f;
            bad[24] = 3;  // bit depth 3
            CHECK(expect_throw(bad));
        }
        std::printf("png fixtures: ok\n");
    }
    std::printf("host_plumbing_test: ok\n");
    return 0;
}
