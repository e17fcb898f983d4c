// dump_golden.rs — an `examples/`-style program for the REFERENCE crate (mpizenberg/visual-odometry-rs), written by this repository.
//
// It runs the reference's OWN code on the raw inputs of tests/golden/*.npz (exported by tests/golden/export_rust_inputs.py to
// tests/golden/rust_inputs/<case>/) and prints one JSON document with every f32 as its bit pattern. tests/test_golden_rust.py compares
// that document with this repository's oracle bit for bit — the one missing piece of evidence (`parity unpinned`): nothing
// the Rust reference ever produced pinned the tracker, because no Rust toolchain exists where this repository is built.
//
// NOT COMPILED HERE (no cargo / rustc in the build image). It uses the crate's PUBLIC API only, so it needs no patch of the reference:
//
//     cp tools/rust_golden/dump_golden.rs  <reference checkout>/examples/dump_golden.rs
//     cd <reference checkout>
//     for c in sparse_128x96_L4 sparse_odd_167x123_L3 dense_80x60_L3; do
//         cargo run --release --example dump_golden -- <this repo>/tests/golden/rust_inputs/$c > <this repo>/tests/golden/rust/$c.json
//     done
//     cd <this repo> && python -m pytest tests/test_golden_rust.py -q
//
// What it pins, per case:
//   * `poses`      Config::init + Tracker::track + Tracker::current_frame for every pair (src/bin/vors_track.rs:46-62 does exactly this):
//                  the whole private pipeline end to end — candidates, inverse-depth pyramid, extract_z's order, Jacobians, the LM loop,
//                  the pose composition. (Candidate modes other than coarse-to-fine do not exist in the reference: skipped for them.)
//   * `pyramid`    multires::mean_pyramid of the first keyframe: FNV-1a of every level.
//   * `mask0`      the level-0 candidate mask from the public functions precompute_multires_data calls (multires::gradients_xy,
//                  gradient::centered, gradient::squared_norm, candidates::coarse_to_fine::select), row-major '0'/'1'.
//   * `idepth`     the inverse-depth pyramid from the public functions it calls (helper::zip_mask_map, inverse_depth::from_depth, fuse +
//                  strategy_dso_mean through multires::limited_sequence / halve): per level the number of known values and the FNV-1a
//                  of their bits in DMatrix iteration order (= extract_z's order).
//   * `lm`         LMOptimizerState::iterative_solve level by level (coarse to fine, the model carried over like Tracker::track does) on
//                  the observation lists of pair 0 that the manifest carries (coordinates, inverse depths, Jacobians; Hessians = j * j^T):
//                  iteration count, final model, energy and lm_coef of every level — Cholesky, se3::exp, warp, interpolate, eval, step and
//                  stop_criterion on known inputs.

extern crate nalgebra as na;
extern crate visual_odometry_rs as vors;

use std::{env, fs, path::Path, process::exit};

use na::DMatrix;
use vors::core::camera::Intrinsics;
use vors::core::candidates::coarse_to_fine as candidates;
use vors::core::inverse_depth::{self, InverseDepth};
use vors::core::track::inverse_compositional as track;
use vors::core::track::lm_optimizer::{self, LMOptimizerState};
use vors::core::{gradient, multires};
use vors::math::optimizer::State as _;
use vors::misc::helper;
use vors::misc::type_aliases::{Float, Iso3, Mat6, Vec6};

fn main() {
    let args: Vec<String> = env::args().collect();
    if args.len() != 2 {
        eprintln!("Usage: cargo run --release --example dump_golden -- <tests/golden/rust_inputs/CASE directory>");
        exit(1);
    }
    run(Path::new(&args[1]));
}

// ---------------------------------------------------------------------------------------------------- manifest + raw files
struct Manifest {
    case: String,
    rows: usize,
    cols: usize,
    levels: usize,
    mode: usize,
    thresh: u16,
    pairs: usize,
    depth_scale: Float,
    idepth_variance: Float,
    intrinsics: [Float; 5], // cu cv fu fv skew
    level_n: Vec<usize>,
}

fn f32_from_hex(s: &str) -> Float {
    Float::from_bits(u32::from_str_radix(s, 16).expect("hex f32"))
}

fn read_manifest(dir: &Path) -> Manifest {
    let text = fs::read_to_string(dir.join("manifest.txt")).expect("manifest.txt");
    let mut m = Manifest {
        case: String::new(),
        rows: 0,
        cols: 0,
        levels: 0,
        mode: 0,
        thresh: 0,
        pairs: 0,
        depth_scale: 0.0,
        idepth_variance: 0.0,
        intrinsics: [0.0; 5],
        level_n: Vec::new(),
    };
    for line in text.lines() {
        let w: Vec<&str> = line.split_whitespace().collect();
        if w.is_empty() {
            continue;
        }
        match w[0] {
            "case" => m.case = w[1].to_string(),
            "rows" => m.rows = w[1].parse().unwrap(),
            "cols" => m.cols = w[1].parse().unwrap(),
            "levels" => m.levels = w[1].parse().unwrap(),
            "mode" => m.mode = w[1].parse().unwrap(),
            "thresh" => m.thresh = w[1].parse().unwrap(),
            "pairs" => m.pairs = w[1].parse().unwrap(),
            "depth_scale_f32" => m.depth_scale = f32_from_hex(w[1]),
            "idepth_variance_f32" => m.idepth_variance = f32_from_hex(w[1]),
            "intrinsics_f32" => {
                for k in 0..5 {
                    m.intrinsics[k] = f32_from_hex(w[1 + k]);
                }
            }
            "level" => m.level_n.push(w[3].parse().unwrap()), // "level L n N" in level order
            _ => {}
        }
    }
    m
}

fn read_u8_images(path: &Path, n: usize, rows: usize, cols: usize) -> Vec<DMatrix<u8>> {
    let bytes = fs::read(path).expect("image file");
    assert_eq!(bytes.len(), n * rows * cols);
    (0..n)
        .map(|p| DMatrix::from_row_slice(rows, cols, &bytes[p * rows * cols..(p + 1) * rows * cols]))
        .collect()
}

fn read_u16_images(path: &Path, n: usize, rows: usize, cols: usize) -> Vec<DMatrix<u16>> {
    let bytes = fs::read(path).expect("depth file");
    assert_eq!(bytes.len(), 2 * n * rows * cols);
    let words: Vec<u16> = bytes
        .chunks(2)
        .map(|b| u16::from(b[0]) | (u16::from(b[1]) << 8))
        .collect();
    (0..n)
        .map(|p| DMatrix::from_row_slice(rows, cols, &words[p * rows * cols..(p + 1) * rows * cols]))
        .collect()
}

fn read_i32(path: &Path) -> Vec<i32> {
    fs::read(path)
        .expect("i32 file")
        .chunks(4)
        .map(|b| (u32::from(b[0]) | (u32::from(b[1]) << 8) | (u32::from(b[2]) << 16) | (u32::from(b[3]) << 24)) as i32)
        .collect()
}

fn read_f32(path: &Path) -> Vec<Float> {
    fs::read(path)
        .expect("f32 file")
        .chunks(4)
        .map(|b| Float::from_bits(u32::from(b[0]) | (u32::from(b[1]) << 8) | (u32::from(b[2]) << 16) | (u32::from(b[3]) << 24)))
        .collect()
}

// ---------------------------------------------------------------------------------------------------- output helpers
fn hex(x: Float) -> String {
    format!("\"{:08x}\"", x.to_bits())
}

fn iso_hex(iso: &Iso3) -> String {
    // translation x y z, then the unit quaternion's coordinates in nalgebra's storage order i j k w (tum_rgbd.rs:78-85 prints them so)
    let t = iso.translation.vector;
    let q = iso.rotation.coords;
    format!(
        "[{}, {}, {}, {}, {}, {}, {}]",
        hex(t[0]),
        hex(t[1]),
        hex(t[2]),
        hex(q[0]),
        hex(q[1]),
        hex(q[2]),
        hex(q[3])
    )
}

fn fnv1a(hash: u64, byte: u8) -> u64 {
    (hash ^ u64::from(byte)).wrapping_mul(0x0000_0100_0000_01b3)
}
const FNV_OFFSET: u64 = 0xcbf2_9ce4_8422_2325;

fn fnv_u32(mut hash: u64, word: u32) -> u64 {
    for k in 0..4 {
        hash = fnv1a(hash, ((word >> (8 * k)) & 0xff) as u8);
    }
    hash
}

// ---------------------------------------------------------------------------------------------------- the dump
fn run(dir: &Path) {
    let m = read_manifest(dir);
    let kf_gray = read_u8_images(&dir.join("kf_gray.bin"), m.pairs, m.rows, m.cols);
    let cur_gray = read_u8_images(&dir.join("cur_gray.bin"), m.pairs, m.rows, m.cols);
    let kf_depth = read_u16_images(&dir.join("kf_depth.bin"), m.pairs, m.rows, m.cols);
    let cur_depth = read_u16_images(&dir.join("cur_depth.bin"), m.pairs, m.rows, m.cols);
    let intrinsics = Intrinsics {
        principal_point: (m.intrinsics[0], m.intrinsics[1]),
        focal: (m.intrinsics[2], m.intrinsics[3]),
        skew: m.intrinsics[4],
    };

    println!("{{");
    println!("  \"case\": \"{}\",", m.case);
    println!("  \"generator\": \"tools/rust_golden/dump_golden.rs on the reference crate\",");

    // ---- poses through the public Tracker API (coarse-to-fine candidates = the reference's only mode)
    if m.mode == 0 {
        let mut poses = Vec::new();
        for p in 0..m.pairs {
            let config = track::Config {
                nb_levels: m.levels,
                candidates_diff_threshold: m.thresh,
                depth_scale: m.depth_scale,
                intrinsics: intrinsics.clone(),
                idepth_variance: m.idepth_variance,
            };
            let mut tracker = config.init(0.0, &kf_depth[p], 0.0, kf_gray[p].clone());
            tracker.track(1.0, &cur_depth[p], 1.0, cur_gray[p].clone());
            let (_, pose) = tracker.current_frame();
            poses.push(iso_hex(&pose));
        }
        println!("  \"poses\": [{}],", poses.join(", "));
    }

    // ---- pyramid of the first keyframe and of the first current frame
    let kf_pyr = multires::mean_pyramid(m.levels, kf_gray[0].clone());
    let cur_pyr = multires::mean_pyramid(m.levels, cur_gray[0].clone());
    let pyr_hash: Vec<String> = kf_pyr
        .iter()
        .map(|img| {
            // row-major walk, so that the hash does not depend on the storage order
            let (r, c) = img.shape();
            let mut h = FNV_OFFSET;
            for i in 0..r {
                for j in 0..c {
                    h = fnv1a(h, img[(i, j)]);
                }
            }
            format!("\"{:016x}\"", h)
        })
        .collect();
    println!("  \"pyramid\": [{}],", pyr_hash.join(", "));

    // ---- level-0 candidates and the inverse-depth pyramid, with the public functions precompute_multires_data is made of
    if m.mode == 0 {
        let mut gradients = multires::gradients_xy(&kf_pyr);
        gradients.insert(0, gradient::centered(&kf_pyr[0]));
        let norms: Vec<_> = gradients.iter().map(|(gx, gy)| gradient::squared_norm(gx, gy)).collect();
        let mask = candidates::select(m.thresh, &norms).pop().unwrap();
        let (r, c) = mask.shape();
        let mut bits = String::with_capacity(r * c);
        for i in 0..r {
            for j in 0..c {
                bits.push(if mask[(i, j)] { '1' } else { '0' });
            }
        }
        println!("  \"mask0\": \"{}\",", bits);

        let scale = m.depth_scale;
        let variance = m.idepth_variance;
        let from_depth = |z| inverse_depth::from_depth(scale, z, variance);
        let idepth0 = helper::zip_mask_map(&kf_depth[0], &mask, InverseDepth::Unknown, from_depth);
        let fuse = |a, b, c, d| inverse_depth::fuse(a, b, c, d, inverse_depth::strategy_dso_mean);
        let idepth_pyr = multires::limited_sequence(m.levels, idepth0, |mat| multires::halve(mat, fuse));
        let mut entries = Vec::new();
        for mat in &idepth_pyr {
            let mut n = 0usize;
            let mut h = FNV_OFFSET;
            for idepth in mat.iter() {
                if let InverseDepth::WithVariance(z, _) = *idepth {
                    n += 1;
                    h = fnv_u32(h, z.to_bits());
                }
            }
            entries.push(format!("{{\"n\": {}, \"fnv\": \"{:016x}\"}}", n, h));
        }
        println!("  \"idepth\": [{}],", entries.join(", "));
    }

    // ---- the LM loop level by level on the manifest's observation lists of pair 0
    let intrinsics_multires = intrinsics.clone().multi_res(m.levels);
    let mut model = Iso3::identity();
    let mut lm = Vec::new();
    for lvl in (0..m.levels).rev() {
        let n = m.level_n[lvl];
        let xy = read_i32(&dir.join(format!("xy{}.bin", lvl)));
        let iz = read_f32(&dir.join(format!("iz{}.bin", lvl)));
        let jac = read_f32(&dir.join(format!("jac{}.bin", lvl)));
        assert_eq!(xy.len(), 2 * n);
        assert_eq!(iz.len(), n);
        assert_eq!(jac.len(), 6 * n);
        let coordinates: Vec<(usize, usize)> = (0..n).map(|i| (xy[2 * i] as usize, xy[2 * i + 1] as usize)).collect();
        let jacobians: Vec<Vec6> = (0..n)
            .map(|i| Vec6::new(jac[6 * i], jac[6 * i + 1], jac[6 * i + 2], jac[6 * i + 3], jac[6 * i + 4], jac[6 * i + 5]))
            .collect();
        let hessians: Vec<Mat6> = jacobians.iter().map(|j| j * j.transpose()).collect();
        let obs = lm_optimizer::Obs {
            intrinsics: &intrinsics_multires[lvl],
            template: &kf_pyr[lvl],
            image: &cur_pyr[lvl],
            coordinates: &coordinates,
            _z_candidates: &iz,
            jacobians: &jacobians,
            hessians: &hessians,
        };
        match LMOptimizerState::iterative_solve(&obs, model) {
            Ok((state, nb_iter)) => {
                model = state.eval_data.model;
                lm.push(format!(
                    "{{\"level\": {}, \"nb_iter\": {}, \"model\": {}, \"energy\": {}, \"lm_coef\": {}}}",
                    lvl,
                    nb_iter,
                    iso_hex(&model),
                    hex(state.eval_data.energy),
                    hex(state.lm_coef)
                ));
            }
            Err(err) => {
                lm.push(format!("{{\"level\": {}, \"error\": \"{}\"}}", lvl, err));
                break;
            }
        }
    }
    println!("  \"lm\": [{}]", lm.join(", "));
    println!("}}");
}
