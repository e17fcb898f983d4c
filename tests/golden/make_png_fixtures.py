"""Writes the PNG fixtures under tests/golden/png/ with Pillow — an encoder INDEPENDENT of visual-odometry-rs_amd/host/png_io.hpp
(which until now had only met files of its own filter-0 writer). Pillow's encoder filters adaptively, so the natural-looking
test images below come out with a mix of PNG filter types 0-4 (the script prints the histogram and insists on all of 1..4).
Next to every NAME.png the expected pixels are stored as raw row-major bytes straight from the numpy SOURCE arrays (never
through a decoder): NAME.u8 (8-bit luma / grey) or NAME.u16le (16-bit depth, little-endian).

Expected luma of colour files = image 0.19's `rgb_to_luma` (src/color.rs of that crate: l = 0.2126 r + 0.7152 g + 0.0722 b in
f32, NumCast to u8 = truncation), which is what `image::open(..).to_luma()` (reference src/bin/vors_track.rs:143) applies;
alpha is ignored; palette entries are expanded to RGB first; grey of 1/2/4 bits is expanded to 8 bits by the png crate's EXPAND
transformation (v * 255 / (2^bits - 1)).

Run once in the build container (needs Pillow; the tests need only the bytes):  python tests/golden/make_png_fixtures.py
"""
import os
import struct
import zlib

import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "png")


def luma(rgb):
    r, g, b = (rgb[..., k].astype(np.float32) for k in range(3))
    l = np.float32(0.2126) * r + np.float32(0.7152) * g + np.float32(0.0722) * b
    return l.astype(np.uint8)  # truncation


def filter_histogram(path):
    data = open(path, "rb").read()
    p, idat, ihdr = 8, b"", None
    while p < len(data):
        n, t = struct.unpack(">I4s", data[p:p + 8])
        if t == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data[p + 8:p + 8 + n])
        if t == b"IDAT":
            idat += data[p + 8:p + 8 + n]
        p += 12 + n
    w, h, depth, ct = ihdr[:4]
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ct]
    stride = (w * ch * depth + 7) // 8
    raw = zlib.decompress(idat)
    hist = [0] * 5
    for y in range(h):
        hist[raw[y * (stride + 1)]] += 1
    return hist


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20260929)
    h, w = 48, 61
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    smooth = 128 + 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + 30 * np.sin((xx + yy) / 3.0)
    noise = rng.normal(0, 6, (h, w))
    grey = np.clip(smooth + noise, 0, 255).astype(np.uint8)
    grey[10:20, 15:40] = 200  # flat patch (filter 0/1/2 territory)
    grey[30:, :] = (np.arange(w, dtype=np.uint8) * 4)[None, :]  # horizontal ramp (Sub), constant down the rows (Up)
    rgb = np.stack([grey, np.clip(255 - smooth + noise, 0, 255).astype(np.uint8),
                    np.clip(90 + 40 * np.cos(yy / 4.0) + rng.normal(0, 10, (h, w)), 0, 255).astype(np.uint8)], axis=-1)
    depth = np.clip(9000 + 2500 * np.sin(xx / 11.0) + 40 * yy + rng.normal(0, 3, (h, w)), 0, 65535).astype(np.uint16)
    depth[rng.random((h, w)) < 0.03] = 0
    alpha = rng.integers(0, 256, (h, w), dtype=np.uint8)

    files = {}
    Image.fromarray(grey, "L").save(os.path.join(OUT, "grey8.png"), optimize=True)
    files["grey8"] = grey
    Image.fromarray(rgb, "RGB").save(os.path.join(OUT, "rgb8.png"), optimize=True)
    files["rgb8"] = luma(rgb)
    Image.fromarray(np.dstack([rgb, alpha]), "RGBA").save(os.path.join(OUT, "rgba8.png"))
    files["rgba8"] = luma(rgb)
    Image.fromarray(np.dstack([grey, alpha]), "LA").save(os.path.join(OUT, "greyalpha8.png"))
    files["greyalpha8"] = grey
    Image.fromarray(depth.astype(np.uint16)).save(os.path.join(OUT, "depth16.png"))  # mode I;16 -> 16-bit grey, big-endian samples
    files["depth16"] = depth
    pal_img = Image.fromarray(rgb, "RGB").quantize(colors=37, method=Image.Quantize.MEDIANCUT)
    pal_img.save(os.path.join(OUT, "palette8.png"))
    files["palette8"] = luma(np.asarray(pal_img.convert("RGB")))
    pal4 = Image.fromarray(rgb, "RGB").quantize(colors=13, method=Image.Quantize.MEDIANCUT)
    pal4.save(os.path.join(OUT, "palette4.png"), bits=4)
    files["palette4"] = luma(np.asarray(pal4.convert("RGB")))
    bw = (grey > 128)
    Image.fromarray(bw).convert("1").save(os.path.join(OUT, "grey1.png"))
    files["grey1"] = (bw * 255).astype(np.uint8)

    total = [0] * 5
    for name, arr in files.items():
        path = os.path.join(OUT, name + ".png")
        im = Image.open(path)
        hist = filter_histogram(path)
        total = [a + b for a, b in zip(total, hist)]
        if arr.dtype == np.uint16:
            arr.astype("<u2").tofile(os.path.join(OUT, name + ".u16le"))
        else:
            arr.tofile(os.path.join(OUT, name + ".u8"))
        print(f"{name}.png: mode {im.mode}, {os.path.getsize(path)} B, PNG filter types per row [0..4] = {hist}")
    assert all(t > 0 for t in total[1:]), f"fixtures must exercise filters 1-4: {total}"
    print("all filter types present:", total)


if __name__ == "__main__":
    main()
