"""Reduces the only renders the reference itself holds in this renderer's display encoding to small tables of block
means (run in the build container only: /root/reference does not exist on the GPU box; PIL is importable here).

  /root/reference/sample_images/cbox.png          1024x1024 screenshot of scenes/cbox/cbox.xml (film resized from the XML's 512x512)
  /root/reference/sample_images/scene1_phong.png  1280x960 screenshot of scenes/spheres/scene1_spherical_light_phong.xml
  /root/reference/sample_images/bunny.png         640x480 screenshot of scenes/bunny/bunny.xml (its native film size)
  /root/reference/output/img.png                  1280x960 screenshot of scenes/spheres/scene4.xml (30 spheres, 8 of them mirrors)

All carry the display encoding of /root/reference/opengl_display.cpp:99-117 — per pixel
`int(255.99f * clamp(sqrtf(accum / samples), 0, 1))` — which is how their corner pixels come to be 181 =
int(255.99 * sqrt(0.5)) for the 0.5-grey default background (parse_scene.cpp:809, radiance.cuh:27-29) and 127 =
int(255.99 * sqrt(0.25)) for bunny.xml's own <background> of 0.25.  (sample_images/buddha.png and Dragon_1000.png are in
the same encoding, but their meshes are missing from the reference snapshot; party.png shows a scene
whose meshes are not in the snapshot either.)

What is written (tests/golden/reference_image_pins.json) is DATA, not the PNGs: per image a GY x GX grid of
  lin   block mean of the linearised pixels  ((d + 0.5) / 255.99)^2, saturated pixels (d = 255) counted as 1.0
  disp  block mean of the 8-bit display values d
  sat   fraction of saturated pixels in the block (such blocks compare loosely: the clamp acts per pixel, and a
        render at another resolution clamps other pixel footprints)
The tests (tests/test_reference_images.py) render the same scenes with the oracle / the HIP path, clamp each
pixel to [0, 1] as the display does, take the same block means and compare in display units."""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
IMAGES = {
    # name -> (png, scene fixture under tests/golden/scenes, grid rows, grid cols)
    "cbox": ("sample_images/cbox.png", "cbox", 32, 32),
    "scene1_phong": ("sample_images/scene1_phong.png", "scene1_phong", 24, 32),
    "bunny": ("sample_images/bunny.png", "bunny", 24, 32),
    "scene4": ("output/img.png", "scene4", 24, 32),
}


def blocks(a, gy, gx):
    h, w = a.shape[:2]
    assert h % gy == 0 and w % gx == 0
    return a.reshape(gy, h // gy, gx, w // gx, -1).mean(axis=(1, 3))


if __name__ == "__main__":
    out = {"_source": "block means of /root/reference/sample_images/*.png, made by tests/golden/make_reference_image_pins.py; "
                      "display encoding /root/reference/opengl_display.cpp:99-117", "images": {}}
    for name, (png, scene, gy, gx) in IMAGES.items():
        d = np.asarray(Image.open(os.path.join(REF, png)).convert("RGB")).astype(np.float64)
        lin = np.where(d >= 255, 1.0, ((d + 0.5) / 255.99) ** 2)
        out["images"][name] = {
            "png": png, "scene": scene, "width": d.shape[1], "height": d.shape[0], "grid": [gy, gx],
            "corner_pixel": [int(v) for v in d[0, 0]],
            "lin": np.round(blocks(lin, gy, gx), 6).tolist(),
            "disp": np.round(blocks(d, gy, gx), 3).tolist(),
            "sat": np.round(blocks((d >= 255).any(axis=2, keepdims=True).astype(np.float64), gy, gx)[..., 0], 4).tolist(),
        }
        print(name, d.shape, "corner", d[0, 0], "mean display", d.mean(axis=(0, 1)).round(2))
    path = os.path.join(HERE, "reference_image_pins.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")
