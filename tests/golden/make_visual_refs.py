#!/usr/bin/env python3
"""Box-downsample the two rendered stills the reference ships (its only outputs: it has no tests
and no golden vectors) into small fixtures for tests/test_visual_reference.py:

    /root/reference/images/glossy_reflections.png  (3200x1800, the bench5 scene, Main.hs:210-219)
    /root/reference/images/depthoffield10.png      (3200x1800, the bench4 scene, Main.hs:195-204)

Each 10x10 block of the original is averaged to one pixel -> 320x180 8-bit RGB PNGs.  These are
DATA (images the reference produced), not source.

And from /root/reference/images/spheres.gif (100 frames, 800x600: the animation of the sample scene
description `sdl`, three spheres orbiting a glass one) the OBJECT COVERAGE of four frames: the
fraction of each 4x4 block that is not background (background is exactly 0 in the GIF) as a 200x150
8-bit grey PNG -- what tests/test_visual_reference.py pins the scene-description parser, the
expression evaluator and the camera with (the frames' shading predates the current material model).  Run here, where /root/reference exists; the
GPU box only sees the committed PNGs."""
import os
import sys

import numpy as np
from PIL import Image

SRC = "/root/reference/images"
HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 320, 180


def main():
    for name in ("glossy_reflections", "depthoffield10"):
        im = Image.open(os.path.join(SRC, name + ".png")).convert("RGB")
        assert im.size == (3200, 1800), im.size
        a = np.asarray(im, dtype=np.float64)
        f = im.size[0] // W
        small = a.reshape(H, f, W, f, 3).mean(axis=(1, 3))
        out = os.path.join(HERE, f"ref_{name}_{W}x{H}.png")
        Image.fromarray(np.floor(small + 0.5).astype(np.uint8)).save(out, optimize=True)
        print(out, os.path.getsize(out), "bytes")
    gif = Image.open(os.path.join(SRC, "spheres.gif"))
    assert gif.size == (800, 600) and gif.n_frames == 100
    for k in (0, 24, 59, 98):
        gif.seek(k)
        obj = np.asarray(gif.convert("RGB"), dtype=np.int32).max(axis=2) > 0
        cov = obj.reshape(150, 4, 200, 4).mean(axis=(1, 3))          # multiples of 1/16
        out = os.path.join(HERE, f"ref_spheres_gif_coverage_frame{k:02d}_200x150.png")
        Image.fromarray(np.floor(cov * 255 + 0.5).astype(np.uint8), mode="L").save(out, optimize=True)
        print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
