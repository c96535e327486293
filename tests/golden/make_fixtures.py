#!/usr/bin/env python3
"""Creates the data fixtures under tests/golden/ from files the reference repository holds.

Run once in a container where /root/reference exists; the outputs are committed.  Every output is DATA
(an image the reference renderer produced, or an input texture it ships), not source text.

  reference_render_top200.png  rows 0..199 of doc/results/render_d94b533_1024_3_25-11-2024_202142.png, the
                               reference's own 500x500 / 1024 spp / 3 bounce render of
                               assets/cornell_box/cornell_box.json (box + glass bunny + two spheres + env map)
  reference_render_bottom60.png  rows 440..499 of the same render: the textured floor in front of the bunny and the spheres
  reference_render_mirror_sphere.png  columns 60..169, rows 320..459 of the same render: the SPECULAR (roughness 0) sphere of
                               cornell_box.json with the red wall, the floor and the environment map in it
  reference_render_glass_sphere.png   columns 310..479, rows 330..499: the TRANSMISSIVE sphere with the refracted, inverted room
  uv-test.png                  assets/textures/uv-test.png (the floor's map_Kd), unchanged
  prater_448x224.png           assets/textures/prater.jpg (the environment map) box-filtered 8x down
"""
import os
import shutil

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

im = Image.open(os.path.join(REF, "doc/results/render_d94b533_1024_3_25-11-2024_202142.png")).convert("RGB")
im.crop((0, 0, 500, 200)).save(os.path.join(OUT, "reference_render_top200.png"), optimize=True)
im.crop((0, 440, 500, 500)).save(os.path.join(OUT, "reference_render_bottom60.png"), optimize=True)
im.crop((60, 320, 170, 460)).save(os.path.join(OUT, "reference_render_mirror_sphere.png"), optimize=True)
im.crop((310, 330, 480, 500)).save(os.path.join(OUT, "reference_render_glass_sphere.png"), optimize=True)
shutil.copyfile(os.path.join(REF, "assets/textures/uv-test.png"), os.path.join(OUT, "uv-test.png"))
env = np.asarray(Image.open(os.path.join(REF, "assets/textures/prater.jpg")).convert("RGB"), np.float64)
h, w, _ = env.shape
small = env.reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))
Image.fromarray(np.clip(small + 0.5, 0, 255).astype(np.uint8)).save(os.path.join(OUT, "prater_448x224.png"), optimize=True)
print("fixtures written to", OUT)
