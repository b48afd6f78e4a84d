"""Decodes the reference's data/mask_0000{0..5}.png (copied verbatim into tests/golden/bunny/)
to raw 8-bit arrays, so that neither the tests nor the GPU box need a PNG decoder.
Run once in the build container:  python tests/golden/make_bunny_masks.py"""
import os

import numpy as np
from PIL import Image

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bunny")
masks = []
for i in range(6):
    im = Image.open(os.path.join(here, "mask_%05d.png" % i))
    assert im.mode == "L" and im.size == (320, 240), (im.mode, im.size)
    masks.append(np.asarray(im, dtype=np.uint8))
masks = np.stack(masks)
assert set(np.unique(masks)) <= {0, 255}
np.savez_compressed(os.path.join(here, "masks.npz"), masks=masks)
print("wrote masks.npz", masks.shape, masks.sum())
