"""Writes tests/golden/preprocess_demo_000000.npz by running THE REFERENCE's DemoDataset (imported from /root/reference through
oracle/tools/ref_shims.py) on one demo frame, in the build container.  TEST INFRASTRUCTURE.  Only data is written: the raw file
contents (640x480 RGB, gravity.txt values, KLT tracks) and the tensors DemoDataset.__getitem__ makes of them.

    python oracle/tools/make_golden_preprocess.py

ref_shims replaces torchvision.transforms.ToTensor by PIL -> CHW float / 255 (torchvision is not installed here); the resize
itself is the reference's own call into Pillow (dataset.py:470)."""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.tools import ref_shims          # noqa: E402

FRAME = "000000"


def main():
    ref = ref_shims.load_reference()
    root = os.path.join(ref_shims.REFERENCE_ROOT, "demo_dataset")
    ds = ref.DemoDataset(root)
    idx = [i for i, f in enumerate(ds.color_files) if f[6:12] == FRAME][0]
    item = ds[idx]
    raw = np.asarray(Image.open(os.path.join(root, "color", "color_%s.png" % FRAME)).convert("RGB"))
    grav = np.loadtxt(os.path.join(root, "gravity", "gravity_%s.txt" % FRAME))
    klt = np.atleast_2d(np.loadtxt(os.path.join(root, "depth_sparse", "depth_sparse_%s.txt" % FRAME), delimiter=" "))
    rc = torch.nonzero(item["sparse_depth"][0] > 0)
    out = os.path.join(ROOT, "tests", "golden", "preprocess_demo_%s.npz" % FRAME)
    np.savez_compressed(out, raw_rgb=raw, gravity_raw=grav, klt_tracks=klt,
                        image_u8=(item["image"] * 255).round().to(torch.uint8).permute(1, 2, 0).numpy(),
                        image=item["image"].numpy(), gravity=item["gravity"].numpy(), aligned=item["aligned_direction"].numpy(),
                        sparse_rc=rc.numpy().astype(np.int16), sparse_val=item["sparse_depth"][0][rc[:, 0], rc[:, 1]].numpy(),
                        homogeneous_probe=item["homogeneous_coordinates"][::40, ::40].numpy())
    print("wrote", out, os.path.getsize(out) // 1024, "KiB;", len(rc), "sparse points")


if __name__ == "__main__":
    main()
