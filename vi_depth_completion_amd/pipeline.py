"""The per-batch hot path behind the reference's operator API.

`DepthCompletionPipeline._call_cnn(input_batch)` is a drop-in for `RunDepthCompletion._call_cnn` (main.py:261-298):
same input-batch dictionary (dataset.py:515-520), same output (B,1,H,W) depth tensor, same attribute names
(`cnn`, `surface_normal_cnn`, `plane_masks_extraction`, `args.enriched_samples`) and the same checkpoint loaders
(network_run.py:319-323, main.py:256-259).  See INTEGRATION.md for how the unchanged main.py/network_run.py bind to it.
"""
import argparse

import numpy as np
import torch

from .networks.depth_completion import ModifiedFPN
from .networks.surface_normal import SurfaceNormalPrediction
from .plane import PlaneBlock


class FixedPlaneMask:
    """Plane-mask provider for the perf configuration ("plane mask fixed", BASELINE.json configs[1]); has the
    `run_on_tensor(image) -> uint8 (H,W) id map` interface of COCODemo (plane_mask_detection/demo/predictor.py:143-150)."""

    def __init__(self, id_map):
        self.id_map = np.ascontiguousarray(id_map, dtype=np.uint8)

    def run_on_tensor(self, image):
        return self.id_map


class DepthCompletionPipeline:
    def __init__(self, enriched_samples=200, fc_img=(202.0, 202.0), cc_img=(0.5 * 319.87654, 0.5 * 239.87603),
                 align_corners=False, device="cuda", network_class_creator=ModifiedFPN, rng=np.random):
        if not torch.cuda.is_available():
            raise RuntimeError("DepthCompletionPipeline needs a GPU: the HIP path has no CPU fallback")
        self.args = argparse.Namespace(enriched_samples=enriched_samples)
        self.device = torch.device(device)
        self.cnn = network_class_creator().to(self.device)                                   # network_run.py:422-424
        self.surface_normal_cnn = SurfaceNormalPrediction(fc_img=np.asarray(fc_img, dtype=np.float64),
                                                          cc_img=np.asarray(cc_img, dtype=np.float64),
                                                          align_corners=align_corners).to(self.device)   # main.py:243
        self.plane_masks_extraction = None
        self.use_gravity = True
        self.planes = PlaneBlock()
        self.rng = rng
        self.eval_mode()

    # ---- the reference's harness methods ------------------------------------------------------------------------
    def eval_mode(self):
        self.surface_normal_cnn.eval()
        self.cnn.eval()

    def load_network_from_file(self, filename):
        state = self.cnn.state_dict()
        state.update(torch.load(filename, map_location=self.device))
        self.cnn.load_state_dict(state)

    def load_surface_normal_network_from_file(self, checkpoint):
        state = self.surface_normal_cnn.state_dict()
        state.update(torch.load(checkpoint, map_location=self.device))
        self.surface_normal_cnn.load_state_dict(state)

    def load_state_dicts(self, sn_state, dc_state):
        for m, sd in ((self.surface_normal_cnn, sn_state), (self.cnn, dc_state)):
            state = m.state_dict()
            state.update(sd)
            m.load_state_dict(state)

    # ---- the hot path ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _call_cnn(self, input_batch, taps=None):
        dev = self.device
        ds = input_batch["sparse_depth"].to(dev, non_blocking=True)
        rgb = input_batch["image"].to(dev, non_blocking=True)
        normals = self.surface_normal_cnn(rgb, input_batch["gravity"].to(dev), input_batch["aligned_direction"].to(dev))
        if taps is not None:
            taps["normals"] = normals
        if self.args.enriched_samples == 0:
            return self.cnn(rgb, normals, ds)
        homo = input_batch["homogeneous_coordinates"].to(dev, non_blocking=True)
        masks = [np.asarray(self.plane_masks_extraction.run_on_tensor(input_batch["image"][i])).reshape(ds.shape[-2], ds.shape[-1])
                 for i in range(ds.shape[0])]
        di, nnz = self.planes.plane_depth(normals, masks, ds, homo, rng=self.rng)
        enriched = self.planes.enrich(ds, di, nnz, self.args.enriched_samples, rng=self.rng)
        self.planes.check_records()
        if taps is not None:
            taps.update(plane_depth=di, enriched=enriched, records=self.planes.last_records)
        return self.cnn(rgb, normals, enriched)
