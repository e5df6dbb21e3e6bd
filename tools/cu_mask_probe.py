#!/usr/bin/env python3
"""Experiment (round 3): two frame programs replayed side by side on two HIP streams whose queues are restricted to disjoint halves of
the CUs (hipExtStreamCreateWithCUMask) against the same replay on ordinary streams.  With ordinary streams the kernels of two lanes
alternate on the CUs (a conv workgroup reserves most of a CU's LDS); on disjoint halves they run truly side by side, each at half
width.  Prints ms per frame for: no mask, lower / upper half of the mask bits, even / odd bits, and 3/4 + 3/4 overlapping masks.
    python tools/cu_mask_probe.py [mixed|fp32]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VIDC_PRECISION"] = sys.argv[1] if len(sys.argv) > 1 else "mixed"
from vi_depth_completion_amd import engine                                               # noqa: E402
from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, build_frame_program  # noqa: E402


def masked_stream(hip, words):
    arr = (C.c_uint32 * len(words))(*words)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(words), arr)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return torch.cuda.ExternalStream(st.value)


def main():
    H, W, B = 256, 320, 1
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    hip = C.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    nw = (n_cu + 31) // 32
    full = [0xFFFFFFFF] * nw
    lower = [0xFFFFFFFF if i < nw // 2 else 0 for i in range(nw)]
    upper = [0 if i < nw // 2 else 0xFFFFFFFF for i in range(nw)]
    even, odd = [0x55555555] * nw, [0xAAAAAAAA] * nw
    q3a = [0xFFFFFFFF if i < 3 * nw // 4 else 0 for i in range(nw)]
    q3b = [0xFFFFFFFF if i >= nw // 4 else 0 for i in range(nw)]
    cc = (0.5 * 319.87654, 0.5 * 239.87603 * H / 240.0)
    pipe = DepthCompletionPipeline(enriched_samples=200, cc_img=cc, device=dev)
    ws = engine.JointWeightStore({"sn": pipe.surface_normal_cnn, "dc": pipe.cnn})

    def pair(streams, iters=24):
        progs = []
        for s in streams:
            with torch.cuda.stream(s):
                p = build_frame_program(pipe.surface_normal_cnn, pipe.cnn, B, H, W, dev, weights=ws)
                p.run()
                p.capture_segments()
            progs.append(p)
        torch.cuda.synchronize()

        def burst(n):
            for _ in range(n):
                for k in (0, 1):
                    for p, s in zip(progs, streams):
                        p.launch_segment(k, stream=s.cuda_stream)
            torch.cuda.synchronize()
        burst(3)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            burst(iters)
            ms = 1e3 * (time.perf_counter() - t0) / (len(streams) * iters)
            best = ms if best is None else min(best, ms)
        return best

    print("%d CUs, %d mask words; %s mode" % (n_cu, nw, os.environ["VIDC_PRECISION"]))
    print("one ordinary stream:            %.3f ms per frame" % pair([torch.cuda.Stream()]))
    print("two ordinary streams:           %.3f ms per frame" % pair([torch.cuda.Stream(), torch.cuda.Stream()]))
    print("one stream, full mask:          %.3f ms per frame" % pair([masked_stream(hip, full)]))
    print("one stream, lower half only:    %.3f ms per frame" % pair([masked_stream(hip, lower)]))
    print("two streams, lower / upper:     %.3f ms per frame" % pair([masked_stream(hip, lower), masked_stream(hip, upper)]))
    print("two streams, even / odd bits:   %.3f ms per frame" % pair([masked_stream(hip, even), masked_stream(hip, odd)]))
    print("two streams, 3/4 + 3/4 overlap: %.3f ms per frame" % pair([masked_stream(hip, q3a), masked_stream(hip, q3b)]))
    print("two streams, full masks:        %.3f ms per frame" % pair([masked_stream(hip, full), masked_stream(hip, full)]))


if __name__ == "__main__":
    main()
