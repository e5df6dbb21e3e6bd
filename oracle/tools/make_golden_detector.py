"""Extracts the known-answer NMS vectors of the reference's own unit tests (plane_mask_detection/tests/test_nms.py:11-54, :60-255) as
DATA into tests/golden/nms_reference_vectors.npz.  TEST INFRASTRUCTURE; runs in the build container only.  The test module itself cannot
be imported (it needs `pytorch_local.maskrcnn_benchmark`, which is not in the tree), so the `np.array([...])` literals are read from
its syntax tree -- numbers only, no code is copied."""
import ast
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = "/root/reference/plane_mask_detection/tests/test_nms.py"


def literal_arrays(fn_node):
    """every list literal that is the first argument of an np.array(...) call inside the function, in source order"""
    out = []
    for node in ast.walk(fn_node):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "array" and node.args and isinstance(node.args[0], ast.List):
            out.append((node.lineno, np.array(ast.literal_eval(node.args[0]))))
    return [a for _, a in sorted(out, key=lambda t: t[0])]


def main():
    tree = ast.parse(open(SRC).read())
    fns = {n.name: n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)}
    a0 = literal_arrays(fns["test_nms_cpu"])
    case0 = a0[0].astype(np.float32).reshape(-1, 5)
    thr, gts = None, None
    for node in ast.walk(fns["test_nms_cpu"]):
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
            if node.targets[0].id == "test_thresh":
                thr = ast.literal_eval(node.value)
            if node.targets[0].id == "gt_indices":
                gts = ast.literal_eval(node.value)
    a1 = literal_arrays(fns["test_nms1_cpu"])
    boxes1, scores1, gt1 = a1[0].astype(np.float32), a1[1].astype(np.float32), a1[2].astype(np.int64)
    thr1 = None
    for node in ast.walk(fns["test_nms1_cpu"]):
        if isinstance(node, ast.Call) and getattr(node.func, "id", "") == "box_nms":
            thr1 = ast.literal_eval(node.args[2])
    out = os.path.join(ROOT, "tests", "golden", "nms_reference_vectors.npz")
    np.savez_compressed(out, c0_boxes=case0[:, :4], c0_scores=case0[:, 4], c0_thresholds=np.array(thr, np.float32),
                        c0_keep=np.array([np.pad(np.array(g), (0, 8 - len(g)), constant_values=-1) for g in gts]),
                        c1_boxes=boxes1, c1_scores=scores1, c1_threshold=np.float32(thr1), c1_keep=gt1)
    print("wrote", out, case0.shape, boxes1.shape, thr1, gt1.shape)


if __name__ == "__main__":
    main()
