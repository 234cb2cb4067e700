"""Shared loader for the hand-written known-answer vectors (tests/golden/kat_spconv_handwritten.json)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    return json.load(open(os.path.join(HERE, "golden", "kat_spconv_handwritten.json")))["cases"]


def build_weight(case):
    """The weight tensor [Cout, k, k, k, Cin] (reference state-dict layout) a case's `weight_rule` describes."""
    k = 3 if case["kind"] == "subm" else 2
    w = np.zeros((case["cout"], k, k, k, case["cin"]), np.float32)
    rule = case["weight_rule"]
    if rule.startswith("W[0,a,b,c,0] = 100*a + 10*b + c"):
        plus = 1.0 if rule.endswith("+ 1") else 0.0
        for a in range(k):
            for b in range(k):
                for c in range(k):
                    w[0, a, b, c, 0] = 100 * a + 10 * b + c + plus
    elif rule.startswith("all zero except W[co,"):
        tap = tuple(int(t) for t in rule.split("W[co,")[1].split(",ci]")[0].split(","))
        mat = np.array(json.loads(rule.split("= ")[1].split("[co][ci]")[0]), np.float32)
        w[:, tap[0], tap[1], tap[2], :] = mat
    else:
        raise ValueError(rule)
    return w


def case_points(case):
    """Points (float32, voxel size 1) and batch ids that voxelize to exactly the case's voxel coordinates."""
    c = np.asarray(case["coords"], np.int64)
    return (c[:, 1:] + 0.5).astype(np.float32), c[:, 0].copy()
