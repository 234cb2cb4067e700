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


def verticality_planes():
    """Hand-derived known answers for the verticality feature (jakteristics: 1 - |<e_z, normal>|, normal = eigenvector of the smallest
    eigenvalue of the neighbourhood covariance, search radius 0.6 m; reference tree_learn/util/data_preparation.py:83-85).  Points on
    a plane through the origin whose normal makes the angle theta with the vertical: every neighbourhood lies in that plane, its
    covariance has eigenvalue 0 along the normal, so verticality = 1 - cos(theta) for EVERY point -- no implementation needed to state
    it.  A 0.1 m lattice inside the plane (two in-plane unit vectors u, v), 41 x 41 points.
    Returns [(points f64[n,3], expected verticality)]."""
    import numpy as np
    out = []
    for theta_deg in (0.0, 30.0, 60.0, 90.0):
        th = np.deg2rad(theta_deg)
        n = np.array([np.sin(th), 0.0, np.cos(th)])               # normal tilted towards +x
        u = np.array([np.cos(th), 0.0, -np.sin(th)]); v = np.array([0.0, 1.0, 0.0])
        a, b = np.meshgrid(np.arange(-20, 21) * 0.1, np.arange(-20, 21) * 0.1)
        pts = a.reshape(-1, 1) * u + b.reshape(-1, 1) * v
        assert np.abs(pts @ n).max() < 1e-12
        out.append((pts, 1.0 - np.cos(th)))
    return out


def voxel_downsample_by_hand():
    """Hand-worked example of the global 0.1 m down-sample (open3d voxel_down_sample_and_trace as the reference calls it,
    tree_learn/util/data_preparation.py:60-79): bound = max|p| + 100 = 101, grid origin = -bound - voxel/2 = -101.05, voxel index =
    floor((p + 101.05) / 0.1).  A = (0, 0, 0) -> 1010.5 -> 1010; B = (.04, .04, .04) -> 1010.9 -> 1010; D = (-.04, 0, 0) -> 1010.1 ->
    1010: one voxel although D and B sit either side of zero (the half-voxel offset of the origin); C = (.06, 0, 0) -> 1011.1 -> 1011;
    E = (1, 1, 1) -> 1020.5 -> 1020.  Voxel {A, B, D}: mean (0, .04/3, .04/3) -> float32 -> 2 decimals = (0, .01, .01), extra columns
    from its first point A.  Output in ascending voxel index.  Returns (data [5,4], voxel, expected rows, first indices, point->voxel)."""
    import numpy as np
    data = np.array([[0.00, 0.00, 0.00, 10.0],      # A
                     [0.04, 0.04, 0.04, 11.0],      # B
                     [0.06, 0.00, 0.00, 12.0],      # C
                     [-0.04, 0.00, 0.00, 13.0],     # D
                     [1.00, 1.00, 1.00, 14.0]])     # E
    expect = np.array([[0.0, 0.01, 0.01, 10.0], [0.06, 0.0, 0.0, 12.0], [1.0, 1.0, 1.0, 14.0]])
    return data, 0.1, expect, np.array([0, 2, 4]), np.array([0, 0, 1, 0, 2])
