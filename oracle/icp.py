"""Oracle twin of the camera tracking: CUDACameraTrackingMultiRes::applyCT
(DSC/CUDACameraTrackingMultiRes.cpp:241-321) in numpy -- per-pixel arithmetic in float32 as the kernels do it, the
sums of the linear system in float64, the 6x6 solve by numpy's SVD.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (the
reference ships no fixtures for this path and cannot be built here: see oracle/vh_oracle.h).
"""
import numpy as np

from . import oracle as O

MINF = np.float32(-np.inf)
f32 = np.float32


def pyramid(maps4, levels):
    """resampleFloat4Map + computeNormals per level, :256-263 -> [(positions, normals)] for levels 1.."""
    out = []
    cur = np.ascontiguousarray(maps4, dtype=np.float32)
    for _ in range(levels - 1):
        h, w = cur.shape[:2]
        nxt = O.image_op("resample_float4_map", cur, w, h, out_channels=4, out_size=(w // 2, h // 2))
        out.append((nxt, O.compute_normals(nxt)))
        cur = nxt
    return out


def _mul_p(m, p):  # float4x4 * float3 as a point, row-major m (16,), float32 left to right
    return np.stack([m[4 * r + 0] * p[..., 0] + m[4 * r + 1] * p[..., 1] + m[4 * r + 2] * p[..., 2] + m[4 * r + 3] * f32(1.0) for r in range(3)], -1).astype(np.float32)


def _mul_d(m, p):
    return np.stack([m[4 * r + 0] * p[..., 0] + m[4 * r + 1] * p[..., 1] + m[4 * r + 2] * p[..., 2] + m[4 * r + 3] * f32(0.0) for r in range(3)], -1).astype(np.float32)


def correspondences(inp, inp_n, tgt, tgt_n, delta, dist_thres, normal_thres, level_factor, cp):
    """projectiveCorrespondencesKernel, DSC/CUDAImageHelper.cu:70-125"""
    h, w = inp.shape[:2]
    m = np.asarray(delta, dtype=np.float32).reshape(16)
    corr = np.full((h, w, 4), MINF, dtype=np.float32)
    corr_n = np.full((h, w, 4), MINF, dtype=np.float32)
    valid = (inp[..., 0] != MINF) & (inp_n[..., 0] != MINF)
    with np.errstate(all="ignore"):
        pt = _mul_p(m, inp[..., :3])
        nt = _mul_d(m, inp_n[..., :3])
        fx, fy, mx, my = f32(cp.fx), f32(cp.fy), f32(cp.mx), f32(cp.my)
        sxf = (pt[..., 0] * fx / pt[..., 2] + mx) + f32(0.5)
        syf = (pt[..., 1] * fy / pt[..., 2] + my) + f32(0.5)
        ok = valid & np.isfinite(sxf) & np.isfinite(syf) & (np.abs(sxf) < 1e9) & (np.abs(syf) < 1e9)
        sx = np.trunc(np.where(ok, sxf, 0)).astype(np.int64)
        sy = np.trunc(np.where(ok, syf, 0)).astype(np.int64)
        sx = np.trunc(sx.astype(np.float32) / f32(level_factor)).astype(np.int64)
        sy = np.trunc(sy.astype(np.float32) / f32(level_factor)).astype(np.int64)
    inside = ok & (sx >= 0) & (sy >= 0) & (sx < w) & (sy < h)
    ys, xs = np.nonzero(inside)
    tx, ty = sx[ys, xs], sy[ys, xs]
    tp, tn = tgt[ty, tx], tgt_n[ty, tx].copy()
    good = (tp[:, 0] != MINF) & (tn[:, 0] != MINF)
    np.seterr(invalid="ignore")  # MINF entries flow through the arithmetic before they are masked out
    p = pt[ys, xs]
    diff = (p - tp[:, :3]).astype(np.float32)
    d = np.sqrt(diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1] + diff[:, 2] * diff[:, 2]).astype(np.float32)
    n = nt[ys, xs]
    dn = (n[:, 0] * tn[:, 0] + n[:, 1] * tn[:, 1] + n[:, 2] * tn[:, 2]).astype(np.float32)
    keep = good & (d <= f32(dist_thres)) & (dn >= f32(normal_thres))
    zmin, zmax = f32(cp.m_sensorDepthWorldMin), f32(cp.m_sensorDepthWorldMax)
    projz = (p[:, 2] - zmin) / (zmax - zmin)
    wgt = np.maximum(f32(0.0), f32(0.5) * ((f32(1.0) - d / f32(dist_thres)) + (f32(1.0) - projz))).astype(np.float32)
    tn[:, 3] = wgt
    corr[ys[keep], xs[keep]] = tp[keep]
    corr_n[ys[keep], xs[keep]] = tn[keep]
    return corr, corr_n


def build_system(inp, corr, corr_n, delta):
    """scanScanElementsCS + reductionSystemCPU -> (ATA 6x6, ATb 6, sumRegError, sumRegWeight, numCorr); float64 sums"""
    m = np.asarray(delta, dtype=np.float32).reshape(16)
    sel = (corr[..., 0] != MINF) & (inp[..., 0] != MINF) & (corr_n[..., 0] != MINF)
    q = _mul_p(m, inp[..., :3][sel])
    p = corr[..., :3][sel]
    n = corr_n[..., :3][sel]
    wgt = corr_n[..., 3][sel].astype(np.float64)
    q64, p64, n64 = q.astype(np.float64), p.astype(np.float64), n.astype(np.float64)
    row = np.stack([n64[:, 0] * q64[:, 1] - n64[:, 1] * q64[:, 0], n64[:, 2] * q64[:, 0] - n64[:, 0] * q64[:, 2],
                    n64[:, 1] * q64[:, 2] - n64[:, 2] * q64[:, 1], -n64[:, 0], -n64[:, 1], -n64[:, 2]], 1)
    b = np.sum(n64 * (q64 - p64), axis=1)
    ata = (row * wgt[:, None]).T @ row
    atb = (row * wgt[:, None]).T @ b
    dn = np.sum((p64 - q64) * n64, axis=1)
    return ata, atb, float(np.sum(wgt * dn * dn)), float(np.sum(wgt)), int(sel.sum())


def solve(ata, atb):
    """JacobiSVD(ATA).solve(ATb) with Eigen's rank threshold; condition = s_max / s_min"""
    u, s, vt = np.linalg.svd(ata)
    keep = s > 6.0 * np.finfo(np.float32).eps * s[0]
    x = (vt.T[:, keep] / s[keep]) @ (u.T[keep] @ atb)
    return x, float(s[0] / s[5]) if s[5] > 0 else float("inf")


def delinearize(x, angle_thres, dist_thres):
    """delinearizeTransformation :186-211 (mean 0, stddev 1) -> 4x4 float32 or None when the step is too large"""
    x = np.asarray(x, dtype=np.float32)
    cz, sz, cy, sy, cx, sx = np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]), np.cos(x[2]), np.sin(x[2])
    r = np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                  [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                  [-sy, cy * sx, cy * cx]], dtype=np.float32)
    angle = np.arccos(np.clip(0.5 * (np.trace(r) - 1.0), -1.0, 1.0))
    if not (angle <= angle_thres) or not (np.linalg.norm(x[3:6]) <= dist_thres):
        return None
    t = np.eye(4, dtype=np.float32)
    t[:3, :3] = r
    t[:3, 3] = x[3:6]
    return t


def apply_ct(inp, inp_n, model, model_n, last_transform, ts, delta_estimate, cp, levels):
    """-> (4x4 pose or None if lost, info dict)"""
    ins = [(np.ascontiguousarray(inp, np.float32), np.ascontiguousarray(inp_n, np.float32))] + pyramid(inp, levels)
    mods = [(np.ascontiguousarray(model, np.float32), np.ascontiguousarray(model_n, np.float32))] + pyramid(model, levels)
    delta = np.asarray(delta_estimate, dtype=np.float32).reshape(4, 4).copy()
    info = dict(iterations=0)
    for level in range(levels - 1, -1, -1):
        last_err = -1.0
        for _ in range(int(ts.s_maxOuterIter[level])):
            corr, corr_n = correspondences(ins[level][0], ins[level][1], mods[level][0], mods[level][1], delta.reshape(16),
                                           ts.s_distThres[level], ts.s_normalThres[level], 2.0 ** level, cp)
            for _i in range(int(ts.s_maxInnerIter[level])):
                ata, atb, err, wsum, ncorr = build_system(ins[level][0], corr, corr_n, delta.reshape(16))
                info["iterations"] += 1
                info.update(sumRegError=err, sumRegWeight=wsum, numCorr=ncorr)
                if not np.any(ata):
                    return None, info
                x, cond = solve(ata, atb)
                info["matrixCondition"] = cond
                t = delinearize(x, ts.s_angleTransThres[level], ts.s_distTransThres[level])
                if t is None:
                    return None, info
                delta = (t @ delta).astype(np.float32)
            if abs(np.float32(last_err) - np.float32(err)) < ts.s_residualEarlyOut[level]:
                break
            last_err = err
    info["delta"] = delta
    return (np.asarray(last_transform, np.float32).reshape(4, 4) @ delta).astype(np.float32), info
