"""Deterministic synthetic workloads of SURVEY.md section 8(d): N^3 grid, sphere of radius 0.35 N,
V pinhole cameras on a Fibonacci sphere at distance 2 N looking at the origin.

Host-side camera arithmetic mirrors the reference in float64 (Camera::set_c2w ->
Affine3d::inverse(), camera.cc:39-42; look-at c2w, common.h:51-75; PinholeCamera fov
constructor, camera.cc:65-72,114-120) and hands the carve path what it reads: w2c cast to
float32, focal length, principal point.
"""
import math

import numpy as np

from .capi import CarverOption, UpdateOption, make_view


def lookat_c2w(position, target, up):
    """common.h:51-75 (T = double): col2 = normalize(target-position), col0 = normalize(col2 x up),
    col1 = col2 x col0, translation = position."""
    position = np.asarray(position, np.float64)
    c2 = np.asarray(target, np.float64) - position
    c2 = c2 / math.sqrt(c2[0] * c2[0] + (c2[1] * c2[1] + c2[2] * c2[2]))
    c0 = _cross(c2, np.asarray(up, np.float64))
    c0 = c0 / math.sqrt(c0[0] * c0[0] + (c0[1] * c0[1] + c0[2] * c0[2]))
    c1 = _cross(c2, c0)
    m = np.empty((3, 4), np.float64)
    m[:, 0], m[:, 1], m[:, 2], m[:, 3] = c0, c1, c2, position
    return m


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


def affine_inverse(c2w):
    """Affine3d::inverse() as Eigen evaluates it: cofactor inverse of the linear part,
    translation = -(inv * t), 3-term sums associated a0 + (a1 + a2)."""
    m = np.asarray(c2w, np.float64)

    def cof(i, j):
        i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
        return m[i1, j1] * m[i2, j2] - m[i1, j2] * m[i2, j1]

    c00, c10, c20 = cof(0, 0), cof(1, 0), cof(2, 0)
    det = c00 * m[0, 0] + (c10 * m[1, 0] + c20 * m[2, 0])
    invdet = 1.0 / det
    inv = np.array([[c00 * invdet, c10 * invdet, c20 * invdet],
                    [cof(0, 1) * invdet, cof(1, 1) * invdet, cof(2, 1) * invdet],
                    [cof(0, 2) * invdet, cof(1, 2) * invdet, cof(2, 2) * invdet]])
    t = m[:, 3]
    out = np.empty((3, 4), np.float64)
    out[:, :3] = inv
    for i in range(3):
        out[i, 3] = -(inv[i, 0] * t[0] + (inv[i, 1] * t[1] + inv[i, 2] * t[2]))
    return out


def pose_from_tum(t, q_xyzw):
    """examples.cc:36-50: Translation3d(t) * Quaterniond(q) (no normalisation)."""
    x, y, z, w = (float(v) for v in q_xyzw)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    m = np.empty((3, 4), np.float64)
    m[0, :3] = [1.0 - (tyy + tzz), txy - twz, txz + twy]
    m[1, :3] = [txy + twz, 1.0 - (txx + tzz), tyz - twx]
    m[2, :3] = [txz - twy, tyz + twx, 1.0 - (txx + tyy)]
    m[:, 3] = t
    return m


def focal_from_fov_y(height, fov_y_deg):
    """PinholeCamera::set_fov_y (camera.cc:114-120) in float32 like the reference."""
    rad = np.float32(fov_y_deg) * np.float32(0.01745329251994329576923690768489)
    return np.float32(np.float32(height) * np.float32(0.5) / np.float32(math.tan(float(rad) * 0.5)))


def sphere_option(n, update_option=None):
    h = n / 2.0
    return CarverOption(bb_min=(-h, -h, -h), bb_max=(h, h, h), resolution=1.0,
                        update_option=update_option or UpdateOption())


def sphere_views(n, n_views, width, height, fov_y_deg=60.0):
    """Returns (views, masks): vcy_view structs and uint8 silhouettes (255 inside the disc)."""
    radius = 0.35 * n
    dist = 2.0 * n
    f = focal_from_fov_y(height, fov_y_deg)
    cx = np.float32(np.float32(width) * np.float32(0.5) - np.float32(0.5))
    cy = np.float32(np.float32(height) * np.float32(0.5) - np.float32(0.5))
    lim = radius * radius / (dist * dist - radius * radius)
    uu = (np.arange(width, dtype=np.float64) - float(cx)) / float(f)
    vv = (np.arange(height, dtype=np.float64) - float(cy)) / float(f)
    mask = np.where(uu[None, :] ** 2 + vv[:, None] ** 2 <= lim, 255, 0).astype(np.uint8)
    views, masks = [], []
    for i in range(n_views):
        y = 1.0 - 2.0 * (i + 0.5) / n_views
        r = math.sqrt(max(0.0, 1.0 - y * y))
        phi = i * math.pi * (3.0 - math.sqrt(5.0))
        d = np.array([r * math.cos(phi), y, r * math.sin(phi)])
        up = (0.0, 1.0, 0.0)
        c2w = lookat_c2w(dist * d, (0.0, 0.0, 0.0), up)
        w2c = affine_inverse(c2w).astype(np.float32)
        views.append(make_view(w2c, f, f, cx, cy, width, height))
        masks.append(mask)
    return views, masks


def blob_views(n, n_views, width, height, fov_y_deg=60.0):
    """A harder scene than the centred sphere, same cameras: two overlapping OFF-CENTRE spheres of different size (their
    union is concave where they meet), so every view sees a different silhouette -- a distinct SDF image per view.
    Returns (views, masks)."""
    dist = 2.0 * n
    f = focal_from_fov_y(height, fov_y_deg)
    cx = np.float32(np.float32(width) * np.float32(0.5) - np.float32(0.5))
    cy = np.float32(np.float32(height) * np.float32(0.5) - np.float32(0.5))
    balls = [((0.16 * n, 0.10 * n, -0.12 * n), 0.24 * n), ((-0.17 * n, -0.08 * n, 0.10 * n), 0.17 * n)]
    uu = (np.arange(width, dtype=np.float64) - float(cx)) / float(f)
    vv = (np.arange(height, dtype=np.float64) - float(cy)) / float(f)
    dx, dy = np.meshgrid(uu, vv)               # ray direction (dx, dy, 1) in camera coordinates
    inv_len2 = 1.0 / (dx * dx + dy * dy + 1.0)
    views, masks = [], []
    for i in range(n_views):
        y = 1.0 - 2.0 * (i + 0.5) / n_views
        r = math.sqrt(max(0.0, 1.0 - y * y))
        phi = i * math.pi * (3.0 - math.sqrt(5.0))
        d = np.array([r * math.cos(phi), y, r * math.sin(phi)])
        c2w = lookat_c2w(dist * d, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
        w2c64 = affine_inverse(c2w)
        views.append(make_view(w2c64.astype(np.float32), f, f, cx, cy, width, height))
        inside = np.zeros((height, width), bool)
        for centre, radius in balls:
            pc = w2c64[:, :3] @ np.asarray(centre, np.float64) + w2c64[:, 3]
            along = pc[0] * dx + pc[1] * dy + pc[2]          # pc . ray
            perp2 = float(pc @ pc) - along * along * inv_len2  # squared distance of the centre from the ray
            inside |= (perp2 <= radius * radius) & (along > 0)
        masks.append(np.where(inside, 255, 0).astype(np.uint8))
    return views, masks
