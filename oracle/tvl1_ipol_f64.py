"""Independent float64 witness of the TV-L1 oracle -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

oracle/tvl1_oracle.c is the arithmetic contract the HIP kernels are held to bit for bit; its inner iteration was
shaped together with the kernels (one division per pixel, a 2^-100 regulariser under the square root, the three
cases of the thresholding operator folded into one clamp with 1/|grad|^2 := 0 below 1e-10).  This file restates the
same algorithm WITHOUT those choices, written directly from the published description, vectorised in numpy and in
float64, so that the C oracle is checked against something other than itself:

  * inner iteration = IPOL 2013 (Sanchez Perez, Meinhardt-Llopis, Facciolo, "TV-L1 Optical Flow Estimation"),
    Algorithm 1: rho = rho_c + <grad I1w, u>; the THREE-CASE thresholding operator TH
        d = +l_t grad I1w            if rho < -l_t |grad I1w|^2
        d = -l_t grad I1w            if rho >  l_t |grad I1w|^2
        d = -rho grad I1w / |grad I1w|^2   otherwise (d = 0 where |grad I1w|^2 < 1e-10, the reference code's
                                            GRAD_IS_ZERO guard)
    v = u + d;  u' = v + theta div p (backward differences, IPOL border rules);
    p' = (p + tau/theta grad u') / (1 + tau/theta |grad u'|)  per flow component, each with its own division,
    no regulariser (forward differences, zero at the last column / row);
  * multi-scale driver = IPOL section 3 (coarse to fine, u upsampled and scaled by 1/step, p = 0 per level, `warps`
    warps of `iters` iterations), fixed-iteration mode.

What it shares with the C oracle are only the choices DESIGN.md section 3 documents as the build's own where the
publications differ or are silent (S0-S3, S8): float frames in [0,255] without normalisation or pre-smoothing, the
pyramid sizes and the Gaussian + bilinear zoom, centred gradients, BILINEAR warping with clamped coordinates (IPOL
uses bicubic; the OpenCV generation the reference's data comes from uses bilinear), no median filtering.  Those are
restated here too, independently (vectorised, float64), not called from the C library.

PARITY UNPINNED against the reference all the same (it holds no TV-L1 code or data): this is a
witness for the restatement, not a pin to the reference.
"""
import numpy as np

GRAD_IS_ZERO = 1e-10


def pyramid_sizes(w, h, nscales=5, step=0.8):
    """Level sizes: the factor is applied in float32 and rounded half up (a SIZE rule, kept exactly)."""
    sizes = [(w, h)]
    step = np.float32(step)
    while len(sizes) < nscales:
        nw = int(np.float32(sizes[-1][0]) * step + np.float32(0.5))
        nh = int(np.float32(sizes[-1][1]) * step + np.float32(0.5))
        if min(nw, nh) < 16:
            break
        sizes.append((nw, nh))
    return sizes


def bilinear(img, x, y):
    """img[h,w] sampled at float coordinate arrays (x, y), coordinates clamped to the image."""
    h, w = img.shape
    x = np.clip(x, 0.0, w - 1.0)
    y = np.clip(y, 0.0, h - 1.0)
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x1 = np.minimum(x0 + 1, w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    ax, ay = x - x0, y - y0
    top = img[y0, x0] + ax * (img[y0, x1] - img[y0, x0])
    bot = img[y1, x0] + ax * (img[y1, x1] - img[y1, x0])
    return top + ay * (bot - top)


def zoom_out(img, ow, oh, step=0.8):
    """Gaussian (sigma = 0.6 sqrt(1/step^2 - 1), radius int(3 sigma) + 1, replicate border) then bilinear sampling."""
    h, w = img.shape
    sigma = 0.6 * np.sqrt(1.0 / (step * step) - 1.0)
    R = min(int(3.0 * sigma) + 1, 8)
    k = np.arange(-R, R + 1, dtype=np.float64)
    g = np.exp(-k * k / (2.0 * sigma * sigma))
    g /= g.sum()
    pad = np.pad(img, ((0, 0), (R, R)), mode="edge")
    t = sum(g[i] * pad[:, i:i + w] for i in range(2 * R + 1))
    pad = np.pad(t, ((R, R), (0, 0)), mode="edge")
    t = sum(g[i] * pad[i:i + h, :] for i in range(2 * R + 1))
    yy, xx = np.mgrid[0:oh, 0:ow].astype(np.float64)
    return bilinear(t, xx * (w / ow), yy * (h / oh))


def centred_gradient(I):
    Ix = 0.5 * (np.concatenate([I[:, 1:], I[:, -1:]], axis=1) - np.concatenate([I[:, :1], I[:, :-1]], axis=1))
    Iy = 0.5 * (np.concatenate([I[1:], I[-1:]], axis=0) - np.concatenate([I[:1], I[:-1]], axis=0))
    return Ix, Iy


def divergence(p1, p2):
    """IPOL: backward differences; first column/row: p itself; last column/row: minus the previous value."""
    h, w = p1.shape
    d1 = np.empty_like(p1)
    d1[:, 0] = p1[:, 0]
    d1[:, 1:w - 1] = p1[:, 1:w - 1] - p1[:, 0:w - 2]
    d1[:, w - 1] = -p1[:, w - 2]
    d2 = np.empty_like(p2)
    d2[0, :] = p2[0, :]
    d2[1:h - 1, :] = p2[1:h - 1, :] - p2[0:h - 2, :]
    d2[h - 1, :] = -p2[h - 2, :]
    return d1 + d2


def forward_gradient(u):
    ux = np.zeros_like(u)
    uy = np.zeros_like(u)
    ux[:, :-1] = u[:, 1:] - u[:, :-1]
    uy[:-1, :] = u[1:, :] - u[:-1, :]
    return ux, uy


def level(I0, I1, I1x, I1y, u1, u2, tau, lam, theta, warps, iters):
    """IPOL Algorithm 1 on one scale (fixed number of inner iterations)."""
    h, w = I0.shape
    l_t = lam * theta
    taut = tau / theta
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    p11 = np.zeros((h, w)); p12 = np.zeros((h, w)); p21 = np.zeros((h, w)); p22 = np.zeros((h, w))
    for _ in range(warps):
        I1w = bilinear(I1, xx + u1, yy + u2)
        I1wx = bilinear(I1x, xx + u1, yy + u2)
        I1wy = bilinear(I1y, xx + u1, yy + u2)
        grad = I1wx * I1wx + I1wy * I1wy
        rho_c = I1w - I1wx * u1 - I1wy * u2 - I0
        safe = np.where(grad < GRAD_IS_ZERO, 1.0, grad)
        for _ in range(iters):
            rho = rho_c + (I1wx * u1 + I1wy * u2)
            lo = rho < -l_t * grad
            hi = rho > l_t * grad
            mid = np.where(grad < GRAD_IS_ZERO, 0.0, -rho / safe)
            fi = np.where(lo, l_t, np.where(hi, -l_t, mid))  # d = fi * grad I1w in all three cases
            v1 = u1 + fi * I1wx
            v2 = u2 + fi * I1wy
            u1 = v1 + theta * divergence(p11, p12)
            u2 = v2 + theta * divergence(p21, p22)
            u1x, u1y = forward_gradient(u1)
            u2x, u2y = forward_gradient(u2)
            g1 = 1.0 + taut * np.hypot(u1x, u1y)
            g2 = 1.0 + taut * np.hypot(u2x, u2y)
            p11 = (p11 + taut * u1x) / g1
            p12 = (p12 + taut * u1y) / g1
            p21 = (p21 + taut * u2x) / g2
            p22 = (p22 + taut * u2y) / g2
    return u1, u2


def tvl1_flow_pair(f0, f1, tau=0.25, lam=0.15, theta=0.3, nscales=5, warps=5, iters=300, step=0.8):
    """Flow (u1, u2) float64 [h,w] from frame f0 to f1 (arrays [h,w], gray values in [0,255])."""
    f0 = np.asarray(f0, dtype=np.float64)
    f1 = np.asarray(f1, dtype=np.float64)
    h, w = f0.shape
    sizes = pyramid_sizes(w, h, nscales, step)
    P0, P1 = [f0], [f1]
    for (lw, lh) in sizes[1:]:
        P0.append(zoom_out(P0[-1], lw, lh, step))
        P1.append(zoom_out(P1[-1], lw, lh, step))
    cw, ch = sizes[-1]
    u1 = np.zeros((ch, cw))
    u2 = np.zeros((ch, cw))
    for s in range(len(sizes) - 1, -1, -1):
        I1x, I1y = centred_gradient(P1[s])
        u1, u2 = level(P0[s], P1[s], I1x, I1y, u1, u2, tau, lam, theta, warps, iters)
        if s > 0:
            fw, fh = sizes[s - 1]
            lw, lh = sizes[s]
            yy, xx = np.mgrid[0:fh, 0:fw].astype(np.float64)
            u1 = bilinear(u1, xx * (lw / fw), yy * (lh / fh)) / step
            u2 = bilinear(u2, xx * (lw / fw), yy * (lh / fh)) / step
    return u1, u2
