"""Seeded synthetic YUV 4:2:0 generator — a pure function of (W, H, frame, bitdepth, seed, chroma_texture)."""
import numpy as np


def synth_frame(width, height, frame=0, bit_depth=8, seed=1234, chroma_texture=0.0):
    """chroma_texture > 0 adds that fraction of the (2x2 averaged) luma texture to Cb and minus half of it to Cr: natural video has
    such cross-component correlation and the LM chroma modes (CCLM) only win on pictures that have it."""
    s = 1 if bit_depth == 8 else 4
    mid, a1, a2 = (128, 60, 40) if bit_depth == 8 else (512, 240, 160)
    mx = (1 << bit_depth) - 1
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width]
    Y = mid + a1 * np.sin(x / 37.0) + a2 * np.cos(y / 23.0) + 30 * s * (((x // 32) + (y // 32) + frame) % 2) + rng.normal(0, 6 * s, (height, width))
    yc, xc = np.mgrid[0:height // 2, 0:width // 2]
    U = mid + 20 * s * np.sin(xc / 50.0) + rng.normal(0, 2 * s, (height // 2, width // 2))
    V = mid + 20 * s * np.cos(yc / 40.0) + rng.normal(0, 2 * s, (height // 2, width // 2))
    if chroma_texture:
        Yd = (Y[0:height // 2 * 2:2, 0:width // 2 * 2:2] + Y[1:height // 2 * 2:2, 0:width // 2 * 2:2] + Y[0:height // 2 * 2:2, 1:width // 2 * 2:2] + Y[1:height // 2 * 2:2, 1:width // 2 * 2:2]) / 4.0 - mid
        U = U + chroma_texture * Yd
        V = V - 0.5 * chroma_texture * Yd
    dt = np.uint8 if bit_depth == 8 else np.uint16
    return [np.clip(np.rint(p), 0, mx).astype(dt) for p in (Y, U, V)]


def slice_params(qp, bit_depth=8, dep_quant=False):
    """Slice-level inputs the hot path consumes (the caller's job in the reference):
    lambda per EL/EncSlice.cpp:754-845 for an I slice (QPFactor 0.57, GOP size 1), chroma QP through the
    cfg's mapping table (BIN/encoder_intra.cfg:94-95, identity below 31 here approximated by the VVC
    default table for 4:2:0), distortion weights per EL/EncSlice.cpp:121-137."""
    lam = 0.57 * 2.0 ** ((qp + 6 * (bit_depth - 8) - 12) / 3.0)
    if dep_quant:
        lam *= 2.0 ** (0.25 / 3.0)
    # chroma QP mapping table of the cfg: QpInValCb "17 22 34 42", QpOutValCb "17 23 35 39" is the VTM6 CTC
    # default; this fork's cfg gives points (2->2?, 31->32, 43->41).  Piecewise-linear through (31,32),(43,41).
    def map_qp(q):
        pts_in, pts_out = [31, 43], [32, 41]
        if q <= pts_in[0]:
            return q + (pts_out[0] - pts_in[0])
        if q >= pts_in[-1]:
            return q + (pts_out[-1] - pts_in[-1])
        num = (pts_out[1] - pts_out[0]) * (q - pts_in[0])
        den = pts_in[1] - pts_in[0]
        return pts_out[0] + (num + den // 2) // den
    qpc = map_qp(qp)
    w = 2.0 ** ((qp - qpc) / 3.0)
    return dict(qp=qp, qp_c=(qpc, qpc), lam=lam, dist_weight=(w, w))
