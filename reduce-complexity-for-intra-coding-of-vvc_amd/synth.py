"""Seeded synthetic YUV 4:2:0 generator — a pure function of (W, H, frame, bitdepth, seed, chroma_texture)."""
import numpy as np


def synth_frame(width, height, frame=0, bit_depth=8, seed=1234, chroma_texture=0.0, oriented=0.0, screen=0.0, limited=False):
    """chroma_texture > 0 adds that fraction of the (2x2 averaged) luma texture to Cb and minus half of it to Cr: natural video has
    such cross-component correlation and the LM chroma modes (CCLM) only win on pictures that have it.  oriented > 0 adds gratings of that
    amplitude (8-bit scale) whose direction and period change from one 32x32 block to the next: directional detail is what the angular modes
    predict and what the mode-dependent secondary transform (LFNST) compacts; white noise alone never selects it.  screen > 0 renders that fraction
    of the 32x32 blocks as noise-free screen content (a flat background with one-sample-wide strokes and isolated dots of high contrast): the
    residuals transform skip is made for.  limited squeezes the luma into the video range (64 .. 223 at 8 bit): full-range pictures make the reference's LMCS
    analysis switch the tool off (EL/EncReshape.cpp:444-446: samples in the first / last of its 16 bins)."""
    s = 1 if bit_depth == 8 else 4
    mid, a1, a2 = (128, 60, 40) if bit_depth == 8 else (512, 240, 160)
    mx = (1 << bit_depth) - 1
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width]
    Y = mid + a1 * np.sin(x / 37.0) + a2 * np.cos(y / 23.0) + 30 * s * (((x // 32) + (y // 32) + frame) % 2) + rng.normal(0, 6 * s, (height, width))
    if oriented:
        bx, by = x // 32, y // 32
        h1 = (bx * 73856093 ^ by * 19349663 ^ (seed * 83492791)) & 0xffff
        theta = (h1 % 32) * (np.pi / 32.0); period = 2.5 + ((h1 >> 5) % 5)
        Y = Y + oriented * s * np.sin((x * np.cos(theta) + y * np.sin(theta)) * (2 * np.pi / period))
    if screen:
        bx, by = x // 32, y // 32
        h2 = ((bx * 2654435761) ^ (by * 40503) ^ (seed * 97 + frame * 7919)) & 0xffffffff
        sel = ((h2 >> 3) % 1000) < int(screen * 1000)
        g2 = np.random.default_rng(seed + 77)
        dots = g2.random((height, width)) < 0.04
        strokes = ((x % 7 == (h2 >> 13) % 7) & ((y // 5) % 3 != 0)) | ((y % 9 == (h2 >> 17) % 9) & ((x // 6) % 2 == 0))
        bg = mid + s * (((h2 >> 21) % 5) * 12 - 24)
        fg = np.where(((h2 >> 25) & 1) == 1, bg + 90 * s, bg - 80 * s)
        Y = np.where(sel, np.where(strokes | dots, fg, bg), Y)
    if limited:
        Y = 64 * s + np.clip(np.rint(Y), 0, mx) * 5 // 8
    yc, xc = np.mgrid[0:height // 2, 0:width // 2]
    U = mid + 20 * s * np.sin(xc / 50.0) + rng.normal(0, 2 * s, (height // 2, width // 2))
    V = mid + 20 * s * np.cos(yc / 40.0) + rng.normal(0, 2 * s, (height // 2, width // 2))
    if chroma_texture:
        Yd = (Y[0:height // 2 * 2:2, 0:width // 2 * 2:2] + Y[1:height // 2 * 2:2, 0:width // 2 * 2:2] + Y[0:height // 2 * 2:2, 1:width // 2 * 2:2] + Y[1:height // 2 * 2:2, 1:width // 2 * 2:2]) / 4.0 - mid
        U = U + chroma_texture * Yd
        V = V - 0.5 * chroma_texture * Yd
    dt = np.uint8 if bit_depth == 8 else np.uint16
    return [np.clip(np.rint(p), 0, mx).astype(dt) for p in (Y, U, V)]


def chroma_qp_table(bit_depth=8, qp_in=(2, 31, 43), qp_out=(2, 32, 41)):
    """ChromaQpMappingTable::derivedChromaQPMappingTables (CL/Slice.cpp:1540-1581) for the cfg's pivots (BIN/encoder_intra.cfg:86-87);
    returns {q: mapped chroma QP} for q = -6*(bit_depth-8) .. 63 (same algorithm as vvcx_chroma_qp_table, pinned by tests/golden/chroma_qp.npz)."""
    off = 6 * (bit_depth - 8)
    clip = lambda v: max(-off, min(63, v))
    t = {qp_in[0]: qp_out[0]}
    for k in range(qp_in[0] - 1, -off - 1, -1):
        t[k] = clip(t[k + 1] - 1)
    for j in range(len(qp_in) - 1):
        d_in, d_out = qp_in[j + 1] - qp_in[j], qp_out[j + 1] - qp_out[j]
        sh = (d_in + 1) >> 1
        for m, k in enumerate(range(qp_in[j] + 1, qp_in[j + 1] + 1), 1):
            v = d_out * m + sh
            t[k] = t[qp_in[j]] + (v // d_in if v >= 0 else -((-v) // d_in))        # C integer division truncates towards zero
    for k in range(qp_in[-1] + 1, 64):
        t[k] = clip(t[k - 1] + 1)
    return t


def slice_params(qp, bit_depth=8, dep_quant=False):
    """Slice-level inputs the hot path consumes (the caller's job in the reference; same values as vvcx_derive_slice):
    lambda per EL/EncSlice.cpp:752-845 for an I slice (QPFactor 0.57, GOP size 1), chroma QP through the cfg's mapping table
    (QpInValCb "2 31 43" -> QpOutValCb "2 32 41", CbQpOffset / CrQpOffset 0), distortion weights per EL/EncSlice.cpp:107-137."""
    lam = 0.57 * 2.0 ** ((qp + 6 * (bit_depth - 8) - 12) / 3.0)
    if dep_quant:
        lam *= 2.0 ** (0.25 / 3.0)
    qpc = chroma_qp_table(bit_depth)[qp]
    w = 2.0 ** ((qp - qpc) / 3.0)
    if dep_quant:
        w *= 2.0 ** (0.2 / 3.0)
    return dict(qp=qp, qp_c=(qpc, qpc), lam=lam, dist_weight=(w, w))


# ---- synthetic per-CTU SAO parameters (inputs of vvcx_sao_bound_frames in the tests and in bench.py's SAO line)
def _tile_of(i, n, t):
    return max(k for k in range(t) if i >= (k * n) // t)


def sao_test_params(seed, w, h, tile_cols=1, tile_rows=1):
    """seeded per-CTU SAO parameters [nctu, 3, 7] int8 = {mode 0 off / 1 new / 2 merge, type (new: 0..3 edge class, 4 band; merge: 0 left, 1 above), band position,
    four coded offsets}; a merge only where the candidate CTU exists in the same tile (shared by the fixture generator and the tests)"""
    g = np.random.default_rng(seed)
    cw, ch = (w + 127) // 128, (h + 127) // 128
    prm = np.zeros((cw * ch, 3, 7), np.int8)
    for a in range(cw * ch):
        cx, cy = a % cw, a // cw
        for c in range(3):
            r = g.random()
            mode = 0 if r < 0.2 else 2 if r < 0.45 else 1
            if mode == 2:
                left = bool(g.integers(0, 2))
                ok_l = cx > 0 and _tile_of(cx - 1, cw, tile_cols) == _tile_of(cx, cw, tile_cols)
                ok_a = cy > 0 and _tile_of(cy - 1, ch, tile_rows) == _tile_of(cy, ch, tile_rows)
                if left and ok_l:
                    prm[a, c, :2] = (2, 0)
                elif ok_a:
                    prm[a, c, :2] = (2, 1)
                elif ok_l:
                    prm[a, c, :2] = (2, 0)
                else:
                    mode = 1
            if mode == 1:
                t = int(g.integers(0, 5))
                prm[a, c] = [1, t, int(g.integers(0, 32)) if t == 4 else 0] + [int(v) for v in g.integers(-7, 8, 4)]
    return prm


ALF_APS_INTS = 732      # one ALF parameter set as a row of ints (the layout tests/golden/make_golden.py hands to the reference and vvcx.AlfAps.from_row reads)


def alf_test_params(seed, w, h, n_aps=3):
    """seeded adaptive-loop-filter inputs: parameter sets [n_aps, 732] int32 = {number of luma filters, class -> filter [25], luma clip flag, luma coefficients [25][12],
    luma clipping indices [25][12], number of chroma alternatives, their clip flags [8], chroma coefficients [8][6], chroma clipping indices [8][6]}; the sets the slice
    filters luma with (filter set 16 + k), the set of the chroma alternatives; per CTU {enable Y, Cb, Cr, luma filter set, alternative Cb, Cr} [nctu, 6] int32
    (shared by the fixture generator, the tests and bench.py)"""
    g = np.random.default_rng(seed)
    nctu = ((w + 127) // 128) * ((h + 127) // 128)
    aps = np.zeros((n_aps, ALF_APS_INTS), np.int32)
    lw = np.array([4, 6, 8, 6, 6, 10, 16, 10, 6, 6, 10, 20])          # larger taps next to the centre
    cwt = np.array([6, 8, 14, 8, 8, 18])
    for i in range(n_aps):
        nf = 25 if i == 0 else int(g.integers(1, 26))
        aps[i, 0] = nf
        aps[i, 1:26] = np.arange(25) if i == 0 else g.integers(0, nf, 25)
        aps[i, 26] = int(g.integers(0, 2)) if i else 1
        aps[i, 27:327] = (g.integers(-1, 2, (25, 12)) * g.integers(0, lw + 1, (25, 12))).ravel()
        aps[i, 327:627] = g.integers(0, 4, 300)
        aps[i, 627] = int(g.integers(1, 9))
        aps[i, 628:636] = g.integers(0, 2, 8)
        aps[i, 636:684] = (g.integers(-1, 2, (8, 6)) * g.integers(0, cwt + 1, (8, 6))).ravel()
        aps[i, 684:732] = g.integers(0, 4, 48)
    n_luma = int(g.integers(1, n_aps + 1))
    luma_aps = [int(v) for v in g.permutation(n_aps)[:n_luma]]
    chroma_aps = int(g.integers(0, n_aps))
    ctu = np.zeros((nctu, 6), np.int32)
    ctu[:, 0:3] = (g.random((nctu, 3)) < 0.8) | (nctu == 1)
    ctu[:, 3] = g.integers(0, 16 + n_luma, nctu)
    ctu[:, 4:6] = g.integers(0, aps[chroma_aps, 627], (nctu, 2))
    return dict(aps=aps, luma_aps=luma_aps, chroma_aps=chroma_aps, ctu=ctu)


def alf_test_frame(width, height, bit_depth, seed):
    """a picture for the adaptive loop filter's classifier: synth_frame's gratings and screen content with the detail of every 16 x 16 block scaled by a factor of its own
    (0 .. 2), so that the blocks spread over the activity classes as well as over the directions"""
    pl = synth_frame(width, height, 0, bit_depth, seed, chroma_texture=0.6, oriented=20.0, screen=0.2)
    g = np.random.default_rng(seed + 5)
    y = pl[0].astype(np.float64)
    by, bx = (height + 15) // 16, (width + 15) // 16
    f = np.array([0.0, 0.05, 0.15, 0.3, 0.6, 1.0, 2.0])[g.integers(0, 7, (by, bx))]
    f = np.kron(f, np.ones((16, 16)))[:height, :width]
    m = np.kron(np.add.reduceat(np.add.reduceat(y, np.arange(0, height, 16), 0), np.arange(0, width, 16), 1) / 256.0, np.ones((16, 16)))[:height, :width]
    pl[0] = np.clip(np.rint(m + f * (y - m)), 0, (1 << bit_depth) - 1).astype(pl[0].dtype)
    return pl
