"""ctypes binding of the C-ABI in include/vvcx.h.

The product path is the gfx950 library `libvvcx.so` built in-tree by csrc/Makefile (hipcc).  There is no
CPU fallback: if the library is missing, loading fails loudly.  (`lib_path` exists so that the CPU *debug*
emulation build under tools/hipemu can be exercised by the CPU test-suite; nothing else uses it.)
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libvvcx.so")
TOOL_MRL = 1
TOOL_LFNST = 1 << 3          # low-frequency non-separable transform: the lfnstIdx passes of the search, residual_lfnst_mode (needs DEPQUANT and MIP)
TOOL_JCCR = 1 << 9           # JointCbCr: joint coding of the chroma residual pair (inter-chroma transform), joint_cb_cr flag (needs DEPQUANT)
TOOL_MIP = 1 << 1            # matrix-based intra prediction searched (cfg MIP 1, FastMIP 1): mip_flag / MIP mode per luma CU
TOOL_MTS = 1 << 4            # explicit intra MTS (cfg MTS 1, MTSIntraMaxCand 3): DST-VII / DCT-VIII pairs for luma TUs up to 32x32
TOOL_DEPQUANT = 1 << 6       # dependent quantisation (cfg DepQuant 1): trellis quantiser, state-driven residual syntax and dequantiser
TOOL_CU_REUSE = 1 << 11      # BestEncInfoCache, REUSE_CU_RESULTS (CL/TypeDef.h:291) - on in the reference build
TOOL_CCLM = 1 << 8           # LM / MDLM chroma modes (cfg LMChroma 1, on in the reference's intra configuration)
TOOL_FAST = 1 << 12          # the fork's FAST_ALGORITHM: features + random forest pick the one partition mode of a luma node
TOOL_TS = 1 << 5             # transform skip of luma TUs up to 32x32 with RDOQ-TS (cfg TransformSkip 1, TransformSkipFast 1; needs DEPQUANT and LFNST)
TOOL_RDOQ = 1 << 7           # cfg RDOQ / RDOQTS beside DepQuant: only reaches transform-skip blocks
TOOL_ISP = 1 << 2            # intra sub-partitions (cfg ISP 1, ISPFast 1)
TOOL_LMCS = 1 << 10          # luma mapping with chroma scaling (cfg LMCSEnable 1): the slice carries the model
TOOL_WPP = 1 << 13           # cfg WaveFrontSynchro 1: CTU rows as lagged streams (context hand-over behind the first CTU of the row above, above-right CTU unavailable), a sub-stream per row
TOOLS_DEFAULT = TOOL_MRL | TOOL_CU_REUSE


class VvcxError(RuntimeError):
    pass


class _Cfg(C.Structure):
    _fields_ = [("pic_w", C.c_int32), ("pic_h", C.c_int32), ("bit_depth", C.c_int32), ("ctu_size", C.c_int32),
                ("min_qt", C.c_int32 * 2), ("max_bt_depth", C.c_int32 * 2), ("max_bt_size", C.c_int32 * 2),
                ("max_tt_size", C.c_int32 * 2), ("dual_tree", C.c_int32), ("tile_cols", C.c_int32), ("tile_rows", C.c_int32),
                ("tools", C.c_uint32), ("chroma", C.c_int32), ("max_frames", C.c_int32), ("device", C.c_int32), ("emit_payload", C.c_int32)]


class _Slice(C.Structure):
    _fields_ = [("qp", C.c_int32), ("qp_c", C.c_int32 * 2), ("lam", C.c_double), ("dist_weight", C.c_double * 2),
                ("lmcs_enable", C.c_int32), ("lmcs_chroma_adj", C.c_int32), ("lmcs_min_bin", C.c_int32), ("lmcs_max_bin", C.c_int32), ("lmcs_delta_cw", C.c_int32 * 16)]


class _Frame(C.Structure):
    _fields_ = [("org", C.c_void_p * 3), ("reco", C.c_void_p * 3), ("stride", C.c_int32 * 3)]


class _Task(C.Structure):
    _fields_ = [("frame", C.c_int32), ("ctu_rs_addr", C.c_int32)]


TU_DTYPE = np.dtype([("cu_index", "<i4"), ("x", "<i2"), ("y", "<i2"), ("w", "<i2"), ("h", "<i2"), ("ch_type", "u1"), ("depth", "u1"), ("mts_idx", "u1"),
                     ("joint_cb_cr", "u1"), ("cbf", "u1", (3,)), ("pad_", "u1"), ("coeff_offset", "<i4", (3,)), ("coeff_stride", "<i4", (3,))], align=True)
CTU_DTYPE = np.dtype([("dist", "<u8"), ("frac_bits", "<u8"), ("cost", "<f8"), ("n_cu", "<i4")], align=True)
CU_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "<i2"), ("h", "<i2"), ("ch_type", "u1"), ("qt_depth", "u1"),
                     ("bt_depth", "u1"), ("mt_depth", "u1"), ("depth", "u1"), ("intra_dir", "u1"), ("mrl_idx", "u1"),
                     ("cbf", "u1"), ("mts_idx", "u1"), ("mip_flag", "u1"), ("lfnst_idx", "u1"), ("joint_cb_cr", "u1"), ("isp_mode", "u1"), ("tu_cbf", "u1"), ("split_series", "<u8")], align=True)

_libs = {}


def load_library(lib_path=None):
    path = lib_path or DEFAULT_LIB
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise VvcxError("HIP extension %s is missing - build it with __graft_entry__.build() (hipcc, gfx950); "
                        "there is no CPU fallback for the product path" % path)
    L = C.CDLL(path)
    L.vvcx_create.argtypes = [C.POINTER(_Cfg), C.POINTER(C.c_void_p)]
    L.vvcx_destroy.argtypes = [C.c_void_p]
    L.vvcx_set_slice.argtypes = [C.c_void_p, C.POINTER(_Slice)]
    L.vvcx_bind_frames.argtypes = [C.c_void_p, C.POINTER(_Frame), C.c_int]
    L.vvcx_compress_ctus.argtypes = [C.c_void_p, C.POINTER(_Task), C.c_int, C.c_void_p, C.c_void_p]
    L.vvcx_compress_bound_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.vvcx_submit_ctus.argtypes = [C.c_void_p, C.POINTER(_Task), C.c_int, C.c_void_p]
    L.vvcx_poll_ctus.argtypes = [C.c_void_p]
    L.vvcx_wait_ctus.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.vvcx_get_cus.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.vvcx_get_tus.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.vvcx_last_kernel_ms.restype = C.c_float
    L.vvcx_last_kernel_ms.argtypes = [C.c_void_p]
    L.vvcx_get_counters.argtypes = [C.c_void_p, C.c_void_p]
    L.vvcx_get_profile.argtypes = [C.c_void_p, C.c_void_p]
    L.vvcx_last_error.restype = C.c_char_p
    L.vvcx_ctus_per_frame.argtypes = [C.c_void_p]
    if hasattr(L, "vvcx_resident_streams"):
        L.vvcx_resident_streams.argtypes = [C.c_void_p]
    if hasattr(L, "vvcx_set_forest"):
        L.vvcx_set_forest.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
    if hasattr(L, "vvcx_forest_predict_batch"):
        L.vvcx_forest_predict_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    if hasattr(L, "vvcx_get_payload"):
        L.vvcx_get_payload.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    if hasattr(L, "vvcx_distortion_batch"):
        L.vvcx_distortion_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    if hasattr(L, "vvcx_intra_pred_batch"):
        L.vvcx_intra_pred_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_void_p]
    if hasattr(L, "vvcx_ctx_init"):
        L.vvcx_ctx_init.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    if hasattr(L, "vvcx_cabac_code_bins"):
        L.vvcx_cabac_code_bins.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    if hasattr(L, "vvcx_rd_cost_batch"):
        L.vvcx_rd_cost_batch.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    if hasattr(L, "vvcx_scan_order"):
        L.vvcx_scan_order.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
    if hasattr(L, "vvcx_transform_quant_batch"):
        L.vvcx_transform_quant_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    _libs[path] = L
    return L


def mip_pred_batch(cases, refs, lib_path=None, device=0):
    """vvcx_mip_pred_batch: cases (n, 4) int32 {w, h, mode, bit_depth}; refs int16 (top | left per case); returns the concatenated predictions"""
    L = load_library(lib_path)
    cases = np.ascontiguousarray(cases, np.int32).reshape(-1, 4); refs = np.ascontiguousarray(refs, np.int16)
    out = np.zeros(int((cases[:, 0] * cases[:, 1]).sum()), np.int16)
    L.vvcx_mip_pred_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    if L.vvcx_mip_pred_batch(cases.ctypes.data, len(cases), refs.ctypes.data, len(refs), out.ctypes.data, len(out), device) != 0:
        raise VvcxError(L.vvcx_last_error().decode())
    return out


class _SliceCfg(C.Structure):
    _fields_ = [("qp", C.c_int32), ("bit_depth", C.c_int32), ("n_pts", C.c_int32), ("qp_in", C.c_int32 * 8), ("qp_out", C.c_int32 * 8),
                ("cb_qp_offset", C.c_int32), ("cr_qp_offset", C.c_int32), ("gop_size", C.c_int32), ("dep_quant", C.c_int32)]


def sao_picture(planes, bit_depth, prm, tile_cols=1, tile_rows=1, lf_across_tiles=1, log2_offset_scale=0, device=0, lib_path=None):
    """vvcx_sao_picture: sample adaptive offset with per-CTU parameters prm int8 [ctus, 3, 7] on three host planes -> filtered uint16 planes"""
    L = load_library(lib_path)
    out = [np.ascontiguousarray(p.astype(np.uint16)) for p in planes]
    h, w = out[0].shape
    prm = np.ascontiguousarray(prm, np.int8)
    L.vvcx_sao_picture.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    _chk(L, L.vvcx_sao_picture(w, h, bit_depth, tile_cols, tile_rows, prm.ctypes.data, int(lf_across_tiles), int(log2_offset_scale), out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, device))
    return out


def sao_decide(stats, width, height, bit_depth, lambdas, slice_qp, tile_cols=1, tile_rows=1, log2_offset_scale=0, lib_path=None):
    """vvcx_sao_decide: the RD half of the SAO decision of one picture from its statistics [ctus, 3, 5, 2, 32] -> int8 [ctus, 3, 7] parameters"""
    L = load_library(lib_path)
    st = np.ascontiguousarray(stats, np.int64); lam = np.ascontiguousarray(lambdas, np.float64)
    prm = np.zeros((st.shape[0], 3, 7), np.int8)
    L.vvcx_sao_decide.argtypes = [C.c_int] * 6 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _chk(L, L.vvcx_sao_decide(width, height, bit_depth, tile_cols, tile_rows, int(slice_qp), lam.ctypes.data, int(log2_offset_scale), st.ctypes.data, prm.ctypes.data))
    return prm


class AlfAps(C.Structure):
    """vvcx_alf_aps: what an ALF parameter set carries (AlfParam of the reference)"""
    _fields_ = [("num_luma_filters", C.c_int32), ("class_to_filter", C.c_uint8 * 25), ("nonlinear_luma", C.c_uint8), ("luma_coeff", (C.c_int16 * 12) * 25), ("luma_clip_idx", (C.c_uint8 * 12) * 25),
                ("num_chroma_alt", C.c_int32), ("nonlinear_chroma", C.c_uint8 * 8), ("chroma_coeff", (C.c_int16 * 6) * 8), ("chroma_clip_idx", (C.c_uint8 * 6) * 8)]

    @classmethod
    def from_row(cls, row):
        """from a row of synth.alf_test_params (ALF_APS_INTS ints)"""
        a = cls()
        a.num_luma_filters = int(row[0]); a.nonlinear_luma = int(row[26]); a.num_chroma_alt = int(row[627])
        for c in range(25):
            a.class_to_filter[c] = int(row[1 + c])
            for k in range(12):
                a.luma_coeff[c][k] = int(row[27 + c * 12 + k]); a.luma_clip_idx[c][k] = int(row[327 + c * 12 + k])
        for t in range(8):
            a.nonlinear_chroma[t] = int(row[628 + t])
            for k in range(6):
                a.chroma_coeff[t][k] = int(row[636 + t * 6 + k]); a.chroma_clip_idx[t][k] = int(row[684 + t * 6 + k])
        return a


class AlfSlice(C.Structure):
    _fields_ = [("n_luma_aps", C.c_int32), ("luma_aps", C.c_int32 * 8), ("chroma_aps", C.c_int32)]


def _alf_args(prms):
    """the C arrays of one or more frames' ALF inputs (dicts of synth.alf_test_params' form; the parameter sets of the first one are the sets of the call)"""
    aps = (AlfAps * len(prms[0]["aps"]))(*[AlfAps.from_row(r) for r in prms[0]["aps"]])
    sl = (AlfSlice * len(prms))()
    for i, p in enumerate(prms):
        sl[i].n_luma_aps = len(p["luma_aps"]); sl[i].chroma_aps = int(p["chroma_aps"])
        for k, v in enumerate(p["luma_aps"]):
            sl[i].luma_aps[k] = int(v)
    ctu = np.ascontiguousarray(np.concatenate([np.asarray(p["ctu"]) for p in prms]).astype(np.int8))      # {flag[3], set, alt[2]}: six bytes per CTU
    return aps, sl, ctu


def alf_picture(planes, bit_depth, prm, want_classes=False, device=0, lib_path=None):
    """vvcx_alf_picture: the adaptive loop filter with parameter sets / slice / per-CTU choices (a dict of synth.alf_test_params' form) on three host planes -> filtered
    uint16 planes (and the class | transpose << 5 bytes of the luma 4 x 4 blocks)"""
    L = load_library(lib_path)
    out = [np.ascontiguousarray(p.astype(np.uint16)) for p in planes]
    h, w = out[0].shape
    aps, sl, ctu = _alf_args([prm])
    cls = np.zeros((h // 4, w // 4), np.uint8)
    L.vvcx_alf_picture.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    _chk(L, L.vvcx_alf_picture(w, h, bit_depth, C.addressof(aps), len(aps), C.addressof(sl), ctu.ctypes.data, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, cls.ctypes.data, device))
    return (out, cls) if want_classes else out


def lmcs_analyze_device(ptrs, strides, width, height, bit_depth, qp, update_ctrl=1, lib_path=None):
    """vvcx_lmcs_analyze_device: the same analysis on planes that are in device memory (ptrs: three device addresses, strides in samples)"""
    L = load_library(lib_path)
    org = (C.c_void_p * 3)(*[int(p) for p in ptrs]); st = (C.c_int * 3)(*[int(v) for v in strides])
    sl = _Slice()
    L.vvcx_lmcs_analyze_device.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    _chk(L, L.vvcx_lmcs_analyze_device(org, st, width, height, bit_depth, qp, update_ctrl, C.byref(sl)))
    return dict(enable=int(sl.lmcs_enable), chroma_adj=int(sl.lmcs_chroma_adj), min_bin=int(sl.lmcs_min_bin), max_bin=int(sl.lmcs_max_bin), delta_cw=[int(v) for v in sl.lmcs_delta_cw])


def lmcs_analyze(planes, bit_depth, qp, update_ctrl=1, lib_path=None):
    """vvcx_lmcs_analyze: the reference encoder's LMCS picture analysis for an intra picture (host planes) -> the model as set_slice(lmcs=...) takes it:
    dict(enable, chroma_adj, min_bin, max_bin, delta_cw[16])"""
    L = load_library(lib_path)
    dt = np.uint8 if bit_depth < 10 else np.uint16
    pl = [np.ascontiguousarray(p, dt) for p in planes]
    h, w = pl[0].shape
    org = (C.c_void_p * 3)(*[p.ctypes.data for p in pl]); st = (C.c_int * 3)(*[p.shape[1] for p in pl])
    sl = _Slice()
    L.vvcx_lmcs_analyze.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    rc = L.vvcx_lmcs_analyze(org, st, w, h, bit_depth, qp, update_ctrl, C.byref(sl))
    _chk(L, rc)
    return dict(enable=int(sl.lmcs_enable), chroma_adj=int(sl.lmcs_chroma_adj), min_bin=int(sl.lmcs_min_bin), max_bin=int(sl.lmcs_max_bin), delta_cw=[int(v) for v in sl.lmcs_delta_cw])


def chroma_qp_table(bit_depth=8, qp_in=(2, 31, 43), qp_out=(2, 32, 41), lib_path=None):
    """vvcx_chroma_qp_table: mapped chroma QP for q = -6*(bit_depth-8) .. 63 (host function of the library)"""
    L = load_library(lib_path)
    a = np.asarray(qp_in, np.int32); b = np.asarray(qp_out, np.int32); t = np.zeros(64 + 6 * (bit_depth - 8), np.int32)
    L.vvcx_chroma_qp_table.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    if L.vvcx_chroma_qp_table(bit_depth, len(a), a.ctypes.data, b.ctypes.data, t.ctypes.data) != 0:
        raise VvcxError(L.vvcx_last_error().decode())
    return t


def derive_slice(qp, bit_depth=8, qp_in=(2, 31, 43), qp_out=(2, 32, 41), cb_qp_offset=0, cr_qp_offset=0, gop_size=1, dep_quant=False, lib_path=None):
    """vvcx_derive_slice: the slice-level inputs (lambda, chroma QPs, distortion weights) in the form set_slice takes"""
    L = load_library(lib_path)
    c = _SliceCfg(); c.qp, c.bit_depth, c.n_pts = qp, bit_depth, len(qp_in)
    for i, (a, b) in enumerate(zip(qp_in, qp_out)):
        c.qp_in[i], c.qp_out[i] = a, b
    c.cb_qp_offset, c.cr_qp_offset, c.gop_size, c.dep_quant = cb_qp_offset, cr_qp_offset, gop_size, int(dep_quant)
    out = _Slice()
    L.vvcx_derive_slice.argtypes = [C.POINTER(_SliceCfg), C.POINTER(_Slice)]
    if L.vvcx_derive_slice(C.byref(c), C.byref(out)) != 0:
        raise VvcxError(L.vvcx_last_error().decode())
    return dict(qp=out.qp, qp_c=(out.qp_c[0], out.qp_c[1]), lam=out.lam, dist_weight=(out.dist_weight[0], out.dist_weight[1]))


class VvcxEncoder:
    """≙ one EncCu instance (EL/EncCu.h:80-230): create/init → per-slice set-up → compressCtu calls → destroy."""

    def __init__(self, width, height, bit_depth=8, tile_cols=1, tile_rows=1, chroma=True, tools=TOOLS_DEFAULT,
                 max_frames=1, device=0, lib_path=None, emit_payload=False, forest=None):
        self.L = load_library(lib_path)
        c = _Cfg()
        c.pic_w, c.pic_h, c.bit_depth, c.ctu_size = width, height, bit_depth, 128
        c.min_qt[0], c.min_qt[1] = 8, 4                   # BIN/encoder_intra.cfg:98-99
        c.max_bt_depth[0], c.max_bt_depth[1] = 3, 3       # :102-103
        c.max_bt_size[0], c.max_bt_size[1] = 32, 64       # CL/CommonDef.h:427,437
        c.max_tt_size[0], c.max_tt_size[1] = 32, 32
        c.dual_tree, c.tile_cols, c.tile_rows, c.tools = 1, tile_cols, tile_rows, tools
        c.chroma, c.max_frames, c.device = int(chroma), max_frames, device
        c.emit_payload = int(emit_payload)
        self.cfg = c
        self.h = C.c_void_p()
        self._chk(self.L.vvcx_create(C.byref(c), C.byref(self.h)))
        self.ctus_per_frame = self.L.vvcx_ctus_per_frame(self.h)
        self.n_frames = 0
        if forest is not None:
            self.set_forest(forest)

    def set_forest(self, forest):
        """forest: dict of flattened sklearn tree arrays (forest.py: load_forest / forest_from_sklearn); needed with TOOL_FAST"""
        from .forest import check_forest
        check_forest(forest)
        self._chk(self.L.vvcx_set_forest(self.h, len(forest["root"]), len(forest["feature"]), len(forest["classes"]), forest["root"].ctypes.data,
                                         forest["feature"].ctypes.data, forest["threshold"].ctypes.data, forest["left"].ctypes.data,
                                         forest["right"].ctypes.data, forest["value"].ctypes.data, forest["classes"].ctypes.data))

    def deblock_bound_frames(self, beta_offset_div2=0, tc_offset_div2=0):
        """in-loop deblocking of every bound (completely coded) picture, in place on the reconstruction planes; returns the kernel time in ms"""
        self.L.vvcx_deblock_bound_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.L.vvcx_last_deblock_ms.restype = C.c_float
        self.L.vvcx_last_deblock_ms.argtypes = [C.c_void_p]
        self._chk(self.L.vvcx_deblock_bound_frames(self.h, beta_offset_div2, tc_offset_div2, None))
        return float(self.L.vvcx_last_deblock_ms(self.h))

    def sao_bound_frames(self, prm, lf_across_tiles=1, log2_offset_scale=0):
        """vvcx_sao_bound_frames: prm int8 [n_frames, ctus, 3, 7] = {mode, type, band, four offsets} per CTU and component; returns the kernel time in ms"""
        prm = np.ascontiguousarray(prm, np.int8).reshape(self.n_frames, self.ctus_per_frame, 3, 7)
        self.L.vvcx_sao_bound_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.L.vvcx_last_sao_ms.restype = C.c_float
        self.L.vvcx_last_sao_ms.argtypes = [C.c_void_p]
        self._chk(self.L.vvcx_sao_bound_frames(self.h, prm.ctypes.data, int(lf_across_tiles), int(log2_offset_scale), None))
        return float(self.L.vvcx_last_sao_ms(self.h))

    def sao_statistics_bound_frames(self, lf_across_tiles=1):
        """vvcx_sao_statistics_bound_frames -> (int64 [n_frames, ctus, 3, 5, 2, 32] = count | diff per type and class, kernel ms)"""
        out = np.zeros((self.n_frames, self.ctus_per_frame, 3, 5, 2, 32), np.int64)
        self.L.vvcx_sao_statistics_bound_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.L.vvcx_last_sao_stats_ms.restype = C.c_float
        self.L.vvcx_last_sao_stats_ms.argtypes = [C.c_void_p]
        self._chk(self.L.vvcx_sao_statistics_bound_frames(self.h, int(lf_across_tiles), out.ctypes.data, None))
        return out, float(self.L.vvcx_last_sao_stats_ms(self.h))

    def alf_bound_frames(self, prms):
        """vvcx_alf_bound_frames: prms = one dict of synth.alf_test_params' form per bound frame (the parameter sets of the first are the call's); returns the kernel time in ms"""
        assert len(prms) == self.n_frames
        aps, sl, ctu = _alf_args(prms)
        self.L.vvcx_alf_bound_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.vvcx_last_alf_ms.restype = C.c_float
        self.L.vvcx_last_alf_ms.argtypes = [C.c_void_p]
        self._chk(self.L.vvcx_alf_bound_frames(self.h, C.addressof(aps), len(aps), C.addressof(sl), ctu.ctypes.data, None))
        return float(self.L.vvcx_last_alf_ms(self.h))

    def get_levels(self, frame):
        """quantised levels of the coded picture: three int16 planes (Y, Cb, Cr)"""
        out = []
        self.L.vvcx_get_levels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        for c in range(3):
            w, h = (self.cfg.pic_w, self.cfg.pic_h) if c == 0 else (self.cfg.pic_w >> 1, self.cfg.pic_h >> 1)
            a = np.zeros((h, w), np.int16)
            self._chk(self.L.vvcx_get_levels(self.h, frame, c, a.ctypes.data, w))
            out.append(a)
        return out

    def forest_predict(self, rows):
        """the forest's class for each row of 26 int32 features (≙ BIN/TEST.py GetPartition)"""
        rows = np.ascontiguousarray(rows, np.int32).reshape(-1, 26)
        out = np.zeros(len(rows), np.int32)
        self._chk(self.L.vvcx_forest_predict_batch(self.h, rows.ctypes.data, len(rows), out.ctypes.data))
        return out

    def get_payload(self, frame, tile):
        """slice_data() bytes of one coded tile (needs emit_payload=True)"""
        cap = 1 << 22
        buf = np.zeros(cap, np.uint8); n = C.c_int()
        self._chk(self.L.vvcx_get_payload(self.h, frame, tile, buf.ctypes.data, cap, C.byref(n)))
        return buf[:n.value].copy()

    def enable_training_dump(self, cap_rows):
        """vvcx_enable_training_dump: every qualifying luma node of the search leaves a row of 26 features + complexity class + chosen partition"""
        self.L.vvcx_enable_training_dump.argtypes = [C.c_void_p, C.c_int]
        self._chk(self.L.vvcx_enable_training_dump(self.h, int(cap_rows)))
        self._train_cap = int(cap_rows)

    def training_rows(self):
        """the (n, 28) int32 rows dumped since the frames were bound (order across streams is not defined)"""
        n = C.c_int()
        rows = np.zeros((self._train_cap, 28), np.int32)
        self.L.vvcx_get_training_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self._chk(self.L.vvcx_get_training_rows(self.h, rows.ctypes.data, len(rows), C.byref(n)))
        if n.value > len(rows):
            raise VvcxError("training dump overflow: %d rows produced, capacity %d" % (n.value, len(rows)))
        return rows[:n.value].copy()

    def get_substream_sizes(self, frame, tile):
        """byte counts of the sub-streams inside get_payload(frame, tile): one, or with TOOL_WPP one per CTU row of the tile"""
        n = C.c_int()
        self.L.vvcx_get_substream_sizes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        self._chk(self.L.vvcx_get_substream_sizes(self.h, frame, tile, None, 0, C.byref(n)))
        sizes = np.zeros(n.value, np.int32)
        self._chk(self.L.vvcx_get_substream_sizes(self.h, frame, tile, sizes.ctypes.data, len(sizes), C.byref(n)))
        return sizes

    def resident_streams(self):
        return int(self.L.vvcx_resident_streams(self.h))

    def intra_pred_batch(self, reco, coded, cases):
        """reco: 3 host planes (uint8 / uint16) of the picture; coded: 2 uint8 maps [uh, uw] (luma tree, chroma tree);
        cases: structured array PRED_CASE_DTYPE -> list of int16 [h, w] predictions"""
        reco = [np.ascontiguousarray(p) for p in reco]
        coded = [np.ascontiguousarray(c, np.uint8) for c in coded]
        cases = np.ascontiguousarray(cases, PRED_CASE_DTYPE)
        total = int((cases["w"].astype(np.int64) * cases["h"]).sum())
        pred = np.zeros(total, np.int16)
        rp = (C.c_void_p * 3)(*[p.ctypes.data for p in reco]); cp = (C.c_void_p * 2)(*[c.ctypes.data for c in coded])
        self._chk(self.L.vvcx_intra_pred_batch(self.h, rp, cp, cases.ctypes.data, len(cases), pred.ctypes.data))
        out, off = [], 0
        for c in cases:
            out.append(pred[off:off + c["w"] * c["h"]].reshape(c["h"], c["w"])); off += int(c["w"]) * int(c["h"])
        return out

    def _chk(self, rc):
        if rc != 0:
            raise VvcxError("vvcx error %d: %s" % (rc, self.L.vvcx_last_error().decode()))

    def close(self):
        if self.h:
            self.L.vvcx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_slice(self, qp, qp_c, lam, dist_weight, lmcs=None):
        """lmcs: None / dict(enable, chroma_adj, min_bin, max_bin, delta_cw[16]) = the LMCS model of the slice (TOOL_LMCS)"""
        s = _Slice()
        s.qp = qp
        s.qp_c[0], s.qp_c[1] = qp_c
        s.lam = lam
        s.dist_weight[0], s.dist_weight[1] = dist_weight
        if lmcs and lmcs.get("enable"):
            s.lmcs_enable = 1; s.lmcs_chroma_adj = int(lmcs["chroma_adj"]); s.lmcs_min_bin = int(lmcs["min_bin"]); s.lmcs_max_bin = int(lmcs["max_bin"])
            for i in range(16):
                s.lmcs_delta_cw[i] = int(lmcs["delta_cw"][i])
        self._chk(self.L.vvcx_set_slice(self.h, C.byref(s)))

    def lmcs_inverse_reco(self, stream=None):
        """vvcx_lmcs_inverse_reco: the luma reconstruction of the bound pictures back to the original domain (before deblocking)"""
        self.L.vvcx_lmcs_inverse_reco.argtypes = [C.c_void_p, C.c_void_p]
        self._chk(self.L.vvcx_lmcs_inverse_reco(self.h, C.c_void_p(stream or 0)))

    def lmcs_tables(self):
        n = 1 << self.cfg.bit_depth
        fwd = np.zeros(n, np.int16); inv = np.zeros(n, np.int16); piv = np.zeros(17, np.int32); cs = np.zeros(16, np.int32)
        self.L.vvcx_lmcs_tables.argtypes = [C.c_void_p] * 5
        self._chk(self.L.vvcx_lmcs_tables(self.h, fwd.ctypes.data, inv.ctypes.data, piv.ctypes.data, cs.ctypes.data))
        return fwd, inv, piv, cs

    def bind_frames(self, frames):
        """frames: list of (org_ptrs[3], reco_ptrs[3], strides[3]) with DEVICE pointers (ints)."""
        arr = (_Frame * len(frames))()
        for i, (org, reco, st) in enumerate(frames):
            for c in range(3):
                arr[i].org[c] = org[c]
                arr[i].reco[c] = reco[c]
                arr[i].stride[c] = st[c]
        self._chk(self.L.vvcx_bind_frames(self.h, arr, len(frames)))
        self.n_frames = len(frames)

    def compress_ctus(self, tasks, stream=None):
        """tasks: list of (frame, ctu_rs_addr) ≙ successive EncCu::compressCtu calls."""
        t = (_Task * len(tasks))()
        for i, (f, a) in enumerate(tasks):
            t[i].frame, t[i].ctu_rs_addr = f, a
        out = np.zeros(len(tasks), CTU_DTYPE)
        self._chk(self.L.vvcx_compress_ctus(self.h, t, len(tasks), out.ctypes.data, stream))
        return out

    def submit_ctus(self, tasks, stream=None):
        """Enqueue only (vvcx_submit_ctus): returns the number of tasks to hand to wait_ctus."""
        t = (_Task * max(1, len(tasks)))()
        for i, (f, a) in enumerate(tasks):
            t[i].frame, t[i].ctu_rs_addr = f, a
        self._chk(self.L.vvcx_submit_ctus(self.h, t, len(tasks), stream))
        return len(tasks)

    def poll_ctus(self):
        r = self.L.vvcx_poll_ctus(self.h)
        if r < 0:
            self._chk(r)
        return bool(r)

    def wait_ctus(self, n):
        out = np.zeros(n, CTU_DTYPE)
        self._chk(self.L.vvcx_wait_ctus(self.h, out.ctypes.data, n))
        return out

    def compress_bound_frames(self, stream=None):
        out = np.zeros(self.n_frames * self.ctus_per_frame, CTU_DTYPE)
        self._chk(self.L.vvcx_compress_bound_frames(self.h, out.ctypes.data, stream))
        return out.reshape(self.n_frames, self.ctus_per_frame)

    def get_cus(self, frame):
        n = C.c_int()
        cus = np.zeros(self.ctus_per_frame * 2048, CU_DTYPE)
        self._chk(self.L.vvcx_get_cus(self.h, frame, cus.ctypes.data, len(cus), C.byref(n)))
        return cus[:n.value].copy()

    def get_tus(self, frame):
        n = C.c_int()
        self._chk(self.L.vvcx_get_tus(self.h, frame, None, 0, C.byref(n)))
        tus = np.zeros(max(1, n.value), TU_DTYPE)
        self._chk(self.L.vvcx_get_tus(self.h, frame, tus.ctypes.data, len(tus), C.byref(n)))
        return tus[:n.value].copy()

    def last_kernel_ms(self):
        return float(self.L.vvcx_last_kernel_ms(self.h))

    def profile(self):
        c = np.zeros(48, np.uint64)
        self._chk(self.L.vvcx_get_profile(self.h, c.ctypes.data))
        return c

    def counters(self):
        c = np.zeros(4, np.uint64)
        self._chk(self.L.vvcx_get_counters(self.h, c.ctypes.data))
        return c


# ---- leaf operators (include/vvcx.h, "leaf operators"): host arrays in, host arrays out, work on the device
PRED_CASE_DTYPE = np.dtype([("comp", "<i4"), ("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("mode", "<i4"), ("mrl", "<i4")])


def _chk(L, rc):
    if rc != 0:
        raise VvcxError("vvcx error %d: %s" % (rc, L.vvcx_last_error().decode()))


def distortion_batch(a, b, w, h, device=0, lib_path=None):
    """a, b: int16 arrays of n*w*h samples (blocks back to back) -> uint64 [n, 3] = SAD, SATD, SSE"""
    L = load_library(lib_path)
    a = np.ascontiguousarray(a, np.int16).ravel(); b = np.ascontiguousarray(b, np.int16).ravel()
    n = a.size // (w * h)
    out = np.zeros((n, 3), np.uint64)
    _chk(L, L.vvcx_distortion_batch(a.ctypes.data, b.ctypes.data, w, h, n, out.ctypes.data, device))
    return out


def ctx_init(qp, lib_path=None):
    L = load_library(lib_path)
    s0 = np.zeros(386, np.uint16); s1 = np.zeros(386, np.uint16)
    _chk(L, L.vvcx_ctx_init(int(qp), s0.ctypes.data, s1.ctypes.data))
    return s0, s1


def cabac_code_bins(s0, s1, ctx, bins, device=0, lib_path=None):
    """-> (frac_bits, s0', s1') after coding the bin string on context ctx"""
    L = load_library(lib_path)
    a = np.array([s0], np.uint16); b = np.array([s1], np.uint16); bins = np.ascontiguousarray(bins, np.uint8); bits = np.zeros(1, np.uint64)
    _chk(L, L.vvcx_cabac_code_bins(a.ctypes.data, b.ctypes.data, int(ctx), bins.ctypes.data, len(bins), bits.ctypes.data, device))
    return int(bits[0]), int(a[0]), int(b[0])


def rd_cost_batch(lam, frac_bits, dist, device=0, lib_path=None):
    L = load_library(lib_path)
    fb = np.ascontiguousarray(frac_bits, np.uint64); d = np.ascontiguousarray(dist, np.uint64); out = np.zeros(len(fb), np.float64)
    _chk(L, L.vvcx_rd_cost_batch(float(lam), fb.ctypes.data, d.ctypes.data, len(fb), out.ctypes.data, device))
    return out


def scan_order(w, h, device=0, lib_path=None):
    L = load_library(lib_path)
    idx = np.zeros(min(w, 32) * min(h, 32), np.uint16)
    _chk(L, L.vvcx_scan_order(w, h, idx.ctypes.data, device))
    return idx


def transform_quant_batch(org, pred, w, h, bit_depth, qp, device=0, lib_path=None):
    """n blocks back to back -> (levels int16 [n, h, w], reconstruction int16 [n, h, w], sse uint64 [n], cbf uint8 [n])"""
    L = load_library(lib_path)
    org = np.ascontiguousarray(org, np.int16).ravel(); pred = np.ascontiguousarray(pred, np.int16).ravel()
    n = org.size // (w * h)
    lev = np.zeros(org.size, np.int16); rec = np.zeros(org.size, np.int16); sse = np.zeros(n, np.uint64); cbf = np.zeros(n, np.uint8)
    _chk(L, L.vvcx_transform_quant_batch(org.ctypes.data, pred.ctypes.data, w, h, bit_depth, qp, n, lev.ctypes.data, rec.ctypes.data, sse.ctypes.data, cbf.ctypes.data, device))
    return lev.reshape(n, h, w), rec.reshape(n, h, w), sse, cbf


def depquant_batch(org, pred, w, h, bit_depth, qp, comp, mts_idx, cbf_cb, lam, s0, s1, device=0, lib_path=None):
    """vvcx_depquant_batch: n blocks back to back through transform + dependent quantiser + its dequantiser + inverse transform"""
    L = load_library(lib_path)
    org = np.ascontiguousarray(org, np.int16).ravel(); pred = np.ascontiguousarray(pred, np.int16).ravel()
    s0 = np.ascontiguousarray(s0, np.uint16); s1 = np.ascontiguousarray(s1, np.uint16)
    n = org.size // (w * h)
    lev = np.zeros(org.size, np.int16); rec = np.zeros(org.size, np.int16); sse = np.zeros(n, np.uint64); cbf = np.zeros(n, np.uint8)
    L.vvcx_depquant_batch.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_double, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int]
    _chk(L, L.vvcx_depquant_batch(org.ctypes.data, pred.ctypes.data, w, h, bit_depth, qp, comp, mts_idx, cbf_cb, lam, s0.ctypes.data, s1.ctypes.data, n,
                                  lev.ctypes.data, rec.ctypes.data, sse.ctypes.data, cbf.ctypes.data, device))
    return lev.reshape(n, h, w), rec.reshape(n, h, w), sse, cbf


def transform_skip_batch(resi, w, h, bit_depth, qp, lam, s0, s1, device=0, lib_path=None):
    """vvcx_transform_skip_batch: n residual blocks through the {DCT2, TS} pruning, xTransformSkip, RDOQ-TS, dequantisation + xITransformSkip and residual_codingTS's estimator"""
    L = load_library(lib_path)
    resi = np.ascontiguousarray(resi, np.int16).ravel()
    s0 = np.ascontiguousarray(s0, np.uint16); s1 = np.ascontiguousarray(s1, np.uint16)
    n = resi.size // (w * h)
    lev = np.zeros(resi.size, np.int16); out = np.zeros(resi.size, np.int16); a = np.zeros(n, np.int32); keep = np.zeros(n, np.uint8); bits = np.zeros(n, np.uint64)
    L.vvcx_transform_skip_batch.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_double, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
    _chk(L, L.vvcx_transform_skip_batch(resi.ctypes.data, w, h, bit_depth, qp, lam, s0.ctypes.data, s1.ctypes.data, n, lev.ctypes.data, out.ctypes.data, a.ctypes.data,
                                        keep.ctypes.data, bits.ctypes.data, device))
    return lev.reshape(n, h, w), out.reshape(n, h, w), a, keep, bits


def deblock_cu_table(planes, rows, bit_depth, qp, qp_c, beta_offset_div2=0, tc_offset_div2=0, device=0, lib_path=None):
    """vvcx_deblock_cu_table: in-loop deblocking of a 4:2:0 picture described by a CU table (rows of {ch, x, y, w, h, ispMode}, luma samples); returns the filtered planes"""
    L = load_library(lib_path)
    out = [np.ascontiguousarray(p).astype(np.uint16) for p in planes]
    rows = np.ascontiguousarray(rows, np.int32)
    h, w = out[0].shape
    L.vvcx_deblock_cu_table.argtypes = [C.c_int] * 8 + [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int]
    _chk(L, L.vvcx_deblock_cu_table(w, h, bit_depth, qp, int(qp_c[0]), int(qp_c[1]), beta_offset_div2, tc_offset_div2, rows.ctypes.data, len(rows),
                                    out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, device))
    return out


def isp_tu_batch(org, pred, tw, th, bit_depth, qp, lam, prev_cbf, cbf_inferred, s0, s1, device=0, lib_path=None):
    """vvcx_isp_tu_batch: n sub-partition blocks of ISP CUs through the implicit transform, the dependent quantiser with the ISP cbf context, dequantiser and inverse"""
    L = load_library(lib_path)
    org = np.ascontiguousarray(org, np.int16).ravel(); pred = np.ascontiguousarray(pred, np.int16).ravel()
    s0 = np.ascontiguousarray(s0, np.uint16); s1 = np.ascontiguousarray(s1, np.uint16)
    n = org.size // (tw * th)
    lev = np.zeros(org.size, np.int16); rec = np.zeros(org.size, np.int16); sse = np.zeros(n, np.uint64); cbf = np.zeros(n, np.uint8)
    L.vvcx_isp_tu_batch.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int]
    _chk(L, L.vvcx_isp_tu_batch(org.ctypes.data, pred.ctypes.data, tw, th, bit_depth, qp, lam, int(prev_cbf), int(cbf_inferred), s0.ctypes.data, s1.ctypes.data, n,
                                lev.ctypes.data, rec.ctypes.data, sse.ctypes.data, cbf.ctypes.data, device))
    return lev.reshape(n, th, tw), rec.reshape(n, th, tw), sse, cbf


def lfnst_depquant_batch(org, pred, w, h, bit_depth, qp, comp, lfnst_idx, intra_dir, cbf_cb, lam, s0, s1, device=0, lib_path=None):
    """vvcx_lfnst_depquant_batch: n blocks through DCT-II (zeroed out), forward LFNST, dependent quantiser, dequantiser, inverse LFNST, inverse DCT-II"""
    L = load_library(lib_path)
    org = np.ascontiguousarray(org, np.int16).ravel(); pred = np.ascontiguousarray(pred, np.int16).ravel()
    s0 = np.ascontiguousarray(s0, np.uint16); s1 = np.ascontiguousarray(s1, np.uint16)
    n = org.size // (w * h)
    lev = np.zeros(org.size, np.int16); rec = np.zeros(org.size, np.int16); sse = np.zeros(n, np.uint64); cbf = np.zeros(n, np.uint8)
    L.vvcx_lfnst_depquant_batch.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_double, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int]
    _chk(L, L.vvcx_lfnst_depquant_batch(org.ctypes.data, pred.ctypes.data, w, h, bit_depth, qp, comp, lfnst_idx, intra_dir, cbf_cb, lam, s0.ctypes.data, s1.ctypes.data, n,
                                        lev.ctypes.data, rec.ctypes.data, sse.ctypes.data, cbf.ctypes.data, device))
    return lev.reshape(n, h, w), rec.reshape(n, h, w), sse, cbf
