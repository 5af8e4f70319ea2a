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
TOOL_CU_REUSE = 1 << 11      # BestEncInfoCache, REUSE_CU_RESULTS (CL/TypeDef.h:291) - on in the reference build
TOOLS_DEFAULT = TOOL_MRL | TOOL_CU_REUSE


class VvcxError(RuntimeError):
    pass


class _Cfg(C.Structure):
    _fields_ = [("pic_w", C.c_int32), ("pic_h", C.c_int32), ("bit_depth", C.c_int32), ("ctu_size", C.c_int32),
                ("min_qt", C.c_int32 * 2), ("max_bt_depth", C.c_int32 * 2), ("max_bt_size", C.c_int32 * 2),
                ("max_tt_size", C.c_int32 * 2), ("dual_tree", C.c_int32), ("tile_cols", C.c_int32), ("tile_rows", C.c_int32),
                ("tools", C.c_uint32), ("chroma", C.c_int32), ("max_frames", C.c_int32), ("device", C.c_int32)]


class _Slice(C.Structure):
    _fields_ = [("qp", C.c_int32), ("qp_c", C.c_int32 * 2), ("lam", C.c_double), ("dist_weight", C.c_double * 2)]


class _Frame(C.Structure):
    _fields_ = [("org", C.c_void_p * 3), ("reco", C.c_void_p * 3), ("stride", C.c_int32 * 3)]


class _Task(C.Structure):
    _fields_ = [("frame", C.c_int32), ("ctu_rs_addr", C.c_int32)]


CTU_DTYPE = np.dtype([("dist", "<u8"), ("frac_bits", "<u8"), ("cost", "<f8"), ("n_cu", "<i4")], align=True)
CU_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "<i2"), ("h", "<i2"), ("ch_type", "u1"), ("qt_depth", "u1"),
                     ("bt_depth", "u1"), ("mt_depth", "u1"), ("depth", "u1"), ("intra_dir", "u1"), ("mrl_idx", "u1"),
                     ("cbf", "u1"), ("split_series", "<u8")], align=True)

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
    L.vvcx_get_cus.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.vvcx_last_kernel_ms.restype = C.c_float
    L.vvcx_last_kernel_ms.argtypes = [C.c_void_p]
    L.vvcx_get_counters.argtypes = [C.c_void_p, C.c_void_p]
    L.vvcx_get_profile.argtypes = [C.c_void_p, C.c_void_p]
    L.vvcx_last_error.restype = C.c_char_p
    L.vvcx_ctus_per_frame.argtypes = [C.c_void_p]
    L.vvcx_resident_streams.argtypes = [C.c_void_p]
    _libs[path] = L
    return L


class VvcxEncoder:
    """≙ one EncCu instance (EL/EncCu.h:80-230): create/init → per-slice set-up → compressCtu calls → destroy."""

    def __init__(self, width, height, bit_depth=8, tile_cols=1, tile_rows=1, chroma=True, tools=TOOLS_DEFAULT,
                 max_frames=1, device=0, lib_path=None):
        self.L = load_library(lib_path)
        c = _Cfg()
        c.pic_w, c.pic_h, c.bit_depth, c.ctu_size = width, height, bit_depth, 128
        c.min_qt[0], c.min_qt[1] = 8, 4                   # BIN/encoder_intra.cfg:98-99
        c.max_bt_depth[0], c.max_bt_depth[1] = 3, 3       # :102-103
        c.max_bt_size[0], c.max_bt_size[1] = 32, 64       # CL/CommonDef.h:427,437
        c.max_tt_size[0], c.max_tt_size[1] = 32, 32
        c.dual_tree, c.tile_cols, c.tile_rows, c.tools = 1, tile_cols, tile_rows, tools
        c.chroma, c.max_frames, c.device = int(chroma), max_frames, device
        self.cfg = c
        self.h = C.c_void_p()
        self._chk(self.L.vvcx_create(C.byref(c), C.byref(self.h)))
        self.ctus_per_frame = self.L.vvcx_ctus_per_frame(self.h)
        self.n_frames = 0

    def resident_streams(self):
        return int(self.L.vvcx_resident_streams(self.h))

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

    def set_slice(self, qp, qp_c, lam, dist_weight):
        s = _Slice()
        s.qp = qp
        s.qp_c[0], s.qp_c[1] = qp_c
        s.lam = lam
        s.dist_weight[0], s.dist_weight[1] = dist_weight
        self._chk(self.L.vvcx_set_slice(self.h, C.byref(s)))

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

    def compress_bound_frames(self, stream=None):
        out = np.zeros(self.n_frames * self.ctus_per_frame, CTU_DTYPE)
        self._chk(self.L.vvcx_compress_bound_frames(self.h, out.ctypes.data, stream))
        return out.reshape(self.n_frames, self.ctus_per_frame)

    def get_cus(self, frame):
        n = C.c_int()
        cus = np.zeros(self.ctus_per_frame * 2048, CU_DTYPE)
        self._chk(self.L.vvcx_get_cus(self.h, frame, cus.ctypes.data, len(cus), C.byref(n)))
        return cus[:n.value].copy()

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
