"""ctypes bindings for libmfcchip.so and the Python mirror of the reference's parameterizer API.

``MfccHip`` keeps the method names, argument meaning and error behaviour of the reference's
``ParamBase`` / ``MfccBase`` interface (parambase.h:6-33, mfccbase.h:21-35) that ``MfccCpu`` and
``MfccOpenCL`` implement: ``set_window, set_input, flush, set_alpha, apply,
get_output_data_width, get_output_data, get_input_buffer_size, estimated_window_count``.
Errors the reference throws as ``std::runtime_error`` surface as ``MfxError`` with the same text.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NORM_NONE, NORM_CMN, NORM_CVN, NORM_MINMAX = 0, 1, 2, 3   # normalizer.h:5
DYN_NONE, DYN_DELTA, DYN_ACC = 0, 1, 2                     # parambase.h:9


class MfxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


class MfxConfig(C.Structure):
    """POD mirror of ``mfx_config`` (include/mfx.h)."""
    _fields_ = [
        ("input_buffer_size", C.c_int32),
        ("window_size", C.c_int32),
        ("shift", C.c_int32),
        ("num_banks", C.c_int32),
        ("sample_rate", C.c_float),
        ("low_freq", C.c_float),
        ("high_freq", C.c_float),
        ("ceps_len", C.c_int32),
        ("want_c0", C.c_int32),
        ("lift_coef", C.c_float),
        ("norm", C.c_int32),
        ("dyn", C.c_int32),
        ("delta_l1", C.c_int32),
        ("delta_l2", C.c_int32),
        ("norm_after_dyn", C.c_int32),
        ("fft_size", C.c_int32),
        ("channels", C.c_int32),
        ("bug_compat", C.c_int32),
        ("batch_norm_stats", C.c_int32),
        ("engine", C.c_int32),
        ("tail_split", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


# every symbol include/mfx.h declares
EXPORTED_SYMBOLS = (
    "mfx_create", "mfx_destroy", "mfx_last_error", "mfx_status_string", "mfx_abi_version",
    "mfx_set_window", "mfx_set_input", "mfx_flush", "mfx_set_alpha", "mfx_apply",
    "mfx_get_output_data_width", "mfx_get_output_data", "mfx_get_input_buffer_size",
    "mfx_apply_alphas", "mfx_get_output_data_alpha", "mfx_host_mel_lane_plan",
    "mfx_host_dct_mfma_operands",
    "mfx_estimated_window_count", "mfx_max_frames_out", "mfx_fft_size",
    "mfx_batch_frames", "mfx_batch_plan", "mfx_batch_run_device", "mfx_batch_run_host", "mfx_batch_overlap",
    "mfx_alloc_pinned", "mfx_free_pinned",
    "mfx_set_stream", "mfx_synchronize", "mfx_profile_enable", "mfx_profile_read",
    "mfx_dominant_kernel_name", "mfx_debug_read", "mfx_plan_create", "mfx_plan_set_aligned",
    "mfx_host_mel_table", "mfx_host_dct_matrix", "mfx_host_frame_count",
)


def library_path():
    # MFX_LIB: developer override to A/B an experimental build of the same ABI
    return os.environ.get("MFX_LIB") or os.path.join(_HERE, "libmfcchip.so")


_lib = None


def load_library():
    """Load libmfcchip.so and declare the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = library_path()
    if not os.path.exists(p):
        raise MfxError(-6, "libmfcchip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  A process must hold ONE HIP
    # runtime: if torch is going to be used alongside (device tensors, torch.distributed), it has to be
    # loaded first so that libmfcchip.so binds to the runtime already in the process instead of
    # bringing /opt/rocm's in as a second one ("no ROCm-capable device" on whichever comes second).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(p)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    fp, sp = C.POINTER(C.c_float), C.POINTER(C.c_int16)
    L.mfx_create.argtypes = [C.POINTER(MfxConfig), C.c_int, C.POINTER(vp)]
    L.mfx_destroy.argtypes = [vp]
    L.mfx_destroy.restype = None
    L.mfx_last_error.argtypes, L.mfx_last_error.restype = [vp], C.c_char_p
    L.mfx_status_string.argtypes, L.mfx_status_string.restype = [C.c_int], C.c_char_p
    L.mfx_abi_version.argtypes = []
    L.mfx_set_window.argtypes = [vp, fp]
    L.mfx_set_input.argtypes = [vp, sp, i32, C.POINTER(i32)]
    L.mfx_flush.argtypes = [vp, C.POINTER(i32)]
    L.mfx_set_alpha.argtypes = [vp, C.c_float]
    L.mfx_apply.argtypes = [vp]
    L.mfx_get_output_data_width.argtypes = [vp]
    L.mfx_get_output_data.argtypes = [vp, fp, i32]
    L.mfx_apply_alphas.argtypes = [vp, fp, i32]
    L.mfx_get_output_data_alpha.argtypes = [vp, i32, fp, i32]
    L.mfx_get_input_buffer_size.argtypes = [vp]
    L.mfx_estimated_window_count.argtypes = [vp, i32]
    L.mfx_max_frames_out.argtypes = [vp]
    L.mfx_fft_size.argtypes = [vp]
    L.mfx_batch_frames.argtypes, L.mfx_batch_frames.restype = [vp, i64], i64
    L.mfx_batch_plan.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.mfx_batch_run_device.argtypes = [vp, vp, i64, vp]
    L.mfx_batch_run_host.argtypes = [vp, sp, i64, fp]
    L.mfx_batch_overlap.argtypes = [vp, C.c_int]
    L.mfx_set_stream.argtypes = [vp, vp]
    L.mfx_synchronize.argtypes = [vp]
    L.mfx_profile_enable.argtypes = [vp, C.c_int]
    L.mfx_profile_read.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_double), C.c_int]
    L.mfx_dominant_kernel_name.argtypes, L.mfx_dominant_kernel_name.restype = [vp], C.c_char_p
    L.mfx_plan_create.argtypes = [C.POINTER(MfxConfig), C.POINTER(vp)]
    L.mfx_plan_set_aligned.argtypes = [vp, C.c_int]
    L.mfx_debug_read.argtypes, L.mfx_debug_read.restype = [vp, C.c_int, vp, i64], i64
    L.mfx_host_mel_table.argtypes = [i32, i32, C.c_float, C.c_float, C.c_float, C.c_float, fp, C.POINTER(i32)]
    L.mfx_host_dct_matrix.argtypes = [i32, i32, i32, C.c_float, fp]
    L.mfx_host_frame_count.argtypes, L.mfx_host_frame_count.restype = [i64, i32, i32], i64
    _lib = L
    return L


def host_mel_table(num_banks, fft_size, sample_rate, low_freq, high_freq, alpha=1.0):
    """Mel table exactly as uploaded to the device (host code, no GPU needed)."""
    L = load_library()
    w = np.zeros((2, fft_size), dtype=np.float32)
    beg = np.zeros(num_banks + 2, dtype=np.int32)
    rc = L.mfx_host_mel_table(num_banks, fft_size, sample_rate, low_freq, high_freq, alpha,
                              w.ctypes.data_as(C.POINTER(C.c_float)), beg.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise MfxError(rc, "mfx_host_mel_table failed")
    return w, beg


def host_mel_lane_plan(lanes, weights, beg, max_read_bin):
    """Lane plan of the fused kernels' mel walk (lanes = 16, 32 or 64): dict(rounds, L, row_stride, start, fid, w)."""
    L = load_library()
    weights = np.ascontiguousarray(weights, dtype=np.float32)
    beg = np.ascontiguousarray(beg, dtype=np.int32)
    nb, fft = beg.size - 2, weights.shape[1]
    ip, fpt = C.POINTER(C.c_int32), C.POINTER(C.c_float)
    fn = L.mfx_host_mel_lane_plan
    fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, fpt, ip, C.c_int32, ip, ip, ip, ip, fpt, C.c_int64]
    fn.restype = C.c_int
    Ls = np.zeros(8, np.int32)
    rs = C.c_int32(0)
    rounds = fn(lanes, nb, fft, weights.ctypes.data_as(fpt), beg.ctypes.data_as(ip), int(max_read_bin),
                Ls.ctypes.data_as(ip), C.byref(rs), None, None, None, 0)
    if rounds < 0:
        raise MfxError(rounds, "mfx_host_mel_lane_plan failed")
    start = np.zeros((rounds, lanes), np.int32)
    fid = np.zeros((rounds, lanes), np.int32)
    w = np.zeros((lanes, rs.value), np.float32)
    rc = fn(lanes, nb, fft, weights.ctypes.data_as(fpt), beg.ctypes.data_as(ip), int(max_read_bin), Ls.ctypes.data_as(ip),
            C.byref(rs), start.ctypes.data_as(ip), fid.ctypes.data_as(ip), w.ctypes.data_as(fpt), w.size)
    if rc != rounds:
        raise MfxError(rc, "mfx_host_mel_lane_plan failed")
    return dict(rounds=rounds, L=Ls[:rounds].copy(), row_stride=rs.value, start=start, fid=fid, w=w)


def host_dct_mfma_operands(matrix):
    """[tile][K step][64] operands of the DCT on the matrix pipe for a [num_banks][dct_len] matrix."""
    L = load_library()
    m = np.ascontiguousarray(matrix, dtype=np.float32)
    nb, dl = m.shape
    fpt = C.POINTER(C.c_float)
    fn = L.mfx_host_dct_mfma_operands
    fn.argtypes = [C.c_int32, C.c_int32, fpt, fpt, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    fn.restype = C.c_int64
    tl, ks = C.c_int32(0), C.c_int32(0)
    n = fn(nb, dl, m.ctypes.data_as(fpt), None, 0, C.byref(tl), C.byref(ks))
    out = np.zeros(n, np.float32)
    fn(nb, dl, m.ctypes.data_as(fpt), out.ctypes.data_as(fpt), out.size, C.byref(tl), C.byref(ks))
    return out.reshape(tl.value, ks.value, 64)


def host_dct_matrix(num_banks, ceps_len, want_c0, lift_coef):
    L = load_library()
    m = np.zeros((num_banks, ceps_len + (1 if want_c0 else 0)), dtype=np.float32)
    rc = L.mfx_host_dct_matrix(num_banks, ceps_len, int(bool(want_c0)), lift_coef,
                               m.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != 0:
        raise MfxError(rc, "mfx_host_dct_matrix failed")
    return m


def host_frame_count(samples, window_size, shift):
    return int(load_library().mfx_host_frame_count(int(samples), int(window_size), int(shift)))


def reference_window(window_size):
    """The window the reference's caller builds (ASR_OCL.cpp:149-151):
    ``(float)(0.56f - 0.46f*cos((2.0f*M_PI*i)/W)) / 32768.f`` -- note 0.56 and a denominator of W."""
    i = np.arange(window_size, dtype=np.float64)
    inner = np.float32(0.56) - np.float32(0.46) * np.cos((2.0 * np.pi * i) / window_size)
    return (inner.astype(np.float32) / np.float32(32768.0)).astype(np.float32)


def plan_kernel(window_size, shift, num_banks, sample_rate, ceps_len, want_c0=False, dyn=DYN_NONE, fft_size=0, channels=1,
                aligned=True, engine=0, low_freq=64.0, high_freq=None, input_buffer_size=0):
    """Which front-end kernel the batch entries run for a shape -- asked of a PLANNING handle (mfx_plan_create: the
    library's own configuration checks, host-built tables and LDS sums, no device, nothing computed).  Works without a
    GPU; returns the kernel's name as rocprofv3 prints it.  Raises MfxError for a configuration mfx_create refuses."""
    L = load_library()
    cfg = MfxConfig(int(input_buffer_size or 100 * shift + window_size), int(window_size), int(shift), int(num_banks),
                    float(sample_rate), float(low_freq), float(sample_rate / 2 if high_freq is None else high_freq),
                    int(ceps_len), int(bool(want_c0)), 22.0, NORM_NONE, int(dyn), 3, 3, 1, int(fft_size), int(channels), 1, 0,
                    int(engine), 0)
    h = C.c_void_p()
    rc = L.mfx_plan_create(C.byref(cfg), C.byref(h))
    if rc != 0:
        raise MfxError(rc, "mfx_plan_create: " + L.mfx_status_string(rc).decode())
    try:
        L.mfx_plan_set_aligned(h, int(bool(aligned)))
        return L.mfx_dominant_kernel_name(h).decode()
    finally:
        L.mfx_destroy(h)


# Shape -> front-end kernel of the batch entries (DESIGN.md section 5 prints this table; tests/test_host.py pins it through
# plan_kernel, tests/test_parity_gpu.py checks a real handle against it).  BASELINE.json configs first, then every row of
# profiles/r03/shapes_beyond_baseline.txt and the limits of each kernel.
KERNEL_TABLE = (
    # (what, plan_kernel keyword arguments, kernel)
    ("C1  a0001.wav, 16 kHz 25/10 ms, 512 pt, 26 mel, 13 MFCC + d + dd", dict(window_size=400, shift=160, num_banks=26, sample_rate=16000.0, ceps_len=13, dyn=DYN_ACC), "k_front512"),
    ("C2 / C4  16 kHz 25/10 ms, 512 pt, 40 mel, 13 MFCC + d + dd", dict(window_size=400, shift=160, num_banks=40, sample_rate=16000.0, ceps_len=13, dyn=DYN_ACC), "k_front512"),
    ("R   reference main() defaults: 15 mel, 12 MFCC + c0, CVN", dict(window_size=400, shift=160, num_banks=15, sample_rate=16000.0, ceps_len=12, want_c0=True), "k_front512"),
    ("C3  16 kHz 25/10 ms zero padded to 1024 pt, 80 mel, 13 MFCC", dict(window_size=400, shift=160, num_banks=80, sample_rate=16000.0, ceps_len=13, fft_size=1024), "k_front1024"),
    ("C5  44.1 kHz stereo, 1102/441, 2048 pt, 128 mel, 40 MFCC + d + dd", dict(window_size=1102, shift=441, num_banks=128, sample_rate=44100.0, ceps_len=40, dyn=DYN_ACC, channels=2), "k_front2048"),
    ("C2 shape, odd shift 161 (unaligned build)", dict(window_size=400, shift=161, num_banks=40, sample_rate=16000.0, ceps_len=13, dyn=DYN_ACC, aligned=False), "k_front512"),
    ("fbank-80 at 16 kHz (80 log mel energies, no DCT), 512 pt", dict(window_size=400, shift=160, num_banks=80, sample_rate=16000.0, ceps_len=0), "k_front512"),
    ("8 kHz telephony 25/10 ms: 256 pt, zero-stuffed", dict(window_size=200, shift=80, num_banks=23, sample_rate=8000.0, ceps_len=13, dyn=DYN_ACC), "k_front512"),
    ("8 kHz stereo, 256 pt", dict(window_size=200, shift=80, num_banks=23, sample_rate=8000.0, ceps_len=13, channels=2), "k_front512"),
    ("16 kHz stereo, 512 pt", dict(window_size=400, shift=160, num_banks=40, sample_rate=16000.0, ceps_len=13, channels=2), "k_front512"),
    ("11.025 kHz 25/10 ms: 512 pt", dict(window_size=276, shift=110, num_banks=40, sample_rate=11025.0, ceps_len=13), "k_front512"),
    ("128-point transform (zero-stuffed twice over)", dict(window_size=128, shift=64, num_banks=20, sample_rate=8000.0, ceps_len=12), "k_front512"),
    ("64-point transform (zero-stuffed three times over)", dict(window_size=64, shift=32, num_banks=12, sample_rate=8000.0, ceps_len=8), "k_front512"),
    ("256 pt with the stuffed form switched off (MFX_ENGINE_NO_STUFF256)", dict(window_size=200, shift=80, num_banks=23, sample_rate=8000.0, ceps_len=13, engine=128), "k_front_wave"),
    ("512 pt, more than 128 filters", dict(window_size=400, shift=160, num_banks=160, sample_rate=16000.0, ceps_len=13), "k_front_wave"),
    ("22.05 kHz 25/10 ms: 1024 pt (552 taps, aligned frames)", dict(window_size=552, shift=220, num_banks=80, sample_rate=22050.0, ceps_len=13), "k_front1024"),
    ("22.05 kHz fbank-80 (no DCT), 1024 pt", dict(window_size=552, shift=220, num_banks=80, sample_rate=22050.0, ceps_len=0), "k_front1024"),
    ("32 kHz 25/10 ms: 1024 pt (800 taps)", dict(window_size=800, shift=320, num_banks=80, sample_rate=32000.0, ceps_len=13), "k_front1024"),
    ("C3 shape, odd shift 161 (unaligned build)", dict(window_size=400, shift=161, num_banks=80, sample_rate=16000.0, ceps_len=13, fft_size=1024, aligned=False), "k_front1024"),
    ("1024 pt, 800 taps on UNALIGNED frames", dict(window_size=800, shift=321, num_banks=80, sample_rate=32000.0, ceps_len=13, aligned=False), "k_front_reg"),
    ("1024 pt, more than 80 filters", dict(window_size=400, shift=160, num_banks=96, sample_rate=16000.0, ceps_len=13, fft_size=1024), "k_front_reg"),
    ("1024 pt, more than 16 columns with a DCT", dict(window_size=400, shift=160, num_banks=80, sample_rate=16000.0, ceps_len=20, fft_size=1024), "k_front_reg"),
    ("1024 pt stereo", dict(window_size=552, shift=220, num_banks=80, sample_rate=22050.0, ceps_len=13, channels=2), "k_front_reg"),
    ("C3 shape with k_front1024 switched off (MFX_ENGINE_NO_FRONT1024)", dict(window_size=400, shift=160, num_banks=80, sample_rate=16000.0, ceps_len=13, fft_size=1024, engine=1), "k_front_reg"),
    ("44.1 kHz mono, odd shift 441 (any-alignment build)", dict(window_size=1102, shift=441, num_banks=128, sample_rate=44100.0, ceps_len=40, dyn=DYN_ACC, aligned=False), "k_front2048"),
    ("48 kHz 25/10 ms: 2048 pt (1200 taps, 20-row build)", dict(window_size=1200, shift=480, num_banks=128, sample_rate=48000.0, ceps_len=40, dyn=DYN_ACC), "k_front2048"),
    ("n_fft = win_length = 2048, hop 512 (32-row build)", dict(window_size=2048, shift=512, num_banks=128, sample_rate=44100.0, ceps_len=40), "k_front2048"),
    ("C5 shape with k_front2048 switched off (MFX_ENGINE_NO_FRONT2048)", dict(window_size=1102, shift=441, num_banks=128, sample_rate=44100.0, ceps_len=40, dyn=DYN_ACC, channels=2, engine=4), "k_front_reg"),
    ("4096 pt (50 ms at 48 kHz): spectrum + k_melcep", dict(window_size=2400, shift=480, num_banks=64, sample_rate=48000.0, ceps_len=13), "k_front_reg"),
    ("4096 pt, 20 wide filters (ADVICE r3)", dict(window_size=2400, shift=480, num_banks=20, sample_rate=48000.0, ceps_len=12), "k_front_reg"),
    ("any shape on the streaming interface's kernels (MFX_ENGINE_STREAM_KERNELS)", dict(window_size=400, shift=160, num_banks=40, sample_rate=16000.0, ceps_len=13, dyn=DYN_ACC, engine=8), "k_front512"),
)


ENGINE_NO_FRONT1024, ENGINE_FUSE_DELTA, ENGINE_NO_FRONT2048, ENGINE_STREAM_KERNELS, ENGINE_NORM_TWO_KERNELS = 1, 2, 4, 8, 16
ENGINE_DMA_SMALL_BLOCKS, ENGINE_NO_DCT_SPLIT, ENGINE_NO_STUFF256, ENGINE_FRONT1024_12_WAVES = 32, 64, 128, 256     # mfx_config.engine bits (include/mfx.h)


class MfccHip:
    """Python mirror of ``class MfccHip : public MfccBase`` (host/afet_param.h).

    Constructor arguments are those of MfccBase (mfccbase.h:21-35) in the same order; ``device``
    replaces MfccOpenCL's trailing ``cl_device_id`` (mfccopencl.h:60).
    """

    def __init__(self, input_buffer_size, window_size, shift, num_banks, sample_rate, low_freq, high_freq,
                 ceps_len, want_c0, lift_coef, norm=NORM_NONE, dyn=DYN_NONE, delta_l1=1, delta_l2=1,
                 norm_after_dyn=True, device=0, fft_size=0, channels=1, bug_compat=True, batch_norm_stats=0, engine=0,
                 tail_split=0):
        self._L = load_library()
        self.cfg = MfxConfig(int(input_buffer_size), int(window_size), int(shift), int(num_banks),
                             float(sample_rate), float(low_freq), float(high_freq), int(ceps_len),
                             int(bool(want_c0)), float(lift_coef), int(norm), int(dyn), int(delta_l1),
                             int(delta_l2), int(bool(norm_after_dyn)), int(fft_size), int(channels),
                             int(bool(bug_compat)), int(batch_norm_stats), int(engine), int(tail_split))
        h = C.c_void_p()
        rc = self._L.mfx_create(C.byref(self.cfg), int(device), C.byref(h))
        if rc != 0:
            raise MfxError(rc, "mfx_create: " + self._L.mfx_status_string(rc).decode())
        self._h = h
        self._plan_rows = None
        self._plan_total = 0

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.mfx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MfxError(rc, self._L.mfx_last_error(self._h).decode() or self._L.mfx_status_string(rc).decode())

    # -- ParamBase interface (parambase.h:23-32) ----------------------------------------------------
    def get_input_buffer_size(self):
        return self._L.mfx_get_input_buffer_size(self._h)

    def estimated_window_count(self, samples):
        return self._L.mfx_estimated_window_count(self._h, int(samples))

    def set_alpha(self, alpha):
        self._chk(self._L.mfx_set_alpha(self._h, float(alpha)))

    def set_window(self, window):
        w = np.ascontiguousarray(window, dtype=np.float32)
        if w.size != self.cfg.window_size:
            raise ValueError("window must have window_size taps")
        self._chk(self._L.mfx_set_window(self._h, w.ctypes.data_as(C.POINTER(C.c_float))))

    def set_input(self, data):
        pcm = np.ascontiguousarray(data, dtype=np.int16)
        n = C.c_int32(0)
        self._chk(self._L.mfx_set_input(self._h, pcm.ctypes.data_as(C.POINTER(C.c_int16)), pcm.size, C.byref(n)))
        return n.value

    def flush(self):
        n = C.c_int32(0)
        self._chk(self._L.mfx_flush(self._h, C.byref(n)))
        return n.value

    def apply(self):
        self._chk(self._L.mfx_apply(self._h))

    def get_output_data_width(self):
        return self._L.mfx_get_output_data_width(self._h)

    def get_output_data(self, window_count):
        out = np.empty((max(int(window_count), 0), self.get_output_data_width()), dtype=np.float32)
        self._chk(self._L.mfx_get_output_data(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), int(window_count)))
        return out

    def apply_alphas(self, alphas):
        """VTLN sweep: every warp factor of `alphas` over the stored spectrum of the current block in
        one call (the reference's alpha loop, ASR_OCL.cpp:236-243)."""
        a = np.ascontiguousarray(alphas, dtype=np.float32)
        self._chk(self._L.mfx_apply_alphas(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), int(a.size)))

    def get_output_data_alpha(self, alpha_index, window_count):
        out = np.empty((max(int(window_count), 0), self.get_output_data_width()), dtype=np.float32)
        self._chk(self._L.mfx_get_output_data_alpha(self._h, int(alpha_index), out.ctypes.data_as(C.POINTER(C.c_float)),
                                                    int(window_count)))
        return out

    # -- extensions ---------------------------------------------------------------------------------
    def max_frames_out(self):
        return self._L.mfx_max_frames_out(self._h)

    def fft_size(self):
        return self._L.mfx_fft_size(self._h)

    def process_stream(self, pcm, block_samples=0, alpha=1.0):
        """The per-file loop of the reference driver (ASR_OCL.cpp:227-301): blocks of at most
        get_input_buffer_size() samples through set_input/apply/get_output_data, then flush."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        limit = self.get_input_buffer_size()
        if block_samples and block_samples < limit:
            limit = int(block_samples)
        rows, pos = [], 0
        while pos < pcm.size:
            n = self.set_input(pcm[pos:pos + limit])
            self.set_alpha(alpha)
            self.apply()
            rows.append(self.get_output_data(n))
            pos += limit
        n = self.flush()
        if n > 0:
            self.set_alpha(alpha)
            self.apply()
            rows.append(self.get_output_data(n))
        return np.concatenate(rows, 0) if rows else np.zeros((0, self.get_output_data_width()), np.float32)

    # -- batch interface ----------------------------------------------------------------------------
    def batch_frames(self, samples):
        return int(self._L.mfx_batch_frames(self._h, int(samples)))

    def batch_plan(self, offsets, lengths):
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        ln = np.ascontiguousarray(lengths, dtype=np.int64)
        assert off.size == ln.size
        rows = np.zeros(off.size, dtype=np.int64)
        total = C.c_int64(0)
        p64 = C.POINTER(C.c_int64)
        self._chk(self._L.mfx_batch_plan(self._h, off.size, off.ctypes.data_as(p64), ln.ctypes.data_as(p64),
                                         rows.ctypes.data_as(p64), C.byref(total)))
        self._plan_rows, self._plan_total = rows, total.value
        return rows, total.value

    def batch_run_device(self, d_pcm_ptr, pcm_samples_total, d_out_ptr):
        """Device pointers in, device pointer out; asynchronous on the handle's stream."""
        self._chk(self._L.mfx_batch_run_device(self._h, C.c_void_p(int(d_pcm_ptr)), int(pcm_samples_total),
                                               C.c_void_p(int(d_out_ptr))))

    def batch_run_host(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        total = pcm.size // max(self.cfg.channels, 1)
        out = np.zeros((self._plan_total, self.get_output_data_width()), dtype=np.float32)
        self._chk(self._L.mfx_batch_run_host(self._h, pcm.ctypes.data_as(C.POINTER(C.c_int16)), total,
                                             out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def batch_overlap(self, enable=True):
        """Let the delta tail of a batch overlap the next batch's front end (results complete after synchronize())."""
        self._chk(self._L.mfx_batch_overlap(self._h, int(bool(enable))))

    def set_stream(self, hip_stream_handle):
        self._chk(self._L.mfx_set_stream(self._h, C.c_void_p(int(hip_stream_handle))))

    def synchronize(self):
        self._chk(self._L.mfx_synchronize(self._h))

    def profile_enable(self, on=True):
        self._chk(self._L.mfx_profile_enable(self._h, int(bool(on))))

    def profile_read(self, reset=True):
        n, ms = C.c_int32(0), C.c_double(0)
        self._chk(self._L.mfx_profile_read(self._h, C.byref(n), C.byref(ms), int(bool(reset))))
        return n.value, ms.value

    def dominant_kernel_name(self):
        return self._L.mfx_dominant_kernel_name(self._h).decode()

    def debug_read(self, kind):
        W2, nb = self.fft_size(), self.cfg.num_banks
        dl = self.cfg.ceps_len + (1 if self.cfg.want_c0 else 0)
        if kind == 1:
            buf = np.zeros(nb + 2, dtype=np.int32)
        elif kind == 0:
            buf = np.zeros(2 * W2, dtype=np.float32)
        elif kind == 2:
            buf = np.zeros(max(nb * dl, 1), dtype=np.float32)
        else:
            buf = np.zeros(64 * 1024 * 1024 // 4, dtype=np.float32)
        n = self._L.mfx_debug_read(self._h, int(kind), buf.ctypes.data_as(C.c_void_p), buf.nbytes)
        if n < 0:
            raise MfxError(int(n), "mfx_debug_read failed")
        return buf[:n].copy()
