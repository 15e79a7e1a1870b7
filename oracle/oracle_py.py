"""ctypes bindings for the parity checker (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  It loads ``oracle/liboracle.so`` (this repo's C restatement of the reference's
CPU path, ``oracle/mfcc_oracle.c``) and, when present, the libraries under ``oracle/_ref/`` (the
reference's own translation units, built in place from /root/reference by ``make -C oracle ref``:
``libref_stages*.so`` segmenter/delta/normalizer/base objects, ``libref_mfcccpu*.so`` mfcccpu.cpp without FFTW;
``*_f32`` = the build whose unqualified libm calls select the float overloads, as the reference's toolchain does).
"""
import ctypes as C
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NORM_NONE, NORM_CMN, NORM_CVN, NORM_MINMAX = 0, 1, 2, 3
DYN_NONE, DYN_DELTA, DYN_ACC = 0, 1, 2


class OrcConfig(C.Structure):
    _fields_ = [
        ("input_buffer_size", C.c_int),
        ("window_size", C.c_int),
        ("shift", C.c_int),
        ("num_banks", C.c_int),
        ("sample_rate", C.c_float),
        ("low_freq", C.c_float),
        ("high_freq", C.c_float),
        ("ceps_len", C.c_int),
        ("want_c0", C.c_int),
        ("lift_coef", C.c_float),
        ("norm", C.c_int),
        ("dyn", C.c_int),
        ("delta_l1", C.c_int),
        ("delta_l2", C.c_int),
        ("norm_after_dyn", C.c_int),
        ("fft_mode", C.c_int),
    ]


def make_config(input_buffer_size, window_size=400, shift=160, num_banks=26, sample_rate=16000.0,
                low_freq=64.0, high_freq=None, ceps_len=13, want_c0=False, lift_coef=22.0,
                norm=NORM_NONE, dyn=DYN_ACC, delta_l1=3, delta_l2=3, norm_after_dyn=True, fft_mode=0):
    if high_freq is None:
        high_freq = sample_rate / 2
    return OrcConfig(int(input_buffer_size), int(window_size), int(shift), int(num_banks),
                     float(sample_rate), float(low_freq), float(high_freq), int(ceps_len),
                     int(bool(want_c0)), float(lift_coef), int(norm), int(dyn), int(delta_l1),
                     int(delta_l2), int(bool(norm_after_dyn)), int(fft_mode))


def reference_window(window_size):
    """Caller-side window of ASR_OCL.cpp:149-151: float32 expression order."""
    i = np.arange(window_size, dtype=np.float64)
    arg = (np.float32(2.0) * np.float64(np.pi) * i) / window_size  # (2.0f * M_PI * i) / window_size, double
    inner = np.float32(0.56) - np.float32(0.46) * np.cos(arg)        # float - float*double -> double
    return (inner.astype(np.float32) / np.float32(32768.0)).astype(np.float32)


def read_wav_pcm16(path):
    """Minimal RIFF/PCM16 reader (the reference uses libsndfile, ASR_OCL.cpp:174-231)."""
    with open(path, "rb") as f:
        b = f.read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE", "not a RIFF/WAVE file"
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, sz = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", b[pos + 8:pos + 24])
        elif cid == b"data":
            data = b[pos + 8:pos + 8 + sz]
        pos += 8 + sz + (sz & 1)
    assert fmt is not None and data is not None and fmt[0] == 1 and fmt[5] == 16
    pcm = np.frombuffer(data, dtype="<i2").astype(np.int16)
    return pcm.reshape(-1, fmt[1]), fmt[2]


_lib = None


def lib(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(p):
        raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
    L = C.CDLL(p)
    fp, ip, sp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_short)
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.POINTER(OrcConfig)]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_set_window.argtypes = [C.c_void_p, fp]
    L.orc_set_input.argtypes = [C.c_void_p, sp, C.c_int]
    L.orc_flush.argtypes = [C.c_void_p]
    L.orc_set_alpha.argtypes = [C.c_void_p, C.c_float]
    L.orc_apply.argtypes = [C.c_void_p]
    L.orc_get_output_data_width.argtypes = [C.c_void_p]
    L.orc_get_output_data.argtypes = [C.c_void_p, fp, C.c_int]
    L.orc_get_input_buffer_size.argtypes = [C.c_void_p]
    L.orc_estimated_window_count.argtypes = [C.c_void_p, C.c_int]
    L.orc_window_limit.argtypes = [C.c_void_p]
    L.orc_fft_size.argtypes = [C.c_void_p]
    L.orc_set_bug_compat.argtypes = [C.c_void_p, C.c_int]
    L.orc_set_libm_binding.argtypes = [C.c_void_p, C.c_int]
    L.orc_stage_filter.argtypes = [C.c_void_p, C.c_int]
    L.orc_stage_dct.argtypes = [C.c_void_p, C.c_int]
    for name in ("frames", "fft", "mel", "mfcc", "filters", "dct_matrix"):
        fn = getattr(L, "orc_tap_" + name)
        fn.restype, fn.argtypes = fp, [C.c_void_p]
    L.orc_tap_filter_beg.restype, L.orc_tap_filter_beg.argtypes = ip, [C.c_void_p]
    L.orc_tap_norm_stats.restype, L.orc_tap_norm_stats.argtypes = fp, [C.c_void_p, C.c_int, C.c_int]
    L.orc_ewc.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_output_width.argtypes = [C.c_int] * 4
    L.orc_delta_apply.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp]
    L.orc_normalize.argtypes = [C.c_int, fp, C.c_int, C.c_int, C.c_int, fp, fp, fp]
    L.orc_segment.argtypes = [sp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp]
    L.orc_rfft_rows.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int]
    L.orc_run_utterance.argtypes = [C.POINTER(OrcConfig), fp, C.c_float, C.c_int, sp, C.c_int, C.c_int, fp]
    L.orc_run_batch.restype = C.c_longlong
    L.orc_run_batch.argtypes = [C.POINTER(OrcConfig), fp, sp, C.c_int, C.c_int, fp, C.c_int]
    L.orc_bench_batch.restype = C.c_longlong
    L.orc_bench_batch.argtypes = [C.POINTER(OrcConfig), fp, sp, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_double)]
    if path is None:
        _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _sp(a):
    return a.ctypes.data_as(C.POINTER(C.c_short))


ERRORS = {
    -1: "Can't process data, buffer is too small",
    -2: "Can't process data, window count is too small",
    -3: "Processed samples <= 0, this should never happen",
    -4: "Window count too high",
    -5: "bad configuration",
}


class OracleMfcc:
    """Object mirror of the reference's MfccCpu (mfcccpu.h:14-67) on top of liboracle.so."""

    def __init__(self, cfg, window=None, bug_compat=True, libpath=None, libm_double=False):
        self.L = lib(libpath)
        self.cfg = cfg
        self.h = self.L.orc_create(C.byref(cfg))
        if not self.h:
            raise RuntimeError("orc_create failed")
        self.L.orc_set_bug_compat(self.h, int(bug_compat))
        if libm_double:   # g++ binding of the reference's unqualified libm calls (see mfcc_oracle.h)
            self.L.orc_set_libm_binding(self.h, 1)
        if window is None:
            window = reference_window(cfg.window_size)
        self.set_window(window)

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(ERRORS.get(rc, "error %d" % rc))
        return rc

    def set_window(self, w):
        w = np.ascontiguousarray(w, dtype=np.float32)
        assert w.size == self.cfg.window_size
        self.L.orc_set_window(self.h, _fp(w))

    def set_input(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        return self._chk(self.L.orc_set_input(self.h, _sp(pcm), pcm.size))

    def flush(self):
        return self._chk(self.L.orc_flush(self.h))

    def set_alpha(self, a):
        self.L.orc_set_alpha(self.h, float(a))

    def apply(self):
        self._chk(self.L.orc_apply(self.h))

    @property
    def width(self):
        return self.L.orc_get_output_data_width(self.h)

    @property
    def input_buffer_size(self):
        return self.L.orc_get_input_buffer_size(self.h)

    @property
    def window_limit(self):
        return self.L.orc_window_limit(self.h)

    @property
    def fft_size(self):
        return self.L.orc_fft_size(self.h)

    def estimated_window_count(self, samples):
        return self.L.orc_estimated_window_count(self.h, int(samples))

    def get_output_data(self, n):
        out = np.empty((max(n, 0), self.width), dtype=np.float32)
        if n > 0:
            self._chk(self.L.orc_get_output_data(self.h, _fp(out), n))
        return out

    def tap(self, name, rows):
        """Copy of an internal stage buffer: frames|fft|mel|mfcc (first `rows` rows)."""
        W2, nb = self.fft_size, self.cfg.num_banks
        dl = self.cfg.ceps_len + (1 if self.cfg.want_c0 else 0)
        per_row = {"frames": W2, "fft": 2 * W2, "mel": nb, "mfcc": dl}[name]
        p = getattr(self.L, "orc_tap_" + name)(self.h)
        return np.ctypeslib.as_array(p, shape=(rows * per_row,)).reshape(rows, per_row).copy()

    def load_fft(self, spec):
        """Put caller-made spectra [rows][W2/2+1] complex64 into the spectrum buffer (rows of W2 complex values)."""
        spec = np.asarray(spec, dtype=np.complex64)
        W2, rows = self.fft_size, spec.shape[0]
        assert spec.shape[1] == W2 // 2 + 1
        buf = np.ctypeslib.as_array(self.L.orc_tap_fft(self.h), shape=(rows * W2 * 2,)).reshape(rows, W2, 2)
        buf[:, :spec.shape[1], 0] = spec.real
        buf[:, :spec.shape[1], 1] = spec.imag

    def filter(self, rows):
        self.L.orc_stage_filter(self.h, int(rows))

    def dct(self, rows):
        self.L.orc_stage_dct(self.h, int(rows))

    def norm_stats(self):
        """(mean, multiplier) of the normaliser instances after apply(): [groups][2][cols]; groups = 1 when the
        statics are normalised before the deltas, else one per column group of the output row."""
        dl = self.cfg.ceps_len + (1 if self.cfg.want_c0 else 0)
        cols = dl if self.cfg.ceps_len > 0 else self.cfg.num_banks
        groups = (1 + self.cfg.dyn) if self.cfg.norm_after_dyn else 1
        out = np.empty((groups, 2, cols), np.float32)
        for g in range(groups):
            for k in range(2):
                out[g, k] = np.ctypeslib.as_array(self.L.orc_tap_norm_stats(self.h, g, k), shape=(cols,))
        if self.cfg.norm == NORM_CMN:   # mean subtraction only: the multiplier slot is unused (normalizercpu.cpp:72-76)
            out[:, 1] = 1.0
        return out

    def tables(self):
        W2, nb = self.fft_size, self.cfg.num_banks
        dl = self.cfg.ceps_len + (1 if self.cfg.want_c0 else 0)
        t = {
            "filters": np.ctypeslib.as_array(self.L.orc_tap_filters(self.h), shape=(2 * W2,)).reshape(2, W2).copy(),
            "filter_beg": np.ctypeslib.as_array(self.L.orc_tap_filter_beg(self.h), shape=(nb + 2,)).copy(),
        }
        if self.cfg.ceps_len > 0:
            t["dct_matrix"] = np.ctypeslib.as_array(self.L.orc_tap_dct_matrix(self.h), shape=(nb * dl,)).reshape(nb, dl).copy()
        return t


def run_utterance(cfg, pcm, window=None, alpha=1.0, bug_compat=True, block_samples=0, libpath=None):
    """Reference call sequence (ASR_OCL.cpp:149-301) over one utterance -> [frames][width]."""
    L = lib(libpath)
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    if window is None:
        window = reference_window(cfg.window_size)
    window = np.ascontiguousarray(window, dtype=np.float32)
    width = L.orc_output_width(cfg.num_banks, cfg.ceps_len, cfg.want_c0, cfg.dyn)
    T = max(L.orc_ewc(pcm.size, cfg.window_size, cfg.shift), 0)
    out = np.zeros((T + 8, width), dtype=np.float32)
    n = L.orc_run_utterance(C.byref(cfg), _fp(window), float(alpha), int(bug_compat), _sp(pcm), pcm.size,
                            int(block_samples), _fp(out))
    if n < 0:
        raise RuntimeError(ERRORS.get(n, "error %d" % n))
    return out[:n].copy()


def run_batch(cfg, pcm2d, window=None, n_threads=1, libpath=None):
    """n_utt equal-length utterances (rows of pcm2d), correct batch semantics, OpenMP threads."""
    L = lib(libpath)
    pcm2d = np.ascontiguousarray(pcm2d, dtype=np.int16)
    n_utt, utt_samples = pcm2d.shape
    if window is None:
        window = reference_window(cfg.window_size)
    window = np.ascontiguousarray(window, dtype=np.float32)
    width = L.orc_output_width(cfg.num_banks, cfg.ceps_len, cfg.want_c0, cfg.dyn)
    fpu = L.orc_ewc(utt_samples, cfg.window_size, cfg.shift)
    out = np.zeros((n_utt * fpu, width), dtype=np.float32)
    n = L.orc_run_batch(C.byref(cfg), _fp(window), _sp(pcm2d), n_utt, utt_samples, _fp(out), int(n_threads))
    if n < 0:
        raise RuntimeError("orc_run_batch failed")
    return out.reshape(n_utt, fpu, width)


def bench_batch(cfg, pcm2d, window=None, n_threads=1, reps=1, libpath=None):
    """Timed CPU pass (setup excluded): returns (frames, seconds)."""
    L = lib(libpath)
    pcm2d = np.ascontiguousarray(pcm2d, dtype=np.int16)
    n_utt, utt_samples = pcm2d.shape
    if window is None:
        window = reference_window(cfg.window_size)
    window = np.ascontiguousarray(window, dtype=np.float32)
    sec = C.c_double(0)
    n = L.orc_bench_batch(C.byref(cfg), _fp(window), _sp(pcm2d), n_utt, utt_samples, int(n_threads), int(reps),
                          C.byref(sec))
    if n < 0:
        raise RuntimeError("orc_bench_batch failed")
    return int(n), float(sec.value)


# ---------------------------------------------------------------------------------------------
# the real reference stage objects (oracle/_ref/libref_stages.so)
# ---------------------------------------------------------------------------------------------
_ref = {}


def ref_available(f32=False):
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_stages_f32.so" if f32 else "libref_stages.so"))


def ref(f32=False):
    """f32 = the build whose unqualified libm names bind to the float overloads (oracle/Makefile ref_f32: the
    reference's own toolchain's overload selection); default = the plain g++ build (C double functions, int abs)."""
    if f32 in _ref:
        return _ref[f32]
    p = os.path.join(_HERE, "_ref", "libref_stages_f32.so" if f32 else "libref_stages.so")
    if not os.path.exists(p):
        raise RuntimeError("oracle/_ref not built (needs /root/reference): make -C oracle ref")
    R = C.CDLL(p)
    fp, ip, sp, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_short), C.c_void_p
    R.ref_base_new.restype = vp
    R.ref_base_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                               C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    R.ref_base_free.argtypes = [vp]
    for n in ("ref_base_input_buffer_size", "ref_base_output_width"):
        getattr(R, n).argtypes = [vp]
    R.ref_base_ewc.argtypes = [vp, C.c_int]
    R.ref_seg_new.restype, R.ref_seg_new.argtypes = vp, [C.c_int] * 4
    R.ref_seg_free.argtypes = [vp]
    R.ref_seg_set_window.argtypes = [vp, fp]
    R.ref_seg_set_input.argtypes = [vp, sp, fp, C.c_int, ip, ip]
    R.ref_seg_flush.argtypes = [vp, fp, ip, ip]
    for n in ("ref_seg_remaining", "ref_seg_samples", "ref_seg_is_flushed", "ref_seg_was_flushed"):
        getattr(R, n).argtypes = [vp]
    R.ref_seg_ewc.argtypes = [vp, C.c_int]
    R.ref_delta_new.restype, R.ref_delta_new.argtypes = vp, [C.c_int] * 3
    R.ref_delta_free.argtypes = [vp]
    R.ref_delta_apply.argtypes = [vp, fp, C.c_int]
    R.ref_delta_output.restype, R.ref_delta_output.argtypes = fp, [vp]
    R.ref_norm_new.restype, R.ref_norm_new.argtypes = vp, [C.c_int, C.c_int]
    R.ref_norm_free.argtypes = [vp]
    R.ref_norm_normalize.argtypes = [vp, fp, C.c_int, C.c_int]
    _ref[f32] = R
    return R


# ---------------------------------------------------------------------------------------------
# the real MfccCpu member functions (oracle/_ref/libref_mfcccpu.so, see oracle/ref_mfcccpu_shim.cpp)
# ---------------------------------------------------------------------------------------------
_refm = {}


def refm_available(f32=False):
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_mfcccpu_f32.so" if f32 else "libref_mfcccpu.so"))


def refm(f32=False):
    """f32: see ref()."""
    if f32 in _refm:
        return _refm[f32]
    p = os.path.join(_HERE, "_ref", "libref_mfcccpu_f32.so" if f32 else "libref_mfcccpu.so")
    if not os.path.exists(p):
        raise RuntimeError("oracle/_ref/%s not built (needs /root/reference): make -C oracle ref" % os.path.basename(p))
    R = C.CDLL(p)
    fp, ip, sp, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_short), C.c_void_p
    R.refm_new.restype = vp
    R.refm_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                           C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    R.refm_free.argtypes = [vp]
    for n in ("input_buffer_size", "window_limit", "cap_rows", "fft_size", "output_width", "was_flushed", "last_block"):
        getattr(R, "refm_" + n).argtypes = [vp]
    R.refm_ewc.argtypes = [vp, C.c_int]
    R.refm_set_window.argtypes = [vp, fp]
    R.refm_set_alpha.argtypes = [vp, C.c_float]
    R.refm_set_input_nofft.argtypes = [vp, sp, C.c_int, ip]
    R.refm_flush_nofft.argtypes = [vp, ip]
    for n in ("data", "fft", "mel", "mfcc", "dct_matrix", "filters", "delta_in", "delta_out", "acc_out"):
        fn = getattr(R, "refm_" + n)
        fn.restype, fn.argtypes = fp, [vp]
    R.refm_filter_beg.restype, R.refm_filter_beg.argtypes = ip, [vp]
    R.refm_norm_stats.restype, R.refm_norm_stats.argtypes = fp, [vp, C.c_int, C.c_int]
    R.refm_refresh_filters.argtypes = [vp]
    R.refm_filter.argtypes = [vp, C.c_int]
    R.refm_dct.argtypes = [vp, C.c_int]
    R.refm_do_delta.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    R.refm_normalize.argtypes = [vp, C.c_int, C.c_int]
    R.refm_apply.argtypes = [vp]
    R.refm_get_output_data.argtypes = [vp, fp, C.c_int]
    _refm[f32] = R
    return R


class RefMfccCpu:
    """The reference's MfccCpu, driven through the call sequence of OracleMfcc.  refresh_filters / filter / dct /
    do_delta / normalize / apply / get_output_data and the segmenter, delta and normaliser members are the
    reference's compiled code; the transform at the FFTW call site (mfcccpu.cpp:187-190) is the double-precision DFT
    rounded to float that the oracle's fft_mode 0 also uses (orc_rfft_rows), written into the object's m_fft."""

    def __init__(self, cfg, window=None, f32=False):
        self.R = refm(f32)
        self.cfg = cfg
        self.h = self.R.refm_new(cfg.input_buffer_size, cfg.window_size, cfg.shift, cfg.num_banks, cfg.sample_rate,
                                 cfg.low_freq, cfg.high_freq, cfg.ceps_len, cfg.want_c0, cfg.lift_coef, cfg.norm,
                                 cfg.dyn, cfg.delta_l1, cfg.delta_l2, cfg.norm_after_dyn)
        self.W2 = self.R.refm_fft_size(self.h)
        self.cap = self.R.refm_cap_rows(self.h)
        if window is None:
            window = reference_window(cfg.window_size)
        self.set_window(window)

    def close(self):
        if self.h:
            self.R.refm_free(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(ERRORS.get(rc, "error %d" % rc))
        return rc

    def set_window(self, w):
        w = np.ascontiguousarray(w, dtype=np.float32)
        assert w.size == self.cfg.window_size
        self.R.refm_set_window(self.h, _fp(w))

    def _fft(self, rows):
        assert rows <= self.cap
        lib().orc_rfft_rows(self.R.refm_data(self.h), self.R.refm_fft(self.h), self.W2, rows, 0)

    def set_input(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        wcnd = C.c_int(0)
        n = self._chk(self.R.refm_set_input_nofft(self.h, _sp(pcm), pcm.size, C.byref(wcnd)))
        if n > 0:
            self._fft(wcnd.value)
        return n

    def flush(self):
        wcnd = C.c_int(0)
        n = self._chk(self.R.refm_flush_nofft(self.h, C.byref(wcnd)))
        if n > 0:
            self._fft(wcnd.value)
        return n

    def set_alpha(self, a):
        self.R.refm_set_alpha(self.h, float(a))

    def apply(self):
        self._chk(self.R.refm_apply(self.h))

    @property
    def width(self):
        return self.R.refm_output_width(self.h)

    @property
    def input_buffer_size(self):
        return self.R.refm_input_buffer_size(self.h)

    @property
    def window_limit(self):
        return self.R.refm_window_limit(self.h)

    @property
    def fft_size(self):
        return self.W2

    def estimated_window_count(self, samples):
        return self.R.refm_ewc(self.h, int(samples))

    def get_output_data(self, n):
        out = np.empty((max(n, 0), self.width), dtype=np.float32)
        if n > 0:
            self._chk(self.R.refm_get_output_data(self.h, _fp(out), n))
        return out

    def tap(self, name, rows):
        W2, nb = self.W2, self.cfg.num_banks
        dl = self.cfg.ceps_len + (1 if self.cfg.want_c0 else 0)
        per_row = {"frames": W2, "fft": 2 * W2, "mel": nb, "mfcc": dl}[name]
        fn = {"frames": self.R.refm_data, "fft": self.R.refm_fft, "mel": self.R.refm_mel, "mfcc": self.R.refm_mfcc}[name]
        return np.ctypeslib.as_array(fn(self.h), shape=(rows * per_row,)).reshape(rows, per_row).copy()

    def load_fft(self, spec):
        """Put caller-made spectra [rows][W2/2+1] complex64 into m_fft (rows of W2 complex, as FFTW's odist)."""
        spec = np.asarray(spec, dtype=np.complex64)
        rows = spec.shape[0]
        assert rows <= self.cap and spec.shape[1] == self.W2 // 2 + 1
        buf = np.ctypeslib.as_array(self.R.refm_fft(self.h), shape=(self.cap * self.W2 * 2,)).reshape(self.cap, self.W2, 2)
        buf[:rows, :spec.shape[1], 0] = spec.real
        buf[:rows, :spec.shape[1], 1] = spec.imag

    def filter(self, rows):
        self.R.refm_filter(self.h, int(rows))

    def dct(self, rows):
        self.R.refm_dct(self.h, int(rows))

    def norm_stats(self):
        dl = self.cfg.ceps_len + (1 if self.cfg.want_c0 else 0)
        cols = dl if self.cfg.ceps_len > 0 else self.cfg.num_banks
        groups = (1 + self.cfg.dyn) if self.cfg.norm_after_dyn else 1
        out = np.ones((groups, 2, cols), np.float32)
        which = {NORM_CMN: None, NORM_CVN: 1, NORM_MINMAX: 2}[self.cfg.norm]
        for g in range(groups):
            out[g, 0] = np.ctypeslib.as_array(self.R.refm_norm_stats(self.h, g, 0), shape=(cols,))
            if which is not None:
                out[g, 1] = np.ctypeslib.as_array(self.R.refm_norm_stats(self.h, g, which), shape=(cols,))
        return out

    def tables(self):
        W2, nb = self.W2, self.cfg.num_banks
        dl = self.cfg.ceps_len + (1 if self.cfg.want_c0 else 0)
        t = {
            "filters": np.ctypeslib.as_array(self.R.refm_filters(self.h), shape=(2 * W2,)).reshape(2, W2).copy(),
            "filter_beg": np.ctypeslib.as_array(self.R.refm_filter_beg(self.h), shape=(nb + 2,)).copy(),
        }
        if self.cfg.ceps_len > 0:
            t["dct_matrix"] = np.ctypeslib.as_array(self.R.refm_dct_matrix(self.h), shape=(nb * dl,)).reshape(nb, dl).copy()
        return t


def run_reference_utterance(cfg, pcm, window=None, alpha=1.0, block_samples=0, f32=False):
    """ASR_OCL.cpp:149-301 over one utterance on the REAL MfccCpu -> [frames][width] (B1 and all)."""
    m = RefMfccCpu(cfg, window, f32=f32)
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    blk = m.input_buffer_size if block_samples <= 0 else min(block_samples, m.input_buffer_size)
    rows = []
    for pos in range(0, pcm.size, blk):
        n = m.set_input(pcm[pos:pos + blk])
        if n > 0:
            m.set_alpha(alpha)
            m.apply()
            rows.append(m.get_output_data(n))
    n = m.flush()
    if n > 0:
        m.set_alpha(alpha)
        m.apply()
        rows.append(m.get_output_data(n))
    m.close()
    return np.concatenate(rows) if rows else np.zeros((0, m.width), np.float32)
