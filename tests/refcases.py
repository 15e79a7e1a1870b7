"""Cases that pin the checker and the HIP path to the REAL reference (`/root/reference/mfcccpu.cpp`, compiled in place into
oracle/_ref/libref_mfcccpu.so -- see oracle/ref_mfcccpu_shim.cpp).

The same list is used by
  * tests/golden/make_golden.py      -> tests/golden/ref_mfcccpu_vectors.npz (outputs of the real MfccCpu functions),
  * tests/test_ref_mfcccpu.py  (CPU) -> oracle/mfcc_oracle.c against those vectors, and live against the .so when present,
  * tests/test_parity_gpu.py   (GPU) -> the HIP path against the same vectors (and live against the .so, which travels).

A case = extractor configuration + PCM + the block size of the reference's per-file loop (ASR_OCL.cpp:227-301) + VTLN alpha.
Inputs come from the reference's own sound files (tests/golden/*.wav) or from conftest.synth_utterance (BASELINE 8d).
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

NORM_NONE, NORM_CMN, NORM_CVN, NORM_MINMAX = 0, 1, 2, 3
DYN_NONE, DYN_DELTA, DYN_ACC = 0, 1, 2


def _synth(n, seed, sr=16000.0):
    rng = np.random.default_rng(0x5EED0000 + seed)
    f = 100.0 + 37.0 * (seed % 64)
    t = np.arange(n)
    x = 3000.0 * rng.standard_normal(n) + 6000.0 * np.sin(2 * np.pi * f * t / sr)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


def _case(name, pcm, ibs, alpha=1.0, window=None, **cfg):
    base = dict(window_size=400, shift=160, num_banks=26, sample_rate=16000.0, low_freq=64.0, high_freq=None,
                ceps_len=13, want_c0=False, lift_coef=22.0, norm=NORM_NONE, dyn=DYN_ACC, delta_l1=3, delta_l2=3,
                norm_after_dyn=True)
    base.update(cfg)
    return dict(name=name, pcm=pcm, ibs=int(ibs), alpha=float(alpha), window=window, cfg=base)


def window_c3():
    """BASELINE configs[2] (1024-point transform of a 25 ms window): window_size 1024 whose taps 400.. are zero."""
    import oracle_py as O
    w = np.zeros(1024, np.float32)
    w[:400] = O.reference_window(400)
    return w


def cases():
    """name -> case.  PCM given as ('wav', file) | ('synth', n, seed, sr) so that the list itself stays cheap."""
    L = []
    # C1 = BASELINE configs[0]: a0001.wav, 26 mel, 13 MFCC + d + dd; multi-block and the single-block run with bug B1
    L.append(_case("c1_multi", ("wav", "a0001.wav"), 32000))
    L.append(_case("c1_single", ("wav", "a0001.wav"), 10000000))
    # the reference main()'s own defaults (ASR_OCL.cpp:560): 15 banks, 12 + c0, CVN, no dyn
    L.append(_case("a1_main_defaults", ("wav", "a1.wav"), 32000, num_banks=15, ceps_len=12, want_c0=True, norm=NORM_CVN,
                   dyn=DYN_NONE))
    L.append(_case("a1_main_defaults_one_block", ("wav", "a1.wav"), 10000000, num_banks=15, ceps_len=12, want_c0=True,
                   norm=NORM_CVN, dyn=DYN_NONE))
    # C2 / C4 shape (512 points, 40 mel, 13 MFCC + d + dd), first / steady / flush blocks, three warps
    for a in (0.88, 1.0, 1.12):
        L.append(_case("c2_alpha%03d" % round(a * 100), ("synth", 24000, 11, 16000.0), 8000, alpha=a, num_banks=40))
    # C3 shape (1024 points, 80 mel, 13 MFCC, no dyn).  One block per file: with W - S > 2 S and dyn off a steady-state
    # block of the reference returns more rows than its m_window_limit and get_output_data throws ...
    for a in (0.88, 1.0, 1.12):
        L.append(_case("c3_alpha%03d" % round(a * 100), ("synth", 20000, 12, 16000.0), 30000, alpha=a, window="c3",
                       window_size=1024, num_banks=80, dyn=DYN_NONE))
    # ... and the same shape streamed, with deltas (their 3 D rows of capacity cover the carry-over)
    L.append(_case("c3_streamed_dyn", ("synth", 30000, 14, 16000.0), 9000, window="c3", window_size=1024, num_banks=80))
    # C5 shape (44.1 kHz, 1102-sample window, 2048 points, 128 mel, 40 MFCC + d + dd); mono: the downmix is the caller's
    for a in (0.88, 1.0, 1.12):
        L.append(_case("c5_alpha%03d" % round(a * 100), ("synth", 40000, 13, 44100.0), 15000, alpha=a, window_size=1102,
                       shift=441, num_banks=128, sample_rate=44100.0, ceps_len=40))
    # dyn x norm x norm_after_dyn on a small configuration with c0 (first, steady, steady, flush)
    k = 0
    for dyn in (DYN_NONE, DYN_DELTA, DYN_ACC):
        for norm in (NORM_CMN, NORM_CVN, NORM_MINMAX):
            for nad in (True, False):
                k += 1
                L.append(_case("dyn%d_norm%d_nad%d" % (dyn, norm, int(nad)), ("synth", 16000, 20 + k, 16000.0), 5000,
                               num_banks=24, ceps_len=12, want_c0=True, norm=norm, dyn=dyn, delta_l1=2, delta_l2=3,
                               norm_after_dyn=nad))
    # mel energies as the output (ceps_len = 0), odd window / shift, other deltas
    L.append(_case("mel_only", ("synth", 12000, 40, 16000.0), 6000, num_banks=31, ceps_len=0, dyn=DYN_DELTA, delta_l1=1))
    L.append(_case("odd_geometry", ("synth", 15000, 41, 16000.0), 7001, window_size=317, shift=97, num_banks=19, ceps_len=9,
                   want_c0=True, low_freq=120.0, high_freq=6500.0, lift_coef=17.0, delta_l1=1, delta_l2=2))
    # silence: every filter sum hits the 1e-30 floor (mfcccpu.cpp:212)
    L.append(_case("silence", ("zeros", 9000), 4000, num_banks=40))
    return {c["name"]: c for c in L}


def load_pcm(spec):
    import oracle_py as O
    if spec[0] == "wav":
        pcm, _ = O.read_wav_pcm16(os.path.join(GOLDEN, spec[1]))
        return pcm[:, 0].copy()
    if spec[0] == "zeros":
        return np.zeros(spec[1], np.int16)
    return _synth(spec[1], spec[2], spec[3])


def make_cfg(O, case, fft_mode=0):
    c = case["cfg"]
    return O.make_config(case["ibs"], window_size=c["window_size"], shift=c["shift"], num_banks=c["num_banks"],
                         sample_rate=c["sample_rate"], low_freq=c["low_freq"], high_freq=c["high_freq"],
                         ceps_len=c["ceps_len"], want_c0=c["want_c0"], lift_coef=c["lift_coef"], norm=c["norm"],
                         dyn=c["dyn"], delta_l1=c["delta_l1"], delta_l2=c["delta_l2"], norm_after_dyn=c["norm_after_dyn"],
                         fft_mode=fft_mode)


def case_window(O, case):
    if case["window"] == "c3":
        return window_c3()
    return O.reference_window(case["cfg"]["window_size"])


def drive(engine, pcm, alpha):
    """The reference's per-file loop (ASR_OCL.cpp:227-301) on any object with the MfccCpu method set.
    Returns (rows [frames][width], frames per call incl. the flush block)."""
    blk = engine.input_buffer_size
    rows, counts = [], []
    for pos in range(0, pcm.size, blk):
        n = engine.set_input(pcm[pos:pos + blk])
        counts.append(n)
        if n > 0:
            engine.set_alpha(alpha)
            engine.apply()
            rows.append(engine.get_output_data(n))
    n = engine.flush()
    counts.append(n)
    if n > 0:
        engine.set_alpha(alpha)
        engine.apply()
        rows.append(engine.get_output_data(n))
    out = np.concatenate(rows) if rows else np.zeros((0, engine.width), np.float32)
    return out, np.array(counts, np.int64)
