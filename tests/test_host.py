"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/mfx.h
declares, the host-built tables equal the oracle's bit for bit, the config struct mirrors the
header, the product refuses to run without a GPU (no CPU fallback), and the N>1 sharding path
works under torch.distributed (gloo, world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "mfx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mfx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.load_library()
    declared = _header_functions()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, "declared in include/mfx.h but not exported: %s" % missing
    assert set(declared) == set(pkg.mfcc.EXPORTED_SYMBOLS)
    assert L.mfx_abi_version() == 2


def test_config_struct_matches_header(pkg, tmp_path):
    """sizeof/offsets of mfx_config as the C compiler sees them == the ctypes mirror."""
    fields = [f[0] for f in pkg.MfxConfig._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "mfx.h"', 'int main(void){',
            'printf("%zu\\n", sizeof(mfx_config));']
    prog += ['printf("%%zu\\n", offsetof(mfx_config, %s));' % f for f in fields]
    prog += ['return 0;}']
    c = tmp_path / "t.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert vals[0] == C.sizeof(pkg.MfxConfig)
    assert vals[1:] == [getattr(pkg.MfxConfig, f).offset for f in fields]


def test_header_is_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "mfx.h"\nint main(void){return MFX_OK;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-c", str(c), "-o", str(tmp_path / "t.o")])


@pytest.mark.parametrize("nb,W2,sr,low,high,alpha", [
    (26, 512, 16000.0, 64.0, 8000.0, 1.0),
    (40, 512, 16000.0, 64.0, 8000.0, 1.0),
    (15, 512, 16000.0, 64.0, 8000.0, 1.0),
    (80, 1024, 16000.0, 64.0, 8000.0, 1.0),
    (128, 2048, 44100.0, 64.0, 22050.0, 1.0),
    (40, 512, 16000.0, 64.0, 8000.0, 0.88),
    (40, 512, 16000.0, 100.0, 7000.0, 1.12),
    (23, 256, 8000.0, 64.0, 4000.0, 1.0),
])
def test_mel_table_equals_oracle(pkg, orc, nb, W2, sr, low, high, alpha):
    w, beg = pkg.host_mel_table(nb, W2, sr, low, high, alpha)
    W = W2  # any window length with ceil2(W) == W2
    o = orc.OracleMfcc(orc.make_config(4 * W2, window_size=W, shift=W // 2, num_banks=nb, sample_rate=sr, low_freq=low,
                                       high_freq=high, ceps_len=0, dyn=orc.DYN_NONE), np.ones(W, np.float32))
    o.set_alpha(alpha)
    o.set_input(np.zeros(3 * W2, np.int16))
    o.apply()  # the oracle (like the reference) rebuilds the table inside apply()
    t = o.tables()
    assert np.array_equal(beg, t["filter_beg"])
    assert np.array_equal(w, t["filters"])
    # structural properties the kernels rely on
    assert np.all(np.diff(beg) >= 0) and beg[0] >= 0 and beg[-1] <= W2 // 2
    assert w.min() >= 0.0 and w.max() <= 1.0


@pytest.mark.parametrize("nb,nc,c0,lift", [(26, 13, False, 22.0), (40, 13, False, 22.0), (15, 12, True, 22.0),
                                           (128, 40, False, 22.0), (80, 13, True, 10.0)])
def test_dct_matrix_equals_oracle(pkg, orc, nb, nc, c0, lift):
    m = pkg.host_dct_matrix(nb, nc, c0, lift)
    o = orc.OracleMfcc(orc.make_config(4000, num_banks=nb, ceps_len=nc, want_c0=c0, lift_coef=lift, dyn=orc.DYN_NONE))
    assert np.array_equal(m, o.tables()["dct_matrix"])
    if c0:
        assert np.all(m[:, nc] == np.float32(np.sqrt(2.0 / nb)))  # c0 is the LAST column


@pytest.mark.parametrize("lanes,nb,W2,sr,alpha,max_read", [
    (16, 40, 512, 16000.0, 1.0, 479), (16, 26, 512, 16000.0, 0.88, 479), (16, 15, 512, 16000.0, 1.12, 479),
    (16, 80, 1024, 16000.0, 1.0, 527), (16, 64, 1024, 16000.0, 0.9, 527), (64, 80, 1024, 16000.0, 1.0, 1023), (64, 128, 2048, 44100.0, 1.0, 1535), (64, 23, 1024, 22050.0, 0.94, 1023),
    (64, 200, 2048, 44100.0, 1.0, 1535),
    # the 64-lane plan now serves every transform size (k_melcep, k_front_wave): buffers of W2 floats
    (64, 40, 512, 16000.0, 1.0, 511), (64, 26, 512, 16000.0, 0.88, 511), (64, 3, 64, 8000.0, 1.0, 63),
    (64, 128, 4096, 96000.0, 1.12, 4095), (64, 15, 256, 8000.0, 1.0, 255),
    # 32 lanes per frame, two frames per wave: k_front2048 (planes of 1040 floats)
    (32, 128, 2048, 44100.0, 1.0, 1039), (32, 96, 2048, 48000.0, 0.93, 1039), (32, 200, 2048, 32000.0, 1.07, 1039),
    (32, 40, 2048, 44100.0, 1.0, 1039)])
def test_mel_lane_plan_walks_every_filter_once(pkg, lanes, nb, W2, sr, alpha, max_read):
    """Lane plans of the kernels' mel walk (16 lanes per frame: k_front512 / k_front1024; 32: k_front2048; 64: k_front_reg,
    k_front_wave, k_melcep): every filter sits in
    exactly one (round, lane) slot; the lane's zero-padded weight row holds the filter's table weights at its bins, in
    ascending order (one chain of multiply-adds = the reference's summation order, mfcccpu.cpp:192-220), and exact zeros
    everywhere else; starts are even, rounds are whole 8-bin trips, reads stay inside the magnitude buffer; row strides
    are odd in 16-byte words (the 16 lanes of a 16-byte LDS access then fall on disjoint bank quads)."""
    wt, beg = pkg.host_mel_table(nb, W2, sr, 64.0, sr / 2, alpha)
    pl = pkg.host_mel_lane_plan(lanes, wt, beg, max_read)
    R, L, rs = pl["rounds"], pl["L"], pl["row_stride"]
    assert R == (nb + lanes - 1) // lanes and rs % 4 == 0 and (rs // 4) % 2 == 1 and rs >= int(L.sum())
    seen = set()
    base = 0
    for r in range(R):
        assert L[r] % 8 == 0 and L[r] >= 8
        for j in range(lanes):
            m, st = int(pl["fid"][r, j]), int(pl["start"][r, j])
            row = pl["w"][j, base:base + L[r]]
            if m < 0:
                assert not row.any()
                continue
            assert m not in seen and 0 <= m < nb
            seen.add(m)
            assert st % (4 if (lanes == 16 and W2 == 1024) else 2) == 0 and st >= 0 and st + L[r] - 1 <= max_read
            want = np.zeros(L[r], np.float32)
            b0, b1 = int(beg[m]), int(beg[m + 2])
            assert st <= b0 and b1 - st <= L[r]
            want[b0 - st:b1 - st] = wt[m & 1, b0:b1]
            assert np.array_equal(row, want)
        base += int(L[r])
    assert seen == set(range(nb))
    assert not pl["w"][:, base:].any()


@pytest.mark.parametrize("lanes,nb,W2,sr,max_read,clashes", [
    (16, 40, 512, 16000.0, 479, 0), (16, 26, 512, 16000.0, 479, 0), (16, 15, 512, 16000.0, 479, 0),
    (16, 80, 1024, 16000.0, 527, 10), (32, 128, 2048, 44100.0, 1039, 6), (64, 40, 512, 16000.0, 511, None)])
def test_mel_lane_plan_starts_are_a_maximum_matching(pkg, lanes, nb, W2, sr, max_read, clashes):
    """The lanes of one LDS access group (16 lanes of a frame / 32 lanes of a wave half) read magnitudes at start + s together:
    they are conflict free when (start / align) mod group differs from lane to lane.  A filter may start early on zero
    weights as long as its round does not grow; the builder picks the starts by a maximum bipartite matching, so no other
    choice of early starts leaves fewer lanes sharing a residue (checked here against an independent matching), and the
    headline shapes are clash free (C2, C1, R) or down to the short last rounds whose filters have nowhere to move."""
    wt, beg = pkg.host_mel_table(nb, W2, sr, 64.0, sr / 2, 1.0)
    pl = pkg.host_mel_lane_plan(lanes, wt, beg, max_read)
    align = 4 if (lanes == 16 and W2 == 1024) else 2
    group = 16 if lanes == 16 else 32
    total = 0
    for r in range(pl["rounds"]):
        for g0 in range(0, lanes, group):
            js = [j for j in range(g0, min(g0 + group, lanes)) if pl["fid"][r, j] >= 0]
            if not js:
                continue
            L = int(pl["L"][r])
            have = len(js) - len({(int(pl["start"][r, j]) // align) % group for j in js})
            cands = []
            for j in js:
                m = int(pl["fid"][r, j])
                b0, b1 = int(beg[m]) & ~(align - 1), int(beg[m + 2])
                cands.append([(c // align) % group for c in range(b0, -1, -align) if b1 - c <= L])
            owner = {}

            def augment(i, seen):
                for q in cands[i]:
                    if q in seen:
                        continue
                    seen.add(q)
                    if q not in owner or augment(owner[q], seen):
                        owner[q] = i
                        return True
                return False
            best = len(js) - sum(augment(i, set()) for i in range(len(js)))
            assert have == best, (r, g0, have, best)
            total += have
    if clashes is not None:
        assert total == clashes


@pytest.mark.parametrize("nb,nc,c0", [(40, 13, False), (26, 13, False), (15, 12, True), (80, 13, False), (128, 40, False)])
def test_dct_mfma_operands_are_the_matrix(pkg, nb, nc, c0):
    """Operand table of the DCT on the matrix pipe: lane (k = lane >> 4, n = lane & 15) of K step j of tile t holds
    dct[4 j + k][16 t + n] (the B fragment of v_mfma_f32_16x16x4_f32), zeros beyond the matrix; multiplying it out in
    the instruction's order reproduces the DCT of a random vector."""
    m = pkg.host_dct_matrix(nb, nc, c0, 22.0)
    dl = m.shape[1]
    ob = pkg.host_dct_mfma_operands(m)
    tiles, ks = ob.shape[0], ob.shape[1]
    assert tiles == (dl + 15) // 16 and ks == (nb + 3) // 4
    full = np.zeros((4 * ks, 16 * tiles), np.float32)
    full[:nb, :dl] = m
    for t in range(tiles):
        for j in range(ks):
            for lane in range(64):
                assert ob[t, j, lane] == full[4 * j + (lane >> 4), 16 * t + (lane & 15)]
    x = np.random.default_rng(3).standard_normal(nb).astype(np.float32)
    xa = np.zeros(4 * ks, np.float32)
    xa[:nb] = x
    out = np.zeros(16 * tiles, np.float64)
    for t in range(tiles):
        for j in range(ks):
            for k in range(4):
                out[16 * t:16 * t + 16] += xa[4 * j + k] * ob[t, j, 16 * k:16 * k + 16].astype(np.float64)
    np.testing.assert_allclose(out[:dl], x.astype(np.float64) @ m.astype(np.float64), rtol=0, atol=1e-5)


def test_frame_count_integer_vs_float32(pkg, orc):
    L = orc.lib()
    rng = np.random.default_rng(5)
    for W, S in ((400, 160), (1102, 441), (1024, 160), (200, 80)):
        for s in list(rng.integers(0, 1 << 24, size=200)) + [0, 1, W - S, W - 1, W, W + S - 1, W + S]:
            assert pkg.host_frame_count(int(s), W, S) == L.orc_ewc(int(s), W, S)
    # beyond 2^24 samples the reference's float32 expression is no longer exact; the product's is
    assert pkg.host_frame_count(57600000, 400, 160) == (57600000 - 240) // 160


def test_no_cpu_fallback(pkg):
    """Without a GPU the product must fail loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.MfxError):
        pkg.MfccHip(8000, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, False, 22.0)


def test_product_does_not_link_the_oracle(pkg):
    """The shipped library must not depend on anything under oracle/."""
    out = subprocess.check_output(["ldd", pkg.library_path()], text=True)
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "asr-featext-opencl_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle_py" not in txt and "mfcc_oracle" not in txt and "liboracle" not in txt, f


def test_sharding_round_robin(graft):
    pkg = graft.load_package()
    import importlib
    sh = importlib.import_module(graft.PKG_NAME + ".sharding")
    lengths = np.array([160000, 1001, 48000, 7, 160000, 32001, 999, 50000, 8000], dtype=np.int64)
    seen = []
    for r in range(4):
        idx, off, ln, total = sh.shard_layout(lengths, r, 4)
        assert np.array_equal(idx, np.arange(r, lengths.size, 4))
        assert np.all(off % 2 == 0) and np.array_equal(ln, lengths[idx])
        assert total >= ln.sum() and np.all(off[1:] >= off[:-1] + ln[:-1])
        seen += list(idx)
    assert sorted(seen) == list(range(lengths.size))
    fr = sh.frames_of(lengths, 400, 160)
    assert list(fr) == [pkg.host_frame_count(int(n), 400, 160) if n >= 400 else 0 for n in lengths]


_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as G
import importlib
pkg = G.load_package()
sh = importlib.import_module(G.PKG_NAME + ".sharding")
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(7)
lengths = rng.integers(400, 200000, size=101)
idx, off, ln, total = sh.shard_layout(lengths, rank, world)
local = int(sh.frames_of(ln, 400, 160).sum())
dist.barrier()
tot = sh.gather_counts(local, dist)
tmax = sh.max_over_ranks(1.0 + rank, dist)
expect = int(sh.frames_of(lengths, 400, 160).sum())
assert tot == expect, (tot, expect)
assert tmax == float(world), tmax
# every utterance owned exactly once
owned = torch.zeros(lengths.size, dtype=torch.int64); owned[torch.from_numpy(idx)] = 1
dist.all_reduce(owned)
assert bool((owned == 1).all())
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_sharding(tmp_path):
    """The N>1 path of bench.py / sharding.py under torch.distributed, world_size 2, gloo, CPU."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), ROOT],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("ok") >= 2


def test_sphere_and_riff_fixtures_hold_the_same_pcm(orc):
    """tests/golden/sample1_sphere.wav (the reference's sample1.wav, NIST_1A header of 1024 bytes,
    sample_byte_format 01) and sample1_riff.wav (soundfiles/sample1_1.wav) are the same utterance; the
    C++ driver's two readers are compared on the GPU box (test_cpp_driver_sphere_input_and_htk_output)."""
    GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    raw = open(os.path.join(GOLDEN, "sample1_sphere.wav"), "rb").read()
    assert raw[:7] == b"NIST_1A" and int(raw[8:16]) == 1024
    head = raw[:1024].decode("ascii", "replace")
    assert "sample_count -i 54682" in head and "sample_byte_format -s2 01" in head
    sph = np.frombuffer(raw[1024:1024 + 2 * 54682], dtype="<i2")
    pcm, sr = orc.read_wav_pcm16(os.path.join(GOLDEN, "sample1_riff.wav"))
    assert sr == 16000 and np.array_equal(pcm[:, 0], sph)


def test_bench_refuses_rank_count_mismatch():
    """bench.py never reports a line whose n_gpus differs from --gpus: under a launcher that started a different
    number of ranks it stops (checked before any GPU use, so this runs without a GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4" in r.stderr and "{" not in r.stdout


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not mounted (GPU box)")
def test_mfcchip_builds_against_reference_headers(tmp_path):
    """Boundary proof: host/mfcchip.cpp compiled against the reference's OWN parambase.h / mfccbase.h /
    normalizer.h (-DAFET_USE_REFERENCE_HEADERS -I/root/reference, the mirror classes of afet_param.h compiled out) and
    linked with the reference's own parambase.cpp / mfccbase.cpp compiled in place -- MfccHip drops into the reference
    tree as `class MfccHip : public MfccBase` with set_alpha left non-virtual (parambase.h:25)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "asr-featext-opencl_amd", "host")
    pkg = os.path.join(root, "asr-featext-opencl_amd")
    inc = ["-I" + REFERENCE, "-I" + os.path.join(REFERENCE, "include")]
    cxx = ["g++", "-std=c++14", "-O1", "-Wall"]
    for tu in ("parambase.cpp", "mfccbase.cpp"):        # compiled where they lie; only objects leave, into tmp_path
        subprocess.check_call(cxx + inc + ["-c", os.path.join(REFERENCE, tu), "-o", str(tmp_path / (tu[:-4] + ".o"))])
    subprocess.check_call(cxx + inc + ["-DAFET_USE_REFERENCE_HEADERS", "-c", os.path.join(host, "mfcchip.cpp"),
                                       "-o", str(tmp_path / "mfcchip_ref.o")])
    probe = tmp_path / "probe.cpp"
    probe.write_text(r'''
#include <cstdio>
#include <type_traits>
#include "afet_param.h"
static_assert(std::is_base_of<MfccBase, MfccHip>::value, "MfccHip derives from the reference's MfccBase");
static_assert(std::is_abstract<ParamBase>::value, "the reference's ParamBase");
int main()
{
    void (ParamBase::*sa)(float) = &ParamBase::set_alpha;   // the reference's own non-virtual member
    (void)sa;
    try {
        MfccHip m(32000, 400, 160, 26, 16000.f, 64.f, 8000.f, 13, false, 22.f, Normalizer::NORM_NONE,
                  ParamBase::DYN_ACC, 3, 3, true, 0, true);
        ParamBase *p = &m;
        p->set_alpha(1.0f);
        std::printf("created %d %d %d\n", p->get_input_buffer_size(), p->estimated_window_count(32000),
                    p->get_output_data_width());
    } catch (const std::exception &e) {
        std::printf("no device: %s\n", e.what());
    }
    return 0;
}
''')
    subprocess.check_call(cxx + inc + ["-DAFET_USE_REFERENCE_HEADERS", "-I" + host, "-c", str(probe), "-o", str(tmp_path / "probe.o")])
    exe = tmp_path / "probe"
    subprocess.check_call(["g++", "-o", str(exe), str(tmp_path / "probe.o"), str(tmp_path / "mfcchip_ref.o"),
                           str(tmp_path / "parambase.o"), str(tmp_path / "mfccbase.o"), "-L" + pkg, "-lmfcchip",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    # without a GPU the constructor reports the device error; with one the reference's own bookkeeping answers
    assert r.stdout.startswith("no device: MfccHip:") or r.stdout.startswith("created 31920 198 39"), r.stdout


def test_cpp_driver_rejects_truncated_wave(tmp_path):
    """A RIFF file cut inside its fmt chunk, and a SPHERE header without a terminator, are refused with a message
    (no read past the buffer): the readers run before any device is touched."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "asr-featext-opencl_amd", "host", "afet_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    bad = tmp_path / "cut.wav"
    bad.write_bytes(b"RIFF" + (36).to_bytes(4, "little") + b"WAVE" + b"fmt " + (16).to_bytes(4, "little") + b"\x01\x00\x01\x00")
    r = subprocess.run([exe, str(bad), str(tmp_path / "o.txt")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "truncated fmt chunk" in r.stderr
    # a well-formed header that announces ZERO channels (the downmix divides by the channel count) / a rate of 0
    for nch, rate in ((0, 16000), (1, 0)):
        z = tmp_path / ("zero_%d_%d.wav" % (nch, rate))
        z.write_bytes(b"RIFF" + (44).to_bytes(4, "little") + b"WAVE" + b"fmt " + (16).to_bytes(4, "little") + (1).to_bytes(2, "little") +
                      nch.to_bytes(2, "little") + rate.to_bytes(4, "little") + (32000).to_bytes(4, "little") + (2).to_bytes(2, "little") +
                      (16).to_bytes(2, "little") + b"data" + (8).to_bytes(4, "little") + bytes(8))
        r = subprocess.run([exe, str(z), str(tmp_path / "o.txt")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 1 and "Error while loading" in r.stderr, (r.returncode, r.stderr)
    sph = tmp_path / "cut.sph"
    sph.write_bytes(b"NIST_1A\n99999999")
    r = subprocess.run([exe, str(sph), str(tmp_path / "o.txt")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "bad SPHERE header" in r.stderr


def test_sanitizer_targets_run_clean():
    """DESIGN.md section 3's sanitizer claim, reproducible: `make asan` in oracle/ (the checker under
    AddressSanitizer + UBSan over 384 call sequences) and in csrc/ (the host table builders over 1500
    configurations) build, run and report nothing.  CPU only -- GPU sanitizers are not available on the pool."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d, tag in ((os.path.join(root, "oracle"), "oracle_asan:"),
                   (os.path.join(root, "asr-featext-opencl_amd", "csrc"), "tables_asan:")):
        r = subprocess.run(["make", "-C", d, "asan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
        assert tag in r.stdout and "clean" in r.stdout and "ERROR" not in r.stdout


def test_front1024_decimation_identity():
    """The factorisation k_front1024 computes (csrc/mfx_front512.hip), restated in numpy in double precision: for a real
    frame that is zero from sample 512 on, the 1024-point DFT is two 256-point complex DFTs of the packed samples
    z[m] = x[2m] + i x[2m+1] -- even bins through the usual real split of FFT256(z), odd bins through the same split
    arithmetic applied to V = FFT256(z W_512^m) with partner bin 255 - k and twiddle -i W_1024^(2k+1); the paired form
    X[bin(k)] = (S + T) / 2, X[bin(partner)] = conj(S - T) / 2 is what the kernel evaluates per lane."""
    rng = np.random.default_rng(5)
    for W in (400, 512, 37):
        x = np.zeros(1024)
        x[:W] = rng.standard_normal(W) * 1000.0
        X = np.fft.fft(x)                      # forward DFT, e^{-2 pi i n k / N}: the reference's convention
        z = x[0:512:2] + 1j * x[1:512:2]
        m = np.arange(256)
        Z = np.fft.fft(z)
        V = np.fft.fft(z * np.exp(-2j * np.pi * m / 512))
        k = np.arange(128)
        # phase E: partner 256 - k (bin 0 pairs with itself and yields the Nyquist bin), twiddle -i W_512^k
        pe = (256 - k) % 256
        S, D = Z[k] + np.conj(Z[pe]), Z[k] - np.conj(Z[pe])
        T = (-1j * np.exp(-2j * np.pi * k / 512)) * D
        assert np.allclose((S + T) / 2, X[2 * k], rtol=0, atol=1e-6 * np.abs(X).max())
        assert np.allclose(np.conj(S - T) / 2, X[2 * (256 - k)], rtol=0, atol=1e-6 * np.abs(X).max())
        assert np.allclose(2 * Z[128].conj() / 2, X[256], rtol=0, atol=1e-6 * np.abs(X).max())   # the self-paired bin
        # phase O: partner 255 - k, twiddle -i W_1024^(2k+1), no self-paired bins
        po = 255 - k
        S, D = V[k] + np.conj(V[po]), V[k] - np.conj(V[po])
        T = (-1j * np.exp(-2j * np.pi * (2 * k + 1) / 1024)) * D
        assert np.allclose((S + T) / 2, X[2 * k + 1], rtol=0, atol=1e-6 * np.abs(X).max())
        assert np.allclose(np.conj(S - T) / 2, X[2 * po + 1], rtol=0, atol=1e-6 * np.abs(X).max())


# ---------------------------------------------------------------------------------------------
# bench.py: job planning (weak / strong scaling) and the per-frame work model -- no GPU needed
# ---------------------------------------------------------------------------------------------

def _import_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_work_model_matches_survey_8d():
    """C2 ~ 15 kflop and 476 B per frame (9.1 KB staged), C3 ~ 31 kflop / 372 B, C5 ~ 75 kflop / 2244 B;
    the reference-defaults workload R carries the normaliser."""
    B = _import_bench()
    c2 = B.work_model(B.WORKLOADS["C2"], 39, 1)
    assert 14000 <= c2["flops"] <= 16000 and c2["fused_bytes"] == 476 and 9000 <= c2["staged_bytes"] <= 9300
    c3 = B.work_model(B.WORKLOADS["C3"], 13, 1)
    assert 29000 <= c3["flops"] <= 33000 and c3["fused_bytes"] == 372
    c5 = B.work_model(B.WORKLOADS["C5"], 120, 2)
    assert 70000 <= c5["flops"] <= 80000 and c5["fused_bytes"] == 2244
    r = B.work_model(B.WORKLOADS["R"], 13, 1)
    assert r["flops_norm"] > 0 and r["flops_delta"] == 0 and r["fused_bytes"] == 320 + 52
    # intensity above the machine balance: the fused path cannot be HBM-bound at any VALU efficiency below 60 %
    assert c2["flops"] / c2["fused_bytes"] > B.FP32_PEAK_TFLOPS * 1e12 / (B.HBM_PEAK_GBS * 1e9)


def test_bench_plan_job_weak_and_strong(pkg):
    B = _import_bench()
    sh = pkg.sharding
    wl = B.WORKLOADS["T"]
    for world in (1, 2, 3, 8):
        seen = []
        for rank in range(world):
            ids, off, ln, total = B.plan_job(wl, rank, world, "strong", sh)
            assert list(ids) == list(range(rank, wl["n_utt_total"], world))          # round-robin, BASELINE configs[3]
            assert list(off) == [i * wl["utt_samples"] for i in range(ids.size)] and total == ids.size * wl["utt_samples"]
            seen += list(ids)
            w_ids, w_off, w_ln, w_total = B.plan_job(wl, rank, world, "weak", sh)
            assert w_ids.size == wl["n_utt"] and w_total == wl["n_utt"] * wl["utt_samples"]
        assert sorted(seen) == list(range(wl["n_utt_total"]))
    assert B.WORKLOADS["C4"]["n_utt_total"] == 100000 and B.WORKLOADS["C4"]["n_utt"] * 8 == 100000


def test_bench_synth_by_index_is_a_function_of_the_utterance_index():
    """Strong scaling computes ONE job at every N: an utterance's PCM depends on its global index only (CPU tensors here)."""
    import torch
    B = _import_bench()
    a = B.synth_pcm_by_index(torch, [0, 1, 2, 3, 4, 5], 4000, 16000.0, "cpu")
    b = B.synth_pcm_by_index(torch, [4, 1], 4000, 16000.0, "cpu")
    assert torch.equal(a[4], b[0]) and torch.equal(a[1], b[1]) and not torch.equal(a[0], a[1])
    x = a.to(torch.float32)
    assert 3000 < float(x.std()) < 6500 and abs(float(x.mean())) < 200          # 3000 N(0,1) + 6000 sin(...)


_STRONG_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import importlib.util
import __graft_entry__ as G
pkg = G.load_package()
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(sys.argv[1], "bench.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
wl = B.WORKLOADS["T"]
ids, off, ln, total = B.plan_job(wl, rank, world, "strong", pkg.sharding)
owned = torch.zeros(wl["n_utt_total"], dtype=torch.int64); owned[torch.from_numpy(ids)] = 1
dist.all_reduce(owned)
assert bool((owned == 1).all())                                   # a partition of the one job
assert list(ids) == list(range(rank, wl["n_utt_total"], world))   # and the round-robin one
frames = int(pkg.sharding.frames_of(ln, wl["W"], wl["S"]).sum())
tot = pkg.sharding.gather_counts(frames, dist)
assert tot == wl["n_utt_total"] * ((wl["utt_samples"] - (wl["W"] - wl["S"])) // wl["S"]), tot
pcm = B.synth_pcm_by_index(torch, ids, wl["utt_samples"], wl["sr"], "cpu")
# utterance 2 (rank 0 at world 2) and utterance 3 (rank 1): the same bits as a single-rank run would generate
ref = B.synth_pcm_by_index(torch, [2 + rank], wl["utt_samples"], wl["sr"], "cpu")
assert torch.equal(pcm[1], ref[0])
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_strong_scaling_job(tmp_path):
    """bench.py --scaling strong, world_size 2 on gloo: the two ranks' utterance sets are the round-robin partition of the
    one job, the frame total is the job's, and each rank generates exactly the PCM a one-rank run generates."""
    script = tmp_path / "worker_strong.py"
    script.write_text(_STRONG_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29537", str(script), ROOT],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("ok") >= 2


def test_cpp_driver_float_formatter_matches_printf():
    """afet_hip formats feature values with its own exact "%f" for floats (one 64-bit product, a shift and round-half-to-even
    on the exact remainder) instead of a general double formatter: it must print what printf("%f") prints -- the
    reference's text (ASR_OCL.cpp:254-257) -- for random bit patterns, ties (1/128), denormals, signed zeros, NaN / inf."""
    exe = os.path.join(ROOT, "asr-featext-opencl_amd", "host", "afet_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    r = subprocess.run([exe, "--selftest-format", "300000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout, r.stderr)


def test_bench_collective_default_is_gloo():
    """bench.py --gpus N: the barrier / max-over-ranks ride on gloo unless RCCL is asked for (no collective is on the data
    path; ASR_OCL.cpp:340-368 has none either), and the RCCL probe is bounded well below the driver's time limit."""
    import bench
    ap = bench.build_parser()
    assert ap.parse_args([]).collective == "gloo"
    assert bench.resolve_backend(ap.parse_args(["--gpus", "8"]).collective, {}) == "gloo"
    assert bench.resolve_backend(ap.parse_args(["--collective", "rccl"]).collective, {}) == "nccl"
    assert bench.resolve_backend("rccl", {"MFX_BENCH_BACKEND": "gloo"}) == "gloo"
    assert bench.RCCL_PROBE_TIMEOUT_S <= 60
    for name, (label, source) in bench.BOUND_DIAGNOSED.items():
        assert os.path.exists(os.path.join(ROOT, source.split(":")[0])), source


def test_shape_to_kernel_table(pkg):
    """ONE table of shape -> front-end kernel of the batch entries (asr-featext-opencl_amd/mfcc.py KERNEL_TABLE; DESIGN.md
    section 5 prints it): the five BASELINE.json configurations, every row of profiles/r03/shapes_beyond_baseline.txt and the
    limits of each kernel, asked of PLANNING handles -- the library's own dispatch rule (choose_front, mfx_api.cpp: the function
    the batch entry launches from) on its own host-built tables, without a device.  tests/test_parity_gpu.py checks real
    handles against the same table."""
    names = set()
    for what, kw, want in pkg.KERNEL_TABLE:
        assert pkg.plan_kernel(**kw) == want, what
        names.add(want)
    assert names == {"k_front512", "k_front1024", "k_front2048", "k_front_reg", "k_front_wave"}
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for what, kw, want in pkg.KERNEL_TABLE:   # the printed table is this table
        assert "| %s | `%s` |" % (what, want) in design, what


def test_planning_handle_computes_nothing(pkg):
    """mfx_plan_create: geometry and the kernel choice only -- every entry that would need the device fails with
    MFX_ERR_DEVICE (-5 family) instead of computing; configurations mfx_create refuses are refused here too."""
    L = pkg.load_library()
    cfg = pkg.MfxConfig(8000, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, 0, 22.0, 0, 2, 3, 3, 1, 0, 1, 1, 0, 0, 0)
    h = C.c_void_p()
    assert L.mfx_plan_create(C.byref(cfg), C.byref(h)) == 0
    assert L.mfx_get_output_data_width(h) == 39 and L.mfx_fft_size(h) == 512
    assert L.mfx_get_input_buffer_size(h) == 48 * 160 + 240 and L.mfx_estimated_window_count(h, 16000) == 98
    w = np.ones(400, np.float32)
    n = C.c_int32()
    pcm = np.zeros(8000, np.int16)
    out = np.zeros(39 * 64, np.float32)
    rcs = [L.mfx_set_window(h, w.ctypes.data_as(C.POINTER(C.c_float))),
           L.mfx_set_input(h, pcm.ctypes.data_as(C.POINTER(C.c_int16)), 8000, C.byref(n)),
           L.mfx_apply(h), L.mfx_flush(h, C.byref(n)),
           L.mfx_get_output_data(h, out.ctypes.data_as(C.POINTER(C.c_float)), 1), L.mfx_synchronize(h)]
    assert len(set(rcs)) == 1 and rcs[0] < 0 and b"planning handle" in L.mfx_last_error(h)
    assert rcs[0] == L.mfx_create(C.byref(cfg), 9999, C.byref(C.c_void_p()))   # = MFX_ERR_DEVICE
    L.mfx_destroy(h)
    bad = pkg.MfxConfig(8000, 400, 160, 40, 16000.0, 64.0, 8000.0, 13, 0, 0.0, 0, 2, 3, 3, 1, 0, 1, 1, 0, 0, 0)   # lifter 0
    assert L.mfx_plan_create(C.byref(bad), C.byref(h)) != 0
    with pytest.raises(pkg.MfxError):
        pkg.plan_kernel(window_size=8192, shift=160, num_banks=40, sample_rate=16000.0, ceps_len=13)   # > 4096 points


def test_cpp_driver_survives_mutated_headers():
    """tools/fuzz_wav.py: 300 mutants of the reference's own sound files' headers (random bytes and fields, zeroed fields,
    truncations) -- the driver ends with a message and an exit code, never by a signal.  Round 4 found two SIGFPEs this way
    (0 channels in a fmt chunk; a 1 Hz SPHERE header: window and shift of 0 samples).  Without a GPU the well-formed mutants stop
    at the extractor's constructor; on the GPU box they run through."""
    exe = os.path.join(ROOT, "asr-featext-opencl_amd", "host", "afet_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_wav
    assert fuzz_wav.run(seed=5, mutants=300, verbose=False) == 0
