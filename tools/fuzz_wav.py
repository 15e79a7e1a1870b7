"""Dev aid: mutated headers of the reference's own sound files (tests/golden: a RIFF/WAVE file, its NIST SPHERE sample1.wav and the
same as RIFF) through the driver's readers and -- on a GPU box -- on through the extractor: random bytes, random 32-bit fields,
zeroed fields, truncations.  Whatever the file says, asr-featext-opencl_amd/host/afet_hip must end with a message and an exit
code, never by a signal (round 4: a fmt chunk with 0 channels and a SPHERE header with a 1 Hz sample rate both died of SIGFPE).

    python tools/fuzz_wav.py [seed] [mutants]      exit code 1 if any mutant killed the driver
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "asr-featext-opencl_amd", "host", "afet_hip")


def run(seed=1, mutants=400, verbose=True):
    rng = np.random.default_rng(seed)
    srcs = [open(os.path.join(ROOT, "tests", "golden", n), "rb").read() for n in ("a0001.wav", "sample1_sphere.wav", "sample1_riff.wav")]
    d = tempfile.mkdtemp(prefix="fuzz_wav_")
    killed = 0
    for i in range(mutants):
        b = bytearray(srcs[i % 3][:20000])
        hdr = 64 if i % 3 != 1 else 1100
        for _ in range(int(rng.integers(1, 6))):
            mode = int(rng.integers(0, 4))
            if mode == 0:
                b[int(rng.integers(0, hdr))] = int(rng.integers(0, 256))
            elif mode == 1:
                p = int(rng.integers(0, hdr - 4))
                b[p:p + 4] = int(rng.integers(0, 2 ** 32)).to_bytes(4, "little")
            elif mode == 2:
                b = b[:int(rng.integers(0, len(b)))]
            else:
                p = int(rng.integers(0, hdr))
                b[p:p + 2] = b"\x00\x00"
            if len(b) < hdr:
                break
        f = os.path.join(d, "m.wav")
        open(f, "wb").write(bytes(b))
        r = subprocess.run([EXE, f, os.path.join(d, "o.txt")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        if r.returncode < 0:
            killed += 1
            if verbose:
                print("mutant %d (seed %d): killed by signal %d" % (i, seed, -r.returncode))
    if verbose:
        print("seed %d: %d mutants, %d killed the driver" % (seed, mutants, killed))
    return killed


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 400) else 0)
