#!/bin/bash
# dev aid (GPU box): one-worker files/s of afet_hip over N copies of a0001.wav for several batch sizes / helper counts
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-2048}; EXE=$R/asr-featext-opencl_amd/host/afet_hip
D=$(mktemp -d /tmp/afet_bench.XXXX)
args=""
for i in $(seq 1 $N); do args="$args $R/tests/golden/a0001.wav $D/o$i"; done
OPT="--banks 26 --ceps 13 --c0 0 --norm 0 --dyn 2 --l1 3 --l2 3"
$EXE $OPT --devs 0 $R/tests/golden/a0001.wav $D/warm > /dev/null 2>&1
for mode in "" "--htk"; do
  for cfg in "" "--io-threads 8" "--batch-mb 0"; do
    s=$(date +%s.%N); $EXE $OPT $mode --timing $cfg $args > /dev/null 2> $D/err; rc=$?; e=$(date +%s.%N)
    python3 -c "print('%-5s %-32s rc %d: %8.1f files/s' % ('$mode' or 'text', '$cfg', $rc, $N / ($e - $s)))"; tail -2 $D/err
  done
done
# the same over 8 x N files (short paths: the argument list stays under the kernel's limit): start-up amortised
cp $R/tests/golden/a0001.wav $D/a.wav; args=""; M=$((8 * N))
for i in $(seq 1 $M); do args="$args $D/a.wav $D/p$i"; done
for mode in "" "--htk"; do
  s=$(date +%s.%N); $EXE $OPT $mode --timing $args > /dev/null 2> $D/err; rc=$?; e=$(date +%s.%N)
  python3 -c "print('%-5s %-32s rc %d: %8.1f files/s' % ('$mode' or 'text', '($M files)', $rc, $M / ($e - $s)))"; tail -2 $D/err
done
rm -rf $D
