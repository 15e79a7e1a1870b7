#!/bin/bash
# dev aid (GPU box): files/s of the afet_hip driver over N copies of the reference's a0001.wav (7.1 s, 711 frames x 39)
#   tools/afet_bench.sh [N] [exe]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-256}; EXE=${2:-$R/asr-featext-opencl_amd/host/afet_hip}
D=$(mktemp -d /tmp/afet_bench.XXXX)
args=""
for i in $(seq 1 $N); do args="$args $R/tests/golden/a0001.wav $D/o$i"; done
TIMING=${AFET_TIMING---timing}
OPT="--banks 26 --ceps 13 --c0 0 --norm 0 --dyn 2 --l1 3 --l2 3"
for mode in "" "--htk"; do
  for devs in 0 0,0,0,0; do
    $EXE $OPT --devs 0 $R/tests/golden/a0001.wav $D/warm > /dev/null 2>&1
    s=$(date +%s.%N); $EXE $OPT $mode $TIMING --devs $devs $args > /dev/null 2> $D/err; rc=$?; e=$(date +%s.%N)
    tail -1 $D/err; python3 -c "print('%-6s workers %-8s rc %d: %7.1f files/s' % ('$mode' or 'text', '$devs', $rc, $N / ($e - $s)))"
  done
done
md5sum $D/o1 $D/o$N | cut -c1-32 | tr '\n' ' '; echo
# the per-file loop (--batch-mb 0) for comparison: same bytes, round-2 speed
for mode in "" "--htk"; do
  s=$(date +%s.%N); $EXE $OPT $mode $TIMING --batch-mb 0 --devs 0 $args > /dev/null 2> $D/err; rc=$?; e=$(date +%s.%N)
  tail -1 $D/err; python3 -c "print('%-6s per-file loop, 1 worker rc %d: %7.1f files/s' % ('$mode' or 'text', $rc, $N / ($e - $s)))"
done
md5sum $D/o1 $D/o$N | cut -c1-32 | tr '\n' ' '; echo
rm -rf $D
