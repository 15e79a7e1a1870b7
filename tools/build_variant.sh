#!/bin/bash
# dev aid: build an experimental libmfcchip under build/var/ (travels to the GPU box; MFX_LIB=... selects it)
#   tools/build_variant.sh <name> ["-DFLAG ..."] [git-rev]      (git-rev: the kernel sources -- csrc/*.hip and the device headers --
#   taken from that commit; the current mfx_kernels.h / host objects must still fit them)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; DEFS=${2:-}; REV=${3:-}
C=$R/asr-featext-opencl_amd/csrc
T=/tmp/var_$NAME
rm -rf $T; mkdir -p $R/build/var $T
TUS="mfx_front512 mfx_front_generic mfx_front2048 mfx_tail"
SRCDIR=$C
if [ -n "$REV" ]; then
  for f in $(git -C $R ls-tree --name-only $REV asr-featext-opencl_amd/csrc/ | grep -E '\.(hip|h)$'); do git -C $R show $REV:$f > $T/$(basename $f); done
  SRCDIR=$T
  TUS=$(cd $T && ls *.hip | sed 's/\.hip$//')
fi
pids=""
for t in $TUS; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize $DEFS -I$SRCDIR -I$C -c $SRCDIR/$t.hip -o $T/$t.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
make -s -C $C mfx_api.o mfx_tables.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build/var/lib_$NAME.so $T/*.o $C/mfx_api.o $C/mfx_tables.o
echo built build/var/lib_$NAME.so
