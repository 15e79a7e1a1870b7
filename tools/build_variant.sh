#!/bin/bash
# dev aid: build an experimental libmfcchip under build/var/ (travels to the GPU box; MFX_LIB=... selects it)
#   tools/build_variant.sh <name> ["-DFLAG ..."] [git-rev]      (git-rev: kernels source taken from that commit)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; DEFS=${2:-}; REV=${3:-}
C=$R/asr-featext-opencl_amd/csrc
mkdir -p $R/build/var /tmp/var_$NAME
SRC=$C/mfx_kernels.hip
if [ -n "$REV" ]; then
  git -C $R show $REV:asr-featext-opencl_amd/csrc/mfx_kernels.hip > /tmp/var_$NAME/mfx_kernels.hip
  git -C $R show $REV:asr-featext-opencl_amd/csrc/mfx_kernels.h > /tmp/var_$NAME/mfx_kernels.h
  SRC=/tmp/var_$NAME/mfx_kernels.hip
fi
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize $DEFS -I$C -c $SRC -o /tmp/var_$NAME/k.o
# F2048_REV=<git-rev>: k_front2048's source taken from that commit (the current headers must still fit it)
SRC2=$C/mfx_front2048.hip
if [ -n "${F2048_REV:-}" ]; then
  git -C $R show $F2048_REV:asr-featext-opencl_amd/csrc/mfx_front2048.hip > /tmp/var_$NAME/mfx_front2048.hip
  SRC2=/tmp/var_$NAME/mfx_front2048.hip
fi
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize $DEFS -I$C -c $SRC2 -o /tmp/var_$NAME/k2.o
make -s -C $C mfx_api.o mfx_tables.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build/var/lib_$NAME.so /tmp/var_$NAME/k.o /tmp/var_$NAME/k2.o $C/mfx_api.o $C/mfx_tables.o
echo built build/var/lib_$NAME.so
