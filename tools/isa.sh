#!/bin/bash
# dev aid: ISA of one kernel instantiation -> /tmp/asm/<tag>.s   usage: tools/isa.sh <tag> [mangled-substring] [translation unit]
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-k}; PAT=${2:-10k_front512ILb1ELb0ELi13ELb0ELb0ELb0EE}; TU=${3:-mfx_front512}
mkdir -p /tmp/asm && cd /tmp/asm
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize ${MFX_DEFS:-} -S --cuda-device-only "$R/asr-featext-opencl_amd/csrc/$TU.hip" -o all_$TAG.s 2>/dev/null
a=$(grep -n "^_ZN.*$PAT.*:" all_$TAG.s | head -1 | cut -d: -f1)
b=$(grep -n "Lfunc_end.*:" all_$TAG.s | awk -F: -v a=$a '$1>a{print $1; exit}')
sed -n ${a},${b}p all_$TAG.s > $TAG.s
wc -l $TAG.s
