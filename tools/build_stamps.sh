#!/bin/bash
# Diagnostic build of libvitk.so with the GEMM's in-kernel cycle stamps (tools/gemm_stamps.py, tools/gemm_timeline.py).
set -e
cd "$(dirname "$0")/.."
python -m vit_amd._build >/dev/null
mkdir -p tools/_stamps
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS -DVITK_GEMM_STAMPS -c vit_amd/csrc/gemm256.hip -o tools/_stamps/gemm256.o
OBJS=$(ls vit_amd/lib/obj/*.o | grep -v gemm256.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_stamps/libvitk.so $OBJS tools/_stamps/gemm256.o
echo built tools/_stamps/libvitk.so
