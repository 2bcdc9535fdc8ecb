#!/bin/bash
# Build tools/_var/libvitk_<tag>.so: one source recompiled with extra flags, the rest from the cached objects.
# usage: tools/build_variant.sh <tag> <source.hip> [-DFOO=1 ...]      (run `python -m vit_amd._build` first)
set -e
TAG=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/vit_amd/lib/obj
[ -d "$OBJ" ] || OBJ=$(python3 -c "import vit_amd._build as b; print(b.OBJDIR)")
mkdir -p $ROOT/tools/_var
BASE=$(basename $SRC .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result "$@" -c $ROOT/vit_amd/csrc/$BASE.hip -o /tmp/var_${TAG}_$BASE.o
OTHERS=$(ls $OBJ/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_var/libvitk_$TAG.so /tmp/var_${TAG}_$BASE.o $OTHERS
echo $ROOT/tools/_var/libvitk_$TAG.so
