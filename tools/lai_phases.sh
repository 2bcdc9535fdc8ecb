#!/bin/bash
# Where the time of the attention_lai.hip kernels goes: variants that leave every head after phase k (wrong results,
# timing only), built here on the CPU, run on the GPU box by tools/lai_phases_run.sh.
set -e
cd "$(dirname "$0")/.."
python -m vit_amd._build >/dev/null
mkdir -p tools/_lai
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
OBJS=$(ls vit_amd/lib/obj/*.o | grep -v attention_lai.o)
DEF=${VITK_STOP_MACRO:-VITK_LAI_STOP}      # VITK_STOP_MACRO=VITK_FULL_STOP: the unmasked forward's phases
for k in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -D$DEF=$k -c vit_amd/csrc/attention_lai.hip -o /tmp/lai_stop$k.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_lai/libvitk_stop$k.so $OBJS /tmp/lai_stop$k.o ) &
done
wait
ls -la tools/_lai
