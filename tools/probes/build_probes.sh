#!/bin/bash
# Probe builds for tools/sharing_probe_*.py: small libraries of ONE ablated kernel file + ac_api (context, error string).
# Run from the repo root after `make -C audio_cut_amd/csrc`; outputs under tools/probes/build/ (git-ignored, shipped by gpurun).
set -e
cd "$(dirname "$0")/../../audio_cut_amd/csrc"
OUT=../../tools/probes/build
mkdir -p $OUT /tmp/probe_objs
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -ffp-contract=off"
build() {  # name source extra-flags
  /opt/rocm/bin/hipcc $FLAGS $3 -c $2 -o /tmp/probe_objs/$1.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/probe_objs/$1.o build/ac_api.o -o $OUT/lib$1.so
}
for bits in 0x04 0x08 0x10 0x20 0x40 0x80 0x60 0x0c 0x8c; do build conv_p$bits ac_conv96.hip -DW9_PROBE=$bits & done
build mdx_notw ac_mdx.hip -DMDX_PROBE=2 &
build mdx_noslp ac_mdx.hip "-fno-slp-vectorize" &
build mdx_nt ac_mdx.hip -DMDX_NT_LOADS=1 &
wait
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -shared -fPIC ../../tools/probes/canary.hip -o $OUT/libcanary.so
ls -la $OUT
