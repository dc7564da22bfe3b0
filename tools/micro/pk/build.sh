#!/bin/bash
# builds tools/micro/pk_next_to_mfma.bin (gfx950; cross-compiles without a GPU)
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc"
hipcc $F -c victim_pk.hip -o /tmp/victim_pk.o
hipcc $F -Xclang -target-feature -Xclang -packed-fp32-ops -c victim_nopk.hip -o /tmp/victim_nopk.o 2>&1 | grep -v "not a recognized feature" || true
hipcc $F -c main.hip -o /tmp/pk_main.o
hipcc --offload-arch=gfx950 -fgpu-rdc /tmp/victim_pk.o /tmp/victim_nopk.o /tmp/pk_main.o -o ../pk_next_to_mfma.bin
