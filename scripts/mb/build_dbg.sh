#!/bin/bash
# builds scripts/mb/dbg_persist.bin: the persistent-GMW watchdog harness with progress markers / time stamps (-DSRUKF_GMW_DBG)
set -e
cd "$(dirname "$0")"
mkdir -p dbg
for f in srukf_factor srukf_gmw_persist; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -DSRUKF_GMW_DBG -w -c ../../cv-monoslam_amd/csrc/$f.hip -o dbg/$f.o
done
hipcc -O2 -std=c++17 --offload-arch=gfx950 -DSRUKF_GMW_DBG -w -c dbg_persist.cpp -o dbg/dbg_persist.o
hipcc --offload-arch=gfx950 dbg/dbg_persist.o dbg/srukf_factor.o dbg/srukf_gmw_persist.o -o dbg_persist.bin
