#!/bin/bash
# A/B builds of libsrukf_hip.so that differ in ONE compile-time macro of one file (measurement only; the product is csrc/Makefile's build):
#   bash scripts/build_variants.sh <file.hip> <MACRO> <value> [<value> ...]   ->  build/variants/libsrukf_hip_<MACRO>_<value>.so
# (build/ is git-ignored and not next to the product library; it still travels to the GPU box with a gpurun snapshot)
# select one at run time with bench.py --lib <path> (or srukf.load_library(path) as the first call of a script).
set -e
cd "$(dirname "$0")/../cv-monoslam_amd/csrc"
make -s -j8
out="$(cd ../.. && pwd)/build/variants"; mkdir -p "$out"
f=$1; m=$2; shift 2
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16"
[ "$f" = srukf_gmw_persist.hip ] && flags="$flags -Os -mllvm -amdgpu-sched-strategy=max-ilp"
[ "$f" = srukf_assoc.hip ] && flags="$flags -ffp-contract=off"
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
for v in "$@"; do
  /opt/rocm/bin/hipcc $flags -D${m}=${v} -c $f -o "$tmp/variant_${m}_${v}.o"
  objs=$(ls *.o | grep -v "^${f%.hip}.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libsrukf_hip_${m}_${v}.so" $objs "$tmp/variant_${m}_${v}.o"
  echo "built $out/libsrukf_hip_${m}_${v}.so"
done
