#!/bin/bash
# A/B against the last commit: builds cv-monoslam_amd/libsrukf_hip_HEAD.so with HEAD's version of ONE source file (the working tree's objects otherwise), so that both
# libraries can be measured in the same gpurun call (SRUKF_LIB selects one).   bash scripts/ab_head.sh srukf_rank.hip
set -e
cd "$(dirname "$0")/../cv-monoslam_amd/csrc"
make -s -j8
f=$1
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -I."
[ "$f" = srukf_gmw_persist.hip ] && flags="$flags -Os -mllvm -amdgpu-sched-strategy=max-ilp"
[ "$f" = srukf_assoc.hip ] && flags="$flags -ffp-contract=off"
git show HEAD:cv-monoslam_amd/csrc/$f > /tmp/ab_head_$f
/opt/rocm/bin/hipcc $flags -c /tmp/ab_head_$f -o /tmp/ab_head.o
objs=$(ls *.o | grep -v "^${f%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsrukf_hip_HEAD.so $objs /tmp/ab_head.o
echo built ../libsrukf_hip_HEAD.so
