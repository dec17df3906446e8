#!/bin/bash
# A/B against the last commit: builds build/variants/libsrukf_hip_HEAD.so from HEAD's csrc/ (a scratch copy under /tmp), so that the committed and the working-tree
# library can be measured in the same gpurun call (bench.py --lib selects one):   bash scripts/ab_head.sh [commit]
set -e -o pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
rev=${1:-HEAD}
rm -rf /tmp/ab_head_src && mkdir -p /tmp/ab_head_src
git -C "$root" archive "$rev" cv-monoslam_amd/csrc include | tar -x -C /tmp/ab_head_src
mkdir -p "$root/build/variants"
rm -f "$root/build/variants/libsrukf_hip_HEAD.so"              # a failed build below must not leave a stale library to be measured
make -s -j8 -C /tmp/ab_head_src/cv-monoslam_amd/csrc 2>&1 | { grep -E "error" || true; }
cp /tmp/ab_head_src/cv-monoslam_amd/libsrukf_hip.so "$root/build/variants/libsrukf_hip_HEAD.so"
make -s -j8 -C "$root/cv-monoslam_amd/csrc"
echo "built build/variants/libsrukf_hip_HEAD.so from $rev"
