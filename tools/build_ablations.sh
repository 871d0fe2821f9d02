#!/bin/bash
# builds libdemon_hip variants with parts of the conv_patch K loop removed (timing experiments only)
set -e
cd "$(dirname "$0")/../demon_amd/csrc"
mkdir -p ../../gpurun_in
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result"
for v in NOLOAD NOSTORE "NOSTORE -DABL_NOBARRIER" "NOLOAD -DABL_NOSTORE -DABL_NOBARRIER" "NOFRAG" "NOLOAD -DABL_NOSTORE -DABL_NOBARRIER -DABL_NOFRAG"; do
  tag=$(echo "$v" | sed 's/ -DABL_/_/g')
  hipcc $FL -DABL_$v -c conv_patch.hip -o /tmp/conv_patch_$tag.o &
done
wait
for v in NOLOAD NOSTORE "NOSTORE -DABL_NOBARRIER" "NOLOAD -DABL_NOSTORE -DABL_NOBARRIER" "NOFRAG" "NOLOAD -DABL_NOSTORE -DABL_NOBARRIER -DABL_NOFRAG"; do
  tag=$(echo "$v" | sed 's/ -DABL_/_/g')
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_in/libdemon_$tag.so demon_api.o conv_mfma.o ops.o /tmp/conv_patch_$tag.o
done
ls ../../gpurun_in
