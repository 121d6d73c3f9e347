#!/bin/bash
# Build a second liblfd_hip.so with extra -D flags for same-session A/B timing:
#   tools/ab_build.sh scratch/alt/liblfd_hip_x.so -DSOME_VARIANT
#   gpurun -- 'python bench.py ...; LFD_HIP_LIB=$PWD/scratch/alt/liblfd_hip_x.so python bench.py ...'
# (run-to-run differences between GPU boxes are 5-20 %: only numbers from one session are comparable)
set -e
OUT=$(realpath -m "$1"); shift
cd "$(dirname "$0")/../lfd-a-light-and-fast-detector_amd/csrc"
B=$(mktemp -d)
for f in $(sed -n 's/^SRCS *= *//p' Makefile | sed 's/\.hip//g'); do
  extra=""
  [ $f = stem_fused ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  [ $f = head ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  [ $f = block128 ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  [ $f = planes ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  [ $f = planes_ml ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  [ $f = planes_head ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  [ $f = planes_stem2x ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  [ $f = planes_stem2xs ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  [ $f = targets ] && extra="-fhip-fp32-correctly-rounded-divide-sqrt"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -DLFD_BUILDING "$@" $extra -c $f.hip -o $B/$f.o &
done
wait
mkdir -p "$(dirname "$OUT")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $B/*.o -Wl,-rpath,/opt/rocm/lib -Wl,-soname,liblfd_hip.so
rm -rf $B; ls -la "$OUT"
