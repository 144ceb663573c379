# Measurement builds of the LAB flavour of libvqa_hip.so (-DVQA_AB_VARIANTS -DVQA_TEST_SEAMS) with one compile-time probe each, into gpurun_ab/ (git-ignored, travels with gpurun):
#   bash scripts/build_probes.sh NMS3_PROBE 1 2 3   ->  gpurun_ab/libvqa_NMS3_PROBE_1.so ...
# Time them on the GPU box with VQA_LIB_PATH=gpurun_ab/libvqa_<macro>_<v>.so bash scripts/gpu_kernel_ms.sh <label> --no-verify
set -e
macro=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/real-time-video-quality-analysis_amd/csrc
mkdir -p $ROOT/gpurun_ab
for v in "$@"; do
  tmp=$(mktemp -d)
  cp $C/*.hip $C/*.hpp $C/Makefile $tmp/
  mkdir -p $tmp/../../include && cp $ROOT/include/vqa.h $tmp/../../include/ 2>/dev/null || true
  (cd $tmp && sed -i "s#../../include/vqa.h#$ROOT/include/vqa.h#" Makefile *.hip *.hpp && make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=fast -Wno-inline-asm -DVQA_AB_VARIANTS -DVQA_TEST_SEAMS -D$macro=$v" > /dev/null)
  cp $tmp/libvqa_hip.so $ROOT/gpurun_ab/libvqa_${macro}_$v.so
  rm -rf $tmp
  echo built gpurun_ab/libvqa_${macro}_$v.so
done
