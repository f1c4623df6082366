#!/bin/bash
# Tuning builds of the Winograd kernel (csrc/ge_wino.hip): one library per -D combination, selected at run time through
# GE_LIB_PATH (graphecho_amd/_lib.py).  WN_DBG variants compute WRONG results on purpose (phase ablation).
# usage: bash tools/build_wino_variants.sh "tag1:-DWN_DBG=1" "tag2:-DWN_DMA_SPREAD=0" ...      (SRC=ge_wino_wgrad: the weight-gradient file)
set -e
cd "$(dirname "$0")/../graphecho_amd/csrc"
make -j8 > /dev/null
mkdir -p variants
SRC=${SRC:-ge_wino}
OBJS=$(ls *.o | grep -v "^$SRC.o")
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $flags -c $SRC.hip -o variants/${SRC}_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$tag.so $OBJS variants/${SRC}_$tag.o
  rm -f variants/${SRC}_$tag.o
  echo "built variants/lib_$tag.so ($flags)"
done
