#!/bin/bash
# A/B of kNN kernel build variants inside ONE gpurun call (timings from different boxes differ by several per cent,
# so every decision recorded in profiles/README.md was taken this way).
#
#   bash profiles/ab.sh build base "" v1 "-DTREEE_NG=8" v2 "-DTREEE_TRANSPOSE_BELOW=24"     (here, no GPU needed)
#   gpurun -- 'bash profiles/ab.sh run base v1 v2 base'                                      (on the GPU box)
#
# `build` compiles knn.hip once per variant into abtmp/<name>.so (the other objects come from the normal build);
# `run` runs the default benchmark step with each library through TREEE_HIP_LIB and prints value + kNN kernel time.
set -e
repo=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; shift
mkdir -p $repo/abtmp
if [ "$mode" = build ]; then
  make -C $repo/treee_amd/csrc -s
  cd $repo/treee_amd/csrc
  while [ $# -gt 0 ]; do
    name=$1; flags=$2; shift 2
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -c knn.hip -o /tmp/knn_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $repo/abtmp/$name.so build/context.o build/radix_sort.o build/octree.o \
      /tmp/knn_$name.o build/splat.o build/ingest.o build/shard.o build/calculations.o build/segmenting.o build/laz_codec.o -Wl,-rpath,/opt/rocm/lib
    echo "built abtmp/$name.so ($flags)"
  done
else
  cd $repo
  for v in "$@"; do
    echo "== $v"
    TREEE_HIP_LIB=$repo/abtmp/$v.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 |
      sed 's/.*"value": \([0-9.]*\).*"knn_curve": {"ms": \([0-9.]*\).*/value=\1 knn_ms=\2/'
  done
fi
