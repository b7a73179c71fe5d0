#!/bin/bash
# A/B of kernel build variants (knn.hip by default; AB_SRC=splat.hip etc. for another translation unit) inside ONE gpurun call (timings from different boxes differ by several per cent,
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
    src=${AB_SRC:-knn.hip}
    stem=${src%.*}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -x hip -c $src -o /tmp/${stem}_$name.o
    others=$(ls build/*.o | grep -v "build/$stem.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $repo/abtmp/$name.so $others /tmp/${stem}_$name.o -Wl,-rpath,/opt/rocm/lib -ldl
    echo "built abtmp/$name.so ($flags)"
  done
else
  cd $repo
  for v in "$@"; do
    echo "== $v"
    TREEE_HIP_LIB=$repo/abtmp/$v.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | tail -1 |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], {k: v['ms'] for k, v in d['stages'].items()})"
  done
fi
