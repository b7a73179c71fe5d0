#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the library's HOST-side phases (LAZ codec, Segmenting phase): the GPU
# sanitizers are not available on the development pool, the host code is plain C++ and runs anywhere.
#   bash tests/sanitize_host.sh        (from the repo root; needs no GPU)
# Builds the host sources (LAZ codec, Segmenting phase, shard plan) with -fsanitize=address,undefined into abtmp/san/libtreee_hip_san.so (the device objects
# come from the normal build) and runs their CPU tests against it.
set -e
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/abtmp/san
mkdir -p $out
make -C $repo/treee_amd/csrc -s
cd $repo/treee_amd/csrc
for f in context comm multi radix_sort octree knn splat ingest shard calculations; do cp build/$f.o $out/; done
for f in laz_codec segmenting shard_plan; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -x hip -c $f.cpp -o $out/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $out/libtreee_hip_san.so $out/*.o -Wl,-rpath,/opt/rocm/lib -ldl
asan=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $repo
rm -f $out/asan.log.*
LD_PRELOAD=$asan ASAN_OPTIONS=detect_leaks=0:log_path=$out/asan.log TREEE_HIP_LIB=$out/libtreee_hip_san.so \
  python -m pytest tests/test_laz.py tests/test_segmenting.py tests/test_abi.py tests/test_host_logic.py -x -q -k "not gloo and not cpp_host_mirror"
if ls $out/asan.log.* >/dev/null 2>&1; then cat $out/asan.log.*; exit 1; fi
echo "sanitizers: clean"
