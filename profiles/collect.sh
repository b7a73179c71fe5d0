#!/bin/bash
# Collects the judged profile set on a one-GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh <tag>       e.g. r01_v6        (BENCH_ARGS="--splat-sorted" adds bench.py flags)
# Kernel stats and PMC counters are taken in SEPARATE rocprofv3 runs (never --pmc together with tracing).
set -e
tag=${1:-r01_vX}
repo=$PWD
out=$repo/gpurun_out/$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $BENCH_ARGS > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $BENCH_ARGS > $out/write.log 2>&1
cd $repo
python profiles/summarize.py $out $tag
python bench.py $BENCH_ARGS > $out/bench_default.json 2> $out/bench_default.err
tail -1 $out/bench_default.json
