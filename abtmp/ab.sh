timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "octree or sort or slice or radius" 2>&1 | tail -2
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export TREEE_SORT_3K=1; echo "== 3-kernel"; else unset TREEE_SORT_3K; echo "== onesweep"; fi
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*"keys_sort_octree": {"ms": \([0-9.]*\).*/value=\1 sort_octree_ms=\2/'
done
