timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "knn or curv or 15m or 150m" 2>&1 | tail -1
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export TREEE_KNN_NO_ORDER=1; echo "== no order"; else unset TREEE_KNN_NO_ORDER; echo "== ordered"; fi
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "knn_curve" | tail -1 | sed 's/.*"value": \([0-9.]*\).*"knn_curve": {"ms": \([0-9.]*\).*"knn_build": {"ms": \([0-9.]*\).*/value=\1 knn_ms=\2 build_ms=\3/'
done
