TREEE_HIP_LIB=$PWD/abtmp/b8.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "knn or curv or 15m" 2>&1 | tail -1
for v in b4 b8 b4 b8; do
  echo "== $v"; TREEE_HIP_LIB=$PWD/abtmp/$v.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "knn_curve" | tail -1 | sed 's/.*"value": \([0-9.]*\).*"knn_curve": {"ms": \([0-9.]*\).*/value=\1 knn_ms=\2/'
done
