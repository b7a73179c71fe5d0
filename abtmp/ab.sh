for v in base c8 c2 s32c8w2 s32c16w2 w2 t12 t28 base; do
  echo "== $v"; TREEE_HIP_LIB=$PWD/abtmp/$v.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "knn_curve" | tail -1 | sed 's/.*"value": \([0-9.]*\).*"knn_curve": {"ms": \([0-9.]*\).*/value=\1 knn_ms=\2/'
done
