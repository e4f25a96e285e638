for v in base nt4; do
  if [ $v = base ]; then unset FRCNN_LIB_PATH; else export FRCNN_LIB_PATH=$PWD/build/libfrcnn_$v.so; fi
  echo "== $v"
  python -m pytest tests/test_winofused_gpu.py -q -x 2>&1 | tail -1
  python tools/layer_bench.py --fused --only conv 2>/dev/null | grep conv
  python tools/layer_bench.py --fused --shape 1024,256,64,256,0 2>/dev/null | grep custom
done
