for v in base lx1 lx2 lx3 lx7; do
  if [ $v = base ]; then unset FRCNN_LIB_PATH; else export FRCNN_LIB_PATH=$PWD/build/libfrcnn_$v.so; fi
  echo "== $v"
  python tools/layer_bench.py --only fc1 2>/dev/null | grep x6
done
