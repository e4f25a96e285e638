for v in base abl16 abl32 abl48 abl64 abl127; do
  if [ $v = base ]; then unset FRCNN_LIB_PATH; else export FRCNN_LIB_PATH=$PWD/build/libfrcnn_$v.so; fi
  echo "== $v"
  python tools/layer_bench.py --fused --only conv3_2 2>/dev/null | grep conv
  python tools/layer_bench.py --fused --shape 1024,256,64,256,0 2>/dev/null | grep custom
done
