cd $GRAFT_REPO_ROOT
for cfg in "2 0" "1 0" "1 1000" "1 1536" "1 100000"; do
  set -- $cfg
  echo "== NSETS=$1 BM64_BELOW=$2"
  FRCNN_WINO_NSETS=$1 FRCNN_WINO_BM64_BELOW=$2 timeout 200 python tools/layer_bench.py --winograd --only conv --reps 20 2>&1 | grep -E "conv3_1|conv3_2|conv4_1|conv4_2|conv5_x"
done
FRCNN_WINO_NSETS=1 FRCNN_WINO_BM64_BELOW=1536 timeout 300 python bench.py --math f32_winograd --no-cpu-baseline --no-secondary --steps 300 2>&1 | tail -1 | cut -c1-120
FRCNN_WINO_NSETS=1 FRCNN_WINO_BM64_BELOW=100000 timeout 300 python bench.py --math f32_winograd --no-cpu-baseline --no-secondary --steps 300 2>&1 | tail -1 | cut -c1-120
FRCNN_WINO_NSETS=2 FRCNN_WINO_BM64_BELOW=0 timeout 300 python bench.py --math f32_winograd --no-cpu-baseline --no-secondary --steps 300 2>&1 | tail -1 | cut -c1-120
