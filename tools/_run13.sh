cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_train_gpu.py -x -q -m gpu -s > gpurun_out/train_tests.log 2>&1; grep -E "winograd dgrad|passed|failed|Error|assert" gpurun_out/train_tests.log | tail -12
for m in f32_winograd f32; do echo "vgg16 $m: $(python tools/train_bench.py --math $m 2>/dev/null | tail -1 | cut -c1-140)"; done
echo "resnet50: $(python tools/train_bench.py --backbone resnet50 2>/dev/null | tail -1 | cut -c1-140)"
