cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_winograd_gpu.py tests/test_model_gpu.py -x -q -m gpu -s 2>&1 | grep -E "winograd (forward|predict)|passed|failed|Error|error" | tail -12
timeout 200 python tools/layer_bench.py --winograd --only conv --reps 20 2>&1 | grep -E "conv3_|conv4_|conv5_"
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 400 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['per_class_ms_per_image'])"
