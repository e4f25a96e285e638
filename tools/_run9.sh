cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_winograd_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --steps 400 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['other_math_modes_images_per_sec'], j['roofline']['per_class_ms_per_image'])"
timeout 300 python bench.py --no-cpu-baseline --no-secondary --inflight 1 --steps 200 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('single image in flight', j['value'])"
