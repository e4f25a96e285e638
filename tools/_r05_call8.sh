cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
rm -f gpurun_out/observed_counts.json
FRCNN_RECORD_OBSERVED=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_resnet_gpu.py -q -m gpu 2>&1 | tail -4
cat gpurun_out/observed_counts.json | python -c "import json,sys; print(json.dumps(json.load(sys.stdin)))"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 | cut -c1-250
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-extra-legs > gpurun_out/r05f/bench_quick.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05f/bench_quick.json')); print('value',d['value'],'single',d.get('single_stream_images_per_sec')); r=d['roofline']; print('roofline',r['frac'],r['avg_launch_us'],r['launches'],r.get('headline_table',{}).get('frac'))"
