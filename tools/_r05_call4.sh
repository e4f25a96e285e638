set -u
OUT=gpurun_out/r05d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $OUT/pytest_tail.log; cat $OUT/pytest_tail.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05d/bench_driver_form.json"))
print("value", d["value"], "single_stream", d.get("single_stream_images_per_sec"), "strict", d["config"].get("strict_f32_images_per_sec"))
r=d["roofline"]; print("roofline frac", r.get("frac"), r.get("avg_launch_us"), r.get("launches"), "| headline_table", json.dumps(r.get("headline_table"))[:900])
print("per_class", r.get("per_class_ms_per_image"))
print("r50", d.get("resnet50_images_per_sec"), d.get("resnet50_batch8_images_per_sec"))
PY
