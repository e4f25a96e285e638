# tools/exp_train_wgrad_stream.sh -- the ResNet train step with the bottlenecks' weight gradients on a second stream (FRCNN_TRAIN_WGRAD_STREAM=1, the default) against one stream (=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python tools/train_bench.py "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step, losses', d['first_total_loss'], d['last_total_loss'])"; }
for rep in 1 2; do for v in 0 1; do
  echo "wgrad stream $v: resnet101 bf16 + RoIAlign $(FRCNN_TRAIN_WGRAD_STREAM=$v run --backbone resnet101 --grad-math bf16 --roi align)   resnet101 f32 $(FRCNN_TRAIN_WGRAD_STREAM=$v run --backbone resnet101)   resnet50 f32 $(FRCNN_TRAIN_WGRAD_STREAM=$v run --backbone resnet50)"
done; done
