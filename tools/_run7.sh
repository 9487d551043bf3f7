pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_update']['avg_ms'], d['with_keyframe_drops']['frames_per_sec'])"; }
for rep in 1 2; do for cfg in 3 1 0 2; do
  echo "== FU_CFG=$cfg"; DPVO_FU_CFG=$cfg python bench.py --no-cpu-baseline 2>&1 | tail -1 | pr
done; done
for cfg in 3 1; do echo "== update_bench FU_CFG=$cfg"; DPVO_FU_CFG=$cfg WHICH=fused REPS=30 python tools/update_bench.py 2>&1 | grep fused; done
