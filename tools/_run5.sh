pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_update']['avg_ms'], d['with_keyframe_drops']['frames_per_sec'])"; }
for occ in 3 2; do for prio in 0 -1; do
  echo "== CORR_OCC=$occ ENC_PRIO=$prio"; DPVO_CORR_OCC=$occ DPVO_ENC_PRIO=$prio python bench.py --no-cpu-baseline 2>&1 | tail -1 | pr
done; done
