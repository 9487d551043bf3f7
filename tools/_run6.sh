nproc; free -g | head -2
pr() { python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['config']['parallelism'], [ (r['rank'], round(r['seconds'],4), r['host_cpu_us_per_frame']) for r in d['per_rank']])
"; }
for n in 2 4 8; do
  echo "== gpus=$n shared device, overlap off (default)"; timeout 300 python bench.py --gpus $n --no-cpu-baseline 2>&1 | tail -3 | pr
done
for n in 2 8; do
  echo "== gpus=$n shared device, DPVO_OVERLAP_ENC=1"; DPVO_OVERLAP_ENC=1 timeout 300 python bench.py --gpus $n --no-cpu-baseline 2>&1 | tail -5
done
