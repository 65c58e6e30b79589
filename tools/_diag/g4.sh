rm -f gpurun_out/parity_record.txt
python -m pytest tests -m gpu -x -q > gpurun_out/t4.log 2>&1; tail -6 gpurun_out/t4.log
for w in recon moco joint spark; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w', j['value'], j['ms_per_step'], j.get('roofline', {}).get('frac'))"; done > gpurun_out/w4.log 2>&1; cat gpurun_out/w4.log
