rm -f gpurun_out/parity_record.txt
python -m pytest tests -m gpu -x -q > gpurun_out/t8.log 2>&1; tail -4 gpurun_out/t8.log
python bench.py > gpurun_out/bench_r04_default.log 2>gpurun_out/bench_r04_default.err; tail -1 gpurun_out/bench_r04_default.log | cut -c1-600
bash tools/profile_round.sh r04 2>&1 | tail -3
bash tools/profile_workloads.sh r04 2>&1 | tail -4
bash tools/pmc_conv.sh r04 > gpurun_out/pmc_conv_r04.txt 2>&1; tail -2 gpurun_out/pmc_conv_r04.txt | cut -c1-200
for w in moco joint; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w', j['value'], j['ms_per_step'], j.get('roofline', {}).get('frac'))"; done
bash tools/ab_bench.sh tools/_diag/libcmunet_r03.so > gpurun_out/ab_r04c.log 2>&1; cat gpurun_out/ab_r04c.log
