R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04a_spark -- python3 $R/bench.py --workload spark --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/prof_r04a_spark.log 2>&1
tail -1 $R/gpurun_out/prof_r04a_spark.log | cut -c1-200
cd $R
for e in "" "CMU_SPARK_TILES=0"; do env $e python bench.py --workload spark --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('spark [$e]', j['value'], j['ms_per_step'])"; done
