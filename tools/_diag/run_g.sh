set -e
export V5_SHAPES="256,128,128;128,256,256;64,512,512"
bash tools/v5_variants.sh - tools/_diag/libcmunet_wgs1.so tools/_diag/libcmunet_wgs2.so tools/_diag/libcmunet_wgs3.so - > gpurun_out/v5_wgstag.log 2>&1
bash tools/profile_workloads.sh r05 > gpurun_out/profile_workloads_r05.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_spark_r05 -- python3 $R/bench.py --workload spark --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/tl_spark_r05.log 2>&1
cd $R
python tools/step_timeline.py gpurun_out/tl_spark_r05 > gpurun_out/r05_spark_timeline.txt
python bench.py > gpurun_out/bench_r05_default2.log 2> gpurun_out/bench_r05_default2.err
tail -c 300 gpurun_out/v5_wgstag.log
