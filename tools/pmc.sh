cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc1.log 2>&1
tail -2 $R/gpurun_out/pmc1.log | cut -c1-300
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc2.log 2>&1
tail -2 $R/gpurun_out/pmc2.log | cut -c1-300
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc4 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc4.log 2>&1
ls -la $R/gpurun_out/pmc*/*/ | head -30
