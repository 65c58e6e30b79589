python -m pytest tests/test_gpu_bwd_ops.py tests/test_gpu_fwd_ops.py tests/test_gpu_sparse_tiles.py tests/test_gpu_pretrain.py -x -q 2>&1 | tail -1
for w in recon spark; do for lib in tools/_diag/lib_base.so ""; do CMU_LIB_PATH=$lib python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --all-kernel-events 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']; print('$w', '${lib:-tree}'.rjust(24), j['ms_per_step'], {n: k[n] for n in k if 'pool' in n or 'mask' in n or 'stats' in n})"; done; done
