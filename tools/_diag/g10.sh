python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for w in recon moco joint spark; do CMU_DIST_BACKEND=gloo CMU_SINGLE_DEVICE=1 python bench.py --gpus 2 --workload $w --batch 8 --size 256 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events 2>&1 | python -c "
import sys, json
ls=[l for l in sys.stdin if l.startswith('{')]
j=json.loads(ls[-1]) if ls else {}
print('$w', j.get('n_gpus'), j.get('value'), j.get('rccl'))"; done
CMU_DP_REHEARSE=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --gpus 1 --workload spark --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events 2>&1 | python -c "
import sys, json
ls=[l for l in sys.stdin if l.startswith('{')]
j=json.loads(ls[-1]) if ls else {}
print('spark one-rank RCCL', j.get('n_gpus'), j.get('value'), j.get('ms_per_step'), j.get('rccl'))"
