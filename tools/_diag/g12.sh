python -m pytest tests/test_gpu_pretrain.py -x -q -k "masked_recon_step_gate_forced" -s 2>&1 | grep -v Warn | tail -12
