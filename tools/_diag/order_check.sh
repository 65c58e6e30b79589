export V5_SHAPES="256,128,128"
python tools/v5_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
CMU_V5_MIN_K_BST=128 python tools/v5_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
python tools/v5_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
CMU_V5_MIN_K_BST=128 python tools/v5_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
