export V5_SHAPES="256,128,128;256,256,128;256,64,128;512,128,128"
CMU_V5_MIN_K_BST=128 CMU_V5_MIN_K=64 python tools/v5_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-230
