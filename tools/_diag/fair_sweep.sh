python tools/v5_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-230
