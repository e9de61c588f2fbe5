timeout 400 python tools/time_variants.py 10000000 2>&1 | grep -v amdgpu
timeout 200 python tools/time_variants.py 262144 2>&1 | grep -v amdgpu | grep "padded64 30-bit"
timeout 100 python tools/time_meshes.py 2>&1 | grep -v amdgpu | tail -8
timeout 300 python tools/cpu_baselines.py > gpurun_out/cpu_baselines.json 2>/dev/null; cat gpurun_out/cpu_baselines.json
