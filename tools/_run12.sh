DPVO_CORR_STAGED=0 python tools/corr_bench.py 2>&1 | grep -v amdgpu
DPVO_HIP_LIB=$PWD/dpvo_amd/libdpvo_hip_glds.so DPVO_CORR_STAGED=1 python tools/corr_bench.py 2>&1 | grep -v amdgpu
DPVO_HIP_LIB=$PWD/dpvo_amd/libdpvo_hip_glds.so DPVO_CORR_STAGED=1 python tools/corr_bench.py 2>&1 | grep -v amdgpu
