set -x
mkdir -p gpurun_out/r3h
cd /root/repo
timeout 300 python tools/ref_protocol_profile.py > gpurun_out/r3h/ref_profile.log 2>&1; grep -v "amdgpu.ids" gpurun_out/r3h/ref_profile.log | head -60 | cut -c1-180
timeout 200 python tools/stage_times.py > gpurun_out/r3h/stage_times.log 2>&1; tail -1 gpurun_out/r3h/stage_times.log
