#!/bin/bash
# GPU box: kernel duration (rocprofv3) of ec_fwdp_kernel<32,4,128> for every timing-only ablation build (tools/ecp_ablate.py build)
ROOT=$PWD
for m in 0 1 2 4 8 16 3 31; do
  lib=$ROOT/puflow_amd/libpuflow_hip.so; [ $m != 0 ] && lib=$ROOT/puflow_amd/libpuflow_hip_ecp$m.so
  PF_LIB_PATH=$lib PF_ECUNIT_ONLY=1 bash tools/prof_script.sh r4_abl_$m tools/time_ecunit.py > /dev/null 2>&1
  echo "mask $m: $(grep 'ec_fwdp_kernel<32' gpurun_out/r4_abl_$m/kernel_stats.txt)"
done
