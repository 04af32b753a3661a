#!/bin/bash
# GPU box: FPS merge A/B over the variant libraries (puflow_amd/build.py tags), random-order and patch-ordered clouds
for v in "" $FPS_VARIANTS; do
  lib=puflow_amd/libpuflow_hip$v.so
  [ -f $lib ] || continue
  echo "--- $lib"
  PF_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/time_fps.py 99840 20024 1,8,32 cube || exit 1
  PF_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/time_fps.py 99840 20024 1,8,32 patch || exit 1
done
