#!/bin/bash
# GPU box: FPS merge A/B over the variant libraries (python puflow_amd/build.py tags fps2, fpsms16, fpskw4)
for v in "" _fps2 _fpsms16 _fpskw4; do
  lib=puflow_amd/libpuflow_hip$v.so
  [ -f $lib ] || continue
  echo "--- $lib"
  PF_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/time_fps.py || exit 1
done
