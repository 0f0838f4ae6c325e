#!/bin/bash
# round 4: instruction-rate calibration of the ingredients of a likelihood-field look-up
set -u
mkdir -p gpurun_out/r4
hipcc --offload-arch=gfx950 -O3 tools/calib_valu_table.hip -o /tmp/calib_valu 2>/dev/null || exit 1
/tmp/calib_valu | tee gpurun_out/r4/calib_valu_table.txt
hipcc --offload-arch=gfx950 -O3 tools/calib_mfma_f32.hip -o /tmp/calib_mfma_f32 2>/dev/null || exit 1
/tmp/calib_mfma_f32 | tee gpurun_out/r4/calib_mfma_f32.txt
