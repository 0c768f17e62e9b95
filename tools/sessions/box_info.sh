#!/bin/bash
# What kind of box is this?  The pool's boxes differ: most run the default bench at 660-671 frames/s, some at 580-590 with EVERY kernel class at its usual time and ~2.7 us more per
# kernel boundary (profiles/r05_unet_small_native.txt, r05_concurrent_streams.txt).  Printed at the head of a session so that the two kinds can be told apart afterwards.
echo "== box info"; uname -r; nproc; cat /proc/cpuinfo | grep -m1 "model name"
for f in /sys/class/drm/card*/device/power_dpm_force_performance_level; do echo "$f: $(cat $f 2>/dev/null)"; done
for f in /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/pp_dpm_mclk /sys/class/drm/card*/device/pp_dpm_socclk /sys/class/drm/card*/device/pp_dpm_fclk; do echo "$f:"; cat $f 2>/dev/null | tr '\n' ' '; echo; done
rocm-smi --showperflevel --showclocks --showpower --showfwinfo 2>/dev/null | grep -v "^$" | head -60
cat /sys/module/amdgpu/version 2>/dev/null; cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -E "max_engine_clk|simd_count|fw_version|sdma_fw" | head -12
env | grep -E "^HSA_|^HIP_|^AMD_|^GPU_|^ROC" | head -20
numactl -H 2>/dev/null | head -8; taskset -p $$ 2>/dev/null
