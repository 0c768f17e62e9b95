#!/bin/bash
# round-5 probe: why do some boxes of the pool run the whole forward 13 % slower with every kernel class at its usual time (+2.7 us per kernel boundary)?  Hypothesis: the kernel-argument blocks
# live in host memory, and on those boxes the process (hence its first-touch memory) sits on the CPU socket far from the GPU.  One process per arm: default, kernargs in device memory
# (HIP_FORCE_DEV_KERNARG=1), and the process bound to each NUMA node.
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r5numa_$(date +%H%M%S); mkdir -p $OUT
{
echo "== topology"; nproc; lscpu | grep -E "NUMA|Socket|Model name" ; which numactl taskset
BDF=$(rocm-smi --showbus 2>/dev/null | grep -oE "[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9]" | head -1); echo "visible GPU BDF: $BDF"
[ -n "$BDF" ] && { echo "numa_node: $(cat /sys/bus/pci/devices/$BDF/numa_node 2>/dev/null)  local_cpulist: $(cat /sys/bus/pci/devices/$BDF/local_cpulist 2>/dev/null)"; }
for n in /sys/devices/system/node/node*; do echo "$(basename $n): cpus $(cat $n/cpulist)"; done
echo "== arms"
python tools/quick_forward_time.py "default"
HIP_FORCE_DEV_KERNARG=1 python tools/quick_forward_time.py "HIP_FORCE_DEV_KERNARG=1"
HIP_FORCE_DEV_KERNARG=0 python tools/quick_forward_time.py "HIP_FORCE_DEV_KERNARG=0"
for n in /sys/devices/system/node/node*; do
  id=${n##*node}; cl=$(cat $n/cpulist)
  if which numactl >/dev/null 2>&1; then numactl --cpunodebind=$id --membind=$id python tools/quick_forward_time.py "numactl node $id (cpu + memory)"
  else taskset -c $cl python tools/quick_forward_time.py "taskset node $id cpus $cl"; fi
done
python tools/quick_forward_time.py "default (again)"
} 2>&1 | grep -v amdgpu.ids | tee $OUT/numa.txt
