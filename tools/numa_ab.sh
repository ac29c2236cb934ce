#!/bin/bash
# Runs ON THE GPU BOX: is the run-to-run spread of the host-bound legs (c1, c3: 6.1 k vs 8.2 k, 2.65 k vs 3.2 k pairs/s) the NUMA node the process
# happens to start on?  bench.py pinned to either node's CPUs with taskset, three runs each.
export VELO_DRIVE_CACHE=/tmp/dc
lscpu | grep -i "numa\|socket\|model name" | head -8
for d in /sys/class/drm/card*/device; do [ -f $d/vendor ] && echo "$d vendor $(cat $d/vendor) numa $(cat $d/numa_node 2>/dev/null) cpus $(cat $d/local_cpulist 2>/dev/null)"; done | head -12
which taskset numactl
run() { timeout 300 "$@" python bench.py --no-legs --no-cpu-baseline --steps 20 --warmup 5 --workload $W 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(l['value']), end=' ')"; }
python bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
echo "node0 $N0 node1 $N1"
for W in c3 c1; do
  echo -n "$W free:  "; for i in 1 2 3 4; do run env; done; echo
  echo -n "$W node0: "; for i in 1 2 3 4; do run taskset -c $N0; done; echo
  [ -n "$N1" ] && { echo -n "$W node1: "; for i in 1 2 3 4; do run taskset -c $N1; done; echo; }
done
