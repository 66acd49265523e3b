ls /sys/devices/system/node/ | head; for n in /sys/devices/system/node/node*; do echo $n $(cat $n/cpulist) $(grep MemTotal $n/meminfo | awk '{print $4}'); done
for d in /sys/class/drm/card*/device; do echo $d $(cat $d/numa_node 2>/dev/null) $(cat $d/local_cpulist 2>/dev/null) $(cat $d/vendor 2>/dev/null); done
python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], '...')"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; cat /sys/fs/cgroup/cpuset.mems.effective 2>/dev/null
cat /sys/kernel/mm/transparent_hugepage/enabled
