echo "cpu.max:"; cat /sys/fs/cgroup/cpu.max 2>&1
echo "cpu.stat before:"; cat /sys/fs/cgroup/cpu.stat 2>&1
grep Cpus_allowed_list /proc/self/status
nproc
cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1
