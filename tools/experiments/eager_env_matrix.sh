# eager layer step (tools/hostprof_layer2.py) under settings that might pin its bimodal host time
for cfg in "" "HOSTPROF_SINGLE_THREAD_AUTOGRAD=1" "HOSTPROF_CPUS=8" "HOSTPROF_CPUS=8,9" "HOSTPROF_SINGLE_THREAD_AUTOGRAD=1 HOSTPROF_CPUS=8" "HOSTPROF_CPUS=72,73"; do
  for rep in 1 2 3; do
    echo "== [$cfg] rep $rep"; env $cfg HOSTPROF_TIMING_ONLY=1 timeout 100 python tools/hostprof_layer2.py 2>&1 | grep "host us" | tail -2
  done
done
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3
