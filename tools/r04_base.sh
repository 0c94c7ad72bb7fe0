set -x
mkdir -p gpurun_out/r04_base
for wl in p30 c2 ns; do
  HIPSTR_HOST_THREADS=2 timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 5 > gpurun_out/r04_base/${wl}_t2.json 2> gpurun_out/r04_base/${wl}_t2.err
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 5 > gpurun_out/r04_base/${wl}_t16.json 2> gpurun_out/r04_base/${wl}_t16.err
done
nproc; cat /sys/fs/cgroup/cpu.max; lscpu | head -20
