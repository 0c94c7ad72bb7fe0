#!/bin/bash
# does the 12-process fuzz hang?  each process under its own timeout; a timed-out one gets its stacks dumped with gdb if there is one
NP=${1:-12}; NC=${2:-20}
which gdb || echo "no gdb"
pids=()
for i in $(seq 1 $NP); do
  ( timeout -s USR1 ${FZ_TIMEOUT:-300} python -X faulthandler -c "
import faulthandler, signal, sys, runpy
faulthandler.register(signal.SIGUSR1, all_threads=True, chain=False)
sys.argv = ['tools/fuzz_align.py', '$NC', '$((1000*$i + 17))']
runpy.run_path('tools/fuzz_align.py', run_name='__main__')
" > /tmp/fz_$i.txt 2>&1; echo "proc $i rc $? $(tail -n 1 /tmp/fz_$i.txt | cut -c1-150)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
uptime
