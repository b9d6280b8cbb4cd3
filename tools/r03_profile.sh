#!/bin/bash
# End-of-round measurement set (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the same
# command, per-entry step tables, in-step HBM traffic of the largest entries.  Outputs under gpurun_out/r03/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; rm -f $R/gpurun_out/pmc_step_keys.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 $R/bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench done"; tail -c 600 $O/bench_line.json; echo
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03 -- python3 $R/bench.py --no-cpu-baseline --no-also --steps 10 --warmup 5 > $O/bench_under_profiler.json 2> $O/prof.err
echo "profile done"
python3 $R/tools/trace_steps.py $O/prof/r03_kernel_trace.csv 3 60 > $O/by_kernel.txt
python3 $R/tools/trace_overlap.py $O/prof/r03_kernel_trace.csv > $O/overlap.txt
rm -f $O/prof/r03_kernel_trace.csv
for wl in xattn60m audio15m film25m; do
  timeout -k 10 400 python3 $R/tools/step_table.py $wl --json $O/step_table_$wl.json > $O/step_table_$wl.txt 2>/dev/null
  head -2 $O/step_table_$wl.txt | tail -1
done
timeout -k 10 300 python3 $R/tools/pmc_step.py > $O/pmc_keys.log 2>&1      # un-profiled: picks the entries to mark (by time)
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd -- python3 $R/tools/pmc_step.py > $O/pmc_rd.log 2>&1
echo "pmc rd done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr -- python3 $R/tools/pmc_step.py > $O/pmc_wr.log 2>&1
echo "pmc wr done"
cp $R/gpurun_out/pmc_step_keys.json $O/
ls $O/prof/* | head
