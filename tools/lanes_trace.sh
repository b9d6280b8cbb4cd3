mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03 && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lane_prof -o lane -- python3 $GRAFT_REPO_ROOT/bench.py --graph 0 --no-cpu-baseline --no-also --no-roofline --steps 6 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r03/lanes_under_profiler.json 2>/dev/null
python3 $GRAFT_REPO_ROOT/tools/trace_overlap.py $GRAFT_REPO_ROOT/gpurun_out/lane_prof/lane_kernel_trace.csv 4000 | tee $GRAFT_REPO_ROOT/gpurun_out/r03/lanes_overlap.txt
tail -c 1500 $GRAFT_REPO_ROOT/gpurun_out/r03/lanes_under_profiler.json | grep -o '"ms_per_step": [0-9.]*' | head -1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/lane_prof
