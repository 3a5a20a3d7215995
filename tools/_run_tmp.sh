R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ytrace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o x -- python3 $R/bench.py --workload Y_train --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --no-mfma --no-roofline --no-secondary > $O/stats.log 2>&1
cd $R; python tools/rocpd_stats.py $(find $O/stats -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1; head -24 $O/kernel_stats.txt | cut -c1-200; tail -2 $O/stats.log | cut -c1-300
