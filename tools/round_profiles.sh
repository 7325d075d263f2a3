#!/bin/bash
# Round-end evidence: bench line (with cpu_baseline) + rocprofv3 kernel-trace summaries of the same commands.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O
cd $R && timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp1 /tmp/rp2
rocprofv3 --kernel-trace --stats -d /tmp/rp1 -o b -- python $R/bench.py --no-cpu-baseline > /tmp/rp1.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/rp1 -name '*.db' | head -1) $O/bench_kernel_trace.txt
rocprofv3 --kernel-trace --stats -d /tmp/rp2 -o e -- python $R/tools/encode_only.py 4 > /tmp/rp2.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/rp2 -name '*.db' | head -1) $O/encode_only_kernel_trace.txt
python $R/tools/search_bench.py 1000 128 16 1 > $O/search_bench.txt 2>/dev/null   # sweep TF/s and stream-kernel index GB/s
tail -c 600 $O/bench_n1.json; cat $O/search_bench.txt; head -12 $O/encode_only_kernel_trace.txt | cut -c1-140
