#!/bin/bash
# Round-end evidence, everything from ONE commit on ONE box: the bench line (with cpu_baseline), rocprofv3 kernel-trace
# summaries of the same commands, the PMC passes (MFMA busy; FETCH / WRITE sizes: separate --pmc runs, kernel trace only),
# the search's per-kernel trace and the generator's per-token trace.  Output under gpurun_out/round/, named rNN_*:
#   bash tools/round_profiles.sh 03        then copy gpurun_out/round/* to profiles/
N=${1:-03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O
cd $R && timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r${N}_bench_n1.json 2> $O/bench.err     # the driver's command line (100k-page corpus embed included)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp1 /tmp/rp2 /tmp/rp3 /tmp/rp4
rocprofv3 --kernel-trace --stats -d /tmp/rp1 -o b -- python $R/bench.py --no-cpu-baseline --no-extras --corpus-pages 3200 > /tmp/rp1.log 2>&1   # (the full corpus embed is 1.5 M launches of the same kernels: 3 200 pages here)
python $R/tools/prof_summary.py $(find /tmp/rp1 -name '*.db' | head -1) $O/r${N}_bench_kernel_trace.txt
rocprofv3 --kernel-trace --stats -d /tmp/rp2 -o e -- python $R/tools/encode_only.py 4 > /tmp/rp2.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/rp2 -name '*.db' | head -1) $O/r${N}_encode_only_kernel_trace.txt
rocprofv3 --kernel-trace --stats -d /tmp/rp3 -o s -- python $R/tools/search_bench.py 1000 > /tmp/rp3.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/rp3 -name '*.db' | head -1) $O/r${N}_search_kernel_trace.txt
rm -rf /tmp/rp5; rocprofv3 --kernel-trace --stats -d /tmp/rp5 -o t -- python $R/tools/search_templated.py 1000 10 > $O/r${N}_search_templated.json 2>/tmp/rp5.log
python $R/tools/prof_summary.py $(find /tmp/rp5 -name '*.db' | head -1) $O/r${N}_search_templated_kernel_trace.txt
python $R/tools/search_bench.py 1000 128 16 1 > $O/r${N}_search_bench.txt 2>/dev/null   # sweep TF/s and stream-kernel index GB/s
python $R/tools/search_diag.py 100000 2304 1,16,256,1000 > $O/r${N}_search_stages.txt 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/rp4 -o g -- python $R/tools/evisrag_bench.py 5 64 2 1 > $O/r${N}_generate_bench.json 2>/tmp/rp4.log
python $R/tools/gen_prof_summary.py $(find /tmp/rp4 -name '*.db' | head -1) $O/r${N}_generate_kernel_trace.txt
bash $R/tools/pmc_mfma.sh > /tmp/pmc_m.log 2>&1; cp $R/gpurun_out/pmc/mfma_util.txt $O/r${N}_pmc_mfma_util.txt
bash $R/tools/pmc_traffic.sh > /tmp/pmc_t.log 2>&1; cp $R/gpurun_out/pmc/table.txt $O/r${N}_pmc_traffic.txt; cp $R/gpurun_out/pmc/traffic.json $O/r${N}_traffic.json
tail -c 400 $O/r${N}_bench_n1.json; echo; cat $O/r${N}_search_bench.txt; head -12 $O/r${N}_encode_only_kernel_trace.txt | cut -c1-140; ls -la $O
