#!/bin/bash
# Round profiles: bench line, kernel stats of the same command, PMC traffic of the Phi and contraction kernels.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R && timeout -k 10 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 > $O/stats.log 2>&1 || exit 2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/eval_trace.py Cm grad 2 > $O/pmc_fetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/eval_trace.py Cm grad 2 > $O/pmc_write.log 2>&1 || exit 4
cd $R
for d in pmc_fetch pmc_write; do python tools/pmc_collect.py $(ls $O/$d/*/*counter_collection.csv | head -1) "gemm_f64_kernel<128, 128" > $O/$d.txt; done
ls $O/stats/*/ > $O/stats_files.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cat $O/pmc_fetch.txt $O/pmc_write.txt
