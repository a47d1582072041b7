#!/bin/bash
# Round-3 evidence in one GPU call (copy gpurun_out/r03/* into profiles/ afterwards):
#   bench line, kernel stats of the same command, PMC counters of the final Phi and contraction kernels, kernel
#   timelines of single evaluations, the other BASELINE configurations, the factorisation alone, config 5.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py > $O/r03_bench_n1.json 2> $O/bench_n1.err || exit 1
echo "bench done"; tail -c 600 $O/r03_bench_n1.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 > $O/stats.log 2>&1 || exit 2
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/r03_bench_kernel_stats.csv; rm -rf $O/stats
echo "stats done"
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
i=0
P5="TCC_HIT_sum TCC_MISS_sum"
P6="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
for set in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE" "$P5" "$P6"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc$i -- python3 tools/eval_trace.py Cm grad 2 > $O/pmc$i.log 2>&1 || exit 3
  cp $(ls $O/pmc$i/*/*counter_collection.csv | head -1) $O/pmc$i.csv; rm -rf $O/pmc$i
done
python tools/pmc_to_json.py $O/r03_pmc_syrk.json "gemm_f64_kernel<128, 128, true, true, 1>" $O/pmc1.csv $O/pmc2.csv $O/pmc3.csv $O/pmc4.csv $O/pmc5.csv $O/pmc6.csv > /dev/null
python tools/pmc_to_json.py $O/r03_pmc_gradgemm.json "gemm_f64_kernel<128, 128, false, true, 2>" $O/pmc1.csv $O/pmc2.csv $O/pmc3.csv $O/pmc4.csv $O/pmc5.csv $O/pmc6.csv > /dev/null
rm -f $O/pmc?.csv
echo "pmc done"
tools/eval_timeline.sh r03/tl Cm > /dev/null 2>&1
cp $O/tl/timeline_lml.txt $O/r03_timeline_cm_bound.txt; cp $O/tl/timeline_grad.txt $O/r03_timeline_cm_bound_grad.txt; rm -rf $O/tl
tools/chol_trace.sh r03/chol > /dev/null 2>&1; cp $O/chol/chol_steps.txt $O/r03_chol_steps.txt; rm -rf $O/chol
python tools/config_times.py 2>&1 | grep -v amdgpu > $O/r03_config_times.log
echo "configs done"
python tools/c5_probe.py 2>&1 | grep -v amdgpu > $O/r03_c5_probe.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5stats -- python3 tools/c5_probe.py > $O/c5stats.log 2>&1 && cp $(ls $O/c5stats/*/*kernel_stats.csv | head -1) $O/r03_c5_kernel_stats.csv
rm -rf $O/c5stats
ls -la $O
