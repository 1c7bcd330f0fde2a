# PMC counters of the loop's other kernels (separate --pmc passes, SDVGN_PMC_SAFE: no look-ahead side stream under the serialising profiler)
R=${1:-r05}
export SDVGN_PMC_SAFE=1 SDVGN_PMC_LOOP=1
for k in k_ef_tail_resub k_ef_acc_fused k_ef_acc_stats k_ef_stitch; do
  echo "## $k (ten loop bodies of optimize on the cfg3 window; mean per launch)"
  timeout 280 python tools/pmc_linearize.py $k 2>&1 | grep -v "^# rocprofv3"
done > gpurun_out/${R}_loop_kernel_counters.txt 2>&1
tail -60 gpurun_out/${R}_loop_kernel_counters.txt
