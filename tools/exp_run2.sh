timeout 300 python tools/exp_batch_trace.py trace 1 EXP_SHARED_STREAM=1 > gpurun_out/exp_call_timeline.txt 2>&1
SDVGN_OPT_TIMING=1 timeout 120 python tools/exp_batch_trace.py run 1 2>&1 | tail -12
sed -n 1,75p gpurun_out/exp_call_timeline.txt | cut -c1-110
