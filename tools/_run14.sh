export AMD_LOG_LEVEL=0
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profkf -o kf -- python /root/repo/tools/exp_keyframe_update.py 8 > /root/repo/gpurun_out/r05/prof_kf.log 2>&1
cd /root/repo
python tools/rocprof_summary.py /tmp/profkf "python tools/exp_keyframe_update.py 8   (round 5: 8+ key-frames on the resident window from Python and C host loops, then the same by reload)" > gpurun_out/r05/keyframe_update_rocprof_summary.txt
grep -a "k_win\|k_ef_copy_out\|k_pyr_level\|k_ef_finish" gpurun_out/r05/keyframe_update_rocprof_summary.txt | cut -c1-160
bash tools/pmc_loop_kernels.sh r05 2>&1 | tail -45
