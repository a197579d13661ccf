bash tools/collect_profiles.sh r03 > gpurun_out/collect_r03.log 2>&1
tail -30 gpurun_out/collect_r03.log
