bash tools/prof_round5.sh r5z cfg3 > gpurun_out/r5z_run.log 2>&1
bash tools/prof_round5.sh r5z_conv cfg3_conv > gpurun_out/r5z_conv_run.log 2>&1
bash tools/prof_round5.sh r5z_cfg2 cfg2 > gpurun_out/r5z_cfg2_run.log 2>&1
# keep what travels back small: drop the raw per-dispatch counter tables
find gpurun_out/r5z* -name "*counter_collection.csv" -size +20M -delete
du -sh gpurun_out/r5z* | tail -8
tail -5 gpurun_out/r5z_run.log
