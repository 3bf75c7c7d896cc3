# kernel trace of the batch detect path at both sizes (gpurun): gpurun_out/r03e/<size>/trace_kernel_stats.csv
mkdir -p gpurun_out/r03e; export TMPDIR=/tmp; R=$PWD; cd /tmp
for cfg in "640 480 1000 56" "1280 960 4000 28"; do set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03e/$1x$2 -o trace -- python $R/tools/bench_detect_batch.py $1 $2 $3 $4 4 > $R/gpurun_out/r03e/$1x$2.json 2> $R/gpurun_out/r03e/$1x$2.err
done
cd $R; find gpurun_out/r03e -name "*.db" -delete; find gpurun_out/r03e -name "*agent_info*" -delete
for d in 640x480 1280x960; do echo $d; cat gpurun_out/r03e/$d.json; head -14 gpurun_out/r03e/$d/trace_kernel_stats.csv | cut -d, -f1-8; done
