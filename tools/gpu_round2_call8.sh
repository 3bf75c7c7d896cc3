mkdir -p gpurun_out/r02h
(timeout 1500 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_sift.py tests/test_gpu_g2o.py tests/test_gpu_async.py tests/test_gpu_multi.py -m gpu -q -x --timeout 900 > gpurun_out/r02h/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02h/tests.log)
tail -5 gpurun_out/r02h/tests.log
timeout 300 python tools/fuzz_pairs.py phased 1 40 > gpurun_out/r02h/fuzz_phased.log 2>&1; tail -2 gpurun_out/r02h/fuzz_phased.log
timeout 300 python tools/check_huge_batch.py > gpurun_out/r02h/huge.log 2>&1; tail -3 gpurun_out/r02h/huge.log
for W in 0.01 0.002 0.005; do
 for CH in 0; do
   timeout 120 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --depth-noise $W --chunk-iterations $CH --ransac-path record_replay > gpurun_out/r02h/b_${W}_${CH}.json 2>/dev/null
   python - <<PY
import json
d=json.load(open("gpurun_out/r02h/b_${W}_${CH}.json"))
print("noise $W chunk $CH: value %.0f ms_per_step %.3f serial ransac %.3f iters %.1f edges %.3f" % (d["value"], d["ms_per_step"], d["timing"]["serial_stage_ms"]["select_ransac"], d["config"]["mean_ransac_iterations"], d["config"]["edge_fraction"]))
PY
 done
done
timeout 300 python tools/bench_all_pairs.py 250 1000 2>/dev/null
timeout 300 python tools/bench_batch_sweep.py 2>/dev/null | tail -12
