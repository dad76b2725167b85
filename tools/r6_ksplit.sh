#!/bin/bash
# round 6: the K-split wave-pair form of the 128 x 128 tile -- parity tests, per-candidate timings on the layers the dominant kernel runs,
# and the training step with fresh measurements with / without the form
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "every_conv_kernel_configuration or conv_fwd_bwd or deconv_fwd_bwd" > gpurun_out/r6_ksplit_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r6_ksplit_tests.log
for shape in "256 256 3 1 64 16 16" "512 512 3 1 64 8 8" "128 128 3 1 64 32 32" "1024 256 1 1 64 16 16" "2048 512 1 1 64 8 8"; do
  LH_TUNE_DB=0 LH_TUNE_CACHE=0 LH_TUNE_LOG=1 LH_TUNE_ITERS=10 LH_WGRAD_TABLE=0 timeout -k 10 300 python tools/conv_bench.py $shape 2>&1 | grep -E "^\[tune (fwd|dgrad)" | sort -t: -k2 -n | awk '{print}' >> gpurun_out/r6_ksplit_layers.txt
  echo "----" >> gpurun_out/r6_ksplit_layers.txt
done
grep -E "3[0-9], 128\)|\(128, 128, 2[0-9], 128\)|----" gpurun_out/r6_ksplit_layers.txt | head -80
for v in 0 1; do
  LH_KSPLIT_TILES=$v LH_TUNE_DB=0 LH_TUNE_ITERS=10 LH_TUNE_CACHE=$PWD/gpurun_out/r6_tune_ks$v.txt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > /dev/null 2>&1
done
for rep in 1 2 3; do
  for v in 0 1; do
    ms=$(LH_KSPLIT_TILES=$v LH_TUNE_DB=0 LH_TUNE_CACHE=$PWD/gpurun_out/r6_tune_ks$v.txt python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'], d.get('infer_images_per_s'))")
    echo "fresh tuning, LH_KSPLIT_TILES=$v  $ms" | tee -a gpurun_out/r6_ksplit_step.txt
  done
  ms=$(python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'], d.get('infer_images_per_s'))")
  echo "shipped database  $ms" | tee -a gpurun_out/r6_ksplit_step.txt
done
grep -c "3[0-9], 128)" gpurun_out/r6_tune_ks1.txt
