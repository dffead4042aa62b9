mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_c4_dispatch_gpu.py tests/test_train_mode_gpu.py tests/test_conv_halo_gpu.py tests/test_encoder_kernels_gpu.py tests/test_metrics_gpu.py tests/test_vae_gpu.py tests/test_vae_train_gpu.py tests/test_gan_step_gpu.py -q -s --durations=15 > gpurun_out/r4b/tests.txt 2>&1
tail -40 gpurun_out/r4b/tests.txt
for v in 0 1; do IPOKE_C4_PER_FRAME=$v timeout 300 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4b/c4_perframe$v.json 2> gpurun_out/r4b/c4_perframe$v.err; tail -c 600 gpurun_out/r4b/c4_perframe$v.json; tail -3 gpurun_out/r4b/c4_perframe$v.err; done
