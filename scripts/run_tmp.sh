python -m pytest tests/test_conv_shapes_gpu.py tests/test_f16x2_gpu.py tests/test_ops_gpu.py -q -x 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py -q -x -k "pre_resnet50" 2>&1 | tail -2
scripts/ab_multi.sh "XV2_AMAX_PASS=1" "XV2_AMAX_PASS=0" -- --steps 30 --warmup 8
