mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_unet.py -q -s -k full_size 2>&1 | grep -E "sd15 eps|passed|failed|Error" > gpurun_out/r05/unet_emul.txt
cat gpurun_out/r05/unet_emul.txt
python tests/diag/diag_loop_divergence.py 12 1.0 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/loop_divergence.txt
cat gpurun_out/r05/loop_divergence.txt
