mkdir -p gpurun_out
FRCNN_LIB_PATH=build/libfrcnn_gxclk.so timeout 200 python tools/gx_clocks.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gx_clocks.log
