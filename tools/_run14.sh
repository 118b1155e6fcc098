mkdir -p gpurun_out/r4; O=gpurun_out/r4
W=$O/ab_offsets2.txt; : > $W
for i in 1 2; do
echo "previous library" >> $W; MPGAN_LIB_PATH=$PWD/cross-modality-minipig-gan_amd/libmpgan_hip_prev.so timeout -k 10 120 python tools/bench_bf16.py --reps 8 --modes wgrad >> $W 2>&1
echo "new library" >> $W; timeout -k 10 120 python tools/bench_bf16.py --reps 8 --modes wgrad >> $W 2>&1
done
grep -v amdgpu.ids $W
timeout -k 10 400 python -m pytest tests/test_bf16_gpu.py -q -x > $O/t_k.log 2>&1; echo "tests rc=$?"; tail -2 $O/t_k.log
