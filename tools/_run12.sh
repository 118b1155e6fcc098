mkdir -p gpurun_out/r4; O=gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/t_all3.log 2>&1; echo "tests rc=$?"; tail -6 $O/t_all3.log | cut -c1-300
