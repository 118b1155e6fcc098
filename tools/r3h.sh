set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
timeout -k 10 200 python tools/kernel_phases.py --what fwd > $O/kp_fwd.txt 2>&1; echo rc=$?
timeout -k 10 200 python tools/kernel_phases.py --what bwd > $O/kp_bwd.txt 2>&1; echo rc=$?
cd /tmp && export TMPDIR=/tmp
echo "[sq] G forward SQ counters"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_gfwd -- python3 $R/tools/gfwd_loop.py --reps 2 > $O/pmc_sq_gfwd.log 2>&1; echo rc=$?
echo done
