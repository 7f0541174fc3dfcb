cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "encode or unusual or schedules or switches" 2>&1 | tail -2
bash tools/ab_prof.sh "--mode encode" "k_wc_integrate"
bash tools/ab_prof.sh "--mode encode" "k_wc_integrate"
bash tools/ab_prof.sh "--mode encode" "k_wc_integrate"
