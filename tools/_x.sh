echo "shipped:"; python tools/window_host_tick.py
for t in 256 512 1024; do echo "ab threads $t:"; CGP_LIB=corenav_gp_amd/libcorenav_gp_ab.so CGP_WIN_THREADS=$t python tools/window_host_tick.py; done
timeout 900 python -m pytest tests/test_gpu_window.py -x -q 2>&1 | tail -5
