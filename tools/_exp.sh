python3 tools/time_small_steps.py 2>/dev/null
echo "poll off"; LSQAMD_POLL=0 python3 tools/time_small_steps.py 4096 256 2>/dev/null
echo "fuse_min 1"; LSQAMD_FUSE_MIN_TILES=1 python3 tools/time_small_steps.py 4096 256 2>/dev/null
python3 -m pytest tests/test_gpu_stepgraph.py tests/test_gpu_parity.py tests/test_gpu_midsize.py tests/test_gpu_edge.py tests/test_gpu_uninit.py tests/test_gpu_fuzz.py tests/test_gpu_qr.py tests/test_gpu_trs.py tests/test_gpu_dist2.py -x -q 2>&1 | tail -12
