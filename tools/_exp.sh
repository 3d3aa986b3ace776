for s in 1 2 4 8; do echo "batch splits $s"; LSQAMD_BATCH_SPLITS=$s PYTHONPATH=. python3 tools/run_c5.py 2>/dev/null | tail -1; done
