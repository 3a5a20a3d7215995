cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm or split" 2>&1 | tail -2
python tools/ubench_gemm_split.py 2>&1 | grep -E "split" | sed 's/(err.*//' | cut -c1-175
