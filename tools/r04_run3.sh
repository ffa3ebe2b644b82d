cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r04c/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04c/pytest.log
tail -15 gpurun_out/r04c/pytest.log
