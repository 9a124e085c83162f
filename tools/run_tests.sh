cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tests
(time python -m pytest tests -m gpu -x -q "$@") > gpurun_out/tests/pytest.log 2>&1
tail -25 gpurun_out/tests/pytest.log
