O=gpurun_out/r2_tests; mkdir -p $O
python -m pytest tests -m gpu -x -q "$@" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -15 $O/pytest.txt
