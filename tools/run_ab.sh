mkdir -p gpurun_out/r03h; O=gpurun_out/r03h
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $O/pytest.log; cat $O/pytest.log
(timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo rc=$? >> $O/bench.err); tail -3 $O/bench.err
