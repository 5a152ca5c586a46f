#!/bin/bash
# round 6, first GPU call: latency table, the new structure tests, the BA legs with the new cases
O=gpurun_out/r06; mkdir -p $O
./tools/ubench/f64_latency > $O/f64_latency.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_ba_structures.py tests/test_gpu_localmapping.py "tests/test_gpu_ba.py::test_persistent_cholesky_is_bit_identical" -x -q -m gpu -s > $O/tests_first.txt 2>&1
echo "tests rc=$?" >> $O/tests_first.txt
GPU_MAX_HW_QUEUES=12 timeout 1200 python bench_ba.py --out $O/bba --cpu 0 > $O/bba.log 2>&1
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06/bba/result.json"))
for k, v in r.items():
    if k != "roofline": print(k, v)
for k, c in r["roofline"]["cases"].items():
    print(k, "frac %.4f incl_schur %.4f dense_equiv %.4f ms %.1f" % (c["frac"], c["executed_incl_schur"], c["dense_equiv_rate"], c["ms"]), c.get("solves_per_s"), c["skyline"], c.get("plan"), c.get("ms_per_solve_latency"), c.get("ms_per_iteration"))
PY
tail -5 $O/tests_first.txt; cat $O/f64_latency.txt
