#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
python tools/chol_persist_prof.py > $O/chain_c4.txt 2>&1; python - <<'PY'
import json; r=json.load(open("gpurun_out/cholprof/chol_persist_prof.json")); print(json.dumps({k:v for k,v in r.items() if k!="per_step_ns"}, indent=1))
PY
python tools/chol_persist_prof.py c5 > $O/chain_c5.txt 2>&1; tail -12 $O/chain_c5.txt
python tools/chol_wg_prof.py > $O/wg_prof.txt 2>&1; tail -25 $O/wg_prof.txt
