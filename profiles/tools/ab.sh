#!/bin/bash
# profiles/tools/ab.sh TAG LIB [bench args...]: one bench line into gpurun_out/TAG.json, kernel times printed
TAG=$1; LIB=$2; shift; shift
GEOSRAD_LIB=$LIB python bench.py --no-pmc --no-cpu --steps 10 "$@" > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$TAG.json").read().strip().splitlines()[-1])
    k=d["kernels_ms_per_step"]; print("$TAG", "step %.2f ms" % d["ms_per_step"], " ".join("%s=%.2f" % (a.replace("k_",""), b) for a, b in k.items()))
except Exception as e:
    print("$TAG", "ERR", e, open("gpurun_out/$TAG.err").read()[-500:])
PY
