#!/bin/bash
# round 6, GPU run 37: the closing sequence on the final tree — full GPU suite + smoke (tools/verify_r06.sh), the default line, second start, kernel stats,
# PMC passes and the timeline (tools/final_r06.sh), then the two side lines again: the translated sub-operator plans and the narrow resident format
ulimit -c 0
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r06_run37
mkdir -p $OUT
bash tools/verify_r06.sh
bash tools/final_r06.sh
cd $R
timeout 900 python bench.py --plans subop --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 > $OUT/bench_subop.json 2> $OUT/bench_subop.err; echo "subop rc=$?"
timeout 900 python bench.py --narrow-decimals 2 --cpu-sample-sf 0 --cpu-reference-legs 0 --record-runs 0 > $OUT/bench_narrow2.json 2> $OUT/bench_narrow2.err; echo "narrow2 rc=$?"
python - <<'PY'
import json
for f in ("final_r06/bench_default", "r06_run37/bench_subop", "r06_run37/bench_narrow2"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"] if d.get("roofline") else None, {k: (v.get("equal") if isinstance(v, dict) and "equal" in v else v) for k, v in d["checks"].items() if "at_bench" in k or k.endswith("_all")})
    except Exception as e:
        print(f, "failed", e)
PY
