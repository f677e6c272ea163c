# Same-box A/B of a variant library (python -m paper_accurate_fast_cheap_amd.csrc.build --extra "<flags>" --out <file>) against the
# tree's own: the headline step, the fp32 + bf16-slot step, c2 and the 2 000-frame windows, alternating.   tools/ab_lib.sh <variant.so> <tag>
set -e
V=$(realpath $1); T=$2
mkdir -p gpurun_out/ab_$T
for rep in 1 2; do
  for which in base var; do
    if [ $which = var ]; then export PAFC_SO_PATH=$V; else unset PAFC_SO_PATH; fi
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$T/${which}_$rep.json 2> gpurun_out/ab_$T/${which}_$rep.err
    python3 - <<PY
import json
d = json.loads(open("gpurun_out/ab_$T/${which}_$rep.json").read().strip().splitlines()[-1])
e = d.get("extra") or {}
g = lambda k: (e.get(k) or {}).get("audio_sec_per_sec") if isinstance(e.get(k), dict) else e.get(k)
print("$which $rep: ms_per_step %.3f scan %.1f us | bf16slot %s ms | c2 %s | windows %s | streaming %s" % (
    d["ms_per_step"], d["roofline"]["avg_launch_us"], e.get("f32_model_bf16_slot_ms_per_step"), g("c2"), g("windows_2000x8"), g("streaming")), flush=True)
PY
  done
done
