#!/bin/bash
# tuning aid: NN-pass time for every coarse-kernel variant (QT x VAR)
for qt in 2 4; do for v in 0 1; do
  echo -n "QT=$qt VAR=$v: "
  ICPMI_COARSE_QT=$qt ICPMI_COARSE_VAR=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nn pass %.3f ms, steady %.0f it/s, whole %.0f it/s' % (d['roofline']['nn_pass_ms'], d['steady_state_it_per_s'], d['value']))"
done; done
