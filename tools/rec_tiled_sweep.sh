#!/bin/bash
# A/B of brnn_recurrent_t_kernel schedules (SCTC_REC_TCFG, recurrent.hip launcher table): us per time step at B = 64 / 128
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for c in ${CFGS:-0 1 2 3 4}; do
  echo "== SCTC_REC_TCFG=$c"
  SCTC_REC_TCFG=$c timeout 300 python tools/rec_tiled_check.py ${BS:-64 128} 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('  B=%d tiled %.2f us (fwd %.2f bwd %.2f) frac %.3f | base %.2f | bit-identical hActs %s/%s graddist %.1e' % (d['B'], d['tiled']['us_per_time_step'], d['tiled']['fwd'], d['tiled']['bwd'], d['tiled']['frac_of_f32_mfma_peak'], d['one_slab_per_cu']['us_per_time_step'], d['bit_identical_to_minibatches_of_32']['hActsFor'], d['bit_identical_to_minibatches_of_32']['hActsBack'], d['grad_rel_distance']))"
done
