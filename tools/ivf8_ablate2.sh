#!/bin/bash
# Round-6 ablation builds of the search filter (csrc/ivfpq_mfma.hip, -DGNNLM_IVF8_EXP=<bits>: 1 no code loads, 4 no LDS look-ups, 32 nothing
# survives, 256 code bytes always from the same four tiles; 512 = phase clocks): build them with
#   for e in 1 4 32 256 512; do tools/build_variant.sh ivf8_e$e ivfpq_mfma.hip -DGNNLM_IVF8_EXP=$e; done
# then run this on the GPU box (medians of 5 searches of 8192 queries at the reference's index shape over 103 M keys).
for e in 0 1 4 32 256; do
  if [ $e == 0 ]; then L=""; else L="GNNLM_LIB=gnn-lm_amd/build/exp/libivf8_e$e.so"; fi
  echo "== GNNLM_IVF8_EXP=$e"; env $L GNNLM_IVF_QUERY_BLOCK=8192 REPS=5 timeout 200 python tools/ivfpq_bench.py 2>&1 | grep 'medians'
done
echo "== phase clocks of thread 0 / wave 8 (GNNLM_IVF8_EXP=512)"
GNNLM_LIB=gnn-lm_amd/build/exp/libivf8_e512.so GNNLM_IVF_QUERY_BLOCK=8192 timeout 300 python tools/ivf8_phases.py 2>&1 | grep -v amdgpu.ids
