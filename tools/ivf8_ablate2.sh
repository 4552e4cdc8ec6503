for e in 0 1 4 8 12 13 32 256; do
  if [ $e == 0 ]; then L=""; else L="GNNLM_LIB=gnn-lm_amd/build/exp/libivf8_e$e.so"; fi
  echo "== EXP=$e"; env $L REPS=5 timeout 200 python tools/ivfpq_bench.py 2>&1 | grep 'medians'
done
