for H in 1 2; do
  for W in 0 1; do
    if [ $W = 1 ]; then export CORA_WIDE1=1; else unset CORA_WIDE1; fi
    timeout -k 10 120 python bench.py --config 12block --hidden-size 128 --num-heads $H --no-cpu-baseline --no-extra --steps 5 --warmup 2 > gpurun_out/r3h_h${H}_w${W}.json 2> gpurun_out/r3h_h${H}_w${W}.err || exit 1
  done
done
