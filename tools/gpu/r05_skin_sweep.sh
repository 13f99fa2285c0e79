for s in 0.8 1.0 1.2 1.8 2.4 3.0; do
BENCH_ARGS="--skin $s --steps 1500" bash tools/ab_env.sh 1 "" | sed "s/^default/skin $s/"
done
