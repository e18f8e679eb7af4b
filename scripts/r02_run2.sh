# "before" profiles of the HBM regime (round-1 kernels + no-contraction build): 10 M / K=97 and 100 M / K=256
TAG=r02a_10M ARGS="--events-per-gpu 10000000 --knots 97" STEPS=10 bash scripts/profile.sh || exit 1
TAG=r02a_100M ARGS="--events-per-gpu 100000000 --knots 256 --pano-h 2048" STEPS=4 bash scripts/profile.sh || exit 1
bash scripts/scaling.sh
