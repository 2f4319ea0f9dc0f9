run() { echo "== $*"; env "$@" bash tools/kstats.sh k3v python /root/repo/tools/k3_time.py 250 5e6 21 2048 2 | grep -E "k3c?_(scatter|bmh_main|bmh_surv|hist|split|refine)"; grep "K3 ng" /tmp/ks_k3v.out; }
run X=1
run D2G_K3_COMPACT=1
