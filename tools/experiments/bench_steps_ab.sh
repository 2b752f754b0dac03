R=${GRAFT_REPO_ROOT}
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for i in 1 2 3; do
for sw in "20 5" "60 15" "150 20" "400 20"; do set -- $sw
  timeout -k 10 200 python3 $R/bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "steps=$1,warmup=$2"
done; done
