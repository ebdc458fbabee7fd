cd /root/repo
for a in "" "nomatch" "nodl" "nomatch,nodl"; do
  python3 bench.py --steps 2000 --warmup 5 --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile --repeat 2 ${a:+--ablate $a} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate [$a]', d['ms_per_step'], d['repeats']['ms_per_step'])"
done
