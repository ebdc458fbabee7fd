for b in 32 64 128 256 512 1024; do python3 bench.py --batch $b --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile --verify 0 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('euroc batch $b', d['value'], 'ms/step', d['ms_per_step'], 'us/frame', round(d['ms_per_step']*1000/$b,3), 'host-input', d['pcie_inclusive']['value'])"; done
