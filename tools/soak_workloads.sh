for wl in ${WLS:-kitti tumvi}; do timeout 400 python3 bench.py --workload $wl --steps ${STEPS:-1500} --warmup 5 --cpu-frames 0 --no-pmc --no-profile --repeat 2 2>/dev/null | tail -1 > /tmp/o_$wl.json; python3 - $wl <<'PY'
import sys, json
wl = sys.argv[1]
d = json.load(open('/tmp/o_%s.json' % wl)); r = d['repeats']
print('soak', wl, d['steps'], 'steps x 2 regions', d['value'], r['ms_per_step_in_order'], [g['max_gap_ms'] for g in r['step_gaps']], d['parity_checked'])
PY
done
