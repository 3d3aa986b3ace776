"""Developer tool: compact view of bench.py JSON lines (stdin or files): value, ms/step, phases."""
import json
import sys
for line in (open(a).read() for a in sys.argv[1:]) if len(sys.argv) > 1 else sys.stdin:
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    ph = {k: round(v, 3) for k, v in (d.get('phases_ms_per_call') or {}).items() if v}
    rf = d.get('roofline') or {}
    print('%-8s %9.2f %s  %8.4f ms/step  frac %.3f (%.3f ms)  %s' % (
        d['config']['workload'][:8], d['value'], d['unit'], d['ms_per_step'], rf.get('frac') or 0, rf.get('avg_launch_ms') or 0, ph))
