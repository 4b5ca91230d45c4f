"""Print the numbers of a tools/sessions/r4_final.sh run (gpurun_out/r4_final)."""
import json, sys
R = (sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r4_final') + '/'
def load(n):
  return json.loads([l for l in open(R + n) if l.startswith('{')][0])
d = load('bench.json')
print('cfg2: value', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'host', round(d['host_ms_per_step'], 4))
r = d['roofline']; print(' roofline launch', round(r['launch_us'], 2), 'frac', round(r['frac'], 4), 'traffic', r['traffic'], 'bracketed', round(r['launch_us_single_bracketed'], 1))
r = d['roofline_step']; print(' step', round(r['step_us'], 2), 'frac', round(r['frac'], 4))
print(' prepared', round(d['prepared_frames']['value']), round(d['prepared_frames']['launch_us'], 1), ' scene', round(d['scene_like_depth']['value']), ' cpu', round(d['cpu_baseline']['value']), d['cpu_baseline']['cores'], round(d['cpu_baseline']['single_thread_value']), d['cpu_baseline']['gpu_matches_cpu_on_sample'])
for k, v in d['other_configs'].items(): print('  ', k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a not in ('workload', 'checked_by', 'algorithmic_bytes')} if isinstance(v, dict) else v)
d = load('bench_rot1.json'); print('rot1: value', round(d['value']), 'launch', round(d['roofline']['launch_us'], 2), round(d['roofline']['frac'], 3), 'step', round(d['roofline_step']['step_us'], 2))
for w in ('cfg3', 'cfg4', 'cfg5'):
  d = load(f'bench_{w}.json')
  print(w, 'value', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'host', round(d['host_ms_per_step'], 4), 'launch', round(d['roofline']['launch_us'], 1), 'frac', round(d['roofline']['frac'], 3))
  if 'roofline_step' in d: print('   step', round(d['roofline_step']['step_us'], 1), round(d['roofline_step']['frac'], 3))
  for k, v in d.get('legs', {}).items(): print('   ', k, round(v['us'], 1), round(v['frac'], 3))
d = load('bench_gloo2.json'); print('gloo2: value', round(d['value']), d['ranks']['backend'], 'ring_of_1', round(d['ring_of_1']['value']))
