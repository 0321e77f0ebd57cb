"""One-paragraph summary of a bench.py JSON line (profiles/rNN_bench_runs.txt): python scripts/bench_summary.py <bench.json> [label]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
lab = sys.argv[2] if len(sys.argv) > 2 else ''
r = d['roofline']; g = d.get('geometry', {}); c3 = d.get('config3', {}); c5 = d.get('config5', {})
err = (d.get('logit_max_abs_err_vs_oracle') or {}).get('value')
print(f"{lab}: split {d['value']} slices/s ({d['ms_per_step']} ms/step) | f16 {d.get('f16_mode', {}).get('value')} | exact {d.get('other_mode', {}).get('exact', {}).get('value') if isinstance(d.get('other_mode'), dict) else None} | "
      f"dominant-kernel frac {r['frac']} ({r.get('kernel_ms_per_launch_avg')} ms/launch), traffic {r.get('traffic')} | logit max-abs-err vs oracle {err}")
def ratio(k): return '/'.join(str(g.get(k, {}).get(m, {}).get('ratio_to_512')) for m in ('split', 'f16'))
print(f"    geometry ratio to 512x512 split/f16: 640x384 {ratio('640x384')}; 448x576(7) {ratio('448x576_7stages')}; config3 f16 {c3.get('f16', {}).get('sub_model_forwards_per_s')} / split "
      f"{c3.get('split', {}).get('sub_model_forwards_per_s')} forwards/s; config5 f16 {c5.get('f16', {}).get('value')} / split {c5.get('split', {}).get('value')} images/s; "
      f"cpu_baseline {d.get('cpu_baseline', {}).get('value')} slices/s ({d.get('cpu_baseline', {}).get('cpu_model')})")
