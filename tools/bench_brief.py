"""One-line summary of a bench.py JSON line read from stdin:  python bench.py ... | python tools/bench_brief.py tag"""
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else '', d['value'], d['ms_per_step'], round(d['final_loss'], 6), d['roofline']['gemm_ms_per_step'],
      [(m['dtype'], m['value'], m['ms_per_step'], round(m['final_loss'], 6)) for m in d['precision_modes']], 'sample', d['inference']['value'],
      'ppo', d['secondary']['value'])
