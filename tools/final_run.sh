python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
grep -E "^real" gpurun_out/final_bench.err
tail -1 gpurun_out/final_bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read())
keep={k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')}
print(json.dumps(keep)); print(json.dumps(d['config'])[:400]); r=d['roofline']; print({k:r[k] for k in ('bound','achieved','peak','unit','frac','traffic')}); c=d['cpu_baseline']; print({k:c[k] for k in ('value','unit','cores','kind')}, c['sample'][:120])"
