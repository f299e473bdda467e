export TMPDIR=/tmp
for rep in 1 2; do for lib in e6 e7; do
  GPH_LDS_SUM=0 timeout 600 python3 bench.py --lib bench_cache/$lib.so --steps 12 --warmup 3 --preroll 120 --no-cpu-baseline 2>/dev/null < /dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); k=l['kernels']
print('$lib: %.3f ms/step, %.1f M evals/s, sweep %.3f ms, tau_eval %.4f, mix_eval %.4f launches %.0f accept %s' % (l['ms_per_step'], l['value']/1e6, l['roofline']['avg_launch_ms'], k['tau_eval']['avg_ms'], k['mix_eval']['avg_ms'], l['config']['kernel_launches_per_iteration'], l['config']['accept_counts_timed'][:4]))"
done; done
