export PYTHONPATH=$PWD
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for lib in base new base new; do
  if [ $lib = base ]; then export PMX_LIB=pharmsol_amd/lib/ab/base.so; else unset PMX_LIB; fi
  python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib c5 %.3f ms  err %.2e' % (d['ms_per_step'], d['max_rel_err_vs_cpu_ref']))"
done
