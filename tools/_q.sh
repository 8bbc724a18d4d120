for shape in "8192 16384" "32768 4096" "2048 65536"; do
  set -- $shape
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch $1 --elems $2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$shape', d['ms_per_step'], 'enc', d['encode_ms'], 'dec', d['decode_ms'], d['value'], {k[6:]:v['avg_us'] for k,v in d['kernels'].items()})"
done
