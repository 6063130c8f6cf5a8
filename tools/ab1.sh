for f in $1; do
    echo -n "hip flags=$f: "
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-convergence-run --no-configs --launches-per-step 1 --steps 10 --warmup 3 --debug-flags $f --reps 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/launch (min %.3f)  %.3e steps/s' % (d['launch_ms']['mean'], d['launch_ms']['min'], d['value']))"
done
