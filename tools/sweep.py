"""N-sweep of the step kernel (dev tool): kernel time per launch vs number of envs."""
import json, subprocess, sys
for n in [int(x) for x in sys.argv[1:]] or [65536, 262144, 1048576, 4194304]:
    steps = max(200, min(20000, int(2e9 // n)))
    out = subprocess.run([sys.executable, 'bench.py', '--envs', str(n), '--steps', str(steps), '--warmup', str(max(50, steps // 10)),
                          '--no-cpu-baseline', '--graph-len', str(min(1000, steps))], capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(n, 'env-steps/s %.3e' % d['value'], 'us/launch %.2f' % d['roofline']['avg_launch_us'], 'GB/s %.0f' % d['roofline']['achieved'], 'frac %.3f' % d['roofline']['frac'])
    except Exception as e:
        print(n, 'FAILED', out.stderr[-500:])
